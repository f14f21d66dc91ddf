// mimir_gpu_shuffler.hpp -- C++ host shim: Mimir's shuffler interface on top
// of the C ABI in mimir_gpu_shuffle.h.
//
// GpuShuffler<K,V> is a drop-in for CollectiveShuffler<K,V>
// (src/collectiveshuffler.h): same constructor arguments, same virtuals
// (open / write / close / make_progress / seek / get_record_count,
// src/baseshuffler.h:255-265), same round, bypass and termination behaviour,
// same error convention (message on stderr, then abort of the whole job,
// src/log.h:70-78).  What changes is where the per-record work runs: write()
// only copies the pair's wire form into a pinned staging buffer; hashing,
// destination and the stable pack into the send image happen on the GPU.
//
// The header is self-contained so that it builds without a Mimir checkout:
// it carries its own copy of the three small interface classes (BaseObject,
// Writable, the DB_POS enum -- src/interface.h:18-62, :80-86) in namespace
// mimir_gpu, and talks to the outside world through `Exchange`, a two-call
// transport (int all-to-all, byte all-to-all-v).  Inside Mimir one derives
// from MIMIR_NS::BaseShuffler instead and implements Exchange with
// MPI_Alltoall / MPI_Alltoallv on the shuffler's communicator.
#ifndef MIMIR_GPU_SHUFFLER_HPP
#define MIMIR_GPU_SHUFFLER_HPP

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "mimir_gpu_shuffle.h"

namespace mimir_gpu {

// ---- interface mirror (src/interface.h) -------------------------------------
enum DB_POS { DB_START, DB_END };

class BaseObject {
  public:
    virtual ~BaseObject() {}
    virtual int open() = 0;
    virtual void close() = 0;
    virtual int seek(DB_POS pos) = 0;
    virtual uint64_t get_record_count() = 0;
};

template <typename KeyType, typename ValType>
class Writable : virtual public BaseObject {
  public:
    virtual ~Writable() {}
    virtual int write(KeyType *, ValType *) = 0;
};

// ---- transport ------------------------------------------------------------------
// The two collectives of exchange_kv (src/collectiveshuffler.h:606-607,
// :643-645) plus the job-wide abort of LOG_ERROR.
class Exchange {
  public:
    virtual ~Exchange() {}
    virtual int rank() const = 0;
    virtual int size() const = 0;
    virtual void alltoall_int(const int *send, int *recv) = 0;
    virtual void alltoallv_bytes(const char *sendbuf, const int *scount, const int *sdispl,
                                 char *recvbuf, const int *rcount, const int *rdispl) = 0;
    // The same exchange counted in units of `unit_bytes` bytes (a power of two): counts and
    // displacements are numbers of units, as with the contiguous type of 2^type_log_bytes
    // bytes the reference commits when the send buffer exceeds 1 GiB
    // (src/collectiveshuffler.h:46-53, :99-100, :630-645).  An MPI transport overrides this
    // with that type; the default turns units into bytes, which fit an int for every buffer
    // size the shufflers accept (up to 2 GiB).
    virtual void alltoallv_units(const char *sendbuf, const int *scount, const int *sdispl, char *recvbuf,
                                 const int *rcount, const int *rdispl, int unit_bytes)
    {
        if (unit_bytes == 1) {
            alltoallv_bytes(sendbuf, scount, sdispl, recvbuf, rcount, rdispl);
            return;
        }
        std::vector<int> b;
        units_to_bytes(scount, sdispl, rcount, rdispl, unit_bytes, b);
        const size_t n = (size_t) size();
        alltoallv_bytes(sendbuf, &b[0], &b[n], recvbuf, &b[2 * n], &b[3 * n]);
    }
    // The payload exchange on DEVICE memory (RCCL grouped send/recv, GPU-aware MPI, peer
    // copies over xGMI): same arguments as alltoallv_bytes, both buffers device pointers,
    // displacements 64-bit.  Blocking like MPI_Alltoallv: on return the receive image is
    // complete and the send image may be overwritten.  Returns false -- having moved nothing
    // -- when the transport cannot move device memory (the default); the answer must be the
    // same on every rank.  The shuffler then exchanges through host memory.
    virtual bool alltoallv_device(const void *d_sendbuf, const int *scount, const int64_t *sdispl,
                                  void *d_recvbuf, const int *rcount, const int64_t *rdispl)
    {
        (void) d_sendbuf; (void) scount; (void) sdispl; (void) d_recvbuf; (void) rcount; (void) rdispl;
        return false;
    }
    // Does alltoallv_device move device memory?  Asked before anything is allocated for it:
    // a transport that answers false (the default, and what an MPI library that is not
    // GPU-aware answers) never sees alltoallv_device called, and the shuffler allocates no
    // receive image on the device and no page-locked mirror of it.  A transport that
    // overrides alltoallv_device overrides this too.
    virtual bool moves_device_memory() const { return false; }
    // Non-blocking forms for NBGpuShuffler (MPI_Ialltoall / MPI_Ialltoallv /
    // MPI_Test / MPI_Wait; the counts and the payloads travel on two separate
    // communicators in the reference, src/nbcollectiveshuffler.h:68-69).  A
    // request is an opaque pointer, NULL once complete.  The defaults complete
    // at once through the blocking calls, which keeps the order of collective
    // calls the same on every rank.
    typedef void *Request;
    virtual Request ialltoall_int(const int *send, int *recv)
    {
        alltoall_int(send, recv);
        return NULL;
    }
    virtual Request ialltoallv_bytes(const char *sendbuf, const int *scount, const int *sdispl,
                                     char *recvbuf, const int *rcount, const int *rdispl)
    {
        alltoallv_bytes(sendbuf, scount, sdispl, recvbuf, rcount, rdispl);
        return NULL;
    }
    // Without an override a unit-counted non-blocking exchange completes at once through the
    // blocking call (same order of collective calls on every rank).
    virtual Request ialltoallv_units(const char *sendbuf, const int *scount, const int *sdispl, char *recvbuf,
                                     const int *rcount, const int *rdispl, int unit_bytes)
    {
        if (unit_bytes == 1) return ialltoallv_bytes(sendbuf, scount, sdispl, recvbuf, rcount, rdispl);
        alltoallv_units(sendbuf, scount, sdispl, recvbuf, rcount, rdispl, unit_bytes);
        return NULL;
    }
    virtual bool test(Request &req) { return req == NULL; } // true: complete, req released and NULL
    virtual void wait(Request &req)
    {
        while (!test(req)) {}
    }
    virtual void abort(int code) { std::exit(code); }

    // counts and displacements in units -> the same in bytes: [scount | sdispl | rcount | rdispl]
    void units_to_bytes(const int *scount, const int *sdispl, const int *rcount, const int *rdispl, int unit_bytes,
                        std::vector<int> &out) const
    {
        const size_t n = (size_t) size();
        out.resize(4 * n);
        const int *src[4] = {scount, sdispl, rcount, rdispl};
        for (int a = 0; a < 4; a++)
            for (size_t i = 0; i < n; i++) {
                const int64_t v = (int64_t) src[a][i] * unit_bytes;
                if (v > 0x7fffffffLL) { // cannot happen below 2 GiB of send buffer
                    fprintf(stderr, "Exchange: %lld bytes do not fit an int count\n", (long long) v);
                    std::abort();
                }
                out[(size_t) a * n + i] = (int) v;
            }
    }
};

// ---- wire form of one typed object (bytestream<T>, src/serializer.h:40-223) ------
template <typename T, typename Enable = void>
struct wire {
    static const int kind = MGS_KEY_FIXED; // == MGS_VAL_FIXED
    static const bool is_void = false;
    static int size(T *, int count) { return (int) sizeof(T) * count; }
    // contiguous bytes of `count` objects; `scratch` unused
    static const char *bytes(T *obj, int, std::vector<char> &) { return reinterpret_cast<const char *>(obj); }
    // rebuild the typed pointer from wire bytes; returns bytes consumed
    static int from(char *buf, int count, T *&out, std::vector<T> &) {
        out = reinterpret_cast<T *>(buf);
        return (int) sizeof(T) * count;
    }
    // the typed pointer of an object known to start at buf (nothing is measured)
    static void at(char *buf, T *&out, std::vector<T> &) { out = reinterpret_cast<T *>(buf); }
};

template <typename T>
struct wire<T, typename std::enable_if<std::is_same<T, char *>::value
                                       || std::is_same<T, const char *>::value>::type> {
    static const int kind = MGS_KEY_CSTR;
    static const bool is_void = false;
    static int size(T *obj, int count) {
        int n = 0;
        for (int i = 0; i < count; i++) n += (int) strlen(obj[i]) + 1;
        return n;
    }
    static const char *bytes(T *obj, int count, std::vector<char> &scratch) {
        if (count == 1) return obj[0];
        scratch.clear();
        for (int i = 0; i < count; i++) scratch.insert(scratch.end(), obj[i], obj[i] + strlen(obj[i]) + 1);
        return scratch.data();
    }
    static int from(char *buf, int count, T *&out, std::vector<T> &ptrs) {
        ptrs.resize((size_t) count);
        int n = 0;
        for (int i = 0; i < count; i++) {
            ptrs[(size_t) i] = buf + n;
            n += (int) strlen(buf + n) + 1;
        }
        out = ptrs.data();
        return n;
    }
    // one string known to start at buf (nothing is measured)
    static void at(char *buf, T *&out, std::vector<T> &ptrs) {
        ptrs.resize(1);
        ptrs[0] = buf;
        out = ptrs.data();
    }
};

struct VoidSlot {};
template <>
struct wire<void, void> {
    static const int kind = MGS_VAL_VOID;
    static const bool is_void = true;
    static int size(void *, int) { return 0; }
    static const char *bytes(void *, int, std::vector<char> &) { return NULL; }
};

template <typename T>
struct ptr_store {
    typedef std::vector<T> type;
};
template <>
struct ptr_store<void> {
    typedef std::vector<char> type;
};

// ---- the shuffler ----------------------------------------------------------------
template <typename KeyType, typename ValType>
class GpuShuffler : public Writable<KeyType, ValType> {
  public:
    // Same parameter list as CollectiveShuffler (src/collectiveshuffler.h:25-31)
    // with the communicator replaced by an Exchange; `h` (HashBucket) is unused
    // because load balancing is not supported.  combine_op >= 0 (mgs_combine_op)
    // makes it the CombineCollectiveShuffler (src/combinecollectiveshuffler.h) for
    // a combiner that is one of the declared operations instead of a callback.
    GpuShuffler(Exchange *comm, Writable<KeyType, ValType> *out,
                int (*user_hash)(KeyType *key, ValType *val, int npartition), int keycount,
                int valcount, bool split_hint = false, void *h = NULL,
                int64_t comm_buf_size = 0, int device = 0, int combine_op = -1)
        : comm(comm), out(out), keycount(keycount), valcount(valcount), device(device),
          combine_op(combine_op)
    {
        (void) split_hint;
        (void) h;
        shuffle_rank = comm->rank();
        shuffle_size = comm->size();
        if (out == NULL) fail("Output shuffler cannot be NULL!\n"); // src/baseshuffler.h:35
        if (user_hash != NULL)
            fail("GpuShuffler: a user hash callback runs on the host per record; "
                 "use CollectiveShuffler for this map\n");
        this->comm_buf_size = comm_buf_size > 0 ? comm_buf_size : env_size("MIMIR_COMM_SIZE", 64 << 20);
        int unit = 0, tlb = 0;
        if (mgs_geometry(this->comm_buf_size, shuffle_size, &buf_size, &unit, &tlb) != MGS_OK)
            fail("The communication buffer is too small!\n"); // src/collectiveshuffler.h:38-41
        // Above 1 GiB the reference counts its all-to-all-v in units of 2^type_log_bytes bytes
        // (src/collectiveshuffler.h:46-53).  Units of two bytes (up to 2 GiB) are taken here; with
        // larger ones the reference's own exchange no longer works -- `i * (int) buf_size`
        // overflows (:635) and save_data steps over `recv_count & (unit - 1)` bytes of padding
        // (:582), which is the padding only for units of two -- so there is nothing to match.
        if (tlb > 1) fail("GpuShuffler: MIMIR_COMM_SIZE above 2 GiB is not supported\n");
        type_log_bytes = tlb;
        handle = NULL;
        stager = NULL;
        memset(&win, 0, sizeof(win));
        shuffle_times = 0;
        done_flag = done_count = 0;
        kvcount = 0;
    }

    virtual ~GpuShuffler() { release(); }

    virtual int open()
    {
        mgs_desc d;
        memset(&d, 0, sizeof(d));
        d.nranks = shuffle_size;
        d.rank = shuffle_rank;
        d.layout.key_kind = wire<KeyType>::kind;
        d.layout.key_bytes = (wire<KeyType>::kind == MGS_KEY_FIXED) ? wire<KeyType>::size(NULL, keycount) : 0;
        d.layout.key_count = keycount;
        d.layout.val_kind = wire<ValType>::is_void ? MGS_VAL_VOID
                            : (wire<ValType>::kind == MGS_KEY_CSTR ? MGS_VAL_CSTR : MGS_VAL_FIXED);
        d.layout.val_bytes = (d.layout.val_kind == MGS_VAL_FIXED) ? wire<ValType>::size(NULL, valcount) : 0;
        d.layout.val_count = valcount;
        d.buf_size = buf_size;
        d.device = device;
        int64_t stage = env_size("MIMIR_GPU_STAGE_SIZE", this->comm_buf_size);
        // with a combiner the current round stays staged: room for a send image of combined
        // records plus one of new ones
        if (combine_op >= 0 && stage < 2 * buf_size * shuffle_size) stage = 2 * buf_size * shuffle_size;
        d.max_records = (uint64_t) stage; // one byte per record is the densest possible stream
        if (mgs_create(&d, &handle) != MGS_OK) fail(mgs_last_error(NULL));
        if (mgs_stager_create(handle, stage, &stager) != MGS_OK) fail(mgs_stager_last_error(NULL));
        // CombineCollectiveShuffler with the user's callback declared as an operation
        if (combine_op >= 0 && mgs_stager_set_combine(stager, combine_op) != MGS_OK)
            fail(mgs_stager_last_error(stager));
        recv_buffer.resize((size_t) buf_size * (size_t) shuffle_size);
        send_count.assign((size_t) shuffle_size, 0);
        recv_count.assign((size_t) shuffle_size, 0);
        d_s_displs.assign((size_t) shuffle_size, 0);
        d_r_displs.assign((size_t) shuffle_size, 0);
        send_bytes = recv_bytes = 0;
        device_exchanges = 0;
        a2a_s_count.assign((size_t) shuffle_size, 0);
        a2a_r_count.assign((size_t) shuffle_size, 0);
        a2a_s_displs.assign((size_t) shuffle_size, 0);
        a2a_r_displs.assign((size_t) shuffle_size, 0);
        done_flag = done_count = 0;
        sync_window();
        return true;
    }

    // kv_to_bytes into the staging buffer (src/collectiveshuffler.h:143-146): the two copies
    // are made right here, in the staging window the stager handed out; the library is only
    // called when the window is full.
    virtual int write(KeyType *key, ValType *val)
    {
        const int klen = wire<KeyType>::size(key, keycount);
        const int vlen = wire<ValType>::size(val, valcount);
        const char *kb = wire<KeyType>::bytes(key, keycount, key_scratch);
        const char *vb = wire<ValType>::bytes(val, valcount, val_scratch);
        const int64_t len = (int64_t) klen + vlen;
        if (win.writable && len > 0 && win.nrec < win.max_rec && win.used + len <= win.cap) {
            memcpy(win.bytes + win.used, kb, (size_t) klen);
            if (vlen) memcpy(win.bytes + win.used + klen, vb, (size_t) vlen);
            win.used += len;
            win.off[++win.nrec] = (uint32_t) win.used;
            return true;
        }
        commit_window();
        int rc = mgs_stager_put(stager, kb, (uint32_t) klen, vb, (uint32_t) vlen);
        if (rc == 1) {
            drain();
            rc = mgs_stager_put(stager, kb, (uint32_t) klen, vb, (uint32_t) vlen);
        }
        if (rc != MGS_OK) fail(mgs_stager_last_error(stager));
        sync_window();
        return true;
    }

    virtual void close()
    {
        wait();
        release();
    }

    // exchange_kv() for the blocking shuffler (src/collectiveshuffler.h:168)
    virtual void make_progress(bool issue_new = false)
    {
        (void) issue_new;
        drain();
        exchange_kv();
    }

    virtual int seek(DB_POS) { return false; }
    virtual uint64_t get_record_count() { return kvcount; } // src/baseshuffler.h:265

    int get_shuffle_times() const { return shuffle_times; }
    // What the reference's profiler counts in exchange_kv (src/stat.h:51-91, COUNTER_SEND_BYTES /
    // COUNTER_RECV_BYTES / COUNTER_SHUFFLE_TIMES; src/collectiveshuffler.h:593,652,661): payload
    // bytes handed to / received from the all-to-all-v, and the exchanges that moved data.
    uint64_t get_send_bytes() const { return send_bytes; }
    uint64_t get_recv_bytes() const { return recv_bytes; }
    // exchanges whose payload travelled between device images (Exchange::alltoallv_device)
    int get_device_exchanges() const { return device_exchanges; }

  protected:
    // Pack everything staged; a full partition ends the round
    // (src/collectiveshuffler.h:148-160).
    void commit_window()
    {
        if (win.writable && mgs_stager_commit(stager, win.used, win.nrec) != MGS_OK)
            fail(mgs_stager_last_error(stager));
        win.writable = 0;
    }
    void sync_window()
    {
        if (mgs_stager_window(stager, &win) != MGS_OK) fail(mgs_stager_last_error(stager));
    }

    void drain()
    {
        commit_window();
        for (;;) {
            mgs_round r;
            const void *local = NULL;
            int drained = 0;
            if (mgs_stager_pack_begin(stager) != MGS_OK) fail(mgs_stager_last_error(stager));
            while_packing(); // the GPU packs; the host may look after its exchanges
            if (mgs_stager_pack_end(stager, &r, &local, &drained) != MGS_OK) fail(mgs_stager_last_error(stager));
            kvcount += r.packed_records;
            if (r.local_bytes > 0) { // out->write at emit time, :135-141
                begin_source(shuffle_rank);
                deliver(const_cast<char *>(static_cast<const char *>(local)), r.local_bytes);
            }
            progress();
            if (drained) break;
            exchange_kv();
        }
        sync_window();
    }

    // Called before the records of one source rank are handed to `out` (the own
    // rank for records that never leave it).  For tests and tracing.
    virtual void begin_source(int src) { (void) src; }
    // Called after every pack; the non-blocking shuffler pushes its exchanges here.
    virtual void progress() {}
    // Called while a pack runs on the GPU (no stager call is allowed in here).
    virtual void while_packing() {}

    // src/collectiveshuffler.h:587-674
    virtual void exchange_kv()
    {
        void *d_image = NULL;
        const int64_t *send_offset = NULL;
        if (mgs_stager_send_image_device(stager, &d_image, &send_offset) != MGS_OK)
            fail(mgs_stager_last_error(stager));
        for (int i = 0; i < shuffle_size; i++) send_count[(size_t) i] = done_flag ? -1 : (int) send_offset[i];
        comm->alltoall_int(send_count.data(), recv_count.data());
        done_count = 0;
        for (int i = 0; i < shuffle_size; i++) {
            if (recv_count[(size_t) i] == -1) {
                done_count++;
                recv_count[(size_t) i] = 0;
            }
            if (send_count[(size_t) i] == -1) send_count[(size_t) i] = 0;
        }
        if (done_count >= shuffle_size) return;
        // counts and displacements of the host exchange in units (:630-639); the device exchange
        // takes bytes and leaves no padding
        const int tb = type_log_bytes, unit = 1 << tb;
        int64_t recvcount = 0, recvunits = 0;
        for (int i = 0; i < shuffle_size; i++) {
            a2a_s_count[(size_t) i] = (send_count[(size_t) i] + unit - 1) >> tb;
            a2a_r_count[(size_t) i] = (recv_count[(size_t) i] + unit - 1) >> tb;
            a2a_s_displs[(size_t) i] = (int) (((int64_t) i * buf_size) >> tb);
            a2a_r_displs[(size_t) i] = (int) recvunits;
            recvunits += a2a_r_count[(size_t) i];
            d_s_displs[(size_t) i] = (int64_t) i * buf_size;
            d_r_displs[(size_t) i] = recvcount;
            recvcount += recv_count[(size_t) i];
            send_bytes += (uint64_t) send_count[(size_t) i];
        }
        recv_bytes += (uint64_t) recvcount;
        // the payload: between the device images when the transport can, else through host memory
        void *d_recv = NULL;
        if (comm->moves_device_memory() && mgs_stager_recv_image_device(stager, &d_recv) != MGS_OK)
            fail(mgs_stager_last_error(stager));
        if (d_recv != NULL
            && comm->alltoallv_device(d_image, send_count.data(), d_s_displs.data(), d_recv, recv_count.data(),
                                      d_r_displs.data())) {
            device_exchanges += 1;
            const void *host = NULL;
            const uint32_t *off = NULL;
            uint64_t nrec = 0;
            if (mgs_stager_recv_to_host(stager, recvcount, &host, &off, &nrec) != MGS_OK)
                fail(mgs_stager_last_error(stager));
            char *bytes = const_cast<char *>(static_cast<const char *>(host));
            uint64_t rec = 0;
            for (int i = 0; i < shuffle_size; i++) { // save_data, :563-585
                if (recv_count[(size_t) i] <= 0) continue;
                begin_source(i);
                const int64_t end = d_r_displs[(size_t) i] + recv_count[(size_t) i];
                if (off == NULL) {
                    deliver(bytes + d_r_displs[(size_t) i], recv_count[(size_t) i]);
                    continue;
                }
                // the record boundaries came from the device: no length is measured here
                while (rec < nrec && (int64_t) off[rec] < end) {
                    deliver_one(bytes + off[rec], (int) (off[rec + 1] - off[rec]));
                    rec++;
                }
            }
        }
        else {
            const void *image = NULL;
            if (mgs_stager_send_image_host(stager, &image, &send_offset) != MGS_OK)
                fail(mgs_stager_last_error(stager));
            comm->alltoallv_units(static_cast<const char *>(image), a2a_s_count.data(), a2a_s_displs.data(),
                                  recv_buffer.data(), a2a_r_count.data(), a2a_r_displs.data(), unit);
            for (int i = 0; i < shuffle_size; i++) { // save_data, :563-585: a source's records, then its padding
                if (recv_count[(size_t) i] > 0) {
                    begin_source(i);
                    deliver(recv_buffer.data() + ((int64_t) a2a_r_displs[(size_t) i] << tb), recv_count[(size_t) i]);
                }
            }
        }
        mgs_stager_round_done(stager);
        shuffle_times += 1;
    }

    // src/collectiveshuffler.h:528-561
    virtual void wait()
    {
        drain();
        while (sendcount() > 0 || shuffle_times == 0) exchange_kv();
        done_flag = 1;
        do {
            exchange_kv();
        } while (done_count < shuffle_size);
    }

    uint64_t sendcount()
    {
        void *d = NULL;
        const int64_t *so = NULL;
        mgs_stager_send_image_device(stager, &d, &so);
        uint64_t n = 0;
        for (int i = 0; i < shuffle_size; i++) n += (uint64_t) so[i];
        return n;
    }

    // Walk back-to-back wire records and hand each to `out`
    // (kv_from_bytes, src/serializer.h:376-403).
    void deliver(char *buf, int64_t nbytes)
    {
        int64_t pos = 0;
        while (pos < nbytes) {
            KeyType *key = NULL;
            pos += wire<KeyType>::from(buf + pos, keycount, key, key_ptrs);
            ValType *val = NULL;
            pos += val_from(buf + pos, val, std::integral_constant<bool, wire<ValType>::is_void>());
            out->write(key, val);
        }
    }

    // One record whose extent is known (layouts mgs_unpack_device parses: one key object, a
    // fixed-width or absent value): the key starts the record, the value ends it.
    void deliver_one(char *rec, int len)
    {
        KeyType *key = NULL;
        ValType *val = NULL;
        const int vlen = wire<ValType>::is_void ? 0 : wire<ValType>::size(NULL, valcount);
        wire<KeyType>::at(rec, key, key_ptrs);
        val_at(rec + len - vlen, val, std::integral_constant<bool, wire<ValType>::is_void>());
        out->write(key, val);
    }
    void val_at(char *, ValType *&val, std::true_type) { val = NULL; }
    void val_at(char *buf, ValType *&val, std::false_type) { wire<ValType>::at(buf, val, val_ptrs); }

    int val_from(char *, ValType *&val, std::true_type)
    {
        val = NULL;
        return 0;
    }
    int val_from(char *buf, ValType *&val, std::false_type)
    {
        return wire<ValType>::from(buf, valcount, val, val_ptrs);
    }

    void release()
    {
        if (stager) mgs_stager_destroy(stager);
        if (handle) mgs_destroy(handle);
        stager = NULL;
        handle = NULL;
    }

    // LOG_ERROR: "rank[size] file:line message" on stderr, then abort the job
    void fail(const char *msg)
    {
        fprintf(stderr, "%d[%d] %s: %s", shuffle_rank, shuffle_size, "GpuShuffler", msg);
        fflush(stderr);
        comm->abort(1);
        std::abort();
    }

    // size with the b/k/m/g suffixes Mimir's env parser accepts (src/mimir.cpp:75-131)
    static int64_t env_size(const char *name, int64_t dflt)
    {
        const char *v = getenv(name);
        if (!v || !*v) return dflt;
        char *end = NULL;
        double x = strtod(v, &end);
        int64_t mul = 1;
        if (end && (*end == 'k' || *end == 'K')) mul = 1024;
        else if (end && (*end == 'm' || *end == 'M')) mul = 1024 * 1024;
        else if (end && (*end == 'g' || *end == 'G')) mul = 1024LL * 1024 * 1024;
        return (int64_t)(x * (double) mul);
    }

    Exchange *comm;
    Writable<KeyType, ValType> *out;
    int keycount, valcount, device, combine_op;
    int shuffle_rank, shuffle_size, shuffle_times;
    int done_flag, done_count;
    uint64_t kvcount;
    int64_t comm_buf_size, buf_size;
    mgs_handle *handle;
    mgs_stager *stager;
    mgs_stage_window win; // the staging buffer, for the inlined put in write()
    std::vector<char> recv_buffer, key_scratch, val_scratch;
    int type_log_bytes; // the exchange counts in units of 1 << type_log_bytes bytes
    std::vector<int> send_count, recv_count, a2a_s_count, a2a_r_count, a2a_s_displs, a2a_r_displs;
    std::vector<int64_t> d_s_displs, d_r_displs;
    uint64_t send_bytes, recv_bytes;
    int device_exchanges;
    typename ptr_store<KeyType>::type key_ptrs;
    typename ptr_store<ValType>::type val_ptrs;
};

// ---- the non-blocking shuffler -------------------------------------------------
// NBCollectiveShuffler (src/nbcollectiveshuffler.h): the same partition image,
// but 2..5 message buffers, so that the exchange of a full round is in flight
// while the next round is packed.  Each exchange is a non-blocking all-to-all
// of the counts (with the -1 vote) followed, in token order, by a non-blocking
// all-to-all-v of the bytes (:222-387).  Records are packed on the device into
// the stager's image as for GpuShuffler; when a round is full its partitions
// are copied from the GPU straight into the current (page-locked) message
// buffer and the image is free again.  While the GPU packs a staged batch the
// exchanges in flight are pushed on (mgs_stager_pack_begin / _end).
template <typename KeyType, typename ValType>
class NBGpuShuffler : public GpuShuffler<KeyType, ValType> {
    typedef GpuShuffler<KeyType, ValType> Base;

  public:
    enum { MIN_SBUF_COUNT = 2, MAX_SBUF_COUNT = 5 }; // src/config.cpp:20-21

    NBGpuShuffler(Exchange *comm, Writable<KeyType, ValType> *out,
                  int (*user_hash)(KeyType *key, ValType *val, int npartition), int keycount,
                  int valcount, bool split_hint = false, void *h = NULL, int64_t comm_buf_size = 0,
                  int device = 0)
        : Base(comm, out, user_hash, keycount, valcount, split_hint, h, comm_buf_size, device)
    {
        cur_idx = 0;
        a2a_token = a2av_token = 0;
        pending_msg = 0;
    }

    virtual ~NBGpuShuffler()
    {
        free_buffers();
    }

    virtual int open() // :72-120
    {
        Base::open();
        free_buffers();
        cur_idx = 0;
        a2a_token = a2av_token = 0;
        pending_msg = 0;
        for (int i = 0; i < MIN_SBUF_COUNT; i++) insert_comm_buffer();
        return true;
    }

    virtual void make_progress(bool issue_new = false) // :185-189
    {
        this->drain();
        if (issue_new && pending_msg == 0) exchange_kv();
        push_kv_exchange();
    }

    int get_buffer_count() const { return (int) msg.size(); }

  protected:
    enum State { ShuffleMsgStart, ShuffleMsgPending, ShuffleMsgComplete }; // :22-34

    struct MsgBuf {
        char *send_buffer; // page-locked: the partitions are copied here straight from the GPU
        std::vector<char> recv_buffer;
        std::vector<int> send_offset, recv_count, s_count, r_count, s_displs, r_displs;
        uint64_t token;
        State state;
        bool counts_known;
        Exchange::Request a2a_req, a2av_req;
    };

    void insert_comm_buffer() // :426-449
    {
        MsgBuf *m = new MsgBuf();
        const size_t P = (size_t) this->shuffle_size;
        m->send_buffer = static_cast<char *>(mgs_host_alloc((size_t) this->buf_size * P));
        if (m->send_buffer == NULL) this->fail("NBGpuShuffler: no page-locked memory for a send buffer\n");
        m->recv_buffer.resize((size_t) this->buf_size * P);
        m->send_offset.assign(P, 0);
        m->recv_count.assign(P, 0);
        m->s_count.assign(P, 0);
        m->r_count.assign(P, 0);
        m->s_displs.assign(P, 0);
        m->r_displs.assign(P, 0);
        m->token = 0;
        m->state = ShuffleMsgComplete;
        m->counts_known = false;
        m->a2a_req = m->a2av_req = NULL;
        msg.push_back(m);
    }

    void free_buffers()
    {
        for (size_t i = 0; i < msg.size(); i++) {
            mgs_host_free(msg[i]->send_buffer);
            delete msg[i];
        }
        msg.clear();
    }

    virtual void progress() { push_kv_exchange(); } // after every record in the reference, :179
    virtual void while_packing() { push_kv_exchange(); }

    // A round is full (or the stream has ended): its partitions leave the stager's image
    // for the current message buffer and the exchange is started.
    virtual void exchange_kv()
    {
        snapshot();
        start_kv_exchange();
    }

    int64_t snapshot()
    {
        const int64_t *send_offset = NULL;
        MsgBuf *m = msg[(size_t) cur_idx];
        if (mgs_stager_send_image_to(this->stager, m->send_buffer, &send_offset) != MGS_OK)
            this->fail(mgs_stager_last_error(this->stager));
        int64_t total = 0;
        for (int i = 0; i < this->shuffle_size; i++) {
            m->send_offset[(size_t) i] = (int) send_offset[i];
            total += send_offset[i];
        }
        mgs_stager_round_done(this->stager);
        return total;
    }

    void start_kv_exchange() // :222-277
    {
        MsgBuf *m = msg[(size_t) cur_idx];
        const int P = this->shuffle_size;
        if (this->done_flag)
            for (int i = 0; i < P; i++) m->send_offset[(size_t) i] = -1;
        this->shuffle_times += 1;
        m->counts_known = false;
        m->a2a_req = this->comm->ialltoall_int(m->send_offset.data(), m->recv_count.data());
        m->state = ShuffleMsgStart;
        m->token = a2a_token++;
        pending_msg += 1;
        push_kv_exchange();
        // the next buffer to pack into: a complete one, else a new one, else wait for one
        cur_idx = -1;
        for (size_t i = 0; i < msg.size(); i++) {
            if (msg[i]->state == ShuffleMsgComplete) {
                cur_idx = (int) i;
                break;
            }
        }
        if (cur_idx == -1) {
            if ((int) msg.size() < MAX_SBUF_COUNT) {
                insert_comm_buffer();
                cur_idx = (int) msg.size() - 1;
            }
            else {
                cur_idx = 0;
                for (;;) {
                    push_kv_exchange();
                    if (msg[(size_t) cur_idx]->state == ShuffleMsgComplete) break;
                    cur_idx = (cur_idx + 1) % (int) msg.size();
                }
            }
        }
    }

    void push_kv_exchange() // :279-387
    {
        const int P = this->shuffle_size;
        for (size_t k = 0; k < msg.size(); k++) {
            MsgBuf *m = msg[k];
            if (m->state == ShuffleMsgComplete) continue;
            if (m->state == ShuffleMsgStart) {
                if (!m->counts_known && this->comm->test(m->a2a_req)) {
                    this->done_count = 0;
                    for (int i = 0; i < P; i++)
                        if (m->recv_count[(size_t) i] == -1) this->done_count++;
                    m->counts_known = true;
                }
                if (m->counts_known && this->done_count == P) { // everybody voted: nothing to move
                    m->state = ShuffleMsgComplete;
                    pending_msg -= 1;
                    reset(m);
                    continue;
                }
                if (m->counts_known && m->token == a2av_token && this->done_count < P) {
                    const int tb = this->type_log_bytes, unit = 1 << tb; // :322-335
                    int64_t recvcount = 0, recvunits = 0;
                    for (int i = 0; i < P; i++) {
                        if (m->send_offset[(size_t) i] == -1) m->send_offset[(size_t) i] = 0;
                        if (m->recv_count[(size_t) i] == -1) m->recv_count[(size_t) i] = 0;
                        m->s_count[(size_t) i] = (m->send_offset[(size_t) i] + unit - 1) >> tb;
                        m->r_count[(size_t) i] = (m->recv_count[(size_t) i] + unit - 1) >> tb;
                        m->s_displs[(size_t) i] = (int) (((int64_t) i * this->buf_size) >> tb);
                        m->r_displs[(size_t) i] = (int) recvunits;
                        recvunits += m->r_count[(size_t) i];
                        recvcount += m->recv_count[(size_t) i];
                        this->send_bytes += (uint64_t) m->send_offset[(size_t) i];
                    }
                    this->recv_bytes += (uint64_t) recvcount;
                    m->a2av_req = this->comm->ialltoallv_units(m->send_buffer, m->s_count.data(),
                                                               m->s_displs.data(), m->recv_buffer.data(),
                                                               m->r_count.data(), m->r_displs.data(), unit);
                    m->state = ShuffleMsgPending;
                    a2av_token += 1;
                }
            }
            // Rounds reach `out` in the order they were sent: an exchange that has completed
            // waits for the ones started before it.  (The reference hands them over in
            // buffer-index order, :362-383, so two rounds of one source can swap when both
            // complete between two polls; a deterministic order costs nothing here.)
            if (m->state == ShuffleMsgPending && oldest_pending(m) && this->comm->test(m->a2av_req)) {
                save_data(m);
                reset(m);
                m->state = ShuffleMsgComplete;
                pending_msg -= 1;
            }
        }
    }

    bool oldest_pending(const MsgBuf *m) const
    {
        for (size_t k = 0; k < msg.size(); k++)
            if (msg[k]->state == ShuffleMsgPending && msg[k]->token < m->token) return false;
        return true;
    }

    void save_data(MsgBuf *m) // :400-424
    {
        for (int i = 0; i < this->shuffle_size; i++) {
            if (m->recv_count[(size_t) i] > 0) {
                this->begin_source(i);
                this->deliver(m->recv_buffer.data() + ((int64_t) m->r_displs[(size_t) i] << this->type_log_bytes),
                              m->recv_count[(size_t) i]);
            }
        }
    }

    void reset(MsgBuf *m)
    {
        m->token = 0;
        m->counts_known = false;
        for (int i = 0; i < this->shuffle_size; i++) m->send_offset[(size_t) i] = 0;
    }

    void wait_all() // :396-399
    {
        while (pending_msg > 0) push_kv_exchange();
    }

    virtual void wait() // :192-220
    {
        this->drain();
        if (this->sendcount() > 0) exchange_kv();
        wait_all();
        this->done_flag = 1;
        do {
            start_kv_exchange();
            wait_all();
        } while (this->done_count < this->shuffle_size);
    }

    std::vector<MsgBuf *> msg;
    int cur_idx;
    uint64_t a2a_token, a2av_token;
    int pending_msg;
};

// ---- the shuffler for device-resident batches ------------------------------------
// CollectiveShuffler for a map whose emitted records already lie in device memory in wire
// form (what mgs_tokenize_device leaves, a generator kernel's output, a container kept in
// HBM): no record crosses the per-record write(key*, val*) boundary on the way in.
//
//     open();  write_device_records(d_blob, n, d_rec_off)  any number of times, in emit
//     order;  close()
//
// Every batch is packed with mgs_pack_streams_device into one stream per destination,
// going on where the batch before stopped, so the rounds are those CollectiveShuffler::write
// would cut over the whole sequence of records (src/collectiveshuffler.h:131-166).  Nothing
// travels before close(): the whole map's partitions stay in HBM (MIMIR_GPU_STREAM_SIZE
// bytes per destination, 1 GiB unless set).  close() tells every rank the send counts of all
// rounds at once (the counts all-to-all of :606-607, for every round), then moves the rounds
// a window at a time -- as many consecutive rounds as MIMIR_GPU_WINDOW_SIZE (256 MiB) holds
// -- with one all-to-all-v per window whose displacements point into the streams: between
// device buffers when the transport moves device memory (Exchange::alltoallv_device, 64-bit
// displacements), else through a host mirror of the window.  `out` sees what the
// reference's `out` sees, in its order: the local records of round k (emitted during it,
// :135-141), then round k's receive image source by source (:563-585), for k = 0, 1, ...
// On the device path the record boundaries come from mgs_unpack_device.  shuffle_times,
// kvcount and the byte counters are those of the reference's exchanges.
//
// write(key*, val*) is not served: a map that emits pair by pair uses GpuShuffler.
template <typename KeyType, typename ValType>
class StreamsGpuShuffler : public Writable<KeyType, ValType> {
  public:
    // CollectiveShuffler's parameter list (src/collectiveshuffler.h:25-31) with the
    // communicator replaced by an Exchange, then: bytes per destination stream (0: the
    // environment or 1 GiB), most records one pack call takes, most rounds one call reports.
    StreamsGpuShuffler(Exchange *comm, Writable<KeyType, ValType> *out,
                       int (*user_hash)(KeyType *key, ValType *val, int npartition), int keycount,
                       int valcount, bool split_hint = false, void *h = NULL, int64_t comm_buf_size = 0,
                       int device = 0, int64_t stream_bytes = 0, uint64_t max_records = 0,
                       uint32_t max_rounds = 4096)
        : comm(comm), out(out), keycount(keycount), valcount(valcount), device(device), max_rounds(max_rounds)
    {
        (void) split_hint;
        (void) h;
        shuffle_rank = comm->rank();
        shuffle_size = comm->size();
        handle = NULL;
        d_streams = d_local = NULL;
        if (out == NULL) fail("Output shuffler cannot be NULL!\n"); // src/baseshuffler.h:35
        if (user_hash != NULL)
            fail("StreamsGpuShuffler: a user hash callback runs on the host per record; "
                 "use CollectiveShuffler for this map\n");
        this->comm_buf_size = comm_buf_size > 0 ? comm_buf_size : env_size("MIMIR_COMM_SIZE", 64 << 20);
        int unit = 0, tlb = 0;
        if (mgs_geometry(this->comm_buf_size, shuffle_size, &buf_size, &unit, &tlb) != MGS_OK)
            fail("The communication buffer is too small!\n"); // src/collectiveshuffler.h:38-41
        if (tlb > 0) fail("StreamsGpuShuffler: MIMIR_COMM_SIZE above 1 GiB is not supported (use GpuShuffler)\n");
        if (stream_bytes <= 0) stream_bytes = env_size("MIMIR_GPU_STREAM_SIZE", (int64_t) 1 << 30);
        stream_cap = stream_bytes / 16 * 16;
        if (stream_cap < buf_size) fail("StreamsGpuShuffler: a destination stream must hold one partition\n");
        this->max_records = max_records ? max_records : ((uint64_t) 1 << 26);
        window_bytes = env_size("MIMIR_GPU_WINDOW_SIZE", 256 << 20);
        if (window_bytes > 0x7fffffffLL) window_bytes = 0x7fffffffLL; // counts and host displacements are ints
        if (this->max_rounds < 2) this->max_rounds = 2;
        reset_counters();
    }

    virtual ~StreamsGpuShuffler() { release(); }

    virtual int open()
    {
        mgs_desc d;
        shuffler_desc(d);
        if (mgs_create(&d, &handle) != MGS_OK) fail(mgs_last_error(NULL));
        const size_t P = (size_t) shuffle_size;
        d_streams = static_cast<char *>(mgs_device_alloc(device, P * (size_t) stream_cap));
        d_local = static_cast<char *>(mgs_device_alloc(device, (size_t) stream_cap));
        if (d_streams == NULL || d_local == NULL)
            fail("StreamsGpuShuffler: no device memory for the destination streams (MIMIR_GPU_STREAM_SIZE)\n");
        reset_counters();
        t_next.resize(max_rounds);
        t_lrec.resize(max_rounds);
        t_lbytes.resize(max_rounds);
        t_so.resize((size_t) max_rounds * P);
        t_base.resize((size_t) max_rounds * P);
        return true;
    }

    // The records [0, n) of a device-resident batch in wire form, emit order; d_rec_off (n + 1
    // offsets on the device) for variable-width layouts, NULL for fixed widths.  Blocking.
    int write_device_records(const void *d_blob, uint64_t n, const uint32_t *d_rec_off = NULL)
    {
        if (handle == NULL) fail("StreamsGpuShuffler: write_device_records before open()\n");
        const size_t P = (size_t) shuffle_size;
        uint64_t start = 0;
        while (start < n) {
            mgs_rounds r;
            const bool going_on = !r_so.empty();
            const int rc = mgs_pack_streams_device(handle, d_blob, d_rec_off, n, start, going_on ? fill.data() : NULL,
                                                   going_on ? open_base.data() : NULL, d_streams, stream_cap, max_rounds,
                                                   d_local + lused, stream_cap - lused, NULL, NULL, &r, t_next.data(),
                                                   t_so.data(), t_base.data(), t_lrec.data(), t_lbytes.data());
            if (rc != MGS_OK) fail(mgs_last_error(handle));
            const uint64_t consumed = r.next - start;
            if (consumed == 0 && r.n_rounds >= max_rounds) fail("StreamsGpuShuffler: max_rounds too small for this batch\n");
            if (r.next < n && r.n_rounds < max_rounds && consumed < this->max_records)
                fail("StreamsGpuShuffler: a destination stream is full (raise MIMIR_GPU_STREAM_SIZE)\n");
            // the first round of this call is the open round of the call before going on
            size_t first = 0;
            if (going_on) {
                const size_t last = r_lbytes.size() - 1;
                for (size_t b = 0; b < P; b++) r_so[last * P + b] = t_so[b];
                r_lbytes[last] += t_lbytes[0];
                first = 1;
            }
            for (size_t k = first; k < r.n_rounds; k++) {
                r_so.insert(r_so.end(), t_so.begin() + (ptrdiff_t)(k * P), t_so.begin() + (ptrdiff_t)((k + 1) * P));
                r_base.insert(r_base.end(), t_base.begin() + (ptrdiff_t)(k * P), t_base.begin() + (ptrdiff_t)((k + 1) * P));
                r_lbytes.push_back(t_lbytes[k]);
            }
            const size_t lastk = (size_t) r.n_rounds - 1;
            open_base.assign(t_base.begin() + (ptrdiff_t)(lastk * P), t_base.begin() + (ptrdiff_t)((lastk + 1) * P));
            fill.resize(P);
            for (size_t b = 0; b < P; b++) fill[b] = (int) b == shuffle_rank ? 0 : open_base[b] + t_so[lastk * P + b];
            lused += r.local_bytes;
            kvcount += r.packed_records;
            start = r.next;
        }
        return true;
    }

    virtual int write(KeyType *, ValType *)
    {
        fail("StreamsGpuShuffler: write(key, val) is not served; emit device batches with write_device_records, "
             "or use GpuShuffler\n");
        return false;
    }

    virtual void close()
    {
        if (handle != NULL) exchange_all();
        release();
    }

    virtual void make_progress(bool issue_new = false) { (void) issue_new; } // nothing travels before close()
    virtual int seek(DB_POS) { return false; }
    virtual uint64_t get_record_count() { return kvcount; } // src/baseshuffler.h:265

    int get_shuffle_times() const { return shuffle_times; }
    uint64_t get_send_bytes() const { return send_bytes; } // as GpuShuffler's
    uint64_t get_recv_bytes() const { return recv_bytes; }
    int get_device_exchanges() const { return device_exchanges; } // all-to-all-v calls between device buffers

  protected:
    virtual void begin_source(int src) { (void) src; }

    void shuffler_desc(mgs_desc &d) const
    {
        memset(&d, 0, sizeof(d));
        d.nranks = shuffle_size;
        d.rank = shuffle_rank;
        d.layout.key_kind = wire<KeyType>::kind;
        d.layout.key_bytes = (wire<KeyType>::kind == MGS_KEY_FIXED) ? wire<KeyType>::size(NULL, keycount) : 0;
        d.layout.key_count = keycount;
        d.layout.val_kind = wire<ValType>::is_void ? MGS_VAL_VOID
                            : (wire<ValType>::kind == MGS_KEY_CSTR ? MGS_VAL_CSTR : MGS_VAL_FIXED);
        d.layout.val_bytes = (d.layout.val_kind == MGS_VAL_FIXED) ? wire<ValType>::size(NULL, valcount) : 0;
        d.layout.val_count = valcount;
        d.buf_size = buf_size;
        d.device = device;
        d.max_records = max_records;
    }

    // Does mgs_unpack_device take this layout?  (fixed widths; one string key with a fixed value of
    // at most 62 bytes or none)  Shortest record of the layout in *minlen.
    bool device_parse(size_t *minlen) const
    {
        mgs_desc d;
        shuffler_desc(d);
        const mgs_layout &l = d.layout;
        if (l.key_kind == MGS_KEY_FIXED && l.val_kind != MGS_VAL_CSTR) {
            *minlen = (size_t) l.key_bytes + (size_t) l.val_bytes;
            return *minlen > 0;
        }
        *minlen = 1 + (l.val_kind == MGS_VAL_FIXED ? (size_t) l.val_bytes : 0);
        return l.key_kind == MGS_KEY_CSTR && l.key_count <= 1
               && (l.val_kind == MGS_VAL_VOID || (l.val_kind == MGS_VAL_FIXED && l.val_bytes <= 62));
    }

    // Every exchange of the map (src/collectiveshuffler.h:587-674, once per round there).
    void exchange_all()
    {
        const size_t P = (size_t) shuffle_size, me = (size_t) shuffle_rank;
        if (r_lbytes.empty()) { // nothing was written: one empty round, as wait() makes (:531-534)
            r_so.assign(P, 0);
            r_base.assign(P, 0);
            r_lbytes.assign(1, 0);
        }
        const size_t r_mine = r_lbytes.size();
        // how many rounds every rank has
        std::vector<int> a(P, (int) r_mine), r_of(P, 0);
        comm->alltoall_int(a.data(), r_of.data());
        size_t R = 0, tab_words = 0;
        for (size_t s = 0; s < P; s++) {
            if ((size_t) r_of[s] > R) R = (size_t) r_of[s];
            tab_words += (size_t) r_of[s];
        }
        // the send counts of all rounds: to destination d the column d of this rank's table
        std::vector<int> tab_s(P * r_mine), tab_r(tab_words ? tab_words : 1);
        std::vector<int> sc(P), sd(P), rcv(P), rd(P);
        size_t pos = 0;
        for (size_t d = 0; d < P; d++) {
            for (size_t k = 0; k < r_mine; k++) {
                tab_s[d * r_mine + k] = (int) r_so[k * P + d];
                send_bytes += (uint64_t) r_so[k * P + d];
            }
            sc[d] = (int) (r_mine * sizeof(int));
            sd[d] = (int) (d * r_mine * sizeof(int));
            rcv[d] = (int) ((size_t) r_of[d] * sizeof(int));
            rd[d] = (int) (pos * sizeof(int));
            pos += (size_t) r_of[d];
        }
        comm->alltoallv_bytes(reinterpret_cast<const char *>(tab_s.data()), sc.data(), sd.data(),
                              reinterpret_cast<char *>(tab_r.data()), rcv.data(), rd.data());
        std::vector<int64_t> rc(R * P, 0); // rc[k*P + s]: bytes of source s's round k for this rank
        for (size_t s = 0; s < P; s++)
            for (size_t k = 0; k < (size_t) r_of[s]; k++) {
                rc[k * P + s] = tab_r[(size_t) rd[s] / sizeof(int) + k];
                recv_bytes += (uint64_t) rc[k * P + s];
            }
        // the local records, all of them (they never leave this rank)
        std::vector<char> h_local((size_t) lused ? (size_t) lused : 1);
        if (mgs_copy_to_host(device, h_local.data(), d_local, (size_t) lused) != MGS_OK)
            fail("StreamsGpuShuffler: copy of the local records failed\n");
        // windows of W consecutive rounds
        const int64_t round_bytes = buf_size * (int64_t) P;
        const size_t W = (size_t) (window_bytes / round_bytes > 0 ? window_bytes / round_bytes : 1);
        int64_t win_max = 0;
        for (size_t k0 = 0; k0 < R; k0 += W) {
            int64_t tot_r = 0, tot_s = 0;
            for (size_t k = k0; k < R && k < k0 + W; k++)
                for (size_t b = 0; b < P; b++) {
                    tot_r += rc[k * P + b];
                    if (k < r_mine) tot_s += r_so[k * P + b];
                }
            if (tot_r > win_max) win_max = tot_r;
            if (tot_s > win_max) win_max = tot_s;
        }
        const bool on_device = comm->moves_device_memory();
        size_t minlen = 1;
        const bool parse = on_device && device_parse(&minlen);
        const size_t win_cap = (size_t) (win_max > 0 ? win_max : 1);
        char *h_recv = static_cast<char *>(mgs_host_alloc(win_cap));
        char *h_send = on_device ? NULL : static_cast<char *>(mgs_host_alloc(win_cap));
        char *d_recv = on_device ? static_cast<char *>(mgs_device_alloc(device, win_cap)) : NULL;
        const uint64_t off_cap = parse ? (uint64_t) win_cap / minlen + 2 : 0;
        uint32_t *d_off = parse ? static_cast<uint32_t *>(mgs_device_alloc(device, (size_t) off_cap * 4)) : NULL;
        uint32_t *h_off = parse ? static_cast<uint32_t *>(mgs_host_alloc((size_t) off_cap * 4)) : NULL;
        if (h_recv == NULL || (!on_device && h_send == NULL) || (on_device && d_recv == NULL)
            || (parse && (d_off == NULL || h_off == NULL)))
            fail("StreamsGpuShuffler: no memory for an exchange window (MIMIR_GPU_WINDOW_SIZE)\n");
        std::vector<int64_t> sd64(P), rd64(P);
        std::vector<uint64_t> cursor(P, 0);
        int64_t lpos = 0;
        for (size_t k0 = 0; k0 < R; k0 += W) {
            const size_t k1 = k0 + W < R ? k0 + W : R;
            int64_t tot_r = 0, tot_s = 0;
            for (size_t b = 0; b < P; b++) {
                int64_t s_b = 0, r_b = 0;
                for (size_t k = k0; k < k1; k++) {
                    r_b += rc[k * P + b];
                    if (k < r_mine) s_b += r_so[k * P + b];
                }
                sc[b] = (int) s_b;
                rcv[b] = (int) r_b;
                sd[b] = (int) tot_s;
                rd[b] = (int) tot_r;
                rd64[b] = tot_r;
                // the rounds of a window lie back to back in a destination's stream
                sd64[b] = (int64_t) b * stream_cap + (k0 < r_mine ? r_base[k0 * P + b] : 0);
                tot_s += s_b;
                tot_r += r_b;
            }
            uint64_t nrec = 0;
            if (on_device) {
                if (!comm->alltoallv_device(d_streams, sc.data(), sd64.data(), d_recv, rcv.data(), rd64.data()))
                    fail("StreamsGpuShuffler: the transport said it moves device memory, then moved nothing\n");
                device_exchanges += 1;
                if (parse && tot_r > 0) {
                    mgs_page_cut cut;
                    int ncuts = 0;
                    if (mgs_unpack_device(handle, d_recv, (uint64_t) tot_r, d_off, off_cap, (int64_t) 1 << 40, 0, &nrec,
                                          &cut, 1, &ncuts, NULL) != MGS_OK)
                        fail(mgs_last_error(handle));
                    if (mgs_copy_to_host(device, h_off, d_off, (size_t) (nrec + 1) * 4) != MGS_OK)
                        fail("StreamsGpuShuffler: copy of the record offsets failed\n");
                }
                if (mgs_copy_to_host(device, h_recv, d_recv, (size_t) tot_r) != MGS_OK)
                    fail("StreamsGpuShuffler: copy of a receive window failed\n");
            }
            else {
                for (size_t b = 0; b < P; b++)
                    if (mgs_copy_to_host(device, h_send + sd[b], d_streams + sd64[b], (size_t) sc[b]) != MGS_OK)
                        fail("StreamsGpuShuffler: copy of a send window failed\n");
                comm->alltoallv_bytes(h_send, sc.data(), sd.data(), h_recv, rcv.data(), rd.data());
            }
            // where every source's records begin in the offsets table
            if (parse && tot_r > 0)
                for (size_t b = 0; b < P; b++) {
                    uint64_t lo = 0, hi = nrec;
                    while (lo < hi) {
                        const uint64_t mid = (lo + hi) / 2;
                        if ((int64_t) h_off[mid] < rd64[b]) lo = mid + 1;
                        else hi = mid;
                    }
                    cursor[b] = lo;
                }
            std::vector<int64_t> at(rd64); // per source: the next byte of its slice to hand over
            for (size_t k = k0; k < k1; k++) {
                if (k < r_mine && r_lbytes[k] > 0) { // out->write at emit time, :135-141
                    begin_source(shuffle_rank);
                    deliver(h_local.data() + lpos, r_lbytes[k]);
                    lpos += r_lbytes[k];
                }
                for (size_t b = 0; b < P; b++) { // save_data, :563-585
                    const int64_t len = rc[k * P + b];
                    if (b == me || len <= 0) continue;
                    begin_source((int) b);
                    const int64_t end = at[b] + len;
                    if (parse) { // the record boundaries came from the device: no length is measured here
                        uint64_t &c = cursor[b];
                        while (c < nrec && (int64_t) h_off[c] < end) {
                            deliver_one(h_recv + h_off[c], (int) (h_off[c + 1] - h_off[c]));
                            c++;
                        }
                    }
                    else
                        deliver(h_recv + at[b], len);
                    at[b] = end;
                }
            }
        }
        shuffle_times = (int) R;
        mgs_host_free(h_recv);
        mgs_host_free(h_send);
        mgs_host_free(h_off);
        mgs_device_free(device, d_recv);
        mgs_device_free(device, d_off);
    }

    // kv_from_bytes over back-to-back records (src/serializer.h:376-403)
    void deliver(char *buf, int64_t nbytes)
    {
        int64_t pos = 0;
        while (pos < nbytes) {
            KeyType *key = NULL;
            pos += wire<KeyType>::from(buf + pos, keycount, key, key_ptrs);
            ValType *val = NULL;
            pos += val_from(buf + pos, val, std::integral_constant<bool, wire<ValType>::is_void>());
            out->write(key, val);
        }
    }
    // one record whose extent is known: the key starts it, the value ends it
    void deliver_one(char *rec, int len)
    {
        KeyType *key = NULL;
        ValType *val = NULL;
        const int vlen = wire<ValType>::is_void ? 0 : wire<ValType>::size(NULL, valcount);
        wire<KeyType>::at(rec, key, key_ptrs);
        val_at(rec + len - vlen, val, std::integral_constant<bool, wire<ValType>::is_void>());
        out->write(key, val);
    }
    void val_at(char *, ValType *&val, std::true_type) { val = NULL; }
    void val_at(char *buf, ValType *&val, std::false_type) { wire<ValType>::at(buf, val, val_ptrs); }
    int val_from(char *, ValType *&val, std::true_type)
    {
        val = NULL;
        return 0;
    }
    int val_from(char *buf, ValType *&val, std::false_type) { return wire<ValType>::from(buf, valcount, val, val_ptrs); }

    void reset_counters()
    {
        shuffle_times = device_exchanges = 0;
        kvcount = send_bytes = recv_bytes = 0;
        lused = 0;
        r_so.clear();
        r_base.clear();
        r_lbytes.clear();
        fill.clear();
        open_base.clear();
    }

    void release()
    {
        if (d_streams) mgs_device_free(device, d_streams);
        if (d_local) mgs_device_free(device, d_local);
        if (handle) mgs_destroy(handle);
        d_streams = d_local = NULL;
        handle = NULL;
    }

    void fail(const char *msg)
    {
        fprintf(stderr, "%d[%d] %s: %s", shuffle_rank, shuffle_size, "StreamsGpuShuffler", msg);
        fflush(stderr);
        comm->abort(1);
        std::abort();
    }

    static int64_t env_size(const char *name, int64_t dflt)
    {
        const char *v = getenv(name);
        if (!v || !*v) return dflt;
        char *end = NULL;
        double x = strtod(v, &end);
        int64_t mul = 1;
        if (end && (*end == 'k' || *end == 'K')) mul = 1024;
        else if (end && (*end == 'm' || *end == 'M')) mul = 1024 * 1024;
        else if (end && (*end == 'g' || *end == 'G')) mul = 1024LL * 1024 * 1024;
        return (int64_t)(x * (double) mul);
    }

    Exchange *comm;
    Writable<KeyType, ValType> *out;
    int keycount, valcount, device;
    uint32_t max_rounds;
    uint64_t max_records;
    int shuffle_rank, shuffle_size, shuffle_times, device_exchanges;
    uint64_t kvcount, send_bytes, recv_bytes;
    int64_t comm_buf_size, buf_size, stream_cap, window_bytes, lused;
    mgs_handle *handle;
    char *d_streams, *d_local;
    // the rounds of the map so far: send counts and stream offsets [round][destination],
    // bytes of local records per round; the open round's fill and base per destination
    std::vector<int64_t> r_so, r_base, r_lbytes, fill, open_base;
    // one call's result tables
    std::vector<uint64_t> t_next, t_lrec;
    std::vector<int64_t> t_so, t_base, t_lbytes;
    typename ptr_store<KeyType>::type key_ptrs;
    typename ptr_store<ValType>::type val_ptrs;
};

} // namespace mimir_gpu
#endif
