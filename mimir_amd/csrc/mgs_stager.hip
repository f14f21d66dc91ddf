// Host-staged front end of the pack path: what a Writable::write() shim needs.
//
// The reference copies every pair into the send buffer inside write()
// (kv_to_bytes, src/collectiveshuffler.h:143-146) because the caller's
// pointers are only valid during the call.  Here the same copy goes into a
// pinned host staging buffer, in emit order and in wire form; a pack call
// moves the staged batch to the GPU, runs hash -> destination -> stable pack
// into the device send image and brings back the local-bypass records.
// Built on the public C ABI only (mgs_pack_device + HIP memory calls).
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "mimir_gpu_shuffle.h"

struct mgs_stager {
    mgs_handle *h = nullptr;
    mgs_desc desc;
    int64_t cap = 0;           // staging bytes
    uint64_t max_rec = 0;      // staging records
    bool fixed = false;
    uint8_t *h_stage = nullptr; // pinned: wire-form records
    uint32_t *h_off = nullptr;  // pinned: record offsets (variable-width layouts)
    uint8_t *d_stage = nullptr;
    uint32_t *d_off = nullptr;
    uint8_t *d_image = nullptr; // nranks * buf_size
    uint8_t *d_local = nullptr; // local-bypass records of one pack call
    uint8_t *h_image = nullptr; // pinned mirror of the send image (host exchange)
    uint8_t *h_local = nullptr; // pinned
    // receive side of an exchange that moves device memory (made on first use)
    uint8_t *d_recv = nullptr;  // nranks * buf_size: source 0's bytes, then source 1's, ...
    uint8_t *h_recv = nullptr;  // pinned mirror
    uint32_t *d_roff = nullptr; // record offsets found by mgs_unpack_device
    uint32_t *h_roff = nullptr; // pinned mirror
    uint64_t roff_cap = 0;
    int64_t used = 0;
    uint64_t nrec = 0;
    uint64_t cursor = 0; // first staged record not packed yet
    bool uploaded = false;
    bool in_flight = false; // mgs_stager_pack_begin without its mgs_stager_pack_end
    int64_t send_offset[MGS_MAX_RANKS];
    // combiner mode (mgs_stager_set_combine): the current round's records stay staged
    int combine_op = -1;
    int64_t loc_done = 0;        // bytes of the round's local stream already handed out
    uint64_t round_keys = 0;     // new keys of the round reported so far
    bool cut_pending = false;    // the last pack ended a round: what is staged belongs to the next
    bool want_refill = false;    // a put found no room
    hipStream_t stream = nullptr;
    std::string err;
};

namespace {
thread_local std::string g_stager_error;

std::string sfmt(const char *f, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return std::string(buf);
}

int fixed_len(const mgs_layout &l)
{
    if (l.key_kind != MGS_KEY_FIXED || l.val_kind == MGS_VAL_CSTR) return 0;
    return l.key_bytes + (l.val_kind == MGS_VAL_FIXED ? l.val_bytes : 0);
}
} // namespace

#define STG_HIP(s, call)                                                         \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            (s)->err = sfmt("%s failed: %s", #call, hipGetErrorString(e_));      \
            return MGS_ERR_HIP;                                                  \
        }                                                                        \
    } while (0)

// Combiner mode.  mgs_combine_device works on a whole round, records arrive in
// batches: the current round's records stay staged and the round is recombined
// from its start.  A cut ends the round: the records before it leave the staging
// buffer.  When the buffer is full and the round goes on, the staged records are
// replaced by what the round has made of them -- its combined partitions one
// after the other (inside a partition the order of first occurrence is kept, so
// partitions, cuts and counters stay those of
// src/combinecollectiveshuffler.h:63-180).  out->local_bytes / packed_records
// are what this call added.
static int stager_record_len(const mgs_stager *s, const uint8_t *p, int64_t room)
{
    const mgs_layout &l = s->desc.layout;
    int64_t k = 0;
    if (l.key_kind == MGS_KEY_FIXED) {
        k = l.key_bytes;
    }
    else {
        for (int c = 0; c < (l.key_count > 0 ? l.key_count : 1); c++) {
            while (k < room && p[k] != 0) k++;
            k++;
        }
    }
    return (k + 8 <= room) ? (int) (k + 8) : -1;
}

static int stager_pack_combine(mgs_stager *s, mgs_round *out, int *drained)
{
    const int P = s->desc.nranks;
    if (s->nrec == 0) {
        for (int b = 0; b < P; b++) out->send_offset[b] = s->send_offset[b];
        *drained = 1;
        return MGS_OK;
    }
    int prev = -1;
    (void) hipGetDevice(&prev);
    struct Restore {
        int p;
        ~Restore() { if (p >= 0) (void) hipSetDevice(p); }
    } restore{prev};
    STG_HIP(s, hipSetDevice(s->desc.device));
    STG_HIP(s, hipMemcpyAsync(s->d_stage, s->h_stage, (size_t) s->used, hipMemcpyHostToDevice, s->stream));
    if (!s->fixed)
        STG_HIP(s, hipMemcpyAsync(s->d_off, s->h_off, (s->nrec + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, s->stream));
    int rc = mgs_combine_device(s->h, s->d_stage, s->fixed ? nullptr : s->d_off, s->nrec, 0, s->combine_op, s->d_image,
                                s->desc.rank >= 0 ? s->d_local : nullptr, s->cap, s->stream, out);
    if (rc != MGS_OK) {
        s->err = mgs_last_error(s->h);
        return rc;
    }
    const int64_t new_local = out->local_bytes - s->loc_done;
    if (new_local > 0) {
        STG_HIP(s, hipMemcpyAsync(s->h_local, s->d_local + s->loc_done, (size_t) new_local, hipMemcpyDeviceToHost, s->stream));
        STG_HIP(s, hipStreamSynchronize(s->stream));
    }
    s->loc_done = out->local_bytes;
    out->local_bytes = new_local > 0 ? new_local : 0;
    const uint64_t keys = out->packed_records;
    out->packed_records = keys > s->round_keys ? keys - s->round_keys : 0;
    s->round_keys = keys;
    for (int b = 0; b < P; b++) s->send_offset[b] = out->send_offset[b];
    if (out->next < s->nrec) { // a new key did not fit: the round ends before it
        if (out->next == 0) {
            s->err = "KV size is larger than buf_size";
            return MGS_ERR_RECORD_SIZE;
        }
        const int64_t cut = s->fixed ? (int64_t) out->next * fixed_len(s->desc.layout) : (int64_t) s->h_off[out->next];
        memmove(s->h_stage, s->h_stage + cut, (size_t) (s->used - cut));
        if (!s->fixed)
            for (uint64_t i = out->next; i <= s->nrec; i++) s->h_off[i - out->next] = s->h_off[i] - (uint32_t) cut;
        s->used -= cut;
        s->nrec -= out->next;
        s->loc_done = 0;
        s->round_keys = 0;
        s->cut_pending = true;
        *drained = 0;
        return MGS_OK;
    }
    if (s->want_refill) { // the buffer is full and the round goes on: keep what it has made so far
        int64_t used = 0;
        uint64_t nrec = 0;
        for (int b = 0; b < P; b++) {
            if (s->send_offset[b] > 0)
                STG_HIP(s, hipMemcpyAsync(s->h_stage + used, s->d_image + (size_t) b * s->desc.buf_size,
                                          (size_t) s->send_offset[b], hipMemcpyDeviceToHost, s->stream));
            used += s->send_offset[b];
        }
        STG_HIP(s, hipStreamSynchronize(s->stream));
        if (!s->fixed) {
            int64_t pos = 0;
            while (pos < used) {
                const int len = stager_record_len(s, s->h_stage + pos, used - pos);
                if (len <= 0 || nrec >= s->max_rec) {
                    s->err = "combined partitions do not parse";
                    return MGS_ERR_STATE;
                }
                pos += len;
                s->h_off[++nrec] = (uint32_t) pos;
            }
        }
        else {
            nrec = (uint64_t) used / (uint64_t) fixed_len(s->desc.layout);
        }
        s->used = used;
        s->nrec = nrec;
        s->loc_done = 0;
        s->want_refill = false;
    }
    *drained = 1;
    return MGS_OK;
}

extern "C" {

// Switch the stager to the combiner shuffler (op: mgs_combine_op, or -1 for the
// plain one).  The staging buffer must hold two send images: one for a refill
// with the round's partitions, one for new records.  Call before the first put.
int mgs_stager_set_combine(mgs_stager *s, int op)
{
    if (!s) return MGS_ERR_ARG;
    if (op >= 0 && (s->cap < 2 * (int64_t) s->desc.nranks * s->desc.buf_size || s->nrec != 0)) {
        s->err = "mgs_stager_set_combine: staging buffer below two send images, or records already staged";
        return MGS_ERR_ARG;
    }
    s->combine_op = op;
    return MGS_OK;
}

const char *mgs_stager_last_error(const mgs_stager *s)
{
    return s ? s->err.c_str() : g_stager_error.c_str();
}

void mgs_stager_destroy(mgs_stager *s)
{
    if (!s) return;
    int prev = -1;
    (void) hipGetDevice(&prev);
    (void) hipSetDevice(s->desc.device);
    if (s->stream) (void) hipStreamSynchronize(s->stream);
    if (s->h_stage) (void) hipHostFree(s->h_stage);
    if (s->h_off) (void) hipHostFree(s->h_off);
    if (s->h_image) (void) hipHostFree(s->h_image);
    if (s->h_local) (void) hipHostFree(s->h_local);
    if (s->d_stage) (void) hipFree(s->d_stage);
    if (s->d_off) (void) hipFree(s->d_off);
    if (s->d_image) (void) hipFree(s->d_image);
    if (s->d_local) (void) hipFree(s->d_local);
    if (s->d_recv) (void) hipFree(s->d_recv);
    if (s->d_roff) (void) hipFree(s->d_roff);
    if (s->h_recv) (void) hipHostFree(s->h_recv);
    if (s->h_roff) (void) hipHostFree(s->h_roff);
    if (s->stream) (void) hipStreamDestroy(s->stream);
    if (prev >= 0) (void) hipSetDevice(prev);
    delete s;
}

int mgs_stager_create(mgs_handle *h, int64_t stage_bytes, mgs_stager **out)
{
    if (!h || !out || stage_bytes < 64) {
        g_stager_error = "mgs_stager_create: bad argument";
        return MGS_ERR_ARG;
    }
    *out = nullptr;
    mgs_stager *s = new (std::nothrow) mgs_stager();
    if (!s) return MGS_ERR_ARG;
    s->h = h;
    if (mgs_get_desc(h, &s->desc) != MGS_OK) {
        delete s;
        return MGS_ERR_ARG;
    }
    const int fl = fixed_len(s->desc.layout);
    s->fixed = fl > 0;
    s->cap = stage_bytes;
    const uint64_t by_bytes = (uint64_t) stage_bytes / (uint64_t)(fl > 0 ? fl : 1);
    s->max_rec = by_bytes < s->desc.max_records ? by_bytes : s->desc.max_records;
    if (s->max_rec == 0) s->max_rec = 1;
    memset(s->send_offset, 0, sizeof(s->send_offset));
    int prev = -1;
    (void) hipGetDevice(&prev);
    const size_t image = (size_t) s->desc.nranks * (size_t) s->desc.buf_size;
    hipError_t e = hipSetDevice(s->desc.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc(&s->h_stage, (size_t) stage_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&s->h_off, (s->max_rec + 1) * sizeof(uint32_t), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&s->h_image, image, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc(&s->h_local, (size_t) stage_bytes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc(&s->d_stage, (size_t) stage_bytes);
    if (e == hipSuccess) e = hipMalloc(&s->d_off, (s->max_rec + 1) * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc(&s->d_image, image);
    if (e == hipSuccess) e = hipMalloc(&s->d_local, (size_t) stage_bytes);
    if (prev >= 0) (void) hipSetDevice(prev);
    if (e != hipSuccess) {
        g_stager_error = sfmt("mgs_stager_create: %s", hipGetErrorString(e));
        mgs_stager_destroy(s);
        return MGS_ERR_HIP;
    }
    s->h_off[0] = 0;
    *out = s;
    return MGS_OK;
}

// Append one pair in wire form.  Returns 0, or 1 when the staging buffer has
// no room for it (nothing was staged: pack, then stage again).
int mgs_stager_put(mgs_stager *s, const void *key, uint32_t klen, const void *val, uint32_t vlen)
{
    if (!s) return MGS_ERR_ARG;
    const int64_t len = (int64_t) klen + (int64_t) vlen;
    if (len <= 0 || len > s->cap) {
        s->err = sfmt("record of %lld bytes cannot be staged (staging buffer %lld)",
                      (long long) len, (long long) s->cap);
        return MGS_ERR_RECORD_SIZE;
    }
    if (s->in_flight) {
        s->err = "a pack is in flight on this stager: call mgs_stager_pack_end first";
        return MGS_ERR_STATE;
    }
    if (s->uploaded || s->nrec >= s->max_rec || s->used + len > s->cap) {
        s->want_refill = true;
        return 1;
    }
    if (klen) memcpy(s->h_stage + s->used, key, klen);
    if (vlen) memcpy(s->h_stage + s->used + klen, val, vlen);
    s->used += len;
    s->nrec++;
    s->h_off[s->nrec] = (uint32_t) s->used;
    return MGS_OK;
}

uint64_t mgs_stager_pending(const mgs_stager *s) { return s ? s->nrec - s->cursor : 0; }

// The staging buffer itself, for a caller that inlines mgs_stager_put: it appends records at
// bytes + used, stores the new end in off[nrec + 1] and hands the new fill back with
// mgs_stager_commit before any other stager call.  writable == 0: the buffer cannot take
// records right now (its contents are on their way through the pack kernels, or the stager
// combines and manages the buffer itself): use mgs_stager_put.
int mgs_stager_window(mgs_stager *s, mgs_stage_window *w)
{
    if (!s || !w) return MGS_ERR_ARG;
    w->bytes = s->h_stage;
    w->off = s->h_off;
    w->cap = s->cap;
    w->max_rec = s->max_rec;
    w->used = s->used;
    w->nrec = s->nrec;
    w->writable = (!s->uploaded && s->combine_op < 0) ? 1 : 0;
    return MGS_OK;
}

int mgs_stager_commit(mgs_stager *s, int64_t used, uint64_t nrec)
{
    if (!s) return MGS_ERR_ARG;
    if (s->uploaded || s->combine_op >= 0 || used < s->used || nrec < s->nrec || used > s->cap || nrec > s->max_rec
        || (int64_t) s->h_off[nrec] != used) {
        s->err = "mgs_stager_commit: the fill does not continue the window handed out";
        return MGS_ERR_STATE;
    }
    s->used = used;
    s->nrec = nrec;
    return MGS_OK;
}

// Pack the staged records that are not packed yet into the current round.
// *drained = 1: every staged record was consumed and the staging buffer is
// empty again.  *drained = 0: a partition is full -- the round must be
// exchanged (mgs_stager_round_done) before the next call continues with the
// record that did not fit (src/collectiveshuffler.h:148-160).
// *local_records points at out->local_bytes bytes of wire-form records whose
// destination is this rank, in emit order (valid until the next call).
//
// In two halves for a caller that has other work while the GPU packs (the
// non-blocking shuffler tests its exchanges in flight,
// src/nbcollectiveshuffler.h:279-387): mgs_stager_pack_begin enqueues the upload
// and the pack kernels and returns, mgs_stager_pack_end waits for them.
int mgs_stager_pack_begin(mgs_stager *s)
{
    if (!s) return MGS_ERR_ARG;
    if (s->in_flight) {
        s->err = "mgs_stager_pack_begin: a pack is in flight already";
        return MGS_ERR_STATE;
    }
    if (s->combine_op >= 0 || s->cursor >= s->nrec) return MGS_OK; // nothing to enqueue: pack_end does it all
    int prev = -1;
    (void) hipGetDevice(&prev);
    struct Restore {
        int p;
        ~Restore() { if (p >= 0) (void) hipSetDevice(p); }
    } restore{prev};
    STG_HIP(s, hipSetDevice(s->desc.device));
    if (!s->uploaded) {
        STG_HIP(s, hipMemcpyAsync(s->d_stage, s->h_stage, (size_t) s->used, hipMemcpyHostToDevice, s->stream));
        if (!s->fixed)
            STG_HIP(s, hipMemcpyAsync(s->d_off, s->h_off, (s->nrec + 1) * sizeof(uint32_t),
                                      hipMemcpyHostToDevice, s->stream));
        s->uploaded = true;
    }
    const int rc = mgs_pack_device_async(s->h, s->d_stage, s->fixed ? nullptr : s->d_off, s->nrec, s->cursor,
                                         s->send_offset, s->d_image, s->desc.rank >= 0 ? s->d_local : nullptr,
                                         s->cap, nullptr, MGS_ALGO_AUTO, s->stream);
    if (rc != MGS_OK) {
        s->err = mgs_last_error(s->h);
        return rc;
    }
    s->in_flight = true;
    return MGS_OK;
}

int mgs_stager_pack_end(mgs_stager *s, mgs_round *out, const void **local_records, int *drained)
{
    if (!s || !out || !drained) return MGS_ERR_ARG;
    memset(out, 0, sizeof(*out));
    if (local_records) *local_records = s->h_local;
    if (s->combine_op >= 0) return stager_pack_combine(s, out, drained);
    if (!s->in_flight) {
        if (s->cursor < s->nrec) {
            s->err = "mgs_stager_pack_end: no pack in flight (call mgs_stager_pack_begin)";
            return MGS_ERR_STATE;
        }
        for (int b = 0; b < s->desc.nranks; b++) out->send_offset[b] = s->send_offset[b];
        out->next = s->cursor;
        s->nrec = 0;
        s->used = 0;
        s->cursor = 0;
        s->uploaded = false;
        *drained = 1;
        return MGS_OK;
    }
    s->in_flight = false;
    int prev = -1;
    (void) hipGetDevice(&prev);
    struct Restore {
        int p;
        ~Restore() { if (p >= 0) (void) hipSetDevice(p); }
    } restore{prev};
    STG_HIP(s, hipSetDevice(s->desc.device));
    const int rc = mgs_pack_result(s->h, out);
    if (rc != MGS_OK) {
        s->err = mgs_last_error(s->h);
        return rc;
    }
    if (out->local_bytes > 0) {
        STG_HIP(s, hipMemcpyAsync(s->h_local, s->d_local, (size_t) out->local_bytes,
                                  hipMemcpyDeviceToHost, s->stream));
        STG_HIP(s, hipStreamSynchronize(s->stream));
    }
    if (out->next == s->cursor && out->local_records == 0) {
        // nothing fitted although the round was handed over empty: cannot happen
        // (a record larger than buf_size is reported by the pack call itself)
        bool empty = true;
        for (int b = 0; b < s->desc.nranks; b++) empty = empty && s->send_offset[b] == 0;
        if (empty) {
            s->err = "pack made no progress on an empty round";
            return MGS_ERR_STATE;
        }
    }
    for (int b = 0; b < s->desc.nranks; b++) s->send_offset[b] = out->send_offset[b];
    s->cursor = out->next;
    if (s->cursor >= s->nrec) {
        s->nrec = 0;
        s->used = 0;
        s->cursor = 0;
        s->uploaded = false;
        *drained = 1;
    }
    else {
        *drained = 0;
    }
    return MGS_OK;
}

int mgs_stager_pack(mgs_stager *s, mgs_round *out, const void **local_records, int *drained)
{
    if (!s || !out || !drained) return MGS_ERR_ARG;
    const int rc = mgs_stager_pack_begin(s);
    if (rc != MGS_OK) return rc;
    return mgs_stager_pack_end(s, out, local_records, drained);
}

// The round's send image.  Host view: the valid bytes of every partition are
// copied to a pinned mirror with the reference's layout (partition d at
// d*buf_size, src/collectiveshuffler.h:635) -- what MPI_Alltoallv reads.
int mgs_stager_send_image_host(mgs_stager *s, const void **host_image, const int64_t **send_offset)
{
    if (!s || !host_image || !send_offset) return MGS_ERR_ARG;
    if (s->in_flight) {
        s->err = "a pack is in flight on this stager: call mgs_stager_pack_end first";
        return MGS_ERR_STATE;
    }
    int prev = -1;
    (void) hipGetDevice(&prev);
    (void) hipSetDevice(s->desc.device);
    hipError_t e = hipSuccess;
    for (int b = 0; b < s->desc.nranks && e == hipSuccess; b++) {
        if (s->send_offset[b] > 0)
            e = hipMemcpyAsync(s->h_image + (size_t) b * s->desc.buf_size,
                               s->d_image + (size_t) b * s->desc.buf_size,
                               (size_t) s->send_offset[b], hipMemcpyDeviceToHost, s->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (prev >= 0) (void) hipSetDevice(prev);
    if (e != hipSuccess) {
        s->err = sfmt("send image copy failed: %s", hipGetErrorString(e));
        return MGS_ERR_HIP;
    }
    *host_image = s->h_image;
    *send_offset = s->send_offset;
    return MGS_OK;
}

// The same copy into a buffer of the caller's (a message buffer of the non-blocking
// shuffler, src/nbcollectiveshuffler.h:426-449: nranks * buf_size bytes): the partitions
// go where the exchange reads them, without a stop in the stager's own mirror.  Memory from
// mgs_host_alloc is copied to directly; any other host memory works, more slowly.
int mgs_stager_send_image_to(mgs_stager *s, void *host_dst, const int64_t **send_offset)
{
    if (!s || !host_dst || !send_offset) return MGS_ERR_ARG;
    if (s->in_flight) {
        s->err = "a pack is in flight on this stager: call mgs_stager_pack_end first";
        return MGS_ERR_STATE;
    }
    int prev = -1;
    (void) hipGetDevice(&prev);
    (void) hipSetDevice(s->desc.device);
    hipError_t e = hipSuccess;
    for (int b = 0; b < s->desc.nranks && e == hipSuccess; b++) {
        if (s->send_offset[b] > 0)
            e = hipMemcpyAsync(static_cast<uint8_t *>(host_dst) + (size_t) b * s->desc.buf_size,
                               s->d_image + (size_t) b * s->desc.buf_size,
                               (size_t) s->send_offset[b], hipMemcpyDeviceToHost, s->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    if (prev >= 0) (void) hipSetDevice(prev);
    if (e != hipSuccess) {
        s->err = sfmt("send image copy failed: %s", hipGetErrorString(e));
        return MGS_ERR_HIP;
    }
    *send_offset = s->send_offset;
    return MGS_OK;
}

// Page-locked host memory for buffers the library copies to or from.
void *mgs_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void mgs_host_free(void *p)
{
    if (p) (void) hipHostFree(p);
}

// Device memory and a blocking device-to-host copy, for host code that holds device buffers
// of its own (the destination streams of a whole map) without being built against HIP.
void *mgs_device_alloc(int device, size_t bytes)
{
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    void *p = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) p = nullptr;
    if (prev >= 0) (void) hipSetDevice(prev);
    return p;
}

void mgs_device_free(int device, void *p)
{
    if (!p) return;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (hipSetDevice(device) == hipSuccess) (void) hipFree(p);
    if (prev >= 0) (void) hipSetDevice(prev);
}

int mgs_copy_to_host(int device, void *host_dst, const void *d_src, size_t bytes)
{
    if (bytes == 0) return MGS_OK;
    if (!host_dst || !d_src) return MGS_ERR_ARG;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMemcpy(host_dst, d_src, bytes, hipMemcpyDeviceToHost);
    if (prev >= 0) (void) hipSetDevice(prev);
    return e == hipSuccess ? MGS_OK : MGS_ERR_HIP;
}

// Device view, for an exchange that reads GPU memory (RCCL, GPU-aware MPI).
int mgs_stager_send_image_device(mgs_stager *s, void **d_image, const int64_t **send_offset)
{
    if (!s || !d_image || !send_offset) return MGS_ERR_ARG;
    if (s->in_flight) {
        s->err = "a pack is in flight on this stager: call mgs_stager_pack_end first";
        return MGS_ERR_STATE;
    }
    *d_image = s->d_image;
    *send_offset = s->send_offset;
    return MGS_OK;
}

// The receive image of an exchange that writes device memory (RCCL, GPU-aware MPI,
// peer copies): nranks * buf_size bytes, the reference's recv_buffer
// (src/collectiveshuffler.h:97-98), filled source by source without gaps (:637-639).
int mgs_stager_recv_image_device(mgs_stager *s, void **d_recv)
{
    if (!s || !d_recv) return MGS_ERR_ARG;
    if (!s->d_recv) {
        int prev = -1;
        (void) hipGetDevice(&prev);
        const size_t image = (size_t) s->desc.nranks * (size_t) s->desc.buf_size;
        // (the table of record offsets is sized by what an exchange actually delivers, in
        // mgs_stager_recv_to_host)
        hipError_t e = hipSetDevice(s->desc.device);
        if (e == hipSuccess) e = hipMalloc(&s->d_recv, image);
        if (e == hipSuccess) e = hipHostMalloc(&s->h_recv, image, hipHostMallocDefault);
        if (prev >= 0) (void) hipSetDevice(prev);
        if (e != hipSuccess) {
            s->err = sfmt("mgs_stager_recv_image_device: %s", hipGetErrorString(e));
            return MGS_ERR_HIP;
        }
    }
    *d_recv = s->d_recv;
    return MGS_OK;
}

// save_data (src/collectiveshuffler.h:563-585) for a receive image that lies in device
// memory: the record boundaries of its first nbytes bytes are found on the GPU
// (mgs_unpack_device) and the bytes and the offsets are copied to pinned host memory, where
// `out->write` needs them.  *host_rec_off: n_records + 1 offsets into *host_records, or
// NULL for a layout the device parse does not take (the caller walks the bytes then).
int mgs_stager_recv_to_host(mgs_stager *s, int64_t nbytes, const void **host_records,
                            const uint32_t **host_rec_off, uint64_t *n_records)
{
    if (!s || !host_records || !host_rec_off || !n_records || nbytes < 0 || !s->d_recv
        || nbytes > (int64_t) s->desc.nranks * s->desc.buf_size) {
        if (s) s->err = "mgs_stager_recv_to_host: bad argument (or no receive image yet)";
        return MGS_ERR_ARG;
    }
    *host_records = s->h_recv;
    *host_rec_off = nullptr;
    *n_records = 0;
    if (nbytes == 0) return MGS_OK;
    const mgs_layout &l = s->desc.layout;
    const bool device_parse = fixed_len(l) > 0
                              || (l.key_kind == MGS_KEY_CSTR && l.key_count <= 1
                                  && (l.val_kind == MGS_VAL_VOID || (l.val_kind == MGS_VAL_FIXED && l.val_bytes <= 62)));
    int prev = -1;
    (void) hipGetDevice(&prev);
    struct Restore {
        int p;
        ~Restore() { if (p >= 0) (void) hipSetDevice(p); }
    } restore{prev};
    STG_HIP(s, hipSetDevice(s->desc.device));
    STG_HIP(s, hipMemcpyAsync(s->h_recv, s->d_recv, (size_t) nbytes, hipMemcpyDeviceToHost, s->stream));
    if (device_parse) {
        // one offset per record: the bytes received over the layout's shortest record
        const int fl = fixed_len(l);
        const size_t minlen = fl > 0 ? (size_t) fl : 1 + (l.val_kind == MGS_VAL_FIXED ? (size_t) l.val_bytes : 0);
        const uint64_t need = (uint64_t) nbytes / minlen + 2;
        if (need > s->roff_cap) {
            STG_HIP(s, hipStreamSynchronize(s->stream)); // (nothing of the old tables is in flight after this)
            if (s->d_roff) STG_HIP(s, hipFree(s->d_roff));
            if (s->h_roff) STG_HIP(s, hipHostFree(s->h_roff));
            s->d_roff = nullptr;
            s->h_roff = nullptr;
            s->roff_cap = 0;
            const uint64_t cap = need + need / 4;
            STG_HIP(s, hipMalloc(&s->d_roff, cap * sizeof(uint32_t)));
            STG_HIP(s, hipHostMalloc(&s->h_roff, cap * sizeof(uint32_t), hipHostMallocDefault));
            s->roff_cap = cap;
        }
        uint64_t n = 0;
        int ncuts = 0;
        mgs_page_cut cut;
        const int rc = mgs_unpack_device(s->h, s->d_recv, (uint64_t) nbytes, s->d_roff, s->roff_cap, (int64_t) 1 << 40, 0,
                                         &n, &cut, 1, &ncuts, s->stream);
        if (rc != MGS_OK) {
            s->err = mgs_last_error(s->h);
            return rc;
        }
        STG_HIP(s, hipMemcpyAsync(s->h_roff, s->d_roff, (n + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, s->stream));
        *n_records = n;
        *host_rec_off = s->h_roff;
    }
    STG_HIP(s, hipStreamSynchronize(s->stream));
    return MGS_OK;
}

// After the exchange: every partition is empty again (send_offset[i] = 0,
// src/collectiveshuffler.h:657).
int mgs_stager_round_done(mgs_stager *s)
{
    if (!s) return MGS_ERR_ARG;
    if (s->in_flight) {
        s->err = "a pack is in flight on this stager: call mgs_stager_pack_end first";
        return MGS_ERR_STATE;
    }
    memset(s->send_offset, 0, sizeof(s->send_offset));
    if (s->combine_op >= 0) {
        // the exchange forgets the round's keys (garbage_collection, src/combinecollectiveshuffler.h:237):
        // after a cut the staged records already belong to the next round, otherwise the
        // round that was open is over and its records go
        if (!s->cut_pending) {
            s->nrec = 0;
            s->used = 0;
            s->loc_done = 0;
            s->round_keys = 0;
        }
        s->cut_pending = false;
    }
    return MGS_OK;
}

} // extern "C"
