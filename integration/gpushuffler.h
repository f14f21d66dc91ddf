// src/gpushuffler.h -- the file a Mimir maintainer adds to bind the GPU pack path
// (this repository: integration/gpushuffler.h; the integration document shows it verbatim).
// Syntax-checked against the reference's own headers by tests/test_integration_binding.py.
#include <mpi.h>
#include "config.h"                    // COMM_BUF_SIZE
#include "interface.h"                 // MIMIR_NS::Writable
#include "baseshuffler.h"              // MIMIR_NS::BaseShuffler, HashBucket
#include "mimir_gpu_shuffler.hpp"      // this repository, include/

namespace MIMIR_NS {

class MpiExchange : public mimir_gpu::Exchange {
  public:
    explicit MpiExchange(MPI_Comm c) : comm(c) { MPI_Comm_rank(c, &r); MPI_Comm_size(c, &n); }
    int rank() const { return r; }
    int size() const { return n; }
    void alltoall_int(const int *s, int *d)               // src/collectiveshuffler.h:606-607
    { MPI_Alltoall((void *) s, 1, MPI_INT, d, 1, MPI_INT, comm); }
    void alltoallv_bytes(const char *sb, const int *sc, const int *sd,
                         char *rb, const int *rc, const int *rd)   // :643-645
    { MPI_Alltoallv((void *) sb, (int *) sc, (int *) sd, MPI_BYTE, rb, (int *) rc, (int *) rd, MPI_BYTE, comm); }
    // send buffers above 1 GiB: counts in units of the contiguous type the reference commits, :99-100, :630-645
    void alltoallv_units(const char *sb, const int *sc, const int *sd, char *rb, const int *rc, const int *rd, int unit)
    { MPI_Datatype t; MPI_Type_contiguous(unit, MPI_BYTE, &t); MPI_Type_commit(&t);
      MPI_Alltoallv((void *) sb, (int *) sc, (int *) sd, t, rb, (int *) rc, (int *) rd, t, comm); MPI_Type_free(&t); }
    void abort(int code) { MPI_Abort(MPI_COMM_WORLD, code); }      // LOG_ERROR, src/log.h:70-78
    // for NBGpuShuffler: counts and payloads on two duplicated communicators,
    // src/nbcollectiveshuffler.h:68-69, :237-239, :347-351, :291/:364 (MPI_Test)
    Request ialltoall_int(const int *s, int *d)
    { dup(); MPI_Request *q = new MPI_Request;
      MPI_Ialltoall((void *) s, 1, MPI_INT, d, 1, MPI_INT, a2a_comm, q); return q; }
    Request ialltoallv_bytes(const char *sb, const int *sc, const int *sd, char *rb, const int *rc, const int *rd)
    { dup(); MPI_Request *q = new MPI_Request;
      MPI_Ialltoallv((void *) sb, (int *) sc, (int *) sd, MPI_BYTE, rb, (int *) rc, (int *) rd, MPI_BYTE, a2av_comm, q);
      return q; }
    Request ialltoallv_units(const char *sb, const int *sc, const int *sd, char *rb, const int *rc, const int *rd,
                             int unit)                             // src/nbcollectiveshuffler.h:109-110, :322-351
    { dup(); MPI_Datatype t; MPI_Type_contiguous(unit, MPI_BYTE, &t); MPI_Type_commit(&t);
      MPI_Request *q = new MPI_Request;
      MPI_Ialltoallv((void *) sb, (int *) sc, (int *) sd, t, rb, (int *) rc, (int *) rd, t, a2av_comm, q);
      MPI_Type_free(&t); return q; }                               // freed once the exchange in flight has ended
    bool test(Request &req)
    { if (!req) return true; int flag = 0; MPI_Test((MPI_Request *) req, &flag, MPI_STATUS_IGNORE);
      if (flag) { delete (MPI_Request *) req; req = NULL; } return flag != 0; }
  private:
    void dup() { if (a2a_comm == MPI_COMM_NULL) { MPI_Comm_dup(comm, &a2a_comm); MPI_Comm_dup(comm, &a2av_comm); } }
    MPI_Comm comm, a2a_comm = MPI_COMM_NULL, a2av_comm = MPI_COMM_NULL; int r, n;
};

// Mimir's `out` seen through the shim's Writable
template <typename K, typename V>
class OutAdapter : public mimir_gpu::Writable<K, V> {
  public:
    explicit OutAdapter(Writable<K, V> *o) : o(o) {}
    int open() { return true; }  void close() {}
    int seek(mimir_gpu::DB_POS) { return false; }
    uint64_t get_record_count() { return o->get_record_count(); }
    int write(K *k, V *v) { return o->write(k, v); }
  private:
    Writable<K, V> *o;
};

// BaseShuffler-shaped wrapper so MimirContext can hold it as BaseShuffler<K,V>*
template <typename K, typename V>
class GpuShuffler : public BaseShuffler<K, V> {
  public:
    GpuShuffler(MPI_Comm comm, Writable<K, V> *out, int (*user_hash)(K *, V *, int),
                int keycount, int valcount, bool split_hint, HashBucket<> *h)
        : BaseShuffler<K, V>(comm, out, user_hash, keycount, valcount, split_hint, h),
          ex(comm), adapter(out),
          impl(&ex, &adapter, user_hash, keycount, valcount, split_hint, h, COMM_BUF_SIZE, gpu_of_rank()) {}
    virtual int open() { return impl.open(); }
    virtual int write(K *k, V *v) { return impl.write(k, v); }
    virtual void close() { impl.close(); }
    virtual void make_progress(bool issue_new = false) { impl.make_progress(issue_new); }
    virtual uint64_t get_record_count() { return impl.get_record_count(); }
  private:
    static int gpu_of_rank() { const char *l = getenv("MPI_LOCALRANKID"); return l ? atoi(l) : 0; }
    MpiExchange ex; OutAdapter<K, V> adapter; mimir_gpu::GpuShuffler<K, V> impl;
};

// The same for a map that emits device-resident batches (wire-form records in HBM, e.g. what
// mgs_tokenize_device leaves): the callback is handed this object as its Writable
// (src/mimircontext.h:257) and reaches write_device_records through a dynamic_cast.
template <typename K, typename V>
class StreamsGpuShuffler : public BaseShuffler<K, V> {
  public:
    StreamsGpuShuffler(MPI_Comm comm, Writable<K, V> *out, int (*user_hash)(K *, V *, int),
                       int keycount, int valcount, bool split_hint, HashBucket<> *h)
        : BaseShuffler<K, V>(comm, out, user_hash, keycount, valcount, split_hint, h),
          ex(comm), adapter(out),
          impl(&ex, &adapter, user_hash, keycount, valcount, split_hint, h, COMM_BUF_SIZE, gpu_of_rank()) {}
    virtual int open() { return impl.open(); }
    virtual int write(K *k, V *v) { return impl.write(k, v); }   // refused: pair-by-pair maps use GpuShuffler
    int write_device_records(const void *d_records, uint64_t n, const uint32_t *d_rec_off = NULL)
    { return impl.write_device_records(d_records, n, d_rec_off); }
    virtual void close() { impl.close(); }                        // every exchange of the map, then `out` is served
    virtual void make_progress(bool issue_new = false) { impl.make_progress(issue_new); }
    virtual uint64_t get_record_count() { return impl.get_record_count(); }
  private:
    static int gpu_of_rank() { const char *l = getenv("MPI_LOCALRANKID"); return l ? atoi(l) : 0; }
    MpiExchange ex; OutAdapter<K, V> adapter; mimir_gpu::StreamsGpuShuffler<K, V> impl;
};

} // namespace MIMIR_NS
