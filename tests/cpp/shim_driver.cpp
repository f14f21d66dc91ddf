// Test driver for include/mimir_gpu_shuffler.hpp: P ranks as P threads of one
// process, every rank feeding its records one write(key*, val*) at a time into
// a GpuShuffler<K,V> the way a Mimir map callback does, with an in-process
// all-to-all standing in for MPI.  Dumps what each rank's `out` received.
//
//   shim_driver <kind> <P> <comm_bytes> <stage_bytes> <indir> <outdir> [keycount] [nb]
//   kind: u64 | words | oct | join;  nb: NBGpuShuffler over non-blocking exchanges;
//   sum (in place of nb): GpuShuffler as the combiner shuffler with MGS_COMBINE_SUM_U64;
//   streams (in place of nb): StreamsGpuShuffler fed device-resident batches (kind longwords too;
//   MGS_SHIM_STREAM_BYTES: bytes per destination stream, MIMIR_GPU_WINDOW_SIZE: exchange window)
// Environment: MGS_SHIM_DEVICE_EXCHANGE=1 -- the transport moves device memory
// (Exchange::alltoallv_device: copies between the ranks' device images); MGS_SHIM_TIMING=1 --
// the sinks only count, and rank<r>.counters also holds the wall time of the write() loop.
#include <pthread.h>

#include <hip/hip_runtime_api.h>

#include <chrono>

#include <map>
#include <mutex>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "mimir_gpu_shuffler.hpp"

using namespace mimir_gpu;

struct Blob16 {
    char b[16];
};

// One non-blocking collective, matched across ranks by its sequence number:
// every rank leaves what it sends (payloads are copied at post time, so the
// sender's buffer is free at once); the operation is complete for a rank once
// all ranks have posted.
struct NbOp {
    int arrived = 0, readers = 0;
    std::vector<int> counts;             // P x P
    std::vector<std::vector<char>> data; // P x P (payload operations)
};

struct Shared {
    int P;
    pthread_barrier_t bar;
    std::vector<int> counts;           // P x P
    std::vector<const char *> sendbuf; // per rank
    std::vector<const int *> scount, sdispl;
    std::vector<const void *> d_send;        // per rank: device send image
    std::vector<const int64_t *> d_sdispl;
    std::mutex mu;
    std::map<uint64_t, NbOp> a2a, a2av;
};

struct NbReq {
    bool payload;
    uint64_t seq;
    int *recv_counts;
    char *recvbuf;
    const int *rcount, *rdispl;
    std::vector<int> own; // byte counts of a unit-counted exchange (they must live as long as the request)
};

class ThreadExchange : public Exchange {
  public:
    ThreadExchange(Shared *s, int r) : s(s), r(r) {}
    int rank() const { return r; }
    int size() const { return s->P; }
    void alltoall_int(const int *send, int *recv)
    {
        for (int d = 0; d < s->P; d++) s->counts[(size_t)(r * s->P + d)] = send[d];
        pthread_barrier_wait(&s->bar);
        for (int src = 0; src < s->P; src++) recv[src] = s->counts[(size_t)(src * s->P + r)];
        pthread_barrier_wait(&s->bar);
    }
    void alltoallv_bytes(const char *sendbuf, const int *scount, const int *sdispl, char *recvbuf,
                         const int *rcount, const int *rdispl)
    {
        s->sendbuf[(size_t) r] = sendbuf;
        s->scount[(size_t) r] = scount;
        s->sdispl[(size_t) r] = sdispl;
        pthread_barrier_wait(&s->bar);
        for (int src = 0; src < s->P; src++) {
            if (rcount[src] != s->scount[(size_t) src][r]) {
                fprintf(stderr, "count mismatch %d<-%d\n", r, src);
                std::abort();
            }
            memcpy(recvbuf + rdispl[src], s->sendbuf[(size_t) src] + s->sdispl[(size_t) src][r],
                   (size_t) rcount[src]);
        }
        pthread_barrier_wait(&s->bar);
    }
    // Device memory: every rank pulls its partitions out of the other ranks' device images
    // (the ranks of this driver share one GPU; between GPUs the same copies are peer copies).
    static bool device_exchange_on()
    {
        static const bool on = getenv("MGS_SHIM_DEVICE_EXCHANGE") && atoi(getenv("MGS_SHIM_DEVICE_EXCHANGE")) != 0;
        return on;
    }
    bool moves_device_memory() const { return device_exchange_on(); }
    bool alltoallv_device(const void *d_sendbuf, const int *scount, const int64_t *sdispl, void *d_recvbuf,
                          const int *rcount, const int64_t *rdispl)
    {
        if (!device_exchange_on()) return false;
        s->d_send[(size_t) r] = d_sendbuf;
        s->scount[(size_t) r] = scount;
        s->d_sdispl[(size_t) r] = sdispl;
        pthread_barrier_wait(&s->bar);
        for (int src = 0; src < s->P; src++) {
            if (rcount[src] != s->scount[(size_t) src][r]) {
                fprintf(stderr, "count mismatch %d<-%d\n", r, src);
                std::abort();
            }
            if (rcount[src] == 0) continue;
            const hipError_t e = hipMemcpyAsync(static_cast<char *>(d_recvbuf) + rdispl[src],
                                                static_cast<const char *>(s->d_send[(size_t) src])
                                                    + s->d_sdispl[(size_t) src][r],
                                                (size_t) rcount[src], hipMemcpyDeviceToDevice, NULL);
            if (e != hipSuccess) {
                fprintf(stderr, "device exchange copy failed: %s\n", hipGetErrorString(e));
                std::abort();
            }
        }
        if (hipStreamSynchronize(NULL) != hipSuccess) std::abort();
        pthread_barrier_wait(&s->bar); // every image has been read: the senders may pack again
        return true;
    }
    Request ialltoall_int(const int *send, int *recv)
    {
        std::lock_guard<std::mutex> g(s->mu);
        NbOp &op = s->a2a[a2a_seq];
        if (op.counts.empty()) op.counts.assign((size_t)(s->P * s->P), 0);
        for (int d = 0; d < s->P; d++) op.counts[(size_t)(r * s->P + d)] = send[d];
        op.arrived++;
        return new NbReq{false, a2a_seq++, recv, NULL, NULL, NULL};
    }
    Request ialltoallv_bytes(const char *sendbuf, const int *scount, const int *sdispl, char *recvbuf,
                             const int *rcount, const int *rdispl)
    {
        std::lock_guard<std::mutex> g(s->mu);
        NbOp &op = s->a2av[a2av_seq];
        if (op.counts.empty()) {
            op.counts.assign((size_t)(s->P * s->P), 0);
            op.data.resize((size_t)(s->P * s->P));
        }
        for (int d = 0; d < s->P; d++) {
            op.counts[(size_t)(r * s->P + d)] = scount[d];
            op.data[(size_t)(r * s->P + d)].assign(sendbuf + sdispl[d], sendbuf + sdispl[d] + scount[d]);
        }
        op.arrived++;
        return new NbReq{true, a2av_seq++, NULL, recvbuf, rcount, rdispl};
    }
    // units of more than one byte (send buffers above 1 GiB): the same exchange on byte counts
    Request ialltoallv_units(const char *sendbuf, const int *scount, const int *sdispl, char *recvbuf,
                             const int *rcount, const int *rdispl, int unit_bytes)
    {
        if (unit_bytes == 1) return ialltoallv_bytes(sendbuf, scount, sdispl, recvbuf, rcount, rdispl);
        std::vector<int> b;
        units_to_bytes(scount, sdispl, rcount, rdispl, unit_bytes, b);
        const size_t n = (size_t) s->P;
        NbReq *q = static_cast<NbReq *>(ialltoallv_bytes(sendbuf, &b[0], &b[n], recvbuf, &b[2 * n], &b[3 * n]));
        std::lock_guard<std::mutex> g(s->mu);
        q->own.swap(b);
        q->rcount = &q->own[2 * n];
        q->rdispl = &q->own[3 * n];
        return q;
    }
    bool test(Request &req)
    {
        if (req == NULL) return true;
        NbReq *q = static_cast<NbReq *>(req);
        std::lock_guard<std::mutex> g(s->mu);
        std::map<uint64_t, NbOp> &ops = q->payload ? s->a2av : s->a2a;
        NbOp &op = ops[q->seq];
        if (op.arrived < s->P) return false;
        for (int src = 0; src < s->P; src++) {
            const int c = op.counts[(size_t)(src * s->P + r)];
            if (!q->payload) {
                q->recv_counts[src] = c;
                continue;
            }
            if (q->rcount[src] != c) {
                fprintf(stderr, "count mismatch %d<-%d\n", r, src);
                std::abort();
            }
            memcpy(q->recvbuf + q->rdispl[src], op.data[(size_t)(src * s->P + r)].data(), (size_t) c);
        }
        if (++op.readers == s->P) ops.erase(q->seq);
        delete q;
        req = NULL;
        return true;
    }
    void abort(int code) { std::_Exit(code); }

  private:
    Shared *s;
    int r;
    uint64_t a2a_seq = 0, a2av_seq = 0;
};

// Sink: re-serialises every pair it is handed, so the dump is wire form again.
template <typename K, typename V>
class DumpSink : public Writable<K, V> {
  public:
    DumpSink(int keycount, int valcount) : cur_src(-1), kc(keycount), vc(valcount), n(0) { off.push_back(0); }
    int open() { return true; }
    void close() {}
    int seek(DB_POS) { return false; }
    uint64_t get_record_count() { return n; }
    int write(K *key, V *val)
    {
        static const bool count_only = getenv("MGS_SHIM_TIMING") && atoi(getenv("MGS_SHIM_TIMING")) != 0;
        if (count_only) {
            n++;
            return true;
        }
        std::vector<char> tmp;
        const int kl = wire<K>::size(key, kc);
        const char *kb = wire<K>::bytes(key, kc, tmp);
        data.insert(data.end(), kb, kb + kl);
        const int vl = wire<V>::size(val, vc);
        const char *vb = wire<V>::bytes(val, vc, tmp);
        if (vl) data.insert(data.end(), vb, vb + vl);
        off.push_back((uint64_t) data.size());
        src.push_back(cur_src);
        n++;
        return true;
    }
    std::vector<char> data;
    std::vector<uint64_t> off;
    std::vector<int32_t> src; // the rank every record came from
    int cur_src;

  private:
    int kc, vc;
    uint64_t n;
};

static std::vector<char> slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

// The shuffler under test, telling the sink whose records come next.
template <typename Sh, typename K, typename V>
class Tagged : public Sh {
  public:
    using Sh::Sh;
    DumpSink<K, V> *sink = NULL;

  protected:
    void begin_source(int src) { sink->cur_src = src; }
};

static bool g_nb = false;
static int g_combine = -1;

template <typename K, typename V, typename Feed>
static void run_rank(Shared *sh, int r, int64_t comm, int64_t /*stage*/, const std::string &indir,
                     const std::string &outdir, int keycount, Feed feed)
{
    std::vector<char> blob = slurp(indir + "/rank" + std::to_string(r) + ".blob");
    std::vector<char> offb = slurp(indir + "/rank" + std::to_string(r) + ".off");
    const uint64_t *off = reinterpret_cast<const uint64_t *>(offb.data());
    const size_t n = offb.size() / 8 - 1;
    ThreadExchange ex(sh, r);
    DumpSink<K, V> sink(keycount, 1);
    GpuShuffler<K, V> *c;
    NBGpuShuffler<K, V> *nb = NULL;
    if (g_nb) {
        Tagged<NBGpuShuffler<K, V>, K, V> *t =
            new Tagged<NBGpuShuffler<K, V>, K, V>(&ex, &sink, NULL, keycount, 1, false, NULL, comm, 0);
        t->sink = &sink;
        c = nb = t;
    }
    else {
        Tagged<GpuShuffler<K, V>, K, V> *t =
            new Tagged<GpuShuffler<K, V>, K, V>(&ex, &sink, NULL, keycount, 1, false, NULL, comm, 0, g_combine);
        t->sink = &sink;
        c = t;
    }
    c->open();
    const auto t0 = std::chrono::steady_clock::now();
    for (size_t i = 0; i < n; i++) feed(c, blob.data() + off[i], (int) (off[i + 1] - off[i]));
    const double write_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const uint64_t send_bytes_at_loop_end = c->get_send_bytes();
    c->close();
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".out", std::ios::binary)
        .write(sink.data.data(), (std::streamsize) sink.data.size());
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".outoff", std::ios::binary)
        .write(reinterpret_cast<const char *>(sink.off.data()), (std::streamsize)(sink.off.size() * 8));
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".outsrc", std::ios::binary)
        .write(reinterpret_cast<const char *>(sink.src.data()), (std::streamsize)(sink.src.size() * 4));
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".txt")
        << c->get_record_count() << " " << c->get_shuffle_times() << " " << (nb ? nb->get_buffer_count() : 1)
        << "\n";
    // send / recv bytes and exchanges as the reference's profiler counts them; device exchanges;
    // records written, seconds inside the write() loop, bytes already sent when it ended
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".counters")
        << c->get_send_bytes() << " " << c->get_recv_bytes() << " " << c->get_device_exchanges() << " " << n << " "
        << write_s << " " << send_bytes_at_loop_end << "\n";
    delete c;
}

// StreamsGpuShuffler: the rank's records are put into device memory and handed over in
// three batches (the third of them cut off-centre), as a map that emits device batches does.
template <typename K, typename V>
static void run_rank_streams(Shared *sh, int r, int64_t comm, const std::string &indir, const std::string &outdir,
                             int keycount, bool fixed_width)
{
    std::vector<char> blob = slurp(indir + "/rank" + std::to_string(r) + ".blob");
    std::vector<char> offb = slurp(indir + "/rank" + std::to_string(r) + ".off");
    const uint64_t *off = reinterpret_cast<const uint64_t *>(offb.data());
    const size_t n = offb.size() / 8 - 1;
    std::vector<uint32_t> off32(n + 1);
    for (size_t i = 0; i <= n; i++) off32[i] = (uint32_t) off[i];
    char *d_blob = NULL;
    uint32_t *d_off = NULL;
    if (hipMalloc(&d_blob, blob.size() + 16) != hipSuccess || hipMalloc(&d_off, (n + 1) * 4) != hipSuccess) std::abort();
    if (hipMemcpy(d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice) != hipSuccess) std::abort();
    if (hipMemcpy(d_off, off32.data(), (n + 1) * 4, hipMemcpyHostToDevice) != hipSuccess) std::abort();
    ThreadExchange ex(sh, r);
    DumpSink<K, V> sink(keycount, 1);
    const char *sb = getenv("MGS_SHIM_STREAM_BYTES");
    Tagged<StreamsGpuShuffler<K, V>, K, V> *c = new Tagged<StreamsGpuShuffler<K, V>, K, V>(
        &ex, &sink, NULL, keycount, 1, false, NULL, comm, 0, sb ? atoll(sb) : (int64_t) 8 << 20, 0, 64);
    c->sink = &sink;
    c->open();
    const auto t0 = std::chrono::steady_clock::now();
    const size_t cut[4] = {0, n / 5, n / 5 + n / 2, n};
    for (int b = 0; b < 3; b++) {
        const size_t i0 = cut[b], i1 = cut[b + 1];
        if (i1 == i0) continue;
        if (fixed_width) c->write_device_records(d_blob + off[i0], i1 - i0, NULL);
        else c->write_device_records(d_blob, i1 - i0, d_off + i0);
    }
    const double write_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    c->close();
    (void) hipFree(d_blob);
    (void) hipFree(d_off);
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".out", std::ios::binary)
        .write(sink.data.data(), (std::streamsize) sink.data.size());
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".outoff", std::ios::binary)
        .write(reinterpret_cast<const char *>(sink.off.data()), (std::streamsize)(sink.off.size() * 8));
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".outsrc", std::ios::binary)
        .write(reinterpret_cast<const char *>(sink.src.data()), (std::streamsize)(sink.src.size() * 4));
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".txt")
        << c->get_record_count() << " " << c->get_shuffle_times() << " 1\n";
    std::ofstream(outdir + "/rank" + std::to_string(r) + ".counters")
        << c->get_send_bytes() << " " << c->get_recv_bytes() << " " << c->get_device_exchanges() << " " << n << " "
        << write_s << " 0\n";
    delete c;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        fprintf(stderr, "usage: %s kind P comm stage indir outdir [keycount]\n", argv[0]);
        return 2;
    }
    const std::string kind = argv[1];
    const int P = atoi(argv[2]);
    const int64_t comm = atoll(argv[3]), stage = atoll(argv[4]);
    const std::string indir = argv[5], outdir = argv[6];
    const int keycount = argc > 7 ? atoi(argv[7]) : 1;
    g_nb = argc > 8 && std::string(argv[8]) == "nb";
    if (argc > 8 && std::string(argv[8]) == "sum") g_combine = MGS_COMBINE_SUM_U64;
    setenv("MIMIR_GPU_STAGE_SIZE", std::to_string(stage).c_str(), 1);
    Shared sh;
    sh.P = P;
    pthread_barrier_init(&sh.bar, NULL, (unsigned) P);
    sh.counts.assign((size_t)(P * P), 0);
    sh.sendbuf.assign((size_t) P, NULL);
    sh.scount.assign((size_t) P, NULL);
    sh.sdispl.assign((size_t) P, NULL);
    sh.d_send.assign((size_t) P, NULL);
    sh.d_sdispl.assign((size_t) P, NULL);
    const bool streams = argc > 8 && std::string(argv[8]) == "streams";
    std::vector<std::thread> th;
    for (int r = 0; r < P; r++) {
        th.emplace_back([&, r]() {
            if (streams) {
                if (kind == "u64") run_rank_streams<uint64_t, uint64_t>(&sh, r, comm, indir, outdir, 1, true);
                else if (kind == "words" || kind == "longwords")
                    run_rank_streams<char *, uint64_t>(&sh, r, comm, indir, outdir, 1, false);
                else if (kind == "oct") run_rank_streams<char, uint64_t>(&sh, r, comm, indir, outdir, keycount, true);
                else run_rank_streams<char *, Blob16>(&sh, r, comm, indir, outdir, 1, false);
                return;
            }
            if (kind == "u64") {
                run_rank<uint64_t, uint64_t>(&sh, r, comm, stage, indir, outdir, 1,
                    [](GpuShuffler<uint64_t, uint64_t> *c, char *rec, int) {
                        uint64_t k, v;
                        memcpy(&k, rec, 8);
                        memcpy(&v, rec + 8, 8);
                        c->write(&k, &v);
                    });
            }
            else if (kind == "words") { // examples/wordcount.cpp:72-76
                run_rank<char *, uint64_t>(&sh, r, comm, stage, indir, outdir, 1,
                    [](GpuShuffler<char *, uint64_t> *c, char *rec, int len) {
                        char *word = rec;
                        uint64_t one;
                        memcpy(&one, rec + len - 8, 8);
                        c->write(&word, &one);
                    });
            }
            else if (kind == "oct") { // examples/octree_clustering.cpp:226-235
                run_rank<char, uint64_t>(&sh, r, comm, stage, indir, outdir, keycount,
                    [keycount](GpuShuffler<char, uint64_t> *c, char *rec, int) {
                        char key[16];
                        memcpy(key, rec, (size_t) keycount);
                        uint64_t count;
                        memcpy(&count, rec + keycount, 8);
                        c->write(key, &count);
                    });
            }
            else { // join: examples/join.cpp:110-117
                run_rank<char *, Blob16>(&sh, r, comm, stage, indir, outdir, 1,
                    [](GpuShuffler<char *, Blob16> *c, char *rec, int len) {
                        char *word = rec;
                        Blob16 v;
                        memcpy(&v, rec + len - 16, 16);
                        c->write(&word, &v);
                    });
            }
        });
    }
    for (auto &t : th) t.join();
    return 0;
}
