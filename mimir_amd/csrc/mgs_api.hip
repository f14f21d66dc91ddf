// Host side of the C ABI declared in include/mimir_gpu_shuffle.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "mgs_common.h"
#include "mgs_generic.h"
#include "mgs_onesweep16.h"
#include "mgs_onesweep16w.h"
#include "mgs_onesweepv.h"
#include "mgs_onesweepd.h"
#include "mgs_onesweepr_api.h"
#include "mgs_kernel_tables.h"
#include "mgs_unpack.h"
#include "mgs_tokenize.h"
#include "mgs_group.h"
#include "mgs_combine.h"

using namespace mgs;

namespace {

thread_local std::string g_create_error = "";

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void) hipSetDevice(prev);
    }
};

std::string fmt(const char *f, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof(buf), f, ap);
    va_end(ap);
    return std::string(buf);
}

int fixed_record_bytes(const mgs_layout &l)
{
    if (l.key_kind != MGS_KEY_FIXED || l.val_kind == MGS_VAL_CSTR) return 0;
    return l.key_bytes + (l.val_kind == MGS_VAL_FIXED ? l.val_bytes : 0);
}

} // namespace

// ---- the no-allocation entry point's device memory ------------------------------
// mgs_pack_device_noalloc runs the one-sweep kernels with their control block and
// chained-scan arrays in this file-scope array instead of hipMalloc'ed buffers.
// The name `scratch` is a host-side stand-in, not a device address: the address of
// each GPU's copy is fetched once per device with hipGetSymbolAddress (mgs_create,
// with that device current) and only that pointer is ever handed to kernels,
// memsets and copies.
constexpr size_t kNoallocCtrlBytes = (sizeof(Ctrl) + 255) / 256 * 256;
constexpr size_t kNoallocRoundBytes = 4096; // round table + result table of a one-round call
constexpr size_t kNoallocScratchBytes = kNoallocCtrlBytes + kNoallocRoundBytes + MGS_NOALLOC_SCAN_BYTES;
static __device__ __align__(256) unsigned char scratch[kNoallocScratchBytes];

constexpr int kMaxDevices = 64;
static std::mutex g_scratch_mutex;
static unsigned char *g_scratch_ptr[kMaxDevices]; // per device, nullptr until mgs_create ran there

// Call with `device` current.
static hipError_t noalloc_scratch_for(int device, unsigned char **out)
{
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = static_cast<unsigned char *>(ptr);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

static const OsVariant *const kOsVariants = os16_variants(nullptr);
static int num_os_variants()
{
    int n = 0;
    os16_variants(&n);
    return n;
}
#define kNumOsVariants (num_os_variants())
constexpr int kDefaultOsVariant = 0;

struct mgs_handle {
    mgs_desc desc;
    int os_variant = kDefaultOsVariant;
    int num_cus = 0;
    uint32_t fixed_len = 0;
    FastMod fm;
    hipStream_t own_stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // around the dominant kernel
    bool ev_valid = false;
    Ctrl *d_ctrl = nullptr;
    unsigned char *d_noalloc = nullptr; // this device's copy of `scratch`
    CtrlHead *h_head = nullptr; // pinned, host -> device
    Ctrl *h_ctrl = nullptr;     // pinned, device -> host
    // generic path scratch
    uint32_t *d_tile_hist = nullptr;
    uint64_t *d_tile_base = nullptr;
    uint8_t *d_dest = nullptr;
    // one-sweep path scratch
    unsigned long long *d_status = nullptr;
    unsigned long long *d_agg16 = nullptr;
    size_t status_bytes = 0, agg16_bytes = 0;
    // round-cutting kernels: round table (device), result table (device + pinned host)
    unsigned long long *d_rt = nullptr, *d_rtab = nullptr, *h_rtab = nullptr;
    size_t rt_words = 0, rtab_words = 0;
    uint32_t pending_rounds = 0; // > 0: the call in flight leaves a result table for that many rounds
    bool pending_streams = false; // ... in the streams layout (rounds are views into destination streams)
    uint16_t *d_seg = nullptr; // streams layout, 16-byte records: units per (tile, worker wave, destination)
    size_t seg_bytes = 0;
    uint8_t *d_round_image = nullptr; // streams layout packed round by round: the send image of one round
    uint8_t *d_unp = nullptr;   // receive-side scratch (chunk and piece tables, cuts, result)
    size_t unp_bytes = 0;
    uint8_t *d_grp = nullptr;   // reduce-side scratch (tables, sort buffers)
    size_t grp_bytes = 0;
    uint64_t *h_unp = nullptr;  // pinned: result + cuts
    bool pending_osv = false; // the byte-granular one-sweep path was enqueued (may have fallen back)
    bool pending_noalloc = false; // ... through mgs_pack_device_noalloc: the generic kernels cannot take over
    struct {                      // the call in flight, should the generic kernels have to redo it
        const void *d_records;
        const uint32_t *d_rec_off;
        uint64_t n_end;
        void *d_send_image, *d_local_out;
        int64_t local_cap;
        uint8_t *d_dest_out;
    } redo = {};
    bool ov_k_known = false;      // variable width: the record length has been looked at
    uint32_t *h_probe = nullptr;  // pinned, two offsets
    int ov_k = 4;             // (unused) records per worker lane of the last variable-width call
    double ov_bytes_per_record = 16.0; // variable width: average record length seen last
    int ov_shrink = 0;        // ... and how much smaller than that length suggests the tile must be
    bool pending_osr = false; // the call in flight runs k_onesweepr
    int pending_osr_nq = 0;   // ... with that many register quads per record (3: a quad and a dword)
    bool osr_q3_off = false;  // string keys: too many records of this handle are longer than a quad and a dword
    const char *last_kernel = "none";
    int pending_ov_k = 4;
    // call in flight
    bool pending = false;
    hipStream_t pending_stream = nullptr;
    uint64_t pending_start = 0;
    int64_t pending_init[MGS_MAX_RANKS];
    int64_t pending_base[MGS_MAX_RANKS]; // streams layout: where the open round began in each stream
    bool pending_empty = false;
    const char *last_algo = "none";
    std::string err;
};

#define MGS_HIP(h, call)                                                        \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            (h)->err = fmt("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                 \
            return MGS_ERR_HIP;                                                 \
        }                                                                       \
    } while (0)

// Round table ((R+1) entries of `columns` words + R+1 start words) and result table.
// Streams layout: behind the finder's round table, the sweep's own two-entry table and
// the clip words (1 + columns); the result table also holds the rounds' stream offsets.
static size_t round_table_words(uint32_t max_rounds, int columns, bool streams)
{
    size_t w = ((size_t) max_rounds + 1) * ((size_t) columns + 1);
    if (streams) w += 2 * ((size_t) columns + 1) + 1 + (size_t) columns;
    return w;
}

static int ensure_round_tables(mgs_handle *h, uint32_t max_rounds, int columns, int P, bool streams = false)
{
    const size_t rtw = round_table_words(max_rounds, columns, streams);
    const size_t tabw = streams ? os_stab_words(max_rounds, (uint32_t) P) : os_rtab_words(max_rounds, (uint32_t) P);
    if (h->rt_words < rtw) {
        if (h->d_rt) MGS_HIP(h, hipFree(h->d_rt));
        h->d_rt = nullptr;
        h->rt_words = 0;
        MGS_HIP(h, hipMalloc(&h->d_rt, rtw * sizeof(unsigned long long)));
        h->rt_words = rtw;
    }
    if (h->rtab_words < tabw) {
        if (h->d_rtab) MGS_HIP(h, hipFree(h->d_rtab));
        if (h->h_rtab) MGS_HIP(h, hipHostFree(h->h_rtab));
        h->d_rtab = h->h_rtab = nullptr;
        h->rtab_words = 0;
        MGS_HIP(h, hipMalloc(&h->d_rtab, tabw * sizeof(unsigned long long)));
        MGS_HIP(h, hipHostMalloc(&h->h_rtab, tabw * sizeof(unsigned long long), hipHostMallocDefault));
        h->rtab_words = tabw;
    }
    return MGS_OK;
}

extern "C" {

int mgs_geometry(int64_t comm_buf_size, int nranks, int64_t *buf_size, int *unit_size,
                 int *type_log_bytes)
{
    if (nranks <= 0 || comm_buf_size <= 0) return MGS_ERR_ARG;
    int unit = 4096; // MEMPAGE_SIZE
    while ((int64_t) unit * (int64_t) nranks > comm_buf_size) unit /= 2;
    if (unit < 64) return MGS_ERR_ARG;
    const int64_t bs = (comm_buf_size / unit / nranks) * unit;
    int tlb = 0;
    int64_t type_bytes = 1;
    const int64_t max_comm = (int64_t) 1 << 30; // MAX_COMM_SIZE
    while (type_bytes * max_comm < bs * nranks) {
        type_bytes <<= 1;
        tlb++;
    }
    if (buf_size) *buf_size = bs;
    if (unit_size) *unit_size = unit;
    if (type_log_bytes) *type_log_bytes = tlb;
    return MGS_OK;
}

uint32_t mgs_hashlittle_host(const void *key, size_t length, uint32_t initval)
{
    // initval is folded in exactly as src/hash.cpp:163 does
    const uint8_t *k = static_cast<const uint8_t *>(key);
    if (initval == 0) return hashlittle_bytes(k, (uint32_t) length);
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + (uint32_t) length + initval;
    size_t len = length;
    while (len > 12) {
        a += (uint32_t) k[0] | ((uint32_t) k[1] << 8) | ((uint32_t) k[2] << 16) | ((uint32_t) k[3] << 24);
        b += (uint32_t) k[4] | ((uint32_t) k[5] << 8) | ((uint32_t) k[6] << 16) | ((uint32_t) k[7] << 24);
        c += (uint32_t) k[8] | ((uint32_t) k[9] << 8) | ((uint32_t) k[10] << 16) | ((uint32_t) k[11] << 24);
        MGS_MIX(a, b, c);
        len -= 12;
        k += 12;
    }
    if (len == 0) return c;
    uint32_t t[3] = {0, 0, 0};
    for (size_t i = 0; i < len; i++) t[i >> 2] |= (uint32_t) k[i] << (8 * (i & 3));
    a += t[0];
    b += t[1];
    c += t[2];
    MGS_FINAL(a, b, c);
    return c;
}

uint32_t mgs_dest_of_hash_host(uint32_t hash, uint32_t nranks)
{
    if (nranks == 0) return 0;
    const FastMod f = make_fastmod(nranks);
    return fastmod(hash, f);
}

const char *mgs_last_error(const mgs_handle *h)
{
    return h ? h->err.c_str() : g_create_error.c_str();
}

const char *mgs_last_algo(const mgs_handle *h) { return h ? h->last_algo : "none"; }

const char *mgs_last_kernel(const mgs_handle *h) { return h ? h->last_kernel : "none"; }

void mgs_destroy(mgs_handle *h)
{
    if (!h) return;
    {
        DeviceGuard g(h->desc.device);
        if (h->pending && h->pending_stream) (void) hipStreamSynchronize(h->pending_stream);
        if (h->d_ctrl) (void) hipFree(h->d_ctrl);
        if (h->d_tile_hist) (void) hipFree(h->d_tile_hist);
        if (h->d_tile_base) (void) hipFree(h->d_tile_base);
        if (h->d_dest) (void) hipFree(h->d_dest);
        if (h->d_status) (void) hipFree(h->d_status);
        if (h->d_agg16) (void) hipFree(h->d_agg16);
        if (h->d_rt) (void) hipFree(h->d_rt);
        if (h->d_rtab) (void) hipFree(h->d_rtab);
        if (h->h_rtab) (void) hipHostFree(h->h_rtab);
        if (h->d_round_image) (void) hipFree(h->d_round_image);
        if (h->d_seg) (void) hipFree(h->d_seg);
        if (h->d_unp) (void) hipFree(h->d_unp);
        if (h->d_grp) (void) hipFree(h->d_grp);
        if (h->h_unp) (void) hipHostFree(h->h_unp);
        if (h->h_probe) (void) hipHostFree(h->h_probe);
        if (h->h_head) (void) hipHostFree(h->h_head);
        if (h->h_ctrl) (void) hipHostFree(h->h_ctrl);
        if (h->ev0) (void) hipEventDestroy(h->ev0);
        if (h->ev1) (void) hipEventDestroy(h->ev1);
        if (h->own_stream) (void) hipStreamDestroy(h->own_stream);
    }
    delete h;
}

int mgs_create(const mgs_desc *desc, mgs_handle **out)
{
    if (!desc || !out) {
        g_create_error = "mgs_create: NULL argument";
        return MGS_ERR_ARG;
    }
    *out = nullptr;
    const mgs_layout &l = desc->layout;
    if (desc->nranks < 1 || desc->nranks > MGS_MAX_RANKS) {
        g_create_error = fmt("mgs_create: nranks %d outside [1,%d]", desc->nranks, MGS_MAX_RANKS);
        return MGS_ERR_ARG;
    }
    if (desc->rank < -1 || desc->rank >= desc->nranks) {
        g_create_error = fmt("mgs_create: rank %d outside [-1,%d)", desc->rank, desc->nranks);
        return MGS_ERR_ARG;
    }
    if (desc->buf_size <= 0 || desc->max_records == 0) {
        g_create_error = "mgs_create: buf_size and max_records must be positive";
        return MGS_ERR_ARG;
    }
    if ((l.key_kind != MGS_KEY_FIXED && l.key_kind != MGS_KEY_CSTR)
        || (l.val_kind != MGS_VAL_VOID && l.val_kind != MGS_VAL_FIXED && l.val_kind != MGS_VAL_CSTR)
        || (l.key_kind == MGS_KEY_FIXED && l.key_bytes < 1)
        || (l.key_kind == MGS_KEY_CSTR && l.key_count < 1)
        || (l.val_kind == MGS_VAL_FIXED && l.val_bytes < 1)
        || (l.val_kind == MGS_VAL_CSTR && l.val_count < 1)) {
        g_create_error = "mgs_create: inconsistent record layout";
        return MGS_ERR_ARG;
    }
    mgs_handle *h = new (std::nothrow) mgs_handle();
    if (!h) {
        g_create_error = "mgs_create: out of host memory";
        return MGS_ERR_ARG;
    }
    h->desc = *desc;
    h->fixed_len = (uint32_t) fixed_record_bytes(l);
    h->fm = make_fastmod((uint32_t) desc->nranks);
    if (const char *env = getenv("MGS_ONESWEEP_VARIANT")) { // tuning aid, see mgs_set_tuning
        const int v = atoi(env);
        if (v >= 0 && v < kNumOsVariants) h->os_variant = v;
    }

    auto fail = [&](hipError_t e, const char *what) {
        g_create_error = fmt("mgs_create: %s failed: %s", what, hipGetErrorString(e));
        mgs_destroy(h);
        return (int) MGS_ERR_HIP;
    };
    DeviceGuard g(desc->device);
    if (!g.ok) return fail(hipErrorInvalidDevice, "hipSetDevice");
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, desc->device);
    if (e != hipSuccess) return fail(e, "hipGetDeviceProperties");
    h->num_cus = prop.multiProcessorCount;
    // A blocking stream: work enqueued on it starts after everything the caller has put
    // on the legacy default stream before (hipMemcpy, kernels launched on stream 0), and
    // later default-stream work waits for it -- the ordering a caller that passes
    // stream == NULL expects.
    if ((e = hipStreamCreateWithFlags(&h->own_stream, hipStreamDefault)) != hipSuccess)
        return fail(e, "hipStreamCreate");
    if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return fail(e, "hipEventCreate");
    if ((e = hipMalloc(&h->d_ctrl, sizeof(Ctrl))) != hipSuccess) return fail(e, "hipMalloc(ctrl)");
    if ((e = noalloc_scratch_for(desc->device, &h->d_noalloc)) != hipSuccess)
        return fail(e, "hipGetSymbolAddress(scratch)");
    // tables of a one-round call (the pinned one also serves mgs_pack_device_noalloc)
    if (ensure_round_tables(h, 1, OS_MAXP, OS_MAXP) != MGS_OK) {
        g_create_error = "mgs_create: " + h->err;
        mgs_destroy(h);
        return MGS_ERR_HIP;
    }
    if ((e = hipHostMalloc(&h->h_head, sizeof(CtrlHead), hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc(head)");
    if ((e = hipHostMalloc(&h->h_ctrl, sizeof(Ctrl), hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc(ctrl)");
    if ((e = hipHostMalloc(&h->h_probe, 2 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess)
        return fail(e, "hipHostMalloc(probe)");
    *out = h;
    return MGS_OK;
}

static int ensure_generic_scratch(mgs_handle *h, bool need_dest)
{
    const uint64_t tiles = (h->desc.max_records + GEN_TILE - 1) / GEN_TILE;
    const uint64_t cols = (uint64_t) h->desc.nranks + 1;
    if (!h->d_tile_hist)
        MGS_HIP(h, hipMalloc(&h->d_tile_hist, tiles * cols * sizeof(uint32_t)));
    if (!h->d_tile_base)
        MGS_HIP(h, hipMalloc(&h->d_tile_base, (tiles + 1) * cols * sizeof(uint64_t)));
    if (need_dest && !h->d_dest) MGS_HIP(h, hipMalloc(&h->d_dest, h->desc.max_records));
    return MGS_OK;
}

// One-sweep scratch: status u64[tiles][columns] and the 16-bit count words
// ((1 << lpt_log) <= 2 * ceil(columns / 4) words per tile), sized for the
// smallest tile any instantiation uses.
static int ensure_onesweep_scratch(mgs_handle *h, int min_tile, int columns)
{
    const uint64_t tiles = (h->desc.max_records + min_tile - 1) / min_tile;
    const size_t sb = tiles * (uint64_t) columns * sizeof(unsigned long long);
    const size_t ab = tiles * (uint64_t) ((columns + 3) / 4 * 2) * sizeof(unsigned long long);
    if (h->status_bytes < sb) {
        if (h->d_status) MGS_HIP(h, hipFree(h->d_status));
        h->d_status = nullptr;
        h->status_bytes = 0;
        MGS_HIP(h, hipMalloc(&h->d_status, sb));
        h->status_bytes = sb;
    }
    if (h->agg16_bytes < ab) {
        if (h->d_agg16) MGS_HIP(h, hipFree(h->d_agg16));
        h->d_agg16 = nullptr;
        h->agg16_bytes = 0;
        MGS_HIP(h, hipMalloc(&h->d_agg16, ab));
        h->agg16_bytes = ab;
    }
    return MGS_OK;
}

static int os16_min_tile()
{
    int min_tile = kOsVariants[0].tile; // any variant mgs_set_tuning may select
    for (int v = 0; v < kNumOsVariants; v++) min_tile = std::min(min_tile, kOsVariants[v].tile);
    return min_tile;
}

// The byte-granular one-sweep kernel: 512 threads, 4 records per worker lane.
// Tile = 448 worker lanes x K records; K = 4, 2 or 1 by record length (a tile's bytes must
// fit the LDS staging buffer).
constexpr int kOvThreads = 512, kOvLanes = (kOvThreads / 64 - 1) * 64;
static int ov_tile(int k) { return kOvLanes * k; }
// largest K whose tile of `bytes_per_record`-byte records fits, 0 if none
static int ov_pick_k_legacy(double bytes_per_record)
{
    for (int k = 4; k >= 1; k >>= 1)
        if ((double) ov_tile(k) * bytes_per_record + 15.0 <= (double) OV_IN_BYTES) return k;
    return 0;
}
// k_onesweepd: the 64*K records of a worker wave must fit its LDS slice
static int od_pick_k(double bytes_per_record)
{
    for (int k = 4; k >= 1; k--)
        if (64.0 * k * bytes_per_record + (double) OD_SLACK <= (double) OD_SB) return k;
    return 0;
}
// Which byte-granular one-sweep kernel serves a call of `max_rounds` rounds: the LDS-image
// kernel (k_onesweepv: one round per call, faster for it) or the round-cutting kernel
// (k_onesweepd).  mgs_desc.flags pins one; MGS_OSV_LEGACY=1 / 0 in the environment does the
// same for every handle (A/B timing).
static bool osv_image_kernel(const mgs_handle *h, uint32_t max_rounds)
{
    static const char *env = getenv("MGS_OSV_LEGACY");
    if (env) return atoi(env) != 0;
    if (h->desc.flags & MGS_FLAG_OSV_IMAGE_KERNEL) return true;
    if (h->desc.flags & MGS_FLAG_OSV_ROUNDS_KERNEL) return false;
    return max_rounds == 1;
}
static int ov_pick_k(bool image_kernel, double bytes_per_record)
{
    return image_kernel ? ov_pick_k_legacy(bytes_per_record) : od_pick_k(bytes_per_record);
}

// k_onesweepr (records straight into registers) serves the LDS-image kernel's layout modes 1
// and 2 for records that fit its registers.  MGS_OSR=0 in the environment, or
// MGS_FLAG_OSV_STAGED_KERNEL in mgs_desc.flags, keep the staged kernel (k_onesweepv) for A/B runs.
static bool osr_enabled(const mgs_handle *h)
{
    static const char *env = getenv("MGS_OSR");
    if (env && atoi(env) == 0) return false;
    return (h->desc.flags & MGS_FLAG_OSV_STAGED_KERNEL) == 0;
}
// Records per lane and register quads for records of about `bytes_per_record` bytes (variable
// width: 1.25 times the average seen).  The order is the order of preference measured on
// MI355X: fixed widths run fastest with early requests and as many records per lane as registers
// and image take (five of up to 13 bytes, four of up to 32), string keys with four and late requests.
// (nq = 3 stands for a quad and a dword: records of up to 20 bytes, allowed with q3)
static bool osr_pick(int lm, bool a4, double bytes_per_record, int shrink, int *k_out, int *nq_out, bool q3 = false)
{
    static const int fixed_order[4][2] = {{5, 1}, {4, 1}, {4, 2}, {3, 2}};
    static const int cstr_order[5][2] = {{4, 3}, {4, 2}, {3, 2}, {2, 2}, {1, 2}};
    const int(*cand)[2] = lm == 2 ? fixed_order : cstr_order;
    const int ncand = lm == 2 ? 4 : 5;
    int skipped = 0;
    for (int i = 0; i < ncand; i++) {
        const int k = cand[i][0], nq = cand[i][1];
        if (!osr_has(k, nq, lm, a4)) continue;
        if ((double) (kOvLanes * k) * bytes_per_record > (double) osr_image_bytes(nq)) continue;
        if (nq == 1 && bytes_per_record > 16.0) continue;
        if (nq == 3 && (!q3 || bytes_per_record > 20.0)) continue;
        if (skipped++ < shrink) continue; // a tile of this size did not fit last time
        *k_out = k;
        *nq_out = nq;
        return true;
    }
    return false;
}
// String keys: below this average record length the register kernel with two quads per record loses to
// the staged kernel (k_onesweepv; measured: 15-byte wordcount records 0.28 against 0.35 of HBM, 22-byte
// join records 0.41 against 0.36).  Records of up to 16 bytes on average take the register kernel that
// holds a quad and a dword (0.357 against 0.348) as long as few of them are longer than that.
constexpr double kOsrMinCstrBytes = 18.0;
constexpr double kOsrQ3MaxBytes = 16.0;

// Enqueue one pack call.  noalloc: the control block and the chained-scan arrays live
// in the device's static scratch array, nothing is allocated, only the one-sweep
// kernels are available.
// Streams layout: d_send_image is the first of nranks destination streams of `cap` bytes each;
// init_offset = bytes already in each stream; round_base = byte of each stream at which the open
// round began (NULL: at init_offset, the open round is empty).
struct StreamsArgs {
    int64_t cap;
    const int64_t *round_base;
};

static int pack_enqueue(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off,
                        uint64_t n_end, uint64_t start, const int64_t *init_offset,
                        void *d_send_image, void *d_local_out, int64_t local_cap,
                        uint8_t *d_dest_out, int algo, void *stream_v, bool noalloc,
                        uint32_t max_rounds = 1, int64_t image_stride = 0, const StreamsArgs *sa = nullptr)
{
    if (!h) return MGS_ERR_ARG;
    const mgs_desc &D = h->desc;
    const int P = D.nranks;
    const int64_t part_cap = sa ? sa->cap : D.buf_size; // what the sweep may fill of each partition
    if (start > n_end) {
        h->err = "mgs_pack_device: start > n_end";
        return MGS_ERR_ARG;
    }
    if (!d_send_image || (!d_records && n_end > start)) {
        h->err = "mgs_pack_device: NULL device buffer";
        return MGS_ERR_ARG;
    }
    if (h->fixed_len == 0 && !d_rec_off) {
        h->err = "mgs_pack_device: variable-width layout needs d_rec_off";
        return MGS_ERR_ARG;
    }
    if (D.rank >= 0 && (!d_local_out || local_cap < 0)) {
        h->err = "mgs_pack_device: rank >= 0 needs d_local_out";
        return MGS_ERR_ARG;
    }
    for (int b = 0; b < P; b++) {
        const int64_t v = init_offset ? init_offset[b] : 0;
        if (v < 0 || v > part_cap) {
            h->err = fmt("mgs_pack_device: init_offset[%d]=%lld outside [0,%lld]", b, (long long) v, (long long) part_cap);
            return MGS_ERR_ARG;
        }
        if (sa && sa->round_base && b != D.rank
            && (sa->round_base[b] < 0 || sa->round_base[b] > v || v - sa->round_base[b] > D.buf_size)) {
            h->err = fmt("mgs_pack_streams_device: open_round_base[%d]=%lld must lie in [fill - buf_size, fill] (fill %lld)",
                         b, (long long) sa->round_base[b], (long long) v);
            return MGS_ERR_ARG;
        }
    }
    DeviceGuard g(D.device);
    if (!g.ok) {
        h->err = "mgs_pack_device: hipSetDevice failed";
        return MGS_ERR_HIP;
    }
    if (h->pending) { // one call in flight per handle
        MGS_HIP(h, hipStreamSynchronize(h->pending_stream));
        h->pending = false;
    }
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->own_stream;
    const uint64_t n = std::min<uint64_t>(n_end - start, D.max_records);
    Ctrl *d_ctrl = noalloc ? reinterpret_cast<Ctrl *>(h->d_noalloc) : h->d_ctrl;
    unsigned long long *d_status = nullptr, *d_agg16 = nullptr;
    // chained-scan arrays for `tiles` tiles of `columns` columns: the handle's buffers
    // (grown on demand), or a carve-up of the static scratch (refused when too small)
    auto scan_arrays = [&](int min_tile, int tile, int columns) -> int {
        if (!noalloc) {
            const int rc = ensure_onesweep_scratch(h, min_tile, columns);
            d_status = h->d_status;
            d_agg16 = h->d_agg16;
            return rc;
        }
        const uint64_t tiles = (n + (uint64_t) tile - 1) / (uint64_t) tile;
        const size_t sb = (tiles * (uint64_t) columns * sizeof(unsigned long long) + 255) / 256 * 256;
        const size_t ab = tiles * (uint64_t) ((columns + 3) / 4 * 2) * sizeof(unsigned long long);
        if (sb + ab > (size_t) MGS_NOALLOC_SCAN_BYTES) {
            h->err = fmt("mgs_pack_device_noalloc: %llu records in tiles of %d need %zu bytes of scan arrays, "
                         "the static scratch holds %zu: pass fewer records per call",
                         (unsigned long long) n, tile, sb + ab, (size_t) MGS_NOALLOC_SCAN_BYTES);
            return MGS_ERR_ARG;
        }
        d_status = reinterpret_cast<unsigned long long *>(h->d_noalloc + kNoallocCtrlBytes + kNoallocRoundBytes);
        d_agg16 = reinterpret_cast<unsigned long long *>(h->d_noalloc + kNoallocCtrlBytes + kNoallocRoundBytes + sb);
        return MGS_OK;
    };

    h->pending_start = start;
    for (int b = 0; b < MGS_MAX_RANKS; b++)
        h->pending_init[b] = (b < P && init_offset && b != D.rank) ? init_offset[b] : 0;
    h->pending_stream = stream;
    h->pending_empty = (n == 0);
    h->pending_rounds = 0;
    h->pending_streams = false;
    h->ev_valid = false;
    if (n == 0) {
        h->pending = true;
        h->pending_noalloc = noalloc;
        return MGS_OK;
    }

    // ---- choose the kernels -------------------------------------------------
    const uint8_t *rec0 = static_cast<const uint8_t *>(d_records);
    bool os_ok = h->fixed_len == 16 && P <= OS_MAXP && D.layout.key_kind == MGS_KEY_FIXED
                 && D.layout.key_bytes <= 16 && (D.buf_size % 16) == 0 && (part_cap % 16) == 0
                 && ((uintptr_t) rec0 % 16) == 0 && ((uintptr_t) d_send_image % 16) == 0
                 && (D.rank < 0 || ((uintptr_t) d_local_out % 16) == 0);
    for (int b = 0; b < P && os_ok; b++)
        os_ok = (h->pending_init[b] % 16) == 0 && (!sa || !sa->round_base || b == D.rank || (sa->round_base[b] % 16) == 0);
    // byte-granular one-sweep: chain columns = P (+1 for the local record count) <= 64, and for
    // fixed widths a tile that fits the LDS staging buffer (variable widths are checked on the device)
    const bool image_kernel = (sa != nullptr || osv_image_kernel(h, max_rounds)) && !noalloc;
    const bool ov_ok = P + (D.rank >= 0 ? 1 : 0) <= OS_MAXP
                       && (h->fixed_len == 0 || ov_pick_k(image_kernel, (double) h->fixed_len) != 0);
    // variable width: K follows the record length seen in earlier calls (mgs_pack_result);
    // the first call looks at the two ends of its offsets (two 4-byte copies, one wait)
    if (h->fixed_len == 0 && !h->ov_k_known && ov_ok && (algo == MGS_ALGO_AUTO || algo == MGS_ALGO_ONESWEEPV)) {
        MGS_HIP(h, hipMemcpyAsync(&h->h_probe[0], d_rec_off + start, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        MGS_HIP(h, hipMemcpyAsync(&h->h_probe[1], d_rec_off + start + n, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        MGS_HIP(h, hipStreamSynchronize(stream));
        if (h->h_probe[1] > h->h_probe[0]) {
            h->ov_bytes_per_record = (double) (h->h_probe[1] - h->h_probe[0]) / (double) n;
        }
        h->ov_k_known = true;
    }
    int ov_k = ov_pick_k(image_kernel, h->fixed_len ? (double) h->fixed_len : 1.25 * h->ov_bytes_per_record);
    if (ov_k == 0) ov_k = 1;
    if (h->fixed_len == 0 && h->ov_shrink > 0) // the LDS-image kernel exists for K = 4, 2, 1 only
        ov_k = image_kernel ? std::max(1, ov_k >> std::min(h->ov_shrink, 2)) : std::max(1, ov_k - h->ov_shrink);
    if (image_kernel && ov_k == 3) ov_k = 2;
    // the register kernel instead of the staged one: layout modes 1 and 2, records that fit its registers
    const mgs_layout &LY = D.layout;
    const int ov_lm = (LY.val_kind == MGS_VAL_CSTR) ? 0 : (LY.key_kind == MGS_KEY_CSTR ? (LY.key_count == 1 ? 1 : 0) : 2);
    bool use_osr = false, osr_a4 = false;
    int osr_k = 0, osr_nq = 0;
    if (image_kernel && ov_ok && ov_lm != 0 && osr_enabled(h) && P + (D.rank >= 0 ? 1 : 0) <= OS_MAXP) {
        if (h->fixed_len) {
            osr_a4 = (h->fixed_len % 4) == 0 && ((uintptr_t) d_send_image % 4) == 0 && (part_cap % 4) == 0
                     && (D.rank < 0 || ((uintptr_t) d_local_out % 4) == 0);
            for (int b = 0; b < P && osr_a4; b++) osr_a4 = (h->pending_init[b] % 4) == 0;
            // (the kernel hashes a fixed-width key out of a record's first 16 bytes)
            use_osr = h->fixed_len <= 32 && LY.key_bytes <= 16
                      && osr_pick(ov_lm, osr_a4, (double) h->fixed_len, 0, &osr_k, &osr_nq);
        }
        else {
            static const char *force = getenv("MGS_OSR_FORCE"); // experiments: whatever the record length
            const bool q3 = !h->osr_q3_off && h->ov_bytes_per_record <= kOsrQ3MaxBytes;
            use_osr = (h->ov_bytes_per_record >= kOsrMinCstrBytes || q3 || (force && atoi(force)))
                      && osr_pick(ov_lm, false, 1.25 * h->ov_bytes_per_record, h->ov_shrink, &osr_k, &osr_nq, q3);
        }
        if (use_osr) {
            // experiments: pin the instantiation (records per lane, register quads)
            static const char *ek = getenv("MGS_OSR_K"), *eq = getenv("MGS_OSR_NQ");
            const int k2 = ek ? atoi(ek) : osr_k, nq2 = eq ? atoi(eq) : osr_nq;
            if ((ek || eq) && osr_has(k2, nq2, ov_lm, osr_a4)) {
                osr_k = k2;
                osr_nq = nq2;
            }
            ov_k = osr_k;
        }
    }
    if (algo == MGS_ALGO_AUTO)
        algo = os_ok ? MGS_ALGO_ONESWEEP16 : (ov_ok ? MGS_ALGO_ONESWEEPV : MGS_ALGO_GENERIC);
    if (sa && algo == MGS_ALGO_GENERIC) {
        h->err = "mgs_pack_streams_device: this layout / rank count / alignment needs the generic kernels, which "
                 "pack one round per call";
        return MGS_ERR_ARG;
    }
    if (noalloc && algo == MGS_ALGO_GENERIC) {
        h->err = fmt("mgs_pack_device_noalloc: this layout / rank count needs the generic kernels, whose buffers "
                     "are allocated per handle (the call takes at most 64 ranks, fixed widths up to %d bytes or "
                     "variable width)", (OD_SB - OD_SLACK) / 64);
        return MGS_ERR_ARG;
    }
    if (algo == MGS_ALGO_ONESWEEPV && !ov_ok) {
        h->err = fmt("mgs_pack_device: ONESWEEPV needs nranks (+1 with a local rank) <= 64 and, for fixed "
                     "widths, records of at most %d bytes", image_kernel ? (OV_IN_BYTES - 15) / kOvLanes : (OD_SB - OD_SLACK) / 64);
        return MGS_ERR_ARG;
    }
    if (algo == MGS_ALGO_ONESWEEP16 && !os_ok) {
        h->err = "mgs_pack_device: ONESWEEP16 needs 16-byte fixed records, nranks <= 64 and "
                 "16-byte aligned buffers/offsets";
        return MGS_ERR_ARG;
    }
    if (algo != MGS_ALGO_ONESWEEP16 && algo != MGS_ALGO_GENERIC && algo != MGS_ALGO_ONESWEEPV) {
        h->err = "mgs_pack_device: unknown algo";
        return MGS_ERR_ARG;
    }
    if (h->fixed_len == 0 && n_end > 0) {
        // rec_off is 32-bit: the caller keeps a batch under 4 GiB
    }

    // ---- control block ---------------------------------------------------------
    CtrlHead *H = h->h_head;
    memset(H, 0, sizeof(*H));
    H->cut_index = ~0ull;
    H->cut_tile = 0xFFFFFFFFu;
    H->ticket = 0;
    H->error = 0;
    for (int b = 0; b < P; b++) {
        if (b == D.rank) {
            H->part_base[b] = (uint64_t)(uintptr_t) d_local_out;
            H->part_cap[b] = local_cap;
            H->part_init[b] = 0;
        }
        else {
            H->part_base[b] = (uint64_t)(uintptr_t) d_send_image + (uint64_t) b * (uint64_t) part_cap;
            H->part_cap[b] = part_cap;
            H->part_init[b] = h->pending_init[b];
            H->round_base0[b] = (sa && sa->round_base) ? sa->round_base[b] : h->pending_init[b];
        }
    }
    MGS_HIP(h, hipMemcpyAsync(d_ctrl, H, sizeof(CtrlHead), hipMemcpyHostToDevice, stream));

    if (algo == MGS_ALGO_ONESWEEP16) {
        int rc = scan_arrays(os16_min_tile(), kOsVariants[h->os_variant].tile, P);
        if (rc) return rc;
        OsParams p;
        p.rec = reinterpret_cast<const uint4 *>(rec0) + start;
        p.n = n;
        p.first = start;
        p.P = (uint32_t) P;
        p.self = D.rank;
        p.klen = (uint32_t) D.layout.key_bytes;
        for (int w = 0; w < 4; w++) {
            const int kb = std::min(4, std::max(0, D.layout.key_bytes - 4 * w));
            p.kmask[w] = kb == 4 ? 0xFFFFFFFFu : ((1u << (8 * kb)) - 1u);
        }
        p.nbits = 1;
        while ((1u << p.nbits) < (uint32_t) P) p.nbits++;
        p.nbits_pad = (P == 1) ? 0 : p.nbits;
        p.fm = h->fm;
        const OsVariant &V = kOsVariants[h->os_variant];
        p.tile_records = (uint32_t) V.tile;
        p.ntiles = (uint32_t)((n + p.tile_records - 1) / p.tile_records);
        p.status = d_status;
        p.agg16 = d_agg16;
        p.lpt_log = p.nbits_pad > 2 ? p.nbits_pad - 2 : 0;
        p.unit_shift = 4;
        p.dest_out = d_dest_out;
        p.ctrl = d_ctrl;
        p.rt = p.rt_start = nullptr;
        p.max_rounds = 1;
        p.image_stride = 0;
        p.finder = 0;
        p.f_max_rounds = 0;
        p.f_cap = 0;
        p.f_rt = p.f_rt_start = p.clip_out = nullptr;
        p.seg = nullptr;
        MGS_HIP(h, hipMemsetAsync(d_status, 0, (size_t) p.ntiles * P * sizeof(unsigned long long), stream));
        if (V.agg16)
            MGS_HIP(h, hipMemsetAsync(d_agg16, 0, ((size_t) p.ntiles << p.lpt_log) * sizeof(unsigned long long), stream));
        uint32_t grid = std::min<uint32_t>(p.ntiles, (uint32_t)(h->num_cus * V.blocks_per_cu));
        unsigned long long *d_tab = nullptr;
        if (sa) {
            // streams layout: the sweep packs one round (the streams); workgroup 0 finds the rounds
            if (!V.agg16 || noalloc) {
                h->err = "the streams layout needs the production one-sweep kernel and the handle's own buffers";
                return MGS_ERR_ARG;
            }
            if (V.tile > FR_MAX_TILE) {
                h->err = "this tuning variant's tile is larger than the round finder's tables";
                return MGS_ERR_ARG;
            }
            rc = ensure_round_tables(h, max_rounds, P, P, true);
            if (rc) return rc;
            p.f_rt = h->d_rt;
            p.f_rt_start = p.f_rt + ((size_t) max_rounds + 1) * P;
            p.rt = p.f_rt_start + ((size_t) max_rounds + 1);
            p.rt_start = p.rt + 2 * (size_t) P;
            p.clip_out = p.rt_start + 2;
            p.max_rounds = 1;
            p.finder = 1;
            p.f_max_rounds = max_rounds;
            p.f_cap = (unsigned long long) (D.buf_size >> 4);
            // the sweep's units per (tile, worker wave, destination): the finder hashes only the
            // 512 records around a cut again
            {
                const size_t seg_bytes = (size_t) p.ntiles * (size_t) (V.threads / 64 - 1) * (size_t) P * sizeof(uint16_t);
                if (h->seg_bytes < seg_bytes) {
                    if (h->d_seg) MGS_HIP(h, hipFree(h->d_seg));
                    h->d_seg = nullptr;
                    h->seg_bytes = 0;
                    MGS_HIP(h, hipMalloc(&h->d_seg, seg_bytes));
                    h->seg_bytes = seg_bytes;
                }
                p.seg = h->d_seg;
                MGS_HIP(h, hipMemsetAsync(h->d_seg, 0xFF, seg_bytes, stream));
            }
            d_tab = h->d_rtab;
            MGS_HIP(h, hipMemsetAsync(h->d_rt, 0, round_table_words(max_rounds, P, true) * sizeof(unsigned long long), stream));
            grid = std::min<uint32_t>(p.ntiles, (uint32_t) (h->num_cus * V.blocks_per_cu) - 1u) + 1u;
        }
        else if (V.agg16) {
            // the round-cutting kernel: round table + result table
            if (noalloc) {
                unsigned long long *base = reinterpret_cast<unsigned long long *>(h->d_noalloc + kNoallocCtrlBytes);
                p.rt = base;
                d_tab = base + 2 * ((size_t) P + 1);
                static_assert((2 * (OS_MAXP + 1) + OS_RTAB_HEAD + 3 + OS_MAXP) * 8 <= kNoallocRoundBytes, "round region");
            }
            else {
                rc = ensure_round_tables(h, max_rounds, P, P);
                if (rc) return rc;
                p.rt = h->d_rt;
                d_tab = h->d_rtab;
            }
            p.rt_start = p.rt + ((size_t) max_rounds + 1) * P;
            p.max_rounds = max_rounds;
            p.image_stride = (uint64_t) image_stride;
            MGS_HIP(h, hipMemsetAsync(p.rt, 0, ((size_t) max_rounds + 1) * ((size_t) P + 1) * sizeof(unsigned long long), stream));
        }
        else if (max_rounds != 1) {
            h->err = "this tuning variant of the one-sweep kernel packs one round per call";
            return MGS_ERR_ARG;
        }
        MGS_HIP(h, hipEventRecord(h->ev0, stream));
        hipLaunchKernelGGL(V.kernel[p.nbits], dim3(grid), dim3(V.threads), 0, stream, p);
        MGS_HIP(h, hipEventRecord(h->ev1, stream));
        if (sa) {
            // the sweep's one round into a scratch table (its clip words are what matters), then the rounds
            unsigned long long *d_tab1 = d_tab + os_stab_words(max_rounds, (uint32_t) P) - os_rtab_words(1, (uint32_t) P);
            hipLaunchKernelGGL(k_os_rounds_final, dim3(1), dim3(256), 0, stream, p, (uint32_t) P,
                               (uint32_t) (D.rank >= 0 ? D.rank : 0), d_tab1);
            const size_t items = (size_t) max_rounds * ((size_t) P + 1);
            hipLaunchKernelGGL(k_os_streams_final, dim3((unsigned) std::min<size_t>(256, (items + 255) / 256)), dim3(256), 0,
                               stream, p, (uint32_t) P, (uint32_t) (D.rank >= 0 ? D.rank : 0), d_tab);
            MGS_HIP(h, hipMemcpyAsync(h->h_rtab, d_tab, os_stab_words(max_rounds, (uint32_t) P) * sizeof(unsigned long long),
                                      hipMemcpyDeviceToHost, stream));
            h->pending_rounds = max_rounds;
            h->pending_streams = true;
        }
        else if (V.agg16) {
            const size_t items = (size_t) max_rounds * ((size_t) P + 1);
            hipLaunchKernelGGL(k_os_rounds_final, dim3((unsigned) std::min<size_t>(256, (items + 255) / 256)), dim3(256), 0,
                               stream, p, (uint32_t) P, (uint32_t) (D.rank >= 0 ? D.rank : 0), d_tab);
            MGS_HIP(h, hipMemcpyAsync(h->h_rtab, d_tab, os_rtab_words(max_rounds, (uint32_t) P) * sizeof(unsigned long long),
                                      hipMemcpyDeviceToHost, stream));
            h->pending_rounds = max_rounds;
        }
        else {
            hipLaunchKernelGGL(k_os_finalize, dim3(1), dim3(64), 0, stream, p);
        }
        h->last_algo = "onesweep16";
        h->last_kernel = "k_onesweep16";
        h->pending_osv = false;
        h->pending_osr = false;
    }
    else {
        const bool osv = algo == MGS_ALGO_ONESWEEPV;
        // variable width: a wave's records may be too long for its LDS slice; the one-sweep
        // kernels then leave at once and the generic kernels enqueued behind them do the work.
        // Without them behind (no-allocation call, several rounds) mgs_pack_result reports it.
        const bool check_spans = osv && h->fixed_len == 0;
        const bool may_fall_back = false; // the generic kernels are enqueued by mgs_pack_result when needed
        h->pending_osv = osv;
        if (!osv && max_rounds != 1) {
            h->err = "the generic kernels pack one round per call";
            return MGS_ERR_ARG;
        }
        if (osv) {
            const bool legacy = image_kernel;
            if (legacy && max_rounds != 1 && !sa) {
                h->err = "the LDS-image kernel packs one round per call";
                return MGS_ERR_ARG;
            }
            const int columns = P + (D.rank >= 0 ? 1 : 0);
            const int kOvTile = ov_tile(ov_k);
            int rc = scan_arrays(kOvTile, kOvTile, columns);
            if (rc) return rc;
            OvParams q;
            memset(&q, 0, sizeof(q));
            q.blob = rec0;
            q.rec_off = h->fixed_len ? nullptr : d_rec_off;
            q.fixed_len = h->fixed_len;
            q.first = start;
            q.n = n;
            q.lay = D.layout;
            q.P = (uint32_t) P;
            OsParams &c = q.os;
            c.n = n;
            c.first = start;
            c.P = (uint32_t) columns;
            c.self = D.rank;
            c.nbits = 1;
            while ((1u << c.nbits) < (uint32_t) columns) c.nbits++;
            c.nbits_pad = (columns == 1) ? 0 : c.nbits;
            c.lpt_log = c.nbits_pad > 2 ? c.nbits_pad - 2 : 0;
            c.unit_shift = 0;
            c.fm = h->fm;
            c.tile_records = (uint32_t) kOvTile;
            c.ntiles = (uint32_t) ((n + kOvTile - 1) / kOvTile);
            c.status = d_status;
            c.agg16 = d_agg16;
            c.dest_out = d_dest_out;
            c.ctrl = d_ctrl;
            MGS_HIP(h, hipMemsetAsync(d_status, 0, (size_t) c.ntiles * columns * sizeof(unsigned long long), stream));
            MGS_HIP(h, hipMemsetAsync(d_agg16, 0, ((size_t) c.ntiles << c.lpt_log) * sizeof(unsigned long long), stream));
            const uint32_t wg_per_cu = 2u;
            q.in_bytes = (uint32_t) OV_IN_BYTES;
            uint32_t grid = std::min<uint32_t>(c.ntiles, (uint32_t) h->num_cus * wg_per_cu);
            if (sa) {
                // streams layout: the sweep packs one round (the streams); workgroup 0 finds the rounds
                rc = ensure_round_tables(h, max_rounds, columns, P, true);
                if (rc) return rc;
                c.f_rt = h->d_rt;
                c.f_rt_start = c.f_rt + ((size_t) max_rounds + 1) * columns;
                c.clip_out = c.f_rt_start + ((size_t) max_rounds + 1) + 2 * ((size_t) columns + 1);
                c.finder = 1;
                c.f_max_rounds = max_rounds;
                c.f_cap = (unsigned long long) D.buf_size;
                MGS_HIP(h, hipMemsetAsync(h->d_rt, 0, round_table_words(max_rounds, columns, true) * sizeof(unsigned long long),
                                          stream));
                grid = std::min<uint32_t>(c.ntiles, (uint32_t) h->num_cus * wg_per_cu - 1u) + 1u;
            }
            const mgs_layout &L = D.layout;
            // layout mode 1: ONE string key and a fixed-width or absent value
            const int lm = (L.val_kind == MGS_VAL_CSTR) ? 0 : (L.key_kind == MGS_KEY_CSTR ? (L.key_count == 1 ? 1 : 0) : 2);
            if (legacy) {
                MGS_HIP(h, hipEventRecord(h->ev0, stream));
                if (use_osr) {
                    q.in_bytes = (uint32_t) osr_image_bytes(osr_nq);
                    static const char *sv = getenv("MGS_OSR_STATS");
                    static const char *pv = getenv("MGS_OSR_PRE"); // experiments: pin early / late requests
                    static const char *fv = getenv("MGS_OSR_FAKECHAIN"); // timing only, wrong output
                    // MGS_OSR_STATS: 1 = worker wave 0 reports, 2 = the last worker wave
                    const int var = ((sv && atoi(sv)) ? (atoi(sv) == 2 ? OSV_STATS | OSV_NO_PREFETCH : OSV_STATS) : 0)
                                    | ((fv && atoi(fv)) ? OSV_FAKE_CHAIN : 0);
                    if (!osr_launch(q, osr_k, osr_nq, lm, osr_a4, pv ? atoi(pv) : -1, var, grid, stream)) {
                        h->err = "mgs_pack_device: no k_onesweepr instantiation for this call";
                        return MGS_ERR_ARG;
                    }
                }
                else {
                    osv_launch(q, ov_k, lm, grid, stream);
                }
                if (!may_fall_back) MGS_HIP(h, hipEventRecord(h->ev1, stream));
                hipLaunchKernelGGL(k_osv_finalize, dim3(1), dim3(64), 0, stream, q);
                if (sa) {
                    const size_t items = (size_t) max_rounds * ((size_t) P + 1);
                    hipLaunchKernelGGL(k_os_streams_final, dim3((unsigned) std::min<size_t>(256, (items + 255) / 256)),
                                       dim3(256), 0, stream, c, (uint32_t) P, (uint32_t) P, h->d_rtab);
                    MGS_HIP(h, hipMemcpyAsync(h->h_rtab, h->d_rtab,
                                              os_stab_words(max_rounds, (uint32_t) P) * sizeof(unsigned long long),
                                              hipMemcpyDeviceToHost, stream));
                    h->pending_rounds = max_rounds;
                    h->pending_streams = true;
                }
            }
            else {
                // the direct kernel: round table + result table, as for 16-byte records
                unsigned long long *d_tab = nullptr;
                if (noalloc) {
                    unsigned long long *base = reinterpret_cast<unsigned long long *>(h->d_noalloc + kNoallocCtrlBytes);
                    c.rt = base;
                    d_tab = base + 2 * ((size_t) columns + 1);
                }
                else {
                    rc = ensure_round_tables(h, max_rounds, columns, P);
                    if (rc) return rc;
                    c.rt = h->d_rt;
                    d_tab = h->d_rtab;
                }
                c.rt_start = c.rt + ((size_t) max_rounds + 1) * columns;
                c.max_rounds = max_rounds;
                c.image_stride = (uint64_t) image_stride;
                MGS_HIP(h, hipMemsetAsync(c.rt, 0, ((size_t) max_rounds + 1) * ((size_t) columns + 1) * sizeof(unsigned long long), stream));
                if (check_spans) {
                    const uint64_t items = (uint64_t) c.ntiles * (kOvThreads / 64 - 1);
                    (void) items;
                    osd_check_launch(q, (uint32_t) (64 * ov_k), (uint32_t) (kOvThreads / 64 - 1), c.ntiles, stream);
                }
                MGS_HIP(h, hipEventRecord(h->ev0, stream));
                const char *sv = getenv("MGS_OSD_FAKECHAIN"); // timing only: no chained scan, wrong output
                const char *cv = getenv("MGS_OSD_CHECKED");   // debugging: stores outside the buffers are counted, not made
                int ki = (sv && atoi(sv) && ov_k == 4) ? 4 : ov_k - 1;
                if (const char *fv = getenv("MGS_OSD_DBG_FLAGS")) q.dbg_flags = (uint32_t) atoi(fv); // timing experiments
                if (const char *tv = getenv("MGS_OSD_STATS")) // diagnostic: per-phase cycles in Ctrl::head.stats
                    if (atoi(tv)) ki = (atoi(tv) == 2 ? 12 : 8) + ov_k; // 2: the scan wave's phases
                if (const char *lv = getenv("MGS_OSD_LOOK")) // experiment: 16 loads per poll of the chained scan
                    if (atoi(lv) == 16) ki = (ki >= 13 ? 20 : 16) + ov_k;
                if (cv && atoi(cv)) {
                    ki = 4 + ov_k;
                    q.dbg_lo[0] = (uint64_t) (uintptr_t) d_send_image;
                    q.dbg_hi[0] = q.dbg_lo[0] + (uint64_t) (max_rounds - 1) * (uint64_t) image_stride
                                  + (uint64_t) P * (uint64_t) D.buf_size;
                    q.dbg_lo[1] = (uint64_t) (uintptr_t) d_local_out;
                    q.dbg_hi[1] = q.dbg_lo[1] + (uint64_t) (d_local_out ? local_cap : 0);
                }
                osd_launch(q, ki, lm, grid, stream);
                if (!may_fall_back) MGS_HIP(h, hipEventRecord(h->ev1, stream));
                const size_t items = (size_t) max_rounds * ((size_t) P + 1);
                hipLaunchKernelGGL(k_os_rounds_final, dim3((unsigned) std::min<size_t>(256, (items + 255) / 256)), dim3(256),
                                   0, stream, c, (uint32_t) P, (uint32_t) P, d_tab);
                MGS_HIP(h, hipMemcpyAsync(h->h_rtab, d_tab, os_rtab_words(max_rounds, (uint32_t) P) * sizeof(unsigned long long),
                                          hipMemcpyDeviceToHost, stream));
                h->pending_rounds = max_rounds;
            }
            h->pending_ov_k = ov_k;
            h->pending_osr = legacy && use_osr;
            h->pending_osr_nq = osr_nq;
            h->last_algo = legacy ? "onesweepv" : "onesweepd";
            h->last_kernel = legacy ? (use_osr ? "k_onesweepr" : "k_onesweepv") : "k_onesweepd";
        }
        if (!osv || may_fall_back) {
        int rc = ensure_generic_scratch(h, d_dest_out == nullptr);
        if (rc) return rc;
        GenParams p;
        p.gate = may_fall_back ? 1u : 0u;
        p.blob = rec0;
        p.rec_off = h->fixed_len ? nullptr : d_rec_off;
        p.fixed_len = h->fixed_len;
        p.first = start;
        p.n = n;
        p.lay = D.layout;
        p.P = (uint32_t) P;
        p.self = D.rank;
        p.fm = h->fm;
        p.ntiles = (uint32_t)((n + GEN_TILE - 1) / GEN_TILE);
        p.tile_hist = h->d_tile_hist;
        p.tile_base = h->d_tile_base;
        p.dest = d_dest_out ? d_dest_out : h->d_dest;
        p.ctrl = d_ctrl;
        hipLaunchKernelGGL(k_gen_hist, dim3(p.ntiles), dim3(GEN_THREADS), 0, stream, p);
        hipLaunchKernelGGL(k_gen_scan, dim3(1), dim3(1024), 0, stream, p);
        if (!osv) MGS_HIP(h, hipEventRecord(h->ev0, stream));
        hipLaunchKernelGGL(k_gen_scatter, dim3(p.ntiles), dim3(GEN_THREADS), 0, stream, p);
        MGS_HIP(h, hipEventRecord(h->ev1, stream));
        hipLaunchKernelGGL(k_gen_finalize, dim3(1), dim3(GEN_THREADS), 0, stream, p);
        if (!osv) {
            h->last_algo = "generic";
            h->last_kernel = "k_gen_scatter";
            h->pending_osr = false;
        }
        }
    }
    MGS_HIP(h, hipGetLastError());
    MGS_HIP(h, hipMemcpyAsync(h->h_ctrl, d_ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, stream));
    h->pending = true;
    h->pending_noalloc = noalloc;
    h->redo = {d_records, d_rec_off, n_end, d_send_image, d_local_out, local_cap, d_dest_out};
    return MGS_OK;
}

int mgs_pack_device_async(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off,
                          uint64_t n_end, uint64_t start, const int64_t *init_offset,
                          void *d_send_image, void *d_local_out, int64_t local_cap,
                          uint8_t *d_dest_out, int algo, void *stream_v)
{
    return pack_enqueue(h, d_records, d_rec_off, n_end, start, init_offset, d_send_image, d_local_out,
                        local_cap, d_dest_out, algo, stream_v, false);
}

int mgs_pack_device_noalloc(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off,
                            uint64_t n_end, uint64_t start, const int64_t *init_offset,
                            void *d_send_image, void *d_local_out, int64_t local_cap,
                            uint8_t *d_dest_out, void *stream, mgs_round *out)
{
    int rc = pack_enqueue(h, d_records, d_rec_off, n_end, start, init_offset, d_send_image, d_local_out,
                          local_cap, d_dest_out, MGS_ALGO_AUTO, stream, true);
    if (rc) return rc;
    return mgs_pack_result(h, out);
}

static int pack_result_impl(mgs_handle *h, mgs_round *out, bool rounds_caller);

int mgs_pack_result(mgs_handle *h, mgs_round *out) { return pack_result_impl(h, out, false); }

// rounds_caller: mgs_pack_rounds_result reads the whole result table itself
static int pack_result_impl(mgs_handle *h, mgs_round *out, bool rounds_caller)
{
    if (!h || !out) return MGS_ERR_ARG;
    if (!h->pending) {
        h->err = "mgs_pack_result: no pack call in flight";
        return MGS_ERR_STATE;
    }
    if (!rounds_caller && !h->pending_empty && (h->pending_rounds > 1 || h->pending_streams)) {
        // the call stays in flight: its own result call can still fetch it
        h->err = "mgs_pack_result: the call in flight packs several rounds (use mgs_pack_rounds_result / "
                 "mgs_pack_streams_result)";
        return MGS_ERR_STATE;
    }
    h->pending = false;
    const int P = h->desc.nranks;
    memset(out, 0, sizeof(*out));
    if (h->pending_empty) {
        out->next = h->pending_start;
        for (int b = 0; b < P; b++) out->send_offset[b] = h->pending_init[b];
        return MGS_OK;
    }
    DeviceGuard g(h->desc.device);
    MGS_HIP(h, hipStreamSynchronize(h->pending_stream));
    h->ev_valid = true;
    if (h->pending_osv && h->h_ctrl->head.fallback != 0) {
        // records too long for the one-sweep kernel's LDS staging: it did nothing
        h->ov_shrink++; // the next call tries a smaller tile (and, when none is left, no one-sweep kernel at all)
        if (h->pending_noalloc) {
            h->err = "mgs_pack_device_noalloc: a wave's records do not fit the kernel's LDS staging buffer "
                     "(records too long); use mgs_pack_device";
            return MGS_ERR_RECORD_SIZE;
        }
        if (h->pending_rounds > 1 || h->pending_streams) {
            h->err = "records too long for the one-sweep kernel's LDS staging: pack this batch round by round";
            return MGS_ERR_RECORD_SIZE;
        }
        // the generic kernels redo the call (same stream, same arguments)
        int64_t init[MGS_MAX_RANKS];
        memcpy(init, h->pending_init, sizeof(init));
        const int rc = pack_enqueue(h, h->redo.d_records, h->redo.d_rec_off, h->redo.n_end, h->pending_start, init,
                                    h->redo.d_send_image, h->redo.d_local_out, h->redo.local_cap, h->redo.d_dest_out,
                                    MGS_ALGO_GENERIC, h->pending_stream, false);
        if (rc) return rc;
        return pack_result_impl(h, out, false);
    }
    const int derr = h->h_ctrl->head.error;
    if (derr != 0) {
        switch (derr) {
        case MGS_ERR_RECORD_SIZE:
            h->err = fmt("KV size is larger than buf_size (%lld)", (long long) h->desc.buf_size);
            break;
        case MGS_ERR_LOCAL_OVERFLOW: h->err = "local bypass stream overflow (local_cap too small)"; break;
        case MGS_ERR_MALFORMED: h->err = "a record does not match the declared layout"; break;
        case MGS_ERR_TIMEOUT: h->err = "chained scan gave up waiting for a predecessor tile"; break;
        default: h->err = fmt("device error %d", derr);
        }
        return derr;
    }
    if (h->pending_rounds) {
        // a round-cutting kernel: round 0 of its result table (a one-round call ends at the
        // first record that does not fit, src/collectiveshuffler.h:148)
        const unsigned long long *tab = h->h_rtab;
        const size_t R = h->pending_rounds;
        out->next = tab[1];
        out->packed_records = tab[2];
        out->local_records = tab[3];
        out->local_bytes = (int64_t) tab[4];
        for (int b = 0; b < P; b++) out->send_offset[b] = (int64_t) tab[OS_RTAB_HEAD + 3 * R + b];
    }
    else {
        *out = h->h_ctrl->result;
    }
    for (int b = P; b < MGS_MAX_RANKS; b++) out->send_offset[b] = 0;
    if (h->fixed_len == 0) {
        // Variable width: the next call picks its tile from this call's record length
        const uint64_t recs = out->next - h->pending_start;
        int64_t bytes = out->local_bytes;
        for (int b = 0; b < P; b++) bytes += out->send_offset[b] - h->pending_init[b];
        if (recs > 0 && bytes > 0 && h->pending_rounds <= 1 && !h->pending_streams) h->ov_bytes_per_record = (double) bytes / (double) recs;
        if (h->pending_osv) h->ov_shrink = 0; // a one-sweep kernel served the call: the length alone decides again
        // records longer than a quad and a dword took the slow way round: more than one in 64 and the
        // handle goes back to the kernels that hold whole records
        const uint64_t recs_call = h->pending_rounds ? (uint64_t) h->h_rtab[1] - h->pending_start : recs;
        if (h->pending_osr && h->pending_osr_nq == 3 && (uint64_t) h->h_ctrl->head.long_records * 64u > recs_call)
            h->osr_q3_off = true;

    }
    return MGS_OK;
}

int mgs_pack_device(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off,
                    uint64_t n_end, uint64_t start, const int64_t *init_offset,
                    void *d_send_image, void *d_local_out, int64_t local_cap,
                    uint8_t *d_dest_out, int algo, void *stream, mgs_round *out)
{
    int rc = mgs_pack_device_async(h, d_records, d_rec_off, n_end, start, init_offset,
                                   d_send_image, d_local_out, local_cap, d_dest_out, algo, stream);
    if (rc) return rc;
    return mgs_pack_result(h, out);
}

// ---- many rounds per call ------------------------------------------------------------
// Which kernel cuts the rounds of this call itself?  MGS_ALGO_ONESWEEP16 / MGS_ALGO_ONESWEEPV
// (same conditions as in pack_enqueue), or 0: one kernel sequence per round.
static int rounds_in_kernel(const mgs_handle *h, const void *d_records, void *d_send_images, int64_t image_stride,
                            void *d_local_out, const int64_t *init_offset, int algo, uint32_t max_rounds)
{
    const mgs_desc &D = h->desc;
    bool os16 = (algo == MGS_ALGO_AUTO || algo == MGS_ALGO_ONESWEEP16) && kOsVariants[h->os_variant].agg16
                && h->fixed_len == 16 && D.nranks <= OS_MAXP && D.layout.key_kind == MGS_KEY_FIXED
                && D.layout.key_bytes <= 16 && (D.buf_size % 16) == 0 && ((uintptr_t) d_records % 16) == 0
                && ((uintptr_t) d_send_images % 16) == 0 && (image_stride % 16) == 0
                && (D.rank < 0 || ((uintptr_t) d_local_out % 16) == 0);
    for (int b = 0; b < D.nranks && os16 && init_offset; b++) os16 = (init_offset[b] % 16) == 0;
    if (os16) return MGS_ALGO_ONESWEEP16;
    if (algo == MGS_ALGO_ONESWEEP16) return 0;
    // A handle whose last one-sweep call met records too long for the staging buffer packs
    // round by round (the generic kernels measure the record length call by call) until that
    // length says that the smallest tile fits again.
    if (h->fixed_len == 0 && h->ov_shrink > 0 && od_pick_k(1.25 * h->ov_bytes_per_record) == 0) return 0;
    // (a one-round call of the byte-granular layouts runs the LDS-image kernel through mgs_pack_device)
    const bool osd = (algo == MGS_ALGO_AUTO || algo == MGS_ALGO_ONESWEEPV) && !osv_image_kernel(h, max_rounds)
                     && D.nranks + (D.rank >= 0 ? 1 : 0) <= OS_MAXP
                     && (h->fixed_len == 0 || od_pick_k((double) h->fixed_len) != 0);
    return osd ? MGS_ALGO_ONESWEEPV : 0;
}

int mgs_pack_rounds_device_async(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off, uint64_t n_end,
                                 uint64_t start, const int64_t *init_offset, void *d_send_images,
                                 int64_t image_stride, uint32_t max_rounds, void *d_local_out, int64_t local_cap,
                                 uint8_t *d_dest_out, int algo, void *stream)
{
    if (!h) return MGS_ERR_ARG;
    if (max_rounds < 1 || image_stride < (int64_t) h->desc.nranks * h->desc.buf_size) {
        h->err = "mgs_pack_rounds_device: max_rounds >= 1 and image_stride >= nranks * buf_size required";
        return MGS_ERR_ARG;
    }
    const int in_kernel = rounds_in_kernel(h, d_records, d_send_images, image_stride, d_local_out, init_offset, algo,
                                           max_rounds);
    if (!in_kernel) {
        h->err = "mgs_pack_rounds_device_async: this layout / rank count packs one round per kernel sequence; "
                 "use the blocking mgs_pack_rounds_device";
        return MGS_ERR_ARG;
    }
    return pack_enqueue(h, d_records, d_rec_off, n_end, start, init_offset, d_send_images, d_local_out, local_cap,
                        d_dest_out, in_kernel, stream, false, max_rounds, image_stride);
}

static int rounds_result_impl(mgs_handle *h, mgs_rounds *out, uint64_t *round_next, int64_t *round_send_offset,
                              int64_t *round_base, uint64_t *round_local_records, int64_t *round_local_bytes);

int mgs_pack_rounds_result(mgs_handle *h, mgs_rounds *out, uint64_t *round_next, int64_t *round_send_offset,
                           uint64_t *round_local_records, int64_t *round_local_bytes)
{
    if (h && h->pending && h->pending_streams) {
        h->err = "mgs_pack_rounds_result: the call in flight uses the streams layout (use mgs_pack_streams_result)";
        return MGS_ERR_STATE;
    }
    return rounds_result_impl(h, out, round_next, round_send_offset, nullptr, round_local_records, round_local_bytes);
}

int mgs_pack_streams_result(mgs_handle *h, mgs_rounds *out, uint64_t *round_next, int64_t *round_send_offset,
                            int64_t *round_base, uint64_t *round_local_records, int64_t *round_local_bytes)
{
    if (h && h->pending && !h->pending_streams && !h->pending_empty) {
        h->err = "mgs_pack_streams_result: the call in flight does not use the streams layout";
        return MGS_ERR_STATE;
    }
    return rounds_result_impl(h, out, round_next, round_send_offset, round_base, round_local_records, round_local_bytes);
}

static int rounds_result_impl(mgs_handle *h, mgs_rounds *out, uint64_t *round_next, int64_t *round_send_offset,
                              int64_t *round_base, uint64_t *round_local_records, int64_t *round_local_bytes)
{
    if (!h || !out) return MGS_ERR_ARG;
    if (!h->pending) {
        h->err = "mgs_pack_rounds_result: no pack call in flight";
        return MGS_ERR_STATE;
    }
    const uint32_t R = h->pending_rounds;
    const int P = h->desc.nranks;
    memset(out, 0, sizeof(*out));
    if (h->pending_empty) {
        h->pending = false;
        out->n_rounds = 1;
        out->next = h->pending_start;
        if (round_next) round_next[0] = h->pending_start;
        if (round_local_records) round_local_records[0] = 0;
        if (round_local_bytes) round_local_bytes[0] = 0;
        if (round_send_offset)
            for (int b = 0; b < P; b++) round_send_offset[b] = h->pending_init[b];
        if (round_base)
            for (int b = 0; b < P; b++) round_base[b] = h->pending_base[b];
        if (h->pending_streams && round_send_offset)
            for (int b = 0; b < P; b++) round_send_offset[b] = h->pending_init[b] - h->pending_base[b];
        return MGS_OK;
    }
    if (R == 0) {
        h->err = "mgs_pack_rounds_result: the call in flight is a one-round call (use mgs_pack_result)";
        return MGS_ERR_STATE;
    }
    mgs_round r0; // errors, and round 0
    const int rc = pack_result_impl(h, &r0, true);
    if (rc) return rc;
    if (h->pending_rounds == 0) { // a one-round call that the generic kernels had to redo
        out->n_rounds = 1;
        out->next = r0.next;
        out->packed_records = r0.packed_records;
        out->local_records = r0.local_records;
        out->local_bytes = r0.local_bytes;
        if (round_next) round_next[0] = r0.next;
        if (round_local_records) round_local_records[0] = r0.local_records;
        if (round_local_bytes) round_local_bytes[0] = r0.local_bytes;
        if (round_send_offset)
            for (int b = 0; b < P; b++) round_send_offset[b] = r0.send_offset[b];
        return MGS_OK;
    }
    const unsigned long long *tab = h->h_rtab;
    out->n_rounds = (uint32_t) tab[0];
    out->next = tab[1];
    out->packed_records = tab[2];
    out->local_records = tab[3];
    out->local_bytes = (int64_t) tab[4];
    for (uint32_t k = 0; k < out->n_rounds; k++) {
        if (round_next) round_next[k] = tab[OS_RTAB_HEAD + k];
        if (round_local_records) round_local_records[k] = tab[OS_RTAB_HEAD + (size_t) R + k];
        if (round_local_bytes) round_local_bytes[k] = (int64_t) tab[OS_RTAB_HEAD + 2 * (size_t) R + k];
        if (round_send_offset)
            for (int b = 0; b < P; b++)
                round_send_offset[(size_t) k * P + b] = (int64_t) tab[OS_RTAB_HEAD + 3 * (size_t) R + (size_t) k * P + b];
        if (round_base && h->pending_streams)
            for (int b = 0; b < P; b++)
                round_base[(size_t) k * P + b] =
                    (int64_t) tab[OS_RTAB_HEAD + 3 * (size_t) R + (size_t) R * P + (size_t) k * P + b];
    }
    return MGS_OK;
}

int mgs_pack_streams_device_async(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off, uint64_t n_end,
                                  uint64_t start, const int64_t *stream_fill, const int64_t *open_round_base,
                                  void *d_streams, int64_t stream_cap, uint32_t max_rounds, void *d_local_out,
                                  int64_t local_cap, uint8_t *d_dest_out, void *stream)
{
    if (!h) return MGS_ERR_ARG;
    if (max_rounds < 1 || stream_cap < 1) {
        h->err = "mgs_pack_streams_device: max_rounds >= 1 and stream_cap >= 1 required";
        return MGS_ERR_ARG;
    }
    const mgs_desc &D = h->desc;
    const bool os16 = h->fixed_len == 16 && D.nranks <= OS_MAXP && D.layout.key_kind == MGS_KEY_FIXED
                      && D.layout.key_bytes <= 16;
    const bool osv = D.nranks + (D.rank >= 0 ? 1 : 0) <= OS_MAXP
                     && (h->fixed_len == 0 || ov_pick_k_legacy((double) h->fixed_len) != 0);
    if (!os16 && !osv) {
        h->err = "mgs_pack_streams_device: at most 64 ranks (63 with a local rank unless records are 16 bytes), fixed "
                 "widths up to 79 bytes or variable width";
        return MGS_ERR_ARG;
    }
    const StreamsArgs sa = {stream_cap, open_round_base};
    for (int b = 0; b < MGS_MAX_RANKS; b++) {
        const int64_t fill = (stream_fill && b < D.nranks && b != D.rank) ? stream_fill[b] : 0;
        h->pending_base[b] = (open_round_base && b < D.nranks && b != D.rank) ? open_round_base[b] : fill;
    }
    const int rc = pack_enqueue(h, d_records, d_rec_off, n_end, start, stream_fill, d_streams, d_local_out, local_cap,
                                d_dest_out, MGS_ALGO_AUTO, stream, false, max_rounds, 0, &sa);
    if (rc == MGS_OK && h->pending_empty) h->pending_streams = true;
    return rc;
}

// The streams layout for records the one-sweep kernels do not take (a tile's records longer
// than their LDS image): one mgs_pack_device call per round with the generic kernels, which
// take any record up to buf_size as CollectiveShuffler::write does
// (src/collectiveshuffler.h:131-166), into a send image of the handle's own; what a call
// added to a partition is appended to that destination's stream.  A round is only begun
// when every stream has room for a whole partition.
static int streams_round_by_round(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off, uint64_t n_end,
                                  uint64_t start, const int64_t *stream_fill, const int64_t *open_round_base,
                                  void *d_streams, int64_t stream_cap, uint32_t max_rounds, void *d_local_out,
                                  int64_t local_cap, uint8_t *d_dest_out, void *stream, mgs_rounds *out,
                                  uint64_t *round_next, int64_t *round_send_offset, int64_t *round_base,
                                  uint64_t *round_local_records, int64_t *round_local_bytes)
{
    const int P = h->desc.nranks, me = h->desc.rank;
    const int64_t buf = h->desc.buf_size;
    DeviceGuard g(h->desc.device);
    if (!h->d_round_image) MGS_HIP(h, hipMalloc(&h->d_round_image, (size_t) P * (size_t) buf));
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : h->own_stream;
    int64_t fill[MGS_MAX_RANKS], base[MGS_MAX_RANKS], init[MGS_MAX_RANKS];
    for (int b = 0; b < MGS_MAX_RANKS; b++) {
        fill[b] = (stream_fill && b < P && b != me) ? stream_fill[b] : 0;
        base[b] = (open_round_base && b < P && b != me) ? open_round_base[b] : fill[b];
        init[b] = 0;
    }
    memset(out, 0, sizeof(*out));
    out->next = start;
    uint64_t cur = start, lrec_k = 0;
    int64_t lused = 0, lbytes_k = 0;
    uint32_t k = 0;
    auto report = [&](uint32_t row) {
        if (round_next) round_next[row] = cur;
        if (round_local_records) round_local_records[row] = lrec_k;
        if (round_local_bytes) round_local_bytes[row] = lbytes_k;
        for (int b = 0; b < P; b++) {
            if (round_send_offset) round_send_offset[(size_t) row * P + b] = b == me ? 0 : fill[b] - base[b];
            if (round_base) round_base[(size_t) row * P + b] = base[b];
        }
        out->n_rounds = row + 1;
    };
    for (;;) {
        bool room = true;
        for (int b = 0; b < P; b++) room = room && (b == me || base[b] + buf <= stream_cap);
        if (!room) { // a stream is full: the round stays open, the caller sees next < n_end
            report(k);
            return MGS_OK;
        }
        for (int b = 0; b < P; b++) init[b] = b == me ? 0 : fill[b] - base[b];
        mgs_round r;
        const int rc = mgs_pack_device(h, d_records, d_rec_off, n_end, cur, init, h->d_round_image,
                                       d_local_out ? static_cast<uint8_t *>(d_local_out) + lused : nullptr,
                                       local_cap - lused, d_dest_out ? d_dest_out + (cur - start) : nullptr,
                                       MGS_ALGO_GENERIC, st, &r);
        if (rc) return rc;
        for (int b = 0; b < P; b++) {
            const int64_t add = b == me ? 0 : r.send_offset[b] - init[b];
            if (add <= 0) continue;
            MGS_HIP(h, hipMemcpyAsync(static_cast<uint8_t *>(d_streams) + (size_t) b * (size_t) stream_cap + fill[b],
                                      h->d_round_image + (size_t) b * (size_t) buf + init[b], (size_t) add,
                                      hipMemcpyDeviceToDevice, st));
            fill[b] += add;
        }
        const uint64_t consumed = r.next - cur;
        cur = r.next;
        lused += r.local_bytes;
        lrec_k += r.local_records;
        lbytes_k += r.local_bytes;
        out->packed_records += r.packed_records;
        out->local_records += r.local_records;
        out->local_bytes += r.local_bytes;
        out->next = cur;
        if (cur < n_end && consumed == h->desc.max_records) continue; // same round, next stretch of records
        report(k);
        k++;
        if (cur >= n_end || k == max_rounds) break;
        for (int b = 0; b < P; b++) base[b] = fill[b]; // the record that did not fit opens the next round
        lrec_k = 0;
        lbytes_k = 0;
    }
    MGS_HIP(h, hipStreamSynchronize(st)); // the scratch image may be reused, the streams read, on return
    return MGS_OK;
}

int mgs_pack_streams_device(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off, uint64_t n_end,
                            uint64_t start, const int64_t *stream_fill, const int64_t *open_round_base, void *d_streams,
                            int64_t stream_cap, uint32_t max_rounds, void *d_local_out, int64_t local_cap,
                            uint8_t *d_dest_out, void *stream, mgs_rounds *out, uint64_t *round_next,
                            int64_t *round_send_offset, int64_t *round_base, uint64_t *round_local_records,
                            int64_t *round_local_bytes)
{
    if (!h || !out) return MGS_ERR_ARG;
    // A handle whose last one-sweep call met records too long for the staging buffer packs
    // round by round until the record length says that the smallest tile fits again.
    const bool known_long = h->fixed_len == 0 && h->ov_shrink > 0 && ov_pick_k_legacy(1.25 * h->ov_bytes_per_record) == 0;
    if (!known_long) {
        int rc = mgs_pack_streams_device_async(h, d_records, d_rec_off, n_end, start, stream_fill, open_round_base,
                                               d_streams, stream_cap, max_rounds, d_local_out, local_cap, d_dest_out,
                                               stream);
        if (rc) return rc;
        rc = mgs_pack_streams_result(h, out, round_next, round_send_offset, round_base, round_local_records,
                                     round_local_bytes);
        if (!(rc == MGS_ERR_RECORD_SIZE && h->pending_osv && h->h_ctrl->head.fallback != 0)) return rc;
    }
    else if (max_rounds < 1 || stream_cap < 1) {
        h->err = "mgs_pack_streams_device: max_rounds >= 1 and stream_cap >= 1 required";
        return MGS_ERR_ARG;
    }
    return streams_round_by_round(h, d_records, d_rec_off, n_end, start, stream_fill, open_round_base, d_streams,
                                  stream_cap, max_rounds, d_local_out, local_cap, d_dest_out, stream, out, round_next,
                                  round_send_offset, round_base, round_local_records, round_local_bytes);
}

int mgs_pack_rounds_device(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off, uint64_t n_end,
                           uint64_t start, const int64_t *init_offset, void *d_send_images, int64_t image_stride,
                           uint32_t max_rounds, void *d_local_out, int64_t local_cap, uint8_t *d_dest_out, int algo,
                           void *stream, mgs_rounds *out, uint64_t *round_next, int64_t *round_send_offset,
                           uint64_t *round_local_records, int64_t *round_local_bytes)
{
    if (!h || !out) return MGS_ERR_ARG;
    if (max_rounds < 1 || image_stride < (int64_t) h->desc.nranks * h->desc.buf_size) {
        h->err = "mgs_pack_rounds_device: max_rounds >= 1 and image_stride >= nranks * buf_size required";
        return MGS_ERR_ARG;
    }
    if (rounds_in_kernel(h, d_records, d_send_images, image_stride, d_local_out, init_offset, algo, max_rounds)) {
        int rc = mgs_pack_rounds_device_async(h, d_records, d_rec_off, n_end, start, init_offset, d_send_images,
                                              image_stride, max_rounds, d_local_out, local_cap, d_dest_out, algo,
                                              stream);
        if (rc) return rc;
        rc = mgs_pack_rounds_result(h, out, round_next, round_send_offset, round_local_records, round_local_bytes);
        // variable-width records too long for the one-sweep kernel's staging: the generic
        // kernels, one round per call
        if (!(rc == MGS_ERR_RECORD_SIZE && h->pending_osv && h->h_ctrl->head.fallback != 0)) return rc;
        algo = MGS_ALGO_GENERIC;
    }
    // Kernels that pack one round per launch sequence: one call per round (more when a
    // round outlasts the handle's max_records), the local stream and the destination
    // bytes advancing with the records.
    const int P = h->desc.nranks;
    memset(out, 0, sizeof(*out));
    out->next = start;
    uint64_t cur = start, lrec_k = 0;
    int64_t lused = 0, lbytes_k = 0;
    int64_t so[MGS_MAX_RANKS];
    for (int b = 0; b < MGS_MAX_RANKS; b++) so[b] = (init_offset && b < P) ? init_offset[b] : 0;
    uint32_t k = 0;
    for (;;) {
        mgs_round r;
        const int rc = mgs_pack_device(h, d_records, d_rec_off, n_end, cur, so,
                                       static_cast<uint8_t *>(d_send_images) + (size_t) k * (size_t) image_stride,
                                       d_local_out ? static_cast<uint8_t *>(d_local_out) + lused : nullptr,
                                       local_cap - lused, d_dest_out ? d_dest_out + (cur - start) : nullptr, algo, stream,
                                       &r);
        if (rc) return rc;
        const uint64_t consumed = r.next - cur;
        cur = r.next;
        lused += r.local_bytes;
        lrec_k += r.local_records;
        lbytes_k += r.local_bytes;
        out->packed_records += r.packed_records;
        out->local_records += r.local_records;
        out->local_bytes += r.local_bytes;
        out->next = cur;
        for (int b = 0; b < P; b++) so[b] = r.send_offset[b];
        if (cur < n_end && consumed == h->desc.max_records) continue; // same round, next stretch of records
        if (round_next) round_next[k] = cur;
        if (round_local_records) round_local_records[k] = lrec_k;
        if (round_local_bytes) round_local_bytes[k] = lbytes_k;
        if (round_send_offset)
            for (int b = 0; b < P; b++) round_send_offset[(size_t) k * P + b] = so[b];
        out->n_rounds = ++k;
        if (cur >= n_end || k == max_rounds) break;
        for (int b = 0; b < P; b++) so[b] = 0;
        lrec_k = 0;
        lbytes_k = 0;
    }
    return MGS_OK;
}

int mgs_get_desc(const mgs_handle *h, mgs_desc *out)
{
    if (!h || !out) return MGS_ERR_ARG;
    *out = h->desc;
    return MGS_OK;
}

// ---- receive side -----------------------------------------------------------------
constexpr uint32_t kUnpMaxCuts = 4096;
// Chunks (one wave each) a stream is cut into by the parse kernels; MGS_PARSE_CHUNKS overrides (A/B timing).
static uint64_t parse_chunks()
{
    static const char *env = getenv("MGS_PARSE_CHUNKS");
    const long v = env ? atol(env) : 0;
    return v >= 2048 && v <= 65536 ? (uint64_t) v : (uint64_t) UNP_MAX_CHUNKS;
}
// The same for the tokenizer's text; MGS_TOK_CHUNKS overrides (A/B timing).
static uint64_t tok_chunks()
{
    static const char *env = getenv("MGS_TOK_CHUNKS");
    const long v = env ? atol(env) : 0;
    return v >= 1024 && v <= 65536 ? (uint64_t) v : (uint64_t) TOK_CHUNKS;
}
constexpr size_t kUnpScratch = 8u << 20; // first allocation; grows with the stream

// Scratch of the blocking parse calls (unpack, tokenize): one buffer per handle.
static int ensure_unp_scratch(mgs_handle *h, size_t need)
{
    if (need <= h->unp_bytes) return MGS_OK;
    if (h->d_unp) {
        MGS_HIP(h, hipDeviceSynchronize());
        MGS_HIP(h, hipFree(h->d_unp));
        h->d_unp = nullptr;
        h->unp_bytes = 0;
    }
    const size_t want = std::max(need, kUnpScratch);
    MGS_HIP(h, hipMalloc(&h->d_unp, want));
    h->unp_bytes = want;
    return MGS_OK;
}

int mgs_unpack_device(mgs_handle *h, const void *d_stream, uint64_t nbytes, uint32_t *d_rec_off,
                      uint64_t rec_off_capacity, int64_t page_bytes, int64_t first_page_fill,
                      uint64_t *n_records, mgs_page_cut *cuts, int max_cuts, int *n_cuts, void *stream_v)
{
    if (!h || !d_rec_off || !n_records || !n_cuts || (nbytes && !d_stream) || (max_cuts > 0 && !cuts)) {
        if (h) h->err = "mgs_unpack_device: NULL argument";
        return MGS_ERR_ARG;
    }
    const mgs_layout &L = h->desc.layout;
    if (nbytes >= (1ull << 32) || rec_off_capacity == 0 || page_bytes <= 0 || first_page_fill < 0
        || first_page_fill > page_bytes || max_cuts < 0) {
        h->err = "mgs_unpack_device: stream must be under 4 GiB, page_bytes > 0, 0 <= first_page_fill <= page_bytes";
        return MGS_ERR_ARG;
    }
    const bool fixed = h->fixed_len != 0;
    const bool cstr_fixed = L.key_kind == MGS_KEY_CSTR && L.key_count == 1 && L.val_kind != MGS_VAL_CSTR
                            && (L.val_kind == MGS_VAL_VOID || L.val_bytes <= UNP_MAX_V);
    if (!fixed && !cstr_fixed) {
        h->err = "mgs_unpack_device: layout not supported on the device (one string key + fixed value of "
                 "at most 62 bytes, or fixed-width records)";
        return MGS_ERR_ARG;
    }
    DeviceGuard g(h->desc.device);
    if (!g.ok) {
        h->err = "mgs_unpack_device: hipSetDevice failed";
        return MGS_ERR_HIP;
    }
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->own_stream;
    if (!h->h_unp) MGS_HIP(h, hipHostMalloc(&h->h_unp, (4 + 2 * kUnpMaxCuts) * sizeof(uint64_t), hipHostMallocDefault));

    UnpParams p;
    memset(&p, 0, sizeof(p));
    p.stream = static_cast<const uint8_t *>(d_stream);
    p.nbytes = nbytes;
    p.V = (L.val_kind == MGS_VAL_FIXED) ? (uint32_t) L.val_bytes : 0u;
    const uint32_t S = p.V + 2;
    // one wave per chunk, one lane per piece; pieces of at least four 64-byte blocks
    uint64_t chunk = (nbytes + parse_chunks() - 1) / parse_chunks();
    chunk = (chunk + 16383) / 16384 * 16384; // 64 pieces, each a multiple of 256 bytes (two steps of two blocks)
    if (chunk == 0) chunk = 16384;
    p.chunk_bytes = (uint32_t) chunk;
    p.piece_bytes = (uint32_t) (chunk / UNP_PIECES);
    p.nchunks = (uint32_t) ((nbytes + chunk - 1) / chunk);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t tables = (size_t) p.nchunks * UNP_PIECES * S;
    p.nruns = (p.nchunks + UNP_RUN - 1) / UNP_RUN;
    const size_t sync_bytes = up((2 + (size_t) p.nruns) * sizeof(uint32_t));
    const size_t need = up((4 + 2 * kUnpMaxCuts) * sizeof(uint64_t)) + sync_bytes
                        + up((size_t) p.nchunks * S * sizeof(uint32_t))
                        + up((size_t) p.nruns * S * sizeof(uint32_t)) + up((size_t) p.nruns * S * sizeof(uint64_t))
                        + up((size_t) p.nruns * sizeof(uint32_t)) + up((size_t) p.nruns * sizeof(uint64_t))
                        + up(tables * sizeof(uint16_t)) + up(tables)
                        + up((size_t) p.nchunks * (size_t) (p.chunk_bytes / 8));
    if (h->pending) { // the control block and the scratch are shared with the pack call in flight
        h->err = "mgs_unpack_device: a pack call is in flight on this handle (call mgs_pack_result first)";
        return MGS_ERR_STATE;
    }
    { const int rc = ensure_unp_scratch(h, need); if (rc) return rc; }
    uint8_t *q = h->d_unp;
    p.result = reinterpret_cast<uint64_t *>(q); // result and cuts leave with one copy
    p.cuts = p.result + 4;                                    q += up((4 + 2 * kUnpMaxCuts) * sizeof(uint64_t));
    p.sync = reinterpret_cast<uint32_t *>(q);                 q += sync_bytes;
    p.table = reinterpret_cast<uint32_t *>(q);                q += up((size_t) p.nchunks * S * sizeof(uint32_t));
    p.run_exit = reinterpret_cast<uint32_t *>(q);             q += up((size_t) p.nruns * S * sizeof(uint32_t));
    p.run_count = reinterpret_cast<uint64_t *>(q);            q += up((size_t) p.nruns * S * sizeof(uint64_t));
    p.run_entry = reinterpret_cast<uint32_t *>(q);            q += up((size_t) p.nruns * sizeof(uint32_t));
    p.run_first = reinterpret_cast<uint64_t *>(q);            q += up((size_t) p.nruns * sizeof(uint64_t));
    p.piece_count = reinterpret_cast<uint16_t *>(q);          q += up(tables * sizeof(uint16_t));
    p.piece_exit = q;                                         q += up(tables);
    p.masks = reinterpret_cast<unsigned long long *>(q);      q += up((size_t) p.nchunks * (size_t) (p.chunk_bytes / 8));
    p.rec_off = d_rec_off;
    p.rec_cap = rec_off_capacity;
    p.ctrl = h->d_ctrl;
    p.page_bytes = page_bytes;
    p.first_fill = first_page_fill;
    p.max_cuts = kUnpMaxCuts;

    MGS_HIP(h, hipMemsetAsync(&h->d_ctrl->head.error, 0, sizeof(int32_t), stream));
    MGS_HIP(h, hipMemsetAsync(p.result, 0, 4 * sizeof(uint64_t), stream));
    if (nbytes && !fixed) MGS_HIP(h, hipMemsetAsync(p.sync, 0, sync_bytes, stream));
    if (nbytes == 0) {
        MGS_HIP(h, hipMemsetAsync(d_rec_off, 0, sizeof(uint32_t), stream));
    }
    else if (fixed) {
        const uint64_t n = nbytes / h->fixed_len;
        if (n + 1 > rec_off_capacity) {
            h->err = "mgs_unpack_device: d_rec_off too small";
            return MGS_ERR_ARG;
        }
        hipLaunchKernelGGL(k_unp_fixed, dim3((unsigned) std::min<uint64_t>(4096, n / 256 + 1)), dim3(256), 0,
                           stream, p, h->fixed_len);
    }
    else if (nbytes < 16) {
        hipLaunchKernelGGL(k_unp_tiny, dim3(1), dim3(64), 0, stream, p);
    }
    else {
        const size_t lds_spec = (((2 * (size_t) S * 65 + 3) & ~(size_t) 3) + 256) * sizeof(uint32_t);
        const size_t lds_emit = ((size_t) unp_emit_table_dwords(S) + 64 * UNP_SLOT_PITCH) * sizeof(uint32_t);
        hipLaunchKernelGGL(k_unp_spec, dim3(p.nchunks), dim3(64), lds_spec, stream, p);
        hipLaunchKernelGGL(k_unp_emit, dim3(p.nchunks), dim3(64), lds_emit, stream, p);
    }
    hipLaunchKernelGGL(k_unp_pages, dim3(1), dim3(64), 0, stream, p);
    MGS_HIP(h, hipGetLastError());
    // records, cuts, status and the first cuts in one small copy; the other cuts only when there are any
    constexpr uint64_t kHeadCuts = 62;
    MGS_HIP(h, hipMemcpyAsync(h->h_unp, p.result, (4 + 2 * kHeadCuts) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    MGS_HIP(h, hipStreamSynchronize(stream));
    const int derr = (int) (int32_t) (uint32_t) h->h_unp[2];
    if (derr != 0) {
        switch (derr) {
        case MGS_ERR_MALFORMED: h->err = "the stream does not end at a record boundary"; break;
        case MGS_ERR_RECORD_SIZE: h->err = fmt("a record is larger than one page (%lld)", (long long) page_bytes); break;
        case MGS_ERR_ARG: h->err = "mgs_unpack_device: d_rec_off too small"; break;
        default: h->err = fmt("device error %d", derr);
        }
        return derr;
    }
    *n_records = h->h_unp[0];
    const uint64_t nc = h->h_unp[1];
    if (nc > (uint64_t) max_cuts || nc > kUnpMaxCuts) {
        h->err = fmt("mgs_unpack_device: %llu page cuts, room for %d", (unsigned long long) nc, max_cuts);
        return MGS_ERR_ARG;
    }
    if (nc > kHeadCuts) {
        MGS_HIP(h, hipMemcpyAsync(h->h_unp + 4 + 2 * kHeadCuts, p.result + 4 + 2 * kHeadCuts,
                                  2 * (nc - kHeadCuts) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        MGS_HIP(h, hipStreamSynchronize(stream));
    }
    for (uint64_t k = 0; k < nc; k++) {
        cuts[k].first_record = h->h_unp[4 + 2 * k];
        cuts[k].first_byte = h->h_unp[5 + 2 * k];
    }
    *n_cuts = (int) nc;
    return MGS_OK;
}

// ---- map side of wordcount --------------------------------------------------------
int mgs_tokenize_device(mgs_handle *h, const void *d_text, uint64_t nbytes, uint64_t value, void *d_records,
                        uint64_t records_capacity, uint32_t *d_rec_off, uint64_t rec_off_capacity,
                        uint64_t *n_records, uint64_t *n_bytes, void *stream_v)
{
    if (!h || !d_rec_off || !n_records || !n_bytes || (nbytes && !d_text) || (records_capacity && !d_records)) {
        if (h) h->err = "mgs_tokenize_device: NULL argument";
        return MGS_ERR_ARG;
    }
    const mgs_layout &L = h->desc.layout;
    if (L.key_kind != MGS_KEY_CSTR || L.key_count != 1 || L.val_kind != MGS_VAL_FIXED || L.val_bytes != 8) {
        h->err = "mgs_tokenize_device: the layout must be one string key and an 8-byte value";
        return MGS_ERR_ARG;
    }
    if (nbytes >= (1ull << 32) || rec_off_capacity == 0) {
        h->err = "mgs_tokenize_device: text must be under 4 GiB, rec_off_capacity > 0";
        return MGS_ERR_ARG;
    }
    DeviceGuard g(h->desc.device);
    if (!g.ok) {
        h->err = "mgs_tokenize_device: hipSetDevice failed";
        return MGS_ERR_HIP;
    }
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->own_stream;
    if (!h->h_unp) MGS_HIP(h, hipHostMalloc(&h->h_unp, (4 + 2 * kUnpMaxCuts) * sizeof(uint64_t), hipHostMallocDefault));

    TokParams p;
    memset(&p, 0, sizeof(p));
    p.text = static_cast<const uint8_t *>(d_text);
    p.nbytes = nbytes;
    // one wave per chunk; the end of the text (position nbytes) is covered too: it closes an
    // open word
    const uint64_t want = tok_chunks();
    uint64_t chunk = (nbytes + 1 + want - 1) / want;
    chunk = (chunk + 4 * TOKS_STEP - 1) / (4 * TOKS_STEP) * (4 * TOKS_STEP); // the kernels take four steps per trip
    p.chunk_bytes = (uint32_t) chunk;
    p.nchunks = (uint32_t) ((nbytes + 1 + chunk - 1) / chunk);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t need = up(2 * sizeof(uint64_t)) + up((size_t) p.nchunks * 3 * sizeof(uint32_t))
                        + up((size_t) p.nchunks * 3 * sizeof(uint64_t));
    if (h->pending) { // the control block and the scratch are shared with the pack call in flight
        h->err = "mgs_tokenize_device: a pack call is in flight on this handle (call mgs_pack_result first)";
        return MGS_ERR_STATE;
    }
    { const int rc = ensure_unp_scratch(h, need); if (rc) return rc; }
    uint8_t *q = h->d_unp;
    p.result = reinterpret_cast<uint64_t *>(q);    q += up(2 * sizeof(uint64_t));
    p.chunk_x = reinterpret_cast<uint64_t *>(q);   q += up((size_t) p.nchunks * 3 * sizeof(uint64_t));
    p.chunk_tot = reinterpret_cast<uint32_t *>(q); q += up((size_t) p.nchunks * 3 * sizeof(uint32_t));
    p.out = static_cast<uint8_t *>(d_records);
    p.out_cap = records_capacity;
    p.rec_off = d_rec_off;
    p.rec_cap = rec_off_capacity;
    p.value = value;
    p.ctrl = h->d_ctrl;

    MGS_HIP(h, hipMemsetAsync(&h->d_ctrl->head.error, 0, sizeof(int32_t), stream));
    if (nbytes < 16) { // the streaming kernels fetch 16 bytes at a time
        hipLaunchKernelGGL(k_tok_tiny, dim3(1), dim3(64), 0, stream, p);
    }
    else {
        hipLaunchKernelGGL(k_tok_count, dim3(p.nchunks), dim3(64), 0, stream, p);
        hipLaunchKernelGGL(k_tok_scan, dim3(1), dim3(1024), 0, stream, p);
        hipLaunchKernelGGL(k_tok_stream, dim3(p.nchunks), dim3(64), 0, stream, p);
    }
    MGS_HIP(h, hipGetLastError());
    MGS_HIP(h, hipMemcpyAsync(h->h_unp, p.result, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    MGS_HIP(h, hipMemcpyAsync(&h->h_ctrl->head.error, &h->d_ctrl->head.error, sizeof(int32_t),
                              hipMemcpyDeviceToHost, stream));
    MGS_HIP(h, hipStreamSynchronize(stream));
    const int derr = h->h_ctrl->head.error;
    if (derr != 0) {
        if (derr == MGS_ERR_ARG)
            h->err = fmt("mgs_tokenize_device: %llu records, %llu bytes do not fit the outputs (or pass 4 GiB)",
                         (unsigned long long) h->h_unp[0], (unsigned long long) h->h_unp[1]);
        else
            h->err = fmt("device error %d", derr);
        return derr;
    }
    *n_records = h->h_unp[0];
    *n_bytes = h->h_unp[1];
    return MGS_OK;
}

// ---- reduce side: KV -> KMV -------------------------------------------------------
int mgs_group_device(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off, uint64_t n_records,
                     uint32_t *d_group_first, uint32_t *d_group_start, uint64_t group_capacity, void *d_values,
                     int64_t page_bytes, uint64_t *n_groups, uint64_t *cuts, int max_cuts, int *n_cuts,
                     void *stream_v)
{
    if (!h || !d_group_start || !n_groups || !n_cuts || (n_records && (!d_records || !d_group_first))
        || (max_cuts > 0 && !cuts)) {
        if (h) h->err = "mgs_group_device: NULL argument";
        return MGS_ERR_ARG;
    }
    const mgs_layout &L = h->desc.layout;
    if (L.val_kind == MGS_VAL_CSTR) {
        h->err = "mgs_group_device: string values are not supported (fixed-width or no value)";
        return MGS_ERR_ARG;
    }
    const uint32_t vbytes = L.val_kind == MGS_VAL_FIXED ? (uint32_t) L.val_bytes : 0u;
    if (n_records > (1ull << 30) || page_bytes <= 0 || max_cuts < 0 || (vbytes && n_records && !d_values)
        || (!h->fixed_len && n_records && !d_rec_off)) {
        h->err = "mgs_group_device: at most 2^30 records, page_bytes > 0, d_values and (for variable width) "
                 "d_rec_off required";
        return MGS_ERR_ARG;
    }
    DeviceGuard g(h->desc.device);
    if (!g.ok) {
        h->err = "mgs_group_device: hipSetDevice failed";
        return MGS_ERR_HIP;
    }
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->own_stream;
    if (!h->h_unp) MGS_HIP(h, hipHostMalloc(&h->h_unp, (4 + 2 * kUnpMaxCuts) * sizeof(uint64_t), hipHostMallocDefault));
    *n_groups = 0;
    *n_cuts = 0;
    if (n_records == 0) {
        MGS_HIP(h, hipMemsetAsync(d_group_start, 0, sizeof(uint32_t), stream));
        MGS_HIP(h, hipStreamSynchronize(stream));
        return MGS_OK;
    }

    GrpParams p;
    memset(&p, 0, sizeof(p));
    p.blob = static_cast<const uint8_t *>(d_records);
    p.rec_off = h->fixed_len ? nullptr : d_rec_off;
    p.fixed_len = h->fixed_len;
    p.n = (uint32_t) n_records;
    p.lay = L;
    p.vbytes = vbytes;
    uint32_t log_slots = 10;
    while ((1ull << log_slots) < 2 * n_records) log_slots++;
    const size_t slots = (size_t) 1 << log_slots;
    p.slot_mask = (uint32_t) (slots - 1);
    p.slot_shift = 32 - log_slots;
    p.ntiles = (uint32_t) ((n_records + GRP_TILE - 1) / GRP_TILE);
    p.group_first = d_group_first;
    p.group_start = d_group_start;
    p.group_cap = group_capacity;
    p.values = static_cast<uint8_t *>(d_values);
    p.page_bytes = page_bytes;
    p.max_cuts = kUnpMaxCuts;
    p.ctrl = h->d_ctrl;

    // the sort: passes of the 16-byte one-sweep kernel over {rank, record, value} records
    const OsVariant &SV = *os16_digit_variant();
    const uint32_t sort_tiles = (uint32_t) ((n_records + (uint64_t) SV.tile - 1) / (uint64_t) SV.tile);
    const size_t sort_status = (size_t) sort_tiles * GRP_SORT_BINS * sizeof(unsigned long long);
    const size_t sort_agg = (size_t) sort_tiles * (GRP_SORT_BINS / 4) * sizeof(unsigned long long);
    const size_t sort_rt = 2 * ((size_t) GRP_SORT_BINS + 1) * sizeof(unsigned long long);
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t need = up((2 + kUnpMaxCuts) * sizeof(uint64_t)) + 5 * up(slots * sizeof(uint32_t))
                        + up((size_t) n_records * sizeof(uint32_t)) + 2 * up((size_t) n_records * sizeof(uint4))
                        + up(((size_t) p.ntiles + 1) * sizeof(uint32_t))
                        + up((size_t) p.ntiles * (GRP_TILE / 64) * sizeof(uint64_t))
                        + up(GRP_SORT_MAX_PASSES * GRP_SORT_BINS * sizeof(uint32_t)) + up(GRP_SORT_MAX_PASSES * sizeof(Ctrl))
                        + up(sort_status) + up(sort_agg) + up(sort_rt);
    if (h->pending) { // the control block and the scratch are shared with the pack call in flight
        h->err = "mgs_group_device: a pack call is in flight on this handle (call mgs_pack_result first)";
        return MGS_ERR_STATE;
    }
    if (need > h->grp_bytes) {
        if (h->d_grp) {
            MGS_HIP(h, hipDeviceSynchronize());
            MGS_HIP(h, hipFree(h->d_grp));
            h->d_grp = nullptr;
            h->grp_bytes = 0;
        }
        MGS_HIP(h, hipMalloc(&h->d_grp, need));
        h->grp_bytes = need;
    }
    uint8_t *q = h->d_grp;
    p.result = reinterpret_cast<uint64_t *>(q);
    p.cuts = p.result + 2;                               q += up((2 + kUnpMaxCuts) * sizeof(uint64_t));
    p.table = reinterpret_cast<unsigned long long *>(q); q += 2 * up(slots * sizeof(unsigned long long));
    // string keys (wordcount, join: many records per key) keep the key's first bytes in the slot;
    // for fixed-width keys the larger table costs more than the saved look-ups into the stream
    p.table_stride = L.key_kind == MGS_KEY_CSTR ? 2 : 1;
    p.cand_bit = 0x80000000u;
    p.rank_table = reinterpret_cast<uint32_t *>(q);      q += up(slots * sizeof(uint32_t));
    p.slot_or_rank = reinterpret_cast<uint32_t *>(q);    q += up((size_t) n_records * sizeof(uint32_t));
    p.rec_a = reinterpret_cast<uint4 *>(q);              q += up((size_t) n_records * sizeof(uint4));
    p.rec_b = reinterpret_cast<uint4 *>(q);              q += up((size_t) n_records * sizeof(uint4));
    p.tile_count = reinterpret_cast<uint32_t *>(q);      q += up(((size_t) p.ntiles + 1) * sizeof(uint32_t));
    p.first_bits = reinterpret_cast<uint64_t *>(q);      q += up((size_t) p.ntiles * (GRP_TILE / 64) * sizeof(uint64_t));
    p.hist = reinterpret_cast<uint32_t *>(q);            q += up(GRP_SORT_MAX_PASSES * GRP_SORT_BINS * sizeof(uint32_t));
    p.sort_ctrl = reinterpret_cast<Ctrl *>(q);           q += up(GRP_SORT_MAX_PASSES * sizeof(Ctrl));
    unsigned long long *d_sort_status = reinterpret_cast<unsigned long long *>(q); q += up(sort_status);
    unsigned long long *d_sort_agg = reinterpret_cast<unsigned long long *>(q);    q += up(sort_agg);
    unsigned long long *d_sort_rt = reinterpret_cast<unsigned long long *>(q);     q += up(sort_rt);

    const unsigned wide = (unsigned) std::min<uint64_t>((n_records + GRP_THREADS - 1) / GRP_THREADS,
                                                        (uint64_t) std::max(h->num_cus, 1) * 32);
    MGS_HIP(h, hipMemsetAsync(&h->d_ctrl->head.error, 0, sizeof(int32_t), stream));
    MGS_HIP(h, hipMemsetAsync(p.result, 0, 2 * sizeof(uint64_t), stream));
    MGS_HIP(h, hipMemsetAsync(p.table, 0xFF, (size_t) p.table_stride * slots * sizeof(unsigned long long), stream));
    MGS_HIP(h, hipMemsetAsync(p.hist, 0, GRP_SORT_MAX_PASSES * GRP_SORT_BINS * sizeof(uint32_t), stream));
    MGS_HIP(h, hipMemsetAsync(p.sort_ctrl, 0, GRP_SORT_MAX_PASSES * sizeof(Ctrl), stream));
    hipLaunchKernelGGL(k_grp_insert, dim3(wide), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_grp_count, dim3(p.ntiles), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_grp_scan, dim3(1), dim3(1024), 0, stream, p);
    hipLaunchKernelGGL(k_grp_rank, dim3(p.ntiles), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_grp_keys, dim3(wide), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_grp_sort_setup, dim3(1), dim3(64), 0, stream, p);
    MGS_HIP(h, hipGetLastError());
    // the sort needs only the bits the ranks use: fetch the number of groups
    MGS_HIP(h, hipMemcpyAsync(h->h_unp, p.result, sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
    MGS_HIP(h, hipMemcpyAsync(&h->h_ctrl->head.error, &h->d_ctrl->head.error, sizeof(int32_t),
                              hipMemcpyDeviceToHost, stream));
    MGS_HIP(h, hipStreamSynchronize(stream));
    const uint64_t G = h->h_unp[0];
    int derr = h->h_ctrl->head.error;
    if (derr == 0) {
        // stable radix sort of the sort records on the ranks' bits, least significant digit first
        const GrpSortPlan plan = grp_sort_plan(G);
        for (int s = 0; s < plan.npass; s++) {
            OsParams sp;
            memset(&sp, 0, sizeof(sp));
            sp.rec = (s & 1) ? p.rec_b : p.rec_a;
            sp.n = n_records;
            sp.first = 0;
            sp.P = 1u << plan.width[s];
            sp.self = -1;
            sp.klen = plan.shift[s];      // OSV_DIGIT: the digit's shift ...
            sp.kmask[0] = sp.P - 1u;      // ... and mask
            sp.nbits = plan.width[s];
            sp.nbits_pad = sp.nbits;
            sp.fm = make_fastmod(sp.P);
            sp.tile_records = (uint32_t) SV.tile;
            sp.ntiles = sort_tiles;
            sp.status = d_sort_status;
            sp.agg16 = d_sort_agg;
            sp.lpt_log = sp.nbits_pad > 2 ? sp.nbits_pad - 2 : 0;
            sp.unit_shift = 4;
            sp.ctrl = p.sort_ctrl + s;
            sp.rt = d_sort_rt;
            sp.rt_start = sp.rt + 2 * (size_t) sp.P;
            sp.max_rounds = 1;
            MGS_HIP(h, hipMemsetAsync(d_sort_status, 0, (size_t) sort_tiles * sp.P * sizeof(unsigned long long), stream));
            MGS_HIP(h, hipMemsetAsync(d_sort_agg, 0, ((size_t) sort_tiles << sp.lpt_log) * sizeof(unsigned long long), stream));
            MGS_HIP(h, hipMemsetAsync(d_sort_rt, 0, 2 * ((size_t) sp.P + 1) * sizeof(unsigned long long), stream));
            const uint32_t grid = std::min<uint32_t>(sort_tiles, (uint32_t) (h->num_cus * SV.blocks_per_cu));
            hipLaunchKernelGGL(SV.kernel[sp.nbits], dim3(grid), dim3(SV.threads), 0, stream, sp);
        }
        p.sorted = (plan.npass & 1) ? p.rec_b : p.rec_a;
        hipLaunchKernelGGL(k_grp_emit, dim3(wide), dim3(GRP_THREADS), 0, stream, p);
        hipLaunchKernelGGL(k_grp_pages, dim3(1), dim3(64), 0, stream, p);
        MGS_HIP(h, hipGetLastError());
        MGS_HIP(h, hipMemcpyAsync(h->h_unp, p.result, (2 + kUnpMaxCuts) * sizeof(uint64_t), hipMemcpyDeviceToHost, stream));
        MGS_HIP(h, hipMemcpyAsync(&h->h_ctrl->head.error, &h->d_ctrl->head.error, sizeof(int32_t),
                                  hipMemcpyDeviceToHost, stream));
        MGS_HIP(h, hipStreamSynchronize(stream));
        derr = h->h_ctrl->head.error;
    }
    if (derr != 0) {
        switch (derr) {
        case MGS_ERR_MALFORMED: h->err = "a record does not match the layout"; break;
        case MGS_ERR_ARG: h->err = fmt("mgs_group_device: %llu groups, room for %llu", (unsigned long long) G,
                                       (unsigned long long) group_capacity); break;
        default: h->err = fmt("device error %d", derr);
        }
        return derr;
    }
    *n_groups = G;
    const uint64_t nc = h->h_unp[1];
    if (nc > (uint64_t) max_cuts || nc > kUnpMaxCuts) {
        h->err = fmt("mgs_group_device: %llu value-buffer cuts, room for %d", (unsigned long long) nc, max_cuts);
        return MGS_ERR_ARG;
    }
    for (uint64_t k = 0; k < nc; k++) cuts[k] = h->h_unp[2 + k];
    *n_cuts = (int) nc;
    return MGS_OK;
}

// ---- send side with a combiner ----------------------------------------------------
int mgs_combine_device(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off, uint64_t n_end,
                       uint64_t start, int op, void *d_send_image, void *d_local_out, int64_t local_cap,
                       void *stream_v, mgs_round *out)
{
    if (!h || !out || !d_send_image || (n_end > start && !d_records)) {
        if (h) h->err = "mgs_combine_device: NULL argument";
        return MGS_ERR_ARG;
    }
    const mgs_layout &L = h->desc.layout;
    const int P = h->desc.nranks, self = h->desc.rank;
    if (L.val_kind != MGS_VAL_FIXED || L.val_bytes != 8 || P > OS_MAXP || op < MGS_COMBINE_SUM_U64
        || op > MGS_COMBINE_MAX_U64) {
        h->err = "mgs_combine_device: needs an 8-byte value, at most 64 ranks and a declared operation";
        return MGS_ERR_ARG;
    }
    if (start > n_end || (!h->fixed_len && n_end > start && !d_rec_off) || (self >= 0 && n_end > start && !d_local_out)
        || local_cap < 0) {
        h->err = "mgs_combine_device: start > n_end, or d_rec_off / d_local_out missing";
        return MGS_ERR_ARG;
    }
    const uint64_t n = n_end - start;
    if (n > (1ull << 30) || (h->fixed_len && n * h->fixed_len >= (1ull << 32))) {
        h->err = "mgs_combine_device: at most 2^30 records and 4 GiB per call";
        return MGS_ERR_ARG;
    }
    DeviceGuard g(h->desc.device);
    if (!g.ok) {
        h->err = "mgs_combine_device: hipSetDevice failed";
        return MGS_ERR_HIP;
    }
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->own_stream;
    memset(out, 0, sizeof(*out));
    out->next = start;
    if (n == 0) return MGS_OK;

    CmbParams p;
    memset(&p, 0, sizeof(p));
    GrpParams &q = p.g;
    // records [start, n_end) as records 0 .. n-1: offsets stay absolute into the blob
    q.blob = static_cast<const uint8_t *>(d_records) + (h->fixed_len ? start * h->fixed_len : 0);
    q.rec_off = h->fixed_len ? nullptr : d_rec_off + start;
    q.fixed_len = h->fixed_len;
    q.n = (uint32_t) n;
    q.lay = L;
    q.vbytes = 8;
    uint32_t log_slots = 10;
    while ((1ull << log_slots) < 2 * n) log_slots++;
    const size_t slots = (size_t) 1 << log_slots;
    q.slot_mask = (uint32_t) (slots - 1);
    q.slot_shift = 32 - log_slots;
    q.ntiles = (uint32_t) ((n + GRP_TILE - 1) / GRP_TILE);
    q.ctrl = h->d_ctrl;
    q.fm = h->fm;
    p.start = start;
    p.P = (uint32_t) P;
    p.self = self;
    p.buf_size = h->desc.buf_size;
    p.local_cap = local_cap;
    p.op = op;
    p.ncol = (uint32_t) P + 2;
    p.image = static_cast<uint8_t *>(d_send_image);
    p.local_out = static_cast<uint8_t *>(d_local_out);
    p.round = &h->d_ctrl->result;

    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t cols = ((size_t) q.ntiles + 1) * p.ncol;
    const size_t need = up(sizeof(unsigned long long)) + 3 * up(slots * sizeof(uint32_t)) + 2 * up(n * sizeof(uint32_t))
                        + up(n) + up((size_t) q.ntiles * (GRP_TILE / 64) * sizeof(uint64_t)) + up(cols * sizeof(uint32_t))
                        + up(cols * sizeof(unsigned long long)) + up(n * sizeof(unsigned long long));
    if (h->pending) { // the control block and the scratch are shared with the pack call in flight
        h->err = "mgs_combine_device: a pack call is in flight on this handle (call mgs_pack_result first)";
        return MGS_ERR_STATE;
    }
    if (need > h->grp_bytes) {
        if (h->d_grp) {
            MGS_HIP(h, hipDeviceSynchronize());
            MGS_HIP(h, hipFree(h->d_grp));
            h->d_grp = nullptr;
            h->grp_bytes = 0;
        }
        MGS_HIP(h, hipMalloc(&h->d_grp, need));
        h->grp_bytes = need;
    }
    uint8_t *w = h->d_grp;
    p.end = reinterpret_cast<unsigned long long *>(w);         w += up(sizeof(unsigned long long));
    q.table = reinterpret_cast<unsigned long long *>(w);       w += up(slots * sizeof(unsigned long long));
    q.table_stride = 1;
    q.rank_table = reinterpret_cast<uint32_t *>(w);            w += up(slots * sizeof(uint32_t));
    q.slot_or_rank = reinterpret_cast<uint32_t *>(w);          w += up(n * sizeof(uint32_t));
    p.pos = reinterpret_cast<uint32_t *>(w);                   w += up(n * sizeof(uint32_t));
    q.dest_out = w;                                            w += up(n);
    q.first_bits = reinterpret_cast<uint64_t *>(w);            w += up((size_t) q.ntiles * (GRP_TILE / 64) * sizeof(uint64_t));
    p.col = reinterpret_cast<uint32_t *>(w);                   w += up(cols * sizeof(uint32_t));
    p.col_x = reinterpret_cast<unsigned long long *>(w);       w += up(cols * sizeof(unsigned long long));
    p.acc = reinterpret_cast<unsigned long long *>(w);         w += up(n * sizeof(unsigned long long));

    const unsigned wide = (unsigned) std::min<uint64_t>((n + GRP_THREADS - 1) / GRP_THREADS,
                                                        (uint64_t) std::max(h->num_cus, 1) * 32);
    MGS_HIP(h, hipMemsetAsync(&h->d_ctrl->head.error, 0, sizeof(int32_t), stream));
    MGS_HIP(h, hipMemsetAsync(q.table, 0xFF, slots * sizeof(unsigned long long), stream));
    hipLaunchKernelGGL(k_grp_insert, dim3(wide), dim3(GRP_THREADS), 0, stream, q);
    hipLaunchKernelGGL(k_cmb_cols, dim3(q.ntiles), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_cmb_colscan, dim3(1), dim3(1024), 0, stream, p);
    hipLaunchKernelGGL(k_cmb_place, dim3(q.ntiles), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_cmb_reduce, dim3(wide), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_cmb_emit, dim3(wide), dim3(GRP_THREADS), 0, stream, p);
    hipLaunchKernelGGL(k_cmb_final, dim3(1), dim3(GRP_THREADS), 0, stream, p);
    MGS_HIP(h, hipGetLastError());
    MGS_HIP(h, hipMemcpyAsync(&h->h_ctrl->result, &h->d_ctrl->result, sizeof(mgs_round), hipMemcpyDeviceToHost, stream));
    MGS_HIP(h, hipMemcpyAsync(&h->h_ctrl->head.error, &h->d_ctrl->head.error, sizeof(int32_t),
                              hipMemcpyDeviceToHost, stream));
    MGS_HIP(h, hipStreamSynchronize(stream));
    const int derr = h->h_ctrl->head.error;
    if (derr != 0) {
        switch (derr) {
        case MGS_ERR_MALFORMED: h->err = "a record does not match the layout"; break;
        case MGS_ERR_RECORD_SIZE: h->err = fmt("KV size is larger than buf_size (%lld)", (long long) p.buf_size); break;
        case MGS_ERR_LOCAL_OVERFLOW: h->err = "d_local_out is too small for the records that stay on this rank"; break;
        default: h->err = fmt("device error %d", derr);
        }
        return derr;
    }
    *out = h->h_ctrl->result;
    return MGS_OK;
}

int mgs_set_tuning(mgs_handle *h, int variant)
{
    if (!h) return MGS_ERR_ARG;
    if (variant < 0 || variant >= kNumOsVariants) {
        h->err = fmt("mgs_set_tuning: variant %d outside [0,%d)", variant, kNumOsVariants);
        return MGS_ERR_ARG;
    }
    h->os_variant = variant;
    return MGS_OK;
}

int mgs_debug_stats(mgs_handle *h, uint64_t out[8])
{
    if (!h || !out) return MGS_ERR_ARG;
    for (int i = 0; i < 8; i++) out[i] = h->h_ctrl->head.stats[i];
    return MGS_OK;
}

int mgs_get_tuning(const mgs_handle *h) { return h ? h->os_variant : -1; }

int mgs_finder_stats(mgs_handle *h, uint64_t out[4])
{
    if (!h || !out) return MGS_ERR_ARG;
    for (int i = 0; i < 4; i++) out[i] = h->h_ctrl->head.fstats[i];
    return MGS_OK;
}

const char *mgs_tuning_name(int variant)
{
    return (variant >= 0 && variant < kNumOsVariants) ? kOsVariants[variant].name : NULL;
}

int mgs_copy_calibration(mgs_handle *h, const void *d_in, void *d_out, uint64_t n_records,
                         void *stream_v, float *ms)
{
    if (!h || !d_in || !d_out || !ms || n_records == 0) return MGS_ERR_ARG;
    DeviceGuard g(h->desc.device);
    hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : h->own_stream;
    // a plain ticketed copy with the production kernel's shape: 512 threads, 8 records each
    constexpr int T = 512, I = 8;
    const uint32_t tile = (uint32_t)(T * I);
    const uint32_t ntiles = (uint32_t)((n_records + tile - 1) / tile);
    uint32_t *ticket = &h->d_ctrl->head.ticket;
    MGS_HIP(h, hipMemsetAsync(ticket, 0, sizeof(uint32_t), stream));
    const uint32_t grid = std::min<uint32_t>(ntiles, (uint32_t)(h->num_cus * 2));
    MGS_HIP(h, hipEventRecord(h->ev0, stream));
    hipLaunchKernelGGL((k_copy16<T, I, 0>), dim3(grid), dim3(T), 0, stream,
                       static_cast<const uint4 *>(d_in), static_cast<uint4 *>(d_out), n_records,
                       ntiles, ticket);
    MGS_HIP(h, hipEventRecord(h->ev1, stream));
    MGS_HIP(h, hipStreamSynchronize(stream));
    MGS_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return MGS_OK;
}

int mgs_last_kernel_ms(mgs_handle *h, float *ms)
{
    if (!h || !ms) return MGS_ERR_ARG;
    if (!h->ev_valid) {
        h->err = "mgs_last_kernel_ms: no completed pack call to report";
        return MGS_ERR_STATE;
    }
    DeviceGuard g(h->desc.device);
    MGS_HIP(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
    return MGS_OK;
}

// ---- synthetic inputs ----------------------------------------------------------
__global__ void k_synth_u64(ulonglong2 *out, uint64_t n, uint64_t first, uint64_t seed)
{
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t ctr = first + i;
        uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull; // splitmix64 state after ctr+1 steps
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z = z ^ (z >> 31);
        out[i] = make_ulonglong2(z, ctr);
    }
}

int mgs_synth_u64_records(void *d_out, uint64_t n, uint64_t first, uint64_t seed, int device,
                          void *stream)
{
    if (!d_out && n) return MGS_ERR_ARG;
    if (n == 0) return MGS_OK;
    DeviceGuard g(device);
    if (!g.ok) return MGS_ERR_HIP;
    const uint64_t blocks = std::min<uint64_t>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_synth_u64, dim3((uint32_t) blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<ulonglong2 *>(d_out), n, first, seed);
    return hipGetLastError() == hipSuccess ? MGS_OK : MGS_ERR_HIP;
}

} // extern "C"
