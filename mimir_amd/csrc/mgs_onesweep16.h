// Single-pass stable hash-partition of 16-byte fixed-width records
// (u64 key + u64 value, 8-char octant key + u64 count, ...), nranks <= 64.
//
// Each record is read from HBM exactly once and written exactly once:
//
//   tile      4096 consecutive records, one 512-thread workgroup (8 waves);
//             wave w owns records [w*512, w*512+512) of the tile, row k of a
//             wave is 64 consecutive records -> every global load is one
//             fully coalesced 1 KiB wave access (global_load_dwordx4).
//   hash      lookup3 on the key words in registers, dest = hash % nranks.
//   rank      per row, a match-any over the destination bits (log2(P) wave
//             ballots) gives each lane the mask of lanes with its destination:
//             rank in row = popcount(mask below me), row count = popcount(mask).
//             Row counts are accumulated per (wave, dest) in LDS, then turned
//             into per-wave slot bases by an exclusive scan over (dest, wave).
//   reorder   records are written to their slot of a 64 KiB LDS image of the
//             tile, grouped by destination, in stable order.
//   chain     the tile's per-destination counts are published to a global
//             status array; wave 0 walks back over the preceding tiles'
//             entries (aggregate / inclusive-prefix words, the data IS the
//             flag: one naturally aligned 8-byte device-scope store) until it
//             reaches an inclusive prefix -> global position of the tile in
//             every partition.  Tiles are handed out by an atomic ticket, so
//             every predecessor of a running tile is itself running or done.
//   store     the LDS image is streamed out: consecutive lanes write
//             consecutive 16-byte slots of one partition (coalesced
//             global_store_dwordx4), bounds-checked against the partition
//             capacity; a tile that holds a record which does not fit records
//             its index (round cut, resolved by k_os_finalize).
#ifndef MGS_ONESWEEP16_H
#define MGS_ONESWEEP16_H

#include "mgs_common.h"

namespace mgs {

constexpr int OS_MAXP = 64;
constexpr uint64_t OS_FLAG_AGG = 1ull << 62;  // value = this tile's count
constexpr uint64_t OS_FLAG_INCL = 2ull << 62; // value = count of tiles 0..t (+ initial fill)
constexpr uint64_t OS_VALUE_MASK = (1ull << 62) - 1;
constexpr uint32_t OS_SPIN_LIMIT = 1u << 20; // polls with nothing new before giving up
constexpr int OS_LOOK = 8;                   // status loads in flight per poll

// Variant bits (template parameter VAR); 0 is the production kernel, the others
// exist to price one design decision at a time on the hardware.
constexpr int OSV_FAKE_CHAIN = 1; // timing only: no chained scan, positions are made up
constexpr int OSV_NT_LOAD = 2;    // non-temporal record loads
constexpr int OSV_NT_STORE = 4;   // non-temporal partition stores
constexpr int OSV_NO_PREFETCH = 8; // load a tile only once it is this workgroup's turn
constexpr int OSV_EARLY_AGG = 32;  // the last worker wave to finish hashing publishes the tile's counts
constexpr int OSV_NODEST = 8192;     // experiment: no dest_out
constexpr int OSV_STATS = 16;      // diagnostic: count polls / depth / cycles of the chained scan
constexpr int OSV_DIGIT = 64;      // the destination is a bit field of a record's first word, not a hash (see OsParams::klen)

struct OsParams {
    const uint4 *rec; // record 0 of this call
    uint64_t n;       // records in this call
    uint64_t first;   // absolute index of record 0 (for the result only)
    uint32_t P;
    int32_t self;
    uint32_t klen;  // key bytes (1..16); OSV_DIGIT kernels: the shift of the bit field that is the destination
    uint32_t kmask[4]; // per key word: mask of the bytes that belong to the key; OSV_DIGIT: [0] = mask of that field
    uint32_t nbits; // ceil(log2(P)), >= 1
    uint32_t nbits_pad; // ceil(log2(P)), >= 0: lanes per predecessor in the chained scan = 1 << nbits_pad
    FastMod fm;
    uint32_t ntiles;
    uint32_t tile_records;
    unsigned long long *status; // [ntiles][P]
    unsigned long long *agg16;  // [ntiles][1 << lpt_log]: 16-bit counts, four per word (windowed scan only)
    uint32_t lpt_log;           // log2 of the words per tile in agg16
    uint32_t unit_shift;        // log2 of the bytes one counted unit stands for (4: 16-byte records, 0: bytes)
    uint8_t *dest_out;          // optional
    Ctrl *ctrl;
    // Rounds (k_onesweep16q and later kernels): one launch fills up to max_rounds send
    // images.  Round k's image is the caller's image k: partition b at part_base[b] +
    // k * image_stride.  Entry k of the round table describes the START of round k:
    // rt[k * P + b] = units of column b before the round's first record (flagged with
    // OS_RT_VALID; entry 0 is all zero by definition and never read), rt_start[k] = that
    // record's index in this call.  Entry max_rounds is the end marker: records from
    // there on are not consumed.
    unsigned long long *rt;
    unsigned long long *rt_start;
    uint32_t max_rounds;
    uint64_t image_stride;
    // Streams layout (mgs_round_finder.h): the sweep packs ONE round whose partitions are
    // the caller's destination streams; workgroup 0 of the grid does not sweep but trails
    // the chained scan and finds where the reference's rounds (partitions of f_cap units)
    // begin inside the streams: f_rt / f_rt_start are a round table of f_max_rounds + 1
    // entries in the format of rt / rt_start, f_cuts receives the number of cuts found.
    uint32_t finder; // 1: workgroup 0 is the round finder
    uint32_t f_max_rounds;
    unsigned long long f_cap;
    unsigned long long *f_rt;
    unsigned long long *f_rt_start;
    // optional, written by the kernel that finishes a one-round call: [0] = records of this
    // call that were packed, [1 + c] = units of column c before the first record that was not
    unsigned long long *clip_out;
    // Streams layout, 16-byte records: units per (tile, worker wave, column) as 16-bit counts,
    // [ntiles][worker waves][P], 0xFFFF = not written yet.  A worker wave's records are 512
    // consecutive ones, so with these the round finder narrows a cut to one such segment and
    // hashes only that again instead of the whole tile.  NULL: not kept.
    uint16_t *seg;
};

constexpr unsigned long long OS_RT_VALID = 1ull << 63;

__device__ __forceinline__ unsigned long long os_load_status(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void os_store_status(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int THREADS, int ITEMS>
struct OsGeom {
    static constexpr int WAVES = THREADS / 64;
    static constexpr int TILE = THREADS * ITEMS;
    // LDS carve-up (bytes): one array, 16-byte aligned pieces
    static constexpr int REC = TILE * 16;               // tile image
    static constexpr int CNT = 2 * WAVES * OS_MAXP * 4; // per (wave,dest) counts, two copies
    static constexpr int SLOT = WAVES * OS_MAXP * 4;    // per (wave,dest) running slot
    static constexpr int BINSTART = (OS_MAXP + 4) * 4;  // dest -> first slot (+ total)
    static constexpr int GPOS = OS_MAXP * 8;            // dest -> global record position
    static constexpr int PBASE = OS_MAXP * 8;           // dest -> partition address
    static constexpr int PCAP = OS_MAXP * 8;            // dest -> capacity in records
    static constexpr int MISC = 16;
    static constexpr int BYTES = REC + CNT + SLOT + BINSTART + GPOS + PBASE + PCAP + MISC;
};

template <int ITEMS, int VAR>
__device__ __forceinline__ void os_load_tile(const OsParams &p, uint32_t tile, int tile_records,
                                             int w, int lane, uint4 (&r)[ITEMS])
{
    const uint64_t idx0 = (uint64_t) tile * tile_records + (uint32_t) w * (ITEMS * 64) + lane;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const uint64_t i = idx0 + (uint32_t) k * 64;
        if (i < p.n) {
            if (VAR & OSV_NT_LOAD) {
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p.rec + i));
                r[k] = make_uint4(v.x, v.y, v.z, v.w);
            }
            else {
                r[k] = p.rec[i];
            }
        }
        else {
            r[k] = make_uint4(0, 0, 0, 0);
        }
    }
}

template <int THREADS, int ITEMS, int VAR, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweep16(OsParams p)
{
    using G = OsGeom<THREADS, ITEMS>;
    constexpr int WAVES = G::WAVES, TILE = G::TILE;
    __shared__ __attribute__((aligned(16))) uint8_t smem[G::BYTES];
    uint4 *srec = reinterpret_cast<uint4 *>(smem);
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + G::REC);
    volatile uint32_t *wslot = reinterpret_cast<uint32_t *>(smem + G::REC + G::CNT);
    uint32_t *binstart = reinterpret_cast<uint32_t *>(smem + G::REC + G::CNT + G::SLOT);
    unsigned long long *gpos = reinterpret_cast<unsigned long long *>(
        smem + G::REC + G::CNT + G::SLOT + G::BINSTART);
    unsigned long long *pbase = gpos + OS_MAXP;
    unsigned long long *pcap = pbase + OS_MAXP;
    volatile uint32_t *misc = reinterpret_cast<uint32_t *>(pcap + OS_MAXP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t P = p.P;

    if (tid < OS_MAXP) {
        const bool live = (uint32_t) tid < P;
        pbase[tid] = live ? p.ctrl->head.part_base[tid] : 0ull;
        pcap[tid] = live ? (unsigned long long) (p.ctrl->head.part_cap[tid] >> 4) : 0ull;
    }
    for (int j = tid; j < 2 * WAVES * OS_MAXP; j += THREADS) wcnt[j] = 0;
    if (tid == 0) misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
    __syncthreads();
    uint32_t tile = misc[0];
    uint4 r[ITEMS];
    if (tile < p.ntiles) os_load_tile<ITEMS, VAR>(p, tile, TILE, w, lane, r);
    int par = 0;

    while (tile < p.ntiles) {
        // the last wave draws the next tile while it waits for its loads
        if (tid == THREADS - 64) misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
        uint32_t *wc = wcnt + par * (WAVES * OS_MAXP);
        const uint64_t idx0 = (uint64_t) tile * TILE + (uint32_t) w * (ITEMS * 64) + lane;

        // ---- hash, destination, rank within row -----------------------------
        uint32_t meta[ITEMS]; // dest | rank<<8 | count<<16 | valid<<31
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint64_t i = idx0 + (uint32_t) k * 64;
            const bool valid = i < p.n;
            const uint32_t h = hash_rec16(r[k].x, r[k].y, r[k].z, r[k].w, p.klen);
            const uint32_t d = fastmod(h, p.fm);
            if (p.dest_out && valid) p.dest_out[i] = (uint8_t) d;
            uint64_t m = __ballot(valid);
            for (uint32_t bit = 0; bit < p.nbits; bit++) {
                const bool one = (d >> bit) & 1u;
                const uint64_t bk = __ballot(one);
                m &= one ? bk : ~bk;
            }
            const uint32_t rank = lanes_below(m);
            const uint32_t cnt = (uint32_t) __popcll(m);
            if (valid && rank == 0) atomicAdd(&wc[w * OS_MAXP + d], cnt);
            meta[k] = d | (rank << 8) | (cnt << 16) | (valid ? 0x80000000u : 0u);
        }
        __syncthreads(); // S1: this tile's counts are complete, misc[1] is set
        const uint32_t next = misc[1];

        // ---- per-destination totals, slot bases (every wave, redundantly) ---
        uint32_t tot = 0, below = 0; // lane = destination
#pragma unroll
        for (int ww = 0; ww < WAVES; ww++) {
            const uint32_t v = wc[ww * OS_MAXP + lane];
            tot += v;
            if (ww < w) below += v;
        }
        const uint32_t incl = wave_incl_scan_u32(tot, lane);
        wslot[w * OS_MAXP + lane] = incl - tot + below;
        if (w == 0) {
            binstart[lane] = incl - tot;
            if (lane == 63) binstart[64] = incl;
            // publish this tile's counts before doing anything slow
            if ((uint32_t) lane < P) {
                unsigned long long *st = p.status + (uint64_t) tile * P + lane;
                if (tile == 0) {
                    const unsigned long long start =
                        (unsigned long long) (p.ctrl->head.part_init[lane] >> 4);
                    gpos[lane] = start;
                    os_store_status(st, OS_FLAG_INCL | (start + tot));
                }
                else {
                    os_store_status(st, OS_FLAG_AGG | tot);
                }
            }
        }
        // the other copy of the counts was last read before the previous S2
        for (int j = tid; j < WAVES * OS_MAXP; j += THREADS)
            wcnt[(par ^ 1) * (WAVES * OS_MAXP) + j] = 0;

        // ---- reorder into the LDS image -------------------------------------
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t d = meta[k] & 0xffu, rank = (meta[k] >> 8) & 0xffu;
            const uint32_t cnt = (meta[k] >> 16) & 0xffu;
            const bool valid = (meta[k] >> 31) != 0;
            if (valid) {
                const uint32_t base = wslot[w * OS_MAXP + d];
                if (rank == 0) wslot[w * OS_MAXP + d] = base + cnt;
                srec[base + rank] = r[k];
            }
        }

        // ---- the registers are free again: start fetching the next tile ------
        // (wave 0 waits until its chained scan is over: its polls must not queue
        // behind 8 KiB of its own record loads)
        if (!(VAR & OSV_NO_PREFETCH) && w != 0) {
            if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        }

        // ---- chained scan: where does this tile start in every partition? ---
        if (VAR & OSV_FAKE_CHAIN) {
            if (w == 0 && tile != 0 && (uint32_t) lane < P)
                gpos[lane] = (unsigned long long) tile * (TILE / P);
        }
        else if (w == 0 && tile != 0) {
            // Lane = (j, b): predecessor slot j of a load, destination b.  One
            // poll fetches OS_LOOK loads x tpw predecessors; every decision is
            // wave-uniform (ballots), so the walk is a scalar loop.
            const uint32_t Pp = 1u << p.nbits_pad;    // P rounded up to a power of two
            const uint32_t tpw = 64u >> p.nbits_pad;  // predecessors per load
            const uint32_t j = (uint32_t) lane >> p.nbits_pad, b = (uint32_t) lane & (Pp - 1);
            const bool bin_ok = b < P;
            const uint64_t gmask = (Pp == 64) ? ~0ull : ((1ull << Pp) - 1ull);
            const uint32_t gshift = j << p.nbits_pad;
            unsigned long long acc = 0;
            unsigned long long st_t0 = 0;
            uint32_t st_polls = 0, st_idle = 0;
            if (VAR & OSV_STATS) st_t0 = __builtin_readcyclecounter();
            int64_t look = (int64_t) tile - 1; // nearest predecessor not yet accounted for
            bool done = false;
            uint32_t idle = 0;
            while (!done) {
                unsigned long long v[OS_LOOK];
#pragma unroll
                for (int q = 0; q < OS_LOOK; q++) {
                    const int64_t t = look - (int64_t)(q * tpw + j);
                    v[q] = (bin_ok && t >= 0) ? os_load_status(p.status + (uint64_t) t * P + b)
                                              : OS_FLAG_INCL; // before tile 0 / padding lane
                }
                uint32_t consumed = 0;
                bool stop = false;
#pragma unroll
                for (int q = 0; q < OS_LOOK; q++) {
                    if (stop) continue;
                    const unsigned long long f = v[q] >> 62;
                    const int64_t t = look - (int64_t)(q * tpw + j);
                    const bool pad = !bin_ok;
                    // state of predecessor (q, j): all of its destinations agree
                    const uint64_t m_agg = __ballot(f == 1 || pad);
                    const uint64_t m_inc = __ballot(f == 2 || pad);
                    const bool g_agg = ((m_agg >> gshift) & gmask) == gmask;
                    const bool g_inc = ((m_inc >> gshift) & gmask) == gmask;
                    const uint64_t t_inc = __ballot(g_inc);
                    const uint64_t t_wait = __ballot(!g_agg && !g_inc);
                    const uint32_t ji = t_inc ? (uint32_t)(__ffsll((unsigned long long) t_inc) - 1) >> p.nbits_pad : 64u;
                    const uint32_t jn = t_wait ? (uint32_t)(__ffsll((unsigned long long) t_wait) - 1) >> p.nbits_pad : 64u;
                    const unsigned long long val = (bin_ok && t >= 0) ? (v[q] & OS_VALUE_MASK) : 0ull;
                    if (ji < jn) { // reached an inclusive prefix: finished
                        if (j <= ji) acc += val;
                        done = true;
                        stop = true;
                    }
                    else if (jn < 64u) { // predecessor jn of this load is not published yet
                        if (j < jn) acc += val;
                        consumed += jn;
                        stop = true;
                    }
                    else { // tpw aggregates
                        acc += val;
                        consumed += tpw;
                    }
                }
                look -= consumed;
                if (VAR & OSV_STATS) {
                    st_polls++;
                    if (consumed == 0 && !done) st_idle++;
                }
                if (!done) {
                    if (consumed == 0) {
                        // Bounded wait: give up after OS_SPIN_LIMIT empty polls, and leave
                        // at once when any other tile has already given up.
                        ++idle;
                        if (idle > OS_SPIN_LIMIT) {
                            raise_error(p.ctrl, MGS_ERR_TIMEOUT);
                            done = true;
                        }
                        else if ((idle & 255u) == 0
                                 && __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED,
                                                      __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            done = true;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    else {
                        idle = 0;
                    }
                }
            }
            if ((VAR & OSV_STATS) && lane == 0) {
                const unsigned long long dt = __builtin_readcyclecounter() - st_t0;
                const unsigned long long depth = (unsigned long long) ((int64_t) tile - 1 - look);
                atomicAdd(&p.ctrl->head.stats[0], 1ull);                        // tiles that scanned
                atomicAdd(&p.ctrl->head.stats[1], (unsigned long long) st_polls); // polls
                atomicAdd(&p.ctrl->head.stats[2], (unsigned long long) st_idle);  // polls with nothing new
                atomicAdd(&p.ctrl->head.stats[3], depth);                       // predecessors summed
                atomicAdd(&p.ctrl->head.stats[4], dt);                          // cycles inside the scan
                atomicMax(&p.ctrl->head.stats[5], depth);
                atomicMax(&p.ctrl->head.stats[6], dt);
                atomicMax(&p.ctrl->head.stats[7], (unsigned long long) st_polls);
            }
            // fold the per-slot partial sums: lanes with equal b, strides Pp..32
            for (uint32_t o = Pp; o < 64u; o <<= 1) {
                const uint32_t lo = __shfl_xor((uint32_t) acc, (int) o, 64);
                const uint32_t hi = __shfl_xor((uint32_t)(acc >> 32), (int) o, 64);
                acc += ((unsigned long long) hi << 32) | lo;
            }
            if ((uint32_t) lane < P) { // j == 0 and b == lane
                os_store_status(p.status + (uint64_t) tile * P + lane,
                                OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
                gpos[lane] = acc;
            }
        }
        if (!(VAR & OSV_NO_PREFETCH) && w == 0) {
            if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        }
        __syncthreads(); // S2: LDS image and gpos complete

        // ---- stream the image out, one partition run after another ----------
        const uint32_t total = binstart[64];
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t s = (uint32_t) k * THREADS + tid;
            if (s < total) {
                uint32_t b = 0;
                for (uint32_t step = 1u << (p.nbits - 1); step > 0; step >>= 1) {
                    const uint32_t nb = b + step;
                    if (nb < P && binstart[nb] <= s) b = nb;
                }
                const unsigned long long pos = gpos[b] + (s - binstart[b]);
                if (pos < pcap[b]) {
                    uint4 *dst = reinterpret_cast<uint4 *>(pbase[b]) + pos;
                    const uint4 v = srec[s];
                    if (VAR & OSV_NT_STORE) {
                        unsigned long long *q = reinterpret_cast<unsigned long long *>(dst);
                        __builtin_nontemporal_store(((unsigned long long) v.y << 32) | v.x, q);
                        __builtin_nontemporal_store(((unsigned long long) v.w << 32) | v.z, q + 1);
                    }
                    else {
                        *dst = v;
                    }
                }
                else if ((int32_t) b == p.self) {
                    raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                }
                else {
                    overflow = true;
                }
            }
        }
        if (overflow) atomicMin(&p.ctrl->head.cut_tile, tile);

        if (VAR & OSV_NO_PREFETCH) {
            if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        }
        tile = next;
        par ^= 1;
        // S1 of the next trip keeps the image intact until every lane has stored its slots
    }
}

// Calibration: the same ticketed tile loop doing nothing but load + store
// (what this load/store structure can stream at on the device at hand).
template <int THREADS, int ITEMS, int VAR>
__global__ __launch_bounds__(THREADS) void k_copy16(const uint4 *in, uint4 *out, uint64_t n,
                                                    uint32_t ntiles, uint32_t *ticket)
{
    constexpr int TILE = THREADS * ITEMS;
    __shared__ uint32_t nexttile;
    const int tid = threadIdx.x;
    for (;;) {
        if (tid == 0) nexttile = atomicAdd(ticket, 1u);
        __syncthreads();
        const uint32_t tile = nexttile;
        __syncthreads();
        if (tile >= ntiles) break;
        uint4 r[ITEMS];
        const uint64_t idx0 = (uint64_t) tile * TILE + tid;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint64_t i = idx0 + (uint32_t) k * THREADS;
            r[k] = (i < n) ? in[i] : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint64_t i = idx0 + (uint32_t) k * THREADS;
            if (i < n) out[i] = r[k];
        }
    }
}

// One wave.  Computes send_offset[] and the counters at the end of the round:
// either everything (no record overflowed) or everything before the first
// record that did not fit, which lies in the lowest tile that reported one.
#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
__global__ __launch_bounds__(64) void k_os_finalize(OsParams p)
{
    __shared__ volatile unsigned long long run[OS_MAXP]; // records of the cut tile seen so far, per dest
    const int lane = threadIdx.x;
    const uint32_t P = p.P;
    mgs_round *res = &p.ctrl->result;
    const uint32_t cut = p.ctrl->head.cut_tile;
    const bool live = (uint32_t) lane < P;

    unsigned long long before = 0; // records in partition `lane` before the cut tile
    if (live) {
        const uint32_t t = (cut == 0xFFFFFFFFu) ? p.ntiles : cut;
        before = (t == 0) ? (unsigned long long) (p.ctrl->head.part_init[lane] >> 4)
                          : (p.status[(uint64_t)(t - 1) * P + lane] & OS_VALUE_MASK);
    }
    run[lane] = 0;
    unsigned long long e = p.n;
    if (cut != 0xFFFFFFFFu) {
        const uint64_t t0 = (uint64_t) cut * p.tile_records;
        bool found = false;
        for (int row = 0; row < (int) (p.tile_records / 64) && !found; row++) {
            const uint64_t i = t0 + (uint32_t) row * 64 + lane;
            const bool valid = i < p.n;
            uint32_t d = 0;
            if (valid) {
                const uint4 r = p.rec[i];
                d = fastmod(hash_rec16(r.x, r.y, r.z, r.w, p.klen), p.fm);
            }
            uint64_t m = __ballot(valid);
            for (uint32_t bit = 0; bit < p.nbits; bit++) {
                const bool one = (d >> bit) & 1u;
                const uint64_t bk = __ballot(one);
                m &= one ? bk : ~bk;
            }
            const uint32_t rank = lanes_below(m);
            bool over = false;
            if (valid && (int32_t) d != p.self) {
                const unsigned long long start =
                    (cut == 0) ? (unsigned long long) (p.ctrl->head.part_init[d] >> 4)
                               : (p.status[(uint64_t)(cut - 1) * P + d] & OS_VALUE_MASK);
                const unsigned long long pos = start + run[d] + rank;
                over = pos >= (unsigned long long) (p.ctrl->head.part_cap[d] >> 4);
            }
            const uint64_t om = __ballot(over);
            uint64_t keep = __ballot(valid); // lanes of this row that still belong to the round
            if (om) {
                const int fl = __ffsll((unsigned long long) om) - 1;
                e = t0 + (uint32_t) row * 64 + fl;
                keep &= (1ull << fl) - 1ull;
                found = true;
            }
            // count the kept lanes per destination (leader of each group adds)
            const uint64_t mk = m & keep;
            const bool mine = valid && ((keep >> lane) & 1ull);
            if (mine && lanes_below(mk) == 0) run[d] += (unsigned long long) __popcll(mk);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    unsigned long long mytot = before + run[lane];
    if (live) {
        if ((int32_t) lane == p.self) {
            res->send_offset[lane] = 0;
            res->local_bytes = (int64_t)(mytot << 4);
        }
        else {
            res->send_offset[lane] = (int64_t)(mytot << 4);
        }
    }
    // local records: everything of `self` before e
    unsigned long long lrec = 0;
    if (p.self >= 0) {
        const unsigned long long init = (unsigned long long) (p.ctrl->head.part_init[p.self] >> 4);
        const uint32_t t = (cut == 0xFFFFFFFFu) ? p.ntiles : cut;
        const unsigned long long b4 =
            (t == 0) ? init : (p.status[(uint64_t)(t - 1) * P + p.self] & OS_VALUE_MASK);
        lrec = b4 + run[p.self] - init;
    }
    if (lane == 0) {
        res->next = p.first + e;
        res->local_records = lrec;
        res->packed_records = e - lrec;
        if (p.self < 0) res->local_bytes = 0;
    }
}
#endif

} // namespace mgs
#endif
