// k_onesweepd: single-pass stable hash-partition of byte-granular records
// (variable width: NUL-terminated keys and/or values located by a uint32 offset
// array; fixed widths other than 16 bytes), nranks <= 64, written straight to the
// partitions -- no LDS image of the output.
//
// Every byte is read from HBM once and written once.  A tile is T = 448 * K
// consecutive records; worker wave w owns records [w*64*K, (w+1)*64*K) of it, lane l
// of the wave K consecutive records, so a wave's records are one contiguous span of
// the input.
//
//   stage     each worker wave loads ITS OWN span with aligned 16-byte loads and parks
//             it in its own slice of LDS (two slices per wave: the tile being hashed
//             and the tile being stored).  Nothing a wave reads from the staging area
//             was written by another wave, so staging needs no workgroup barrier.
//   hash      key length by the layout's rule (src/serializer.h:418-421), lookup3 over
//             the key bytes read from LDS with one unaligned 16-byte read (the kernel is
//             bound by instruction issue, not by LDS: an unaligned read costs LDS cycles,
//             a funnel shift costs VALU instructions), dest = hash % nranks.
//   rank      as in k_onesweepv: per pair of destinations one wave scan of the lanes'
//             byte totals; wave totals per destination go to LDS, the last worker wave
//             to finish publishes the tile's counts for the chained scan.
//   barrier   ONE workgroup barrier per tile: after it every wave knows the bytes of
//             each destination in the waves before it, and the scan wave's tables for
//             the PREVIOUS tile are visible.
//   chain     the windowed chained scan of mgs_onesweep16w.h over byte counts, run by
//             the scan wave while the workers hash the next tile.
//   store     one trip later every lane copies its K records from its LDS slice to
//             partition address + bytes of the destination before the record, with
//             overlapping unaligned 8/16-byte pieces (first piece at the record's
//             start, last piece ending at its end).  The L2 merges the pieces of
//             neighbouring records into whole lines before they reach HBM.
//   rounds    as in k_onesweep16q: the scan wave keeps the round state, the tile that
//             holds a cut walks its records (destinations kept in LDS, lengths from the
//             offsets) and publishes the new round; tiles inside one round store
//             without any check.
//
// The kernel needs every wave's span to fit its LDS slice; for variable-width input
// k_osd_check tests that on the device first and raises Ctrl::head.fallback otherwise.
#ifndef MGS_ONESWEEPD_H
#define MGS_ONESWEEPD_H

#include "mgs_onesweepv.h"

namespace mgs {

constexpr int OD_SB = 4864;        // LDS staging bytes per worker wave and tile
constexpr int OD_FRONT = 16;       // bytes kept free at the start of a slice
constexpr int OD_SLACK = OD_FRONT + 15 + 20; // front + alignment shift of a span + bytes read past its last record

__device__ __forceinline__ uint4 od_read16(const uint8_t *p)
{
    const OvU128 v = *reinterpret_cast<const OvU128 *>(p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// len bytes from LDS to global memory, both at any alignment: pieces of 16 (8, 4) bytes,
// the last one ending at the record's end (it may overlap the one before: same bytes).
#define OD_GLOBAL __attribute__((address_space(1)))
typedef uint32_t od_v4 __attribute__((ext_vector_type(4)));
typedef od_v4 od_v4_u __attribute__((aligned(1)));            // 16 bytes at any address
typedef unsigned long long od_u64_u __attribute__((aligned(1)));
typedef uint32_t od_u32_u __attribute__((aligned(1)));
__device__ __forceinline__ void od_copy_out(unsigned long long gaddr, const uint8_t *src, uint32_t len)
{
    OD_GLOBAL uint8_t *ga = (OD_GLOBAL uint8_t *) gaddr; // a global-memory address, not a generic one
    if (len >= 16) {
        uint32_t j = 0;
        for (; j + 16 <= len; j += 16) *(OD_GLOBAL od_v4_u *) (ga + j) = *reinterpret_cast<const od_v4_u *>(src + j);
        if (j < len) *(OD_GLOBAL od_v4_u *) (ga + len - 16) = *reinterpret_cast<const od_v4_u *>(src + len - 16);
    }
    else if (len >= 8) {
        *(OD_GLOBAL od_u64_u *) ga = *reinterpret_cast<const od_u64_u *>(src);
        *(OD_GLOBAL od_u64_u *) (ga + len - 8) = *reinterpret_cast<const od_u64_u *>(src + len - 8);
    }
    else if (len >= 4) {
        *(OD_GLOBAL od_u32_u *) ga = *reinterpret_cast<const od_u32_u *>(src);
        *(OD_GLOBAL od_u32_u *) (ga + len - 4) = *reinterpret_cast<const od_u32_u *>(src + len - 4);
    }
    else {
        if (len >= 1) ga[0] = src[0];
        if (len >= 2) ga[1] = src[1];
        if (len >= 3) ga[2] = src[2];
    }
}

// Checked build: is [ga, ga + n) inside the caller's buffers?  Counts the stores that are not.
template <int VAR>
__device__ __forceinline__ bool od_store_ok(const OvParams &q, unsigned long long ga, uint32_t n, uint32_t what)
{
    if (!(VAR & OSV_NT_LOAD)) return true;
    const bool ok = (ga >= q.dbg_lo[0] && ga + n <= q.dbg_hi[0]) || (ga >= q.dbg_lo[1] && ga + n <= q.dbg_hi[1]);
    if (!ok) {
        atomicAdd(&q.os.ctrl->head.stats[0], 1ull);
        q.os.ctrl->head.stats[1] = ga;
        q.os.ctrl->head.stats[2] = ((unsigned long long) what << 32) | n;
        q.os.ctrl->head.stats[3] = ((unsigned long long) blockIdx.x << 32) | threadIdx.x;
    }
    return ok;
}

// flags (0x80 per byte) of the zero bytes of v; the lowest flag is exact
__device__ __forceinline__ uint32_t od_zero_bytes(uint32_t v)
{
    return (v - 0x01010101u) & ~v & 0x80808080u;
}

// Does every worker wave's span of every tile fit its LDS slice?  One thread per
// (tile, wave).  records_per_wave = 64 * K.
#ifdef MGS_OSD_UNIT // defined by the one translation unit that launches it
__global__ void k_osd_check(OvParams q, uint32_t records_per_wave, uint32_t waves)
{
    const uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t t = g / waves, w = g % waves;
    if (t >= q.os.ntiles) return;
    const uint64_t T = q.os.tile_records;
    const uint64_t i0 = t * T + w * records_per_wave;
    if (i0 >= q.n) return;
    const uint64_t i1 = min(q.n, min(t * T + T, i0 + records_per_wave));
    const uint64_t s = ov_offset(q, i0), e = ov_offset(q, i1);
    const uint64_t shift = ((uintptr_t) q.blob + s) & 15u;
    if (e < s || e - s + shift + OD_FRONT + 20 > (uint64_t) OD_SB)
        __hip_atomic_store(&q.os.ctrl->head.fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// The tile starts in round k_cur and holds at least one cut: find every cut in emit
// order, publish the rounds they begin (see os_find_cuts16; here positions are bytes,
// a record's destination comes from the tile's destination bytes `sd` in LDS and its
// length from the offsets).  One wave; lane = column for acc / tot / base / lim.
template <int K>
__device__ __forceinline__ void od_find_cuts(const OvParams &q, uint32_t tile, int nseg, int lane,
                                             unsigned long long acc, uint32_t tot, const uint32_t *wc,
                                             const uint8_t *sd, bool is_part, unsigned long long my_cap,
                                             uint32_t &k_cur, unsigned long long &base, unsigned long long &lim,
                                             unsigned long long &first_nb, OsCutLds *x)
{
    const OsParams &p = q.os;
    const uint32_t P = q.P, PC = p.P;
    const uint32_t k_in = k_cur;
    uint32_t c = 0, wst = (uint32_t) nseg;
    for (int ww = 0; ww < nseg; ww++) {
        const uint32_t v = wc[ww * OS_MAXP + lane];
        if (is_part && wst == (uint32_t) nseg && acc + c + v > lim) wst = (uint32_t) ww;
        c += v;
    }
    uint32_t wstar = (uint32_t) nseg;
    for (int ww = nseg - 1; ww >= 0; ww--)
        if (__ballot(wst == (uint32_t) ww)) wstar = (uint32_t) ww;
    uint32_t run = 0;
    for (int ww = 0; ww < nseg; ww++)
        if ((uint32_t) ww < wstar) run += wc[ww * OS_MAXP + lane];
    x->pos[lane] = acc + run;
    x->lim[lane] = is_part ? lim : ~0ull;
    __builtin_amdgcn_wave_barrier();
    const uint64_t t0 = (uint64_t) tile * p.tile_records;
    bool stop = false;
    for (int seg = (int) wstar; seg < nseg && !stop; seg++) {
        // lengths of the segment's K rows of 64 records, requested at once
        uint32_t ln[K];
#pragma unroll
        for (int j = 0; j < K; j++) {
            const uint64_t i = t0 + (uint32_t) (seg * K + j) * 64 + lane;
            ln[j] = 0;
            if (i < q.n) {
                const uint64_t o0 = ov_offset(q, i), o1 = ov_offset(q, i + 1);
                if (o1 > o0 && o1 - o0 <= GEN_MAX_RECORD) ln[j] = (uint32_t) (o1 - o0);
            }
        }
#pragma unroll
        for (int j = 0; j < K; j++) {
            if (stop) continue;
            const uint32_t in_tile = (uint32_t) (seg * K + j) * 64 + (uint32_t) lane;
            const uint32_t d = sd[in_tile], len = ln[j];
            const bool valid = d != 0xFFu && len != 0;
            // bytes of the same destination in the lanes below, one destination at a time
            unsigned long long pos = 0;
            uint64_t remaining = __ballot(valid);
            while (remaining) {
                const int leader = __ffsll((unsigned long long) remaining) - 1;
                const uint32_t b = (uint32_t) __shfl((int) d, leader);
                const bool inb = valid && d == b;
                const uint32_t v = inb ? len : 0u;
                const uint32_t incl = wave_incl_scan_u32(v, lane);
                if (inb) pos = x->pos[b] + (incl - v);
                remaining &= ~__ballot(inb);
            }
            uint64_t open = ~0ull;
            for (;;) {
                const bool over = valid && ((open >> lane) & 1ull) && (int32_t) d != p.self && pos + len > x->lim[d];
                const uint64_t om = __ballot(over);
                if (!om) break;
                const int fl = __ffsll((unsigned long long) om) - 1;
                const uint64_t below = (1ull << fl) - 1ull;
                x->cnt[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                if (valid && lane < fl) {
                    atomicAdd(const_cast<uint32_t *>(&x->cnt[d]), len);
                    if ((int32_t) d == p.self) atomicAdd(const_cast<uint32_t *>(&x->cnt[P]), 1u);
                }
                __builtin_amdgcn_wave_barrier();
                k_cur++;
                const unsigned long long nb = x->pos[lane] + x->cnt[lane];
                if (k_cur == k_in + 1) first_nb = nb;
                if ((uint32_t) lane < PC) os_store_status(p.rt + (uint64_t) k_cur * PC + lane, OS_RT_VALID | nb);
                if (lane == 0) {
                    os_store_status(p.rt_start + k_cur, OS_RT_VALID | (t0 + in_tile - (uint32_t) lane + (uint32_t) fl));
                    atomicMax(&p.ctrl->head.rounds_cut, k_cur);
                }
                if (is_part) {
                    base = nb;
                    lim = nb + my_cap;
                    x->lim[lane] = lim;
                }
                __builtin_amdgcn_wave_barrier();
                open &= ~below;
                // a record that does not fit an empty partition (src/collectiveshuffler.h:156-159)
                const bool never = valid && lane == fl && pos + len > x->lim[d];
                if (__ballot(never)) {
                    raise_error(p.ctrl, MGS_ERR_RECORD_SIZE);
                    stop = true;
                    break;
                }
                if (k_cur >= p.max_rounds) {
                    stop = true;
                    break;
                }
            }
            if (valid) {
                atomicAdd(const_cast<unsigned long long *>(&x->pos[d]), (unsigned long long) len);
                if ((int32_t) d == p.self) atomicAdd(const_cast<unsigned long long *>(&x->pos[P]), 1ull);
            }
            __builtin_amdgcn_wave_barrier();
            if (!__ballot(is_part && acc + tot > lim)) stop = true;
        }
    }
}

// LM (layout mode) spares the per-record layout decisions: 1 = string key(s) and a
// fixed-width or absent value (key = the record minus the value, ends with its only
// NUL); 2 = fixed-width key and a fixed-width or absent value; 0 = any layout.
template <int THREADS, int K, int VAR, int MINW, int LOOK, int LM>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweepd(OvParams q)
{
    const OsParams &p = q.os;
    constexpr int WORKERS = THREADS / 64 - 1, WT = WORKERS * 64, T = WT * K;
    constexpr int CPT = (OD_SB / 16 + 63) / 64; // 16-byte chunks per lane of a wave's slice
    __shared__ __attribute__((aligned(16))) uint8_t sin[2 * WORKERS * OD_SB];
    __shared__ uint32_t wcnt[2 * WORKERS * OS_MAXP];   // bytes per (wave, dest), two tiles
    __shared__ uint8_t sdst[2 * T];                    // destination of every record of the tile, emit order
    __shared__ unsigned long long gaddr2[2 * OS_MAXP]; // address of the tile's first byte of each destination
    __shared__ unsigned long long gpos2[2 * OS_MAXP];  // running total of each column before the tile
    __shared__ unsigned long long rb2[2 * 2 * OS_MAXP]; // checked path: totals at the tile's first two rounds
    __shared__ volatile uint32_t misc[12]; // [0],[1] first two tiles; [2+par] ticket; [4+par] mode; [6+par],[8+par] rounds
    __shared__ uint32_t arrive[2];
    __shared__ OsCutLds cutlds;

    if (p.ctrl->head.fallback != 0) return; // the generic path handles this call

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t P = q.P;   // destinations
    const uint32_t PC = p.P;  // chain columns

    for (int j = tid; j < 2 * WORKERS * OS_MAXP; j += THREADS) wcnt[j] = 0;
    if (tid == 0) {
        arrive[0] = 0;
        arrive[1] = 0;
        misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
        misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
        misc[10] = atomicAdd(&p.ctrl->head.ticket, 1u);
    }
    __syncthreads();
    uint32_t tile = misc[0], next = misc[1];
    int par = 0;

    // ------------------------------------------------------------------ scan wave
    if (w == WORKERS) {
        __builtin_amdgcn_s_setprio(3);
        OsWalkStats ws;
        const bool live = (uint32_t) lane < PC, is_dest = (uint32_t) lane < P;
        const bool is_part = is_dest && lane != p.self, is_self = lane == p.self;
        const unsigned long long my_base = is_dest ? p.ctrl->head.part_base[lane] : 0ull;
        const unsigned long long my_cap = is_dest ? (unsigned long long) p.ctrl->head.part_cap[lane] : 0ull;
        uint32_t k_cur = 0;
        unsigned long long base = 0, lim = my_cap;
        // diagnostic build (scan-wave flavour): cycles [0] waiting at the barrier [1] in the chained scan
        // [2] everything else, [3] tiles, [4] polls
        unsigned long long sph[3] = {0, 0, 0}, stiles = 0, st0 = 0;
        constexpr bool SCAN_STATS = (VAR & OSV_STATS) && (VAR & OSV_NO_PREFETCH);
        if (SCAN_STATS) st0 = __builtin_readcyclecounter();
#define OD_SPHASE(i)                                                     \
    if (SCAN_STATS) {                                                    \
        const unsigned long long tnow = __builtin_readcyclecounter();    \
        sph[i] += tnow - st0;                                            \
        st0 = tnow;                                                      \
    }
        // Tickets are drawn one trip before they are handed to the workers: the atomic's round
        // trip runs beside the chained scan instead of in front of the barrier.
        uint32_t drawn = misc[10];
        while (tile < p.ntiles) {
            if (lane == 0) misc[2 + par] = drawn; // the tile after next, for everybody
            OD_SPHASE(2)
            __syncthreads(); // A: counts(tile) complete
            OD_SPHASE(0)
            if (lane == 0) drawn = atomicAdd(&p.ctrl->head.ticket, 1u); // lane 0 alone needs it, a trip from now
            const uint32_t next2 = misc[2 + par];
            const uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
            uint32_t tot = 0; // lane = column
#pragma unroll
            for (int ww = 0; ww < WORKERS; ww++) tot += wc[ww * OS_MAXP + lane];
            unsigned long long acc = 0;
            if (tile == 0) {
                acc = is_dest ? (unsigned long long) p.ctrl->head.part_init[lane] : 0ull;
                if (live) os_store_status(p.status + lane, OS_FLAG_INCL | (acc + tot));
            }
            else if (VAR & OSV_FAKE_CHAIN) {
                acc = (unsigned long long) tile * 3600ull;
            }
            else {
                OD_SPHASE(2)
                acc = os_window_walk<VAR, LOOK>(p, tile, lane, &ws);
                if (live)
                    os_store_status(p.status + (uint64_t) tile * PC + lane, OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
                OD_SPHASE(1)
                stiles++;
            }
            // rounds (see k_onesweep16q)
            if (!(VAR & OSV_FAKE_CHAIN) && k_cur < p.max_rounds) {
                while (__ballot(is_part && acc > lim)) {
                    k_cur++;
                    const unsigned long long nb = os_rt_wait(p, k_cur, lane, live);
                    if (is_part) {
                        base = nb;
                        lim = nb + my_cap;
                    }
                    if (k_cur >= p.max_rounds
                        || __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                        break;
                }
            }
            const uint32_t kfirst = k_cur;
            uint32_t mode = 1;
            if (VAR & OSV_FAKE_CHAIN) mode = 1;
            else if (k_cur >= p.max_rounds) mode = 2;
            else if (__ballot(is_part && acc + tot > lim)) {
                unsigned long long first_nb = 0;
                rb2[(par * 2 + 0) * OS_MAXP + lane] = base;
                od_find_cuts<K>(q, tile, WORKERS, lane, acc, tot, wc, sdst + par * T, is_part, my_cap, k_cur, base, lim,
                                first_nb, &cutlds);
                rb2[(par * 2 + 1) * OS_MAXP + lane] = first_nb;
                mode = 0;
            }
            else if (__ballot(is_self && acc + tot > my_cap)) {
                rb2[(par * 2 + 0) * OS_MAXP + lane] = base;
                mode = 0;
            }
            gpos2[par * OS_MAXP + lane] = acc;
            gaddr2[par * OS_MAXP + lane] =
                my_base + (is_part ? (unsigned long long) kfirst * p.image_stride : 0ull) + (acc - base);
            if (lane == 0) {
                misc[4 + par] = mode;
                misc[6 + par] = kfirst;
                misc[8 + par] = k_cur;
            }
            tile = next;
            next = next2;
            par ^= 1;
        }
        __syncthreads(); // matches the workers' last barrier
        if (SCAN_STATS && lane == 0) {
            atomicAdd(&p.ctrl->head.stats[0], sph[0]);
            atomicAdd(&p.ctrl->head.stats[1], sph[1]);
            atomicAdd(&p.ctrl->head.stats[2], sph[2]);
            atomicAdd(&p.ctrl->head.stats[3], stiles);
            atomicAdd(&p.ctrl->head.stats[4], ws.polls);
            atomicAdd(&p.ctrl->head.stats[5], ws.idle);
            atomicAdd(&p.ctrl->head.stats[6], ws.depth);
        }
#undef OD_SPHASE
        return;
    }

    // ------------------------------------------------------------------- workers
    // Written for instruction-level parallelism: a wave has three other waves on its SIMD
    // at most, so every phase first issues all its LDS / memory reads for the lane's K
    // records and only then uses them; the per-record decisions are selects, not branches.
    const int wt = w * 64 + lane; // owns records wt*K .. wt*K+K-1 of a tile
    // this wave's two LDS regions: `stg` holds the tile being hashed as it lies in the input,
    // `img` the previous tile's records grouped by destination (this wave's share of every run)
    uint8_t *const stg = sin + (0 * WORKERS + w) * OD_SB, *const img = sin + (1 * WORKERS + w) * OD_SB;
    uint32_t lo[K], len[K];                    // current tile: first byte in this wave's slice, length (0 = no record)
    // previous tile, waiting to be stored, packed to spare registers (used by the checked path only):
    // pl = first byte in the image | length << 16, pt = bytes of its destination before it in the tile | destination << 16
    uint32_t pl[K], pt[K];
    uint32_t below_prev = 0; // lane = destination: its bytes of the previous tile in the waves before this one
    const uint32_t vlen = (q.lay.val_kind == MGS_VAL_FIXED) ? (uint32_t) q.lay.val_bytes : 0u;

    // Offsets of a tile's records as fetched: the K+1 record boundaries of this lane, as
    // 32-bit values relative to `base` (the offset array's own values with base 0, or
    // multiples of the fixed width with base = the tile's first byte).
    struct Offsets {
        uint64_t base;
        uint32_t rec[K + 1];
        uint32_t nrec; // records of the tile this lane owns (0..K)
    };
    auto fetch_offsets = [&](uint32_t t, Offsets &f) {
        const uint64_t i0 = (uint64_t) t * T, i1 = min(q.n, i0 + T);
        const uint64_t mine = i0 + (uint64_t) wt * K;
        f.nrec = mine >= i1 ? 0u : (uint32_t) min((uint64_t) K, i1 - mine);
        if (q.rec_off) {
            const uint32_t *ro = q.rec_off + q.first;
            f.base = 0;
#pragma unroll
            for (int k = 0; k <= K; k++) f.rec[k] = ro[min(i1, mine + k)];
        }
        else {
            f.base = (q.first + i0) * (uint64_t) q.fixed_len;
#pragma unroll
            for (int k = 0; k <= K; k++) f.rec[k] = (uint32_t) (min(i1, mine + k) - i0) * q.fixed_len;
        }
    };
    // this wave's span of the tile: from its first lane's first record to its last lane's end
    auto wave_span = [&](const Offsets &f, uint64_t &s, uint64_t &e) {
        s = f.base + (uint32_t) __builtin_amdgcn_readfirstlane((int) f.rec[0]);
        e = f.base + (uint32_t) __builtin_amdgcn_readlane((int) f.rec[K], 63);
    };
    auto fetch_chunks = [&](uint64_t s, uint64_t e, uint4 (&ch)[CPT]) {
        const uintptr_t a0 = ((uintptr_t) q.blob + s) & ~(uintptr_t) 15;
        const uint32_t nch = e > s ? (uint32_t) (((uintptr_t) q.blob + e - a0 + 15) >> 4) : 0u;
        const OD_GLOBAL od_v4 *g = (const OD_GLOBAL od_v4 *) a0;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint32_t c = (uint32_t) j * 64 + (uint32_t) lane;
            ch[j] = make_uint4(0, 0, 0, 0);
            if (c < nch) { // nothing outside the span is touched: it may end where the caller's buffer ends
                const od_v4 t = g[c];
                ch[j] = make_uint4(t.x, t.y, t.z, t.w);
            }
        }
    };
    // (the slice's first 16 bytes stay free: a short record's last 16 bytes may begin before it)
    auto park_chunks = [&](uint8_t *slice, uint64_t s, uint64_t e, const uint4 (&ch)[CPT]) {
        const uintptr_t a0 = ((uintptr_t) q.blob + s) & ~(uintptr_t) 15;
        const uint32_t nch = e > s ? (uint32_t) (((uintptr_t) q.blob + e - a0 + 15) >> 4) : 0u;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint32_t c = (uint32_t) j * 64 + (uint32_t) lane;
            if (c < nch && c < (uint32_t) (OD_SB / 16 - 1)) reinterpret_cast<uint4 *>(slice + OD_FRONT)[c] = ch[j];
        }
    };
    auto set_records = [&](const Offsets &f, uint64_t s, uint64_t e) {
        const uint32_t shift = (uint32_t) (((uintptr_t) q.blob + s) & 15u) + OD_FRONT;
        const uint32_t s32 = (uint32_t) (s - f.base);
        const uint64_t span = (e - s) + shift;
        bool bad = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t l = f.rec[k + 1] - f.rec[k]; // wraps to a huge value when not monotonic
            const uint32_t r = f.rec[k] - s32;
            const bool mine = (uint32_t) k < f.nrec;
            const bool ok = l != 0 && l <= GEN_MAX_RECORD && f.rec[k] >= s32 && (uint64_t) r + shift + l <= span
                            && (uint64_t) r + shift + l + 20 <= (uint64_t) OD_SB;
            bad |= mine && !ok;
            len[k] = (mine && ok) ? l : 0u;
            lo[k] = (mine && ok) ? r + shift : (uint32_t) OD_FRONT; // a safe place to read from
        }
        if (bad) raise_error(p.ctrl, MGS_ERR_MALFORMED);
    };

    Offsets fn; // offsets of the tile after the current one, fetched one trip ahead of its bytes
    fn.base = 0;
    fn.nrec = 0;
#pragma unroll
    for (int k = 0; k <= K; k++) fn.rec[k] = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        lo[k] = OD_FRONT;
        len[k] = 0;
        pl[k] = OD_FRONT;
        pt[k] = 0;
    }
    {
        Offsets f0;
        uint4 ch[CPT];
        if (tile < p.ntiles) {
            fetch_offsets(tile, f0);
            if (next < p.ntiles) fetch_offsets(next, fn);
            uint64_t s, e;
            wave_span(f0, s, e);
            fetch_chunks(s, e, ch);
            park_chunks(stg, s, e, ch);
            set_records(f0, s, e);
        }
    }
    bool have_prev = false;
    // diagnostic build: cycles of worker wave 0 per phase of the trip
    // [0] hash  [1] rank + publish  [2] wait at the barrier  [3] counts + issue loads  [4] store  [5] rotate
    // [6] park + records + offsets  [7] trips
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (VAR & OSV_STATS) tprev = __builtin_readcyclecounter();
#define OD_PHASE(i)                                                      \
    if (VAR & OSV_STATS) {                                               \
        const unsigned long long tnow = __builtin_readcyclecounter();    \
        ph[i] += tnow - tprev;                                           \
        tprev = tnow;                                                    \
    }

    for (;;) {
        const bool live = tile < p.ntiles; // the last trip only stores the last tile
        uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
        const uint8_t *sw = stg;
        uint32_t d[K], pw[K];
        if (live) {
            // ---- hash: destination of each of this lane's records
            uint32_t h[K];
            if (LM != 0) {
                uint4 v[K];
                uint32_t klen[K];
                bool bad = false, longkey = false;
#pragma unroll
                for (int k = 0; k < K; k++) v[k] = od_read16(sw + lo[k]);
#pragma unroll
                for (int k = 0; k < K; k++) {
                    bool ok;
                    if (LM == 1) {
                        klen[k] = len[k] - vlen;
                        ok = len[k] > vlen;
                    }
                    else {
                        klen[k] = (uint32_t) q.lay.key_bytes;
                        ok = klen[k] <= len[k];
                    }
                    const uint32_t kl = klen[k];
                    // the whole key as lookup3's tail: three words masked to the key's bytes
                    const uint32_t m0 = kl >= 4 ? 0xFFFFFFFFu : ((1u << (8 * (kl & 3u))) - 1u);
                    const uint32_t m1 = kl >= 8 ? 0xFFFFFFFFu : (kl > 4 ? ((1u << (8 * (kl & 3u))) - 1u) : 0u);
                    const uint32_t m2 = kl >= 12 ? 0xFFFFFFFFu : (kl > 8 ? ((1u << (8 * (kl & 3u))) - 1u) : 0u);
                    if (LM == 1) {
                        // the key's first NUL must be its last byte (strlen + 1, src/serializer.h:131-136)
                        const uint32_t z0 = od_zero_bytes(v[k].x | ~m0), z1 = od_zero_bytes(v[k].y | ~m1),
                                       z2 = od_zero_bytes(v[k].z | ~m2);
                        const uint32_t fz = z0 ? ((uint32_t) __builtin_ctz(z0) >> 3)
                                               : (z1 ? 4u + ((uint32_t) __builtin_ctz(z1) >> 3)
                                                     : (z2 ? 8u + ((uint32_t) __builtin_ctz(z2) >> 3) : 99u));
                        ok = ok && (kl > 12 || fz == kl - 1);
                    }
                    uint32_t a, b, c;
                    a = b = c = 0xdeadbeefu + kl;
                    a += v[k].x & m0;
                    b += v[k].y & m1;
                    c += v[k].z & m2;
                    MGS_FINAL(a, b, c);
                    h[k] = c;
                    longkey |= ok && len[k] != 0 && kl > 12;
                    if (len[k] != 0 && !ok) {
                        bad = true;
                        len[k] = 0;
                    }
                }
                if (__any(longkey)) { // keys beyond lookup3's 12-byte tail: the block loop (rare)
#pragma unroll 1
                    for (int k = 0; k < K; k++) {
                        if (len[k] == 0 || klen[k] <= 12) continue;
                        const uint8_t *r = sw + lo[k];
                        bool ok = true;
                        if (LM == 1) {
                            ok = r[klen[k] - 1] == 0;
                            for (uint32_t j = 0; j + 1 < klen[k]; j++) ok = ok && r[j] != 0;
                        }
                        h[k] = ov_hash_lds(sw, lo[k], klen[k]);
                        if (!ok) {
                            bad = true;
                            len[k] = 0;
                        }
                    }
                }
                if (bad) raise_error(p.ctrl, MGS_ERR_MALFORMED);
            }
            else {
#pragma unroll 1
                for (int k = 0; k < K; k++) {
                    h[k] = 0;
                    if (len[k] == 0) continue;
                    uint32_t klen = 0;
                    if (ov_key_len(q.lay, sw + lo[k], len[k], klen)) {
                        h[k] = ov_hash_lds(sw, lo[k], klen);
                    }
                    else {
                        raise_error(p.ctrl, MGS_ERR_MALFORMED);
                        len[k] = 0;
                    }
                }
            }
            if (p.fm.pow2) {
#pragma unroll
                for (int k = 0; k < K; k++) d[k] = len[k] ? fastmod_pow2(h[k], p.fm) : 0xFFu;
            }
            else {
#pragma unroll
                for (int k = 0; k < K; k++) d[k] = len[k] ? fastmod_magic(h[k], p.fm) : 0xFFu;
            }
            if (p.dest_out) {
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (len[k]) p.dest_out[(uint64_t) tile * T + (uint64_t) wt * K + k] = (uint8_t) d[k];
            }
            // destinations in emit order, for the walk of a tile that holds a round cut
#pragma unroll
            for (int k = 0; k < K; k++) sdst[par * T + wt * K + k] = (uint8_t) d[k];
            OD_PHASE(0)

            // ---- rank: byte position of each record inside the wave's share of its destination.
            // Two destinations share one scan: 16-bit fields (a wave's bytes fit: one LDS slice).
            uint32_t field[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                pw[k] = 0;
                field[k] = len[k] << (16 * (d[k] & 1u));
            }
            for (uint32_t b2 = 0; b2 < (P + 1) / 2; b2++) {
                uint32_t mine = 0;
#pragma unroll
                for (int k = 0; k < K; k++) mine += ((d[k] >> 1) == b2) ? field[k] : 0u;
                const uint32_t incl = wave_incl_scan_u32(mine, lane);
                uint32_t run = incl - mine;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const bool in = (d[k] >> 1) == b2;
                    pw[k] = in ? ((run >> (16 * (d[k] & 1u))) & 0xffffu) : pw[k];
                    run += in ? field[k] : 0u;
                }
                if (lane == 63) {
                    wc[w * OS_MAXP + 2 * b2] = incl & 0xffffu;
                    wc[w * OS_MAXP + 2 * b2 + 1] = incl >> 16;
                }
            }
            if (p.self >= 0) { // chain column P: records of the local-bypass destination
                uint32_t cnt = 0;
#pragma unroll
                for (int k = 0; k < K; k++) cnt += (uint32_t) __popcll(__ballot(d[k] == (uint32_t) p.self));
                if (lane == 63) wc[w * OS_MAXP + P] = cnt;
            }
            {
                // the last worker wave to get here publishes the tile's byte counts
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                uint32_t arrived = 0;
                if (lane == 0) arrived = atomicAdd(&arrive[par], 1u);
                arrived = __builtin_amdgcn_readfirstlane(arrived);
                if (arrived % (uint32_t) WORKERS == (uint32_t) WORKERS - 1u) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    uint32_t tot_all = 0;
#pragma unroll
                    for (int ww = 0; ww < WORKERS; ww++)
                        tot_all += ((volatile uint32_t *) wc)[ww * OS_MAXP + lane];
                    if ((uint32_t) lane >= PC) tot_all = 0;
                    os_publish_agg16(p, tile, lane, tot_all);
                }
            }
        }
        OD_PHASE(1)
        __syncthreads(); // A: counts(tile) complete; the scan wave's tables of the previous tile are ready
        OD_PHASE(2)
        uint32_t next2 = 0xFFFFFFFFu, below = 0;
        uint4 ch[CPT];
        uint64_t ns = 0, ne = 0;
        if (live) {
            next2 = misc[2 + par];
            __builtin_amdgcn_s_setprio(1);
            // lane = destination: its bytes in the waves before this one
            uint32_t cw[WORKERS];
#pragma unroll
            for (int ww = 0; ww < WORKERS; ww++) cw[ww] = wc[ww * OS_MAXP + lane];
#pragma unroll
            for (int ww = 0; ww < WORKERS; ww++) below += (ww < w) ? cw[ww] : 0u;
            // bytes of the next tile (its offsets arrived during the last trip): in flight during the store
            if (next < p.ntiles) {
                wave_span(fn, ns, ne);
                fetch_chunks(ns, ne, ch);
            }
        }
        OD_PHASE(3)
        // ---- store the previous tile: from this wave's other slice straight to the partitions
        if (have_prev) {
            const int pr = par ^ 1;
            const uint32_t mode = misc[4 + pr];
            const uint8_t *slice = img;
            if (mode == 1) {
                // This wave's share of every destination's run lies contiguous in `img`.  It leaves as
                // 16-byte chunks aligned to the GLOBAL address, one chunk per lane, consecutive lanes =
                // consecutive chunks of a run (then the next destination's run); the two ragged ends of
                // a run leave as unaligned 16-byte stores that overlap their neighbours (same bytes).
                // lane = destination:
                const uint32_t *wcp = wcnt + pr * (WORKERS * OS_MAXP);
                const uint32_t rlen = ((uint32_t) lane < P) ? wcp[w * OS_MAXP + lane] : 0u;
                const uint32_t rsrc = wave_incl_scan_u32(rlen, lane) - rlen + OD_FRONT;   // run's first byte in img
                const unsigned long long rdst = gaddr2[pr * OS_MAXP + lane] + below_prev; // ... and in the partition
                const uint32_t ra = (uint32_t) rdst & 15u;
                const uint32_t rnch = rlen >= 16u ? (ra + rlen + 15u) >> 4 : 0u;          // shorter runs: below
                const uint32_t cincl = wave_incl_scan_u32(rnch, lane);
                const uint32_t rcst = cincl - rnch;
                const uint32_t total = (uint32_t) __builtin_amdgcn_readlane((int) cincl, 63);
                const uint32_t rlo = (uint32_t) rdst, rhi = (uint32_t) (rdst >> 32);
                // lane = chunk: find the chunk's run (the last destination whose first chunk is at or
                // before it), then its place in the run
                auto chunk = [&](uint32_t c, uint32_t &soff, unsigned long long &gdst) -> bool {
                    uint32_t pos = 0;
                    for (uint32_t step = 1u << (p.nbits_pad ? p.nbits_pad - 1 : 0); step > 0; step >>= 1) {
                        const uint32_t cand = pos + step;
                        const uint32_t cs = (uint32_t) __shfl((int) rcst, (int) (cand & 63u));
                        pos = (cand < P && cs <= c) ? cand : pos;
                    }
                    const uint32_t len = (uint32_t) __shfl((int) rlen, (int) pos);
                    const uint32_t src = (uint32_t) __shfl((int) rsrc, (int) pos);
                    const uint32_t j = c - (uint32_t) __shfl((int) rcst, (int) pos);
                    const unsigned long long G = ((unsigned long long) (uint32_t) __shfl((int) rhi, (int) pos) << 32)
                                                 | (uint32_t) __shfl((int) rlo, (int) pos);
                    const uint32_t aa = (uint32_t) G & 15u;
                    const bool full = 16u * j >= aa && 16u * j - aa + 16u <= len;
                    const uint32_t off = full ? 16u * j - aa : (j == 0 ? 0u : len - 16u);
                    soff = src + off;
                    gdst = G + off;
                    return c < total;
                };
                if (!(q.dbg_flags & 1u)) {
                    constexpr int NIT = 6; // covers a full region (OD_SB / 16 / 64 rounded up); a loop takes the rest
#pragma unroll
                    for (int g = 0; g < NIT; g += 2) { // two chunks per lane at a time: reads first, then stores
                        if ((uint32_t) g * 64u >= total) break;
                        uint32_t so[2];
                        unsigned long long gd[2];
                        bool on[2];
                        od_v4_u val[2];
#pragma unroll
                        for (int it = 0; it < 2; it++) {
                            on[it] = chunk((uint32_t) (g + it) * 64u + (uint32_t) lane, so[it], gd[it]);
                            if (!on[it]) so[it] = OD_FRONT;
                            val[it] = *reinterpret_cast<const od_v4_u *>(img + so[it]);
                        }
#pragma unroll
                        for (int it = 0; it < 2; it++)
                            if (on[it] && od_store_ok<VAR>(q, gd[it], 16u, 500u)) *(OD_GLOBAL od_v4_u *) gd[it] = val[it];
                    }
                    for (uint32_t c0 = NIT * 64u; c0 < total; c0 += 64u) {
                        uint32_t so1;
                        unsigned long long gd1;
                        if (chunk(c0 + (uint32_t) lane, so1, gd1) && od_store_ok<VAR>(q, gd1, 16u, 501u))
                            *(OD_GLOBAL od_v4_u *) gd1 = *reinterpret_cast<const od_v4_u *>(img + so1);
                    }
                    // runs under 16 bytes: their destination's lane copies them
                    if (rlen != 0u && rlen < 16u && od_store_ok<VAR>(q, rdst, rlen, 502u)) od_copy_out(rdst, img + rsrc, rlen);
                }
            }
            else if (mode == 0) {
                // checked path: round cuts inside the tile, or the local stream runs full
                const unsigned long long *gpos = gpos2 + pr * OS_MAXP;
                const uint32_t kfirst = misc[6 + pr], klast = misc[8 + pr];
#pragma unroll 1
                for (int k = 0; k < K; k++) {
                    const uint32_t plo = pl[k] & 0xffffu, plen = pl[k] >> 16;
                    if (!plen) continue;
                    const uint32_t b = pt[k] >> 16;
                    const unsigned long long g = gpos[b] + (pt[k] & 0xffffu); // bytes of b before this record
                    const unsigned long long cap = (unsigned long long) p.ctrl->head.part_cap[b];
                    if ((int32_t) b == p.self) {
                        const unsigned long long lbase = p.ctrl->head.part_base[b];
                        if (g + plen <= cap) {
                            if (od_store_ok<VAR>(q, lbase + g, plen, 200u)) od_copy_out(lbase + g, slice + plo, plen);
                        }
                        else raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                        continue;
                    }
                    uint32_t kk = kfirst;
                    unsigned long long bk = rb2[(pr * 2 + 0) * OS_MAXP + b];
                    if (kk < klast && rb2[(pr * 2 + 1) * OS_MAXP + b] <= g) {
                        bk = rb2[(pr * 2 + 1) * OS_MAXP + b];
                        kk++;
                        while (kk < klast) {
                            const unsigned long long bn = os_load_status(p.rt + (uint64_t) (kk + 1) * PC + b) & ~OS_RT_VALID;
                            if (bn > g) break;
                            bk = bn;
                            kk++;
                        }
                    }
                    if (kk < p.max_rounds && g - bk + plen <= cap
                        && od_store_ok<VAR>(q, p.ctrl->head.part_base[b] + (unsigned long long) kk * p.image_stride + (g - bk),
                                            plen, 300u + b))
                        od_copy_out(p.ctrl->head.part_base[b] + (unsigned long long) kk * p.image_stride + (g - bk),
                                    slice + plo, plen);
                }
            }
            // mode 2: the tile lies after the last round, nothing of it is consumed
        }
        OD_PHASE(4)
        if (!live) break;
        // ---- reorder: this tile's records from `stg` into `img`, grouped by destination, emit order
        // inside a group (wave-private: only this wave reads or writes these regions)
        {
            const uint32_t wl = ((uint32_t) lane < P) ? wc[w * OS_MAXP + lane] : 0u; // lane = destination
            const uint32_t wstart = wave_incl_scan_u32(wl, lane) - wl + OD_FRONT;
            uint32_t ipos[K];
            bool rare = false;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const uint32_t dk = len[k] ? d[k] : 0u;
                ipos[k] = (uint32_t) __shfl((int) wstart, (int) dk) + pw[k];
                rare |= len[k] > 32 || (len[k] != 0 && len[k] < 8);
                pl[k] = ipos[k] | (len[k] << 16);
                pt[k] = ((uint32_t) __shfl((int) below, (int) dk) + pw[k]) | (dk << 16);
            }
#pragma unroll
            for (int k0 = 0; k0 < K; k0 += 2) {
                uint4 r1[2], r2[2];
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (k0 + j >= K) continue;
                    r1[j] = od_read16(stg + lo[k0 + j]);
                    r2[j] = od_read16(stg + lo[k0 + j] + len[k0 + j] - 16); // lo >= 16: stays inside the region
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    if (k0 + j >= K) continue;
                    const uint32_t l = len[k0 + j];
                    uint8_t *o = img + ipos[k0 + j];
                    if (l >= 16) {
                        *reinterpret_cast<od_v4_u *>(o) = od_v4{r1[j].x, r1[j].y, r1[j].z, r1[j].w};
                        *reinterpret_cast<od_v4_u *>(o + l - 16) = od_v4{r2[j].x, r2[j].y, r2[j].z, r2[j].w};
                    }
                    else if (l >= 8) {
                        *reinterpret_cast<od_u64_u *>(o) = ((unsigned long long) r1[j].y << 32) | r1[j].x;
                        *reinterpret_cast<od_u64_u *>(o + l - 8) = ((unsigned long long) r2[j].w << 32) | r2[j].z;
                    }
                }
            }
            if (__any(rare)) { // the middle of long records, and records under 8 bytes
#pragma unroll 1
                for (int k = 0; k < K; k++) {
                    if (len[k] > 32) ov_copy_lds(img + ipos[k] + 16, stg + lo[k] + 16, len[k] - 32);
                    else if (len[k] != 0 && len[k] < 8) ov_copy_lds(img + ipos[k], stg + lo[k], len[k]);
                }
            }
            below_prev = below;
        }
        OD_PHASE(5)
        // the staging region is free (this wave's own reads of it are done): stage the next tile
        if (next < p.ntiles) {
            park_chunks(stg, ns, ne, ch);
            set_records(fn, ns, ne);
        }
        else {
#pragma unroll
            for (int k = 0; k < K; k++) {
                len[k] = 0;
                lo[k] = OD_FRONT;
            }
        }
        if (next2 < p.ntiles) fetch_offsets(next2, fn);
        __builtin_amdgcn_s_setprio(0);
        OD_PHASE(6)
        ph[7] += 1;
        have_prev = true;
        tile = next;
        next = next2;
        par ^= 1;
    }
    if ((VAR & OSV_STATS) && !(VAR & OSV_NO_PREFETCH) && wt == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(&p.ctrl->head.stats[i], ph[i]);
    }
#undef OD_PHASE
}

} // namespace mgs
#endif
