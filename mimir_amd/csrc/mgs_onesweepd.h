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
constexpr int OD_SLACK = 15 + 20;  // alignment shift of a span + bytes read past its last record

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

// flags (0x80 per byte) of the zero bytes of v; the lowest flag is exact
__device__ __forceinline__ uint32_t od_zero_bytes(uint32_t v)
{
    return (v - 0x01010101u) & ~v & 0x80808080u;
}

// Does every worker wave's span of every tile fit its LDS slice?  One thread per
// (tile, wave).  records_per_wave = 64 * K.
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
    if (e < s || e - s + shift + 20 > (uint64_t) OD_SB)
        __hip_atomic_store(&q.os.ctrl->head.fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
        while (tile < p.ntiles) {
            // the tile after next, for everybody, drawn before the barrier
            if (lane == 0) misc[2 + par] = atomicAdd(&p.ctrl->head.ticket, 1u);
            __syncthreads(); // A: counts(tile) complete
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
                acc = os_window_walk<VAR, LOOK>(p, tile, lane, &ws);
                if (live)
                    os_store_status(p.status + (uint64_t) tile * PC + lane, OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
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
        if ((VAR & OSV_STATS) && lane == 0) ws.flush(p.ctrl);
        return;
    }

    // ------------------------------------------------------------------- workers
    const int wt = w * 64 + lane; // owns records wt*K .. wt*K+K-1 of a tile
    uint8_t *const slice0 = sin + (0 * WORKERS + w) * OD_SB, *const slice1 = sin + (1 * WORKERS + w) * OD_SB;
    uint32_t lo[K], len[K];       // current tile: first byte in this wave's slice, length (0 = no record)
    uint32_t plo[K], plen[K], ptoff[K], pd[K]; // previous tile, waiting to be stored

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
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint32_t c = (uint32_t) j * 64 + (uint32_t) lane;
            ch[j] = (c < nch) ? reinterpret_cast<const uint4 *>(a0)[c] : make_uint4(0, 0, 0, 0);
        }
    };
    auto park_chunks = [&](uint8_t *slice, uint64_t s, uint64_t e, const uint4 (&ch)[CPT]) {
        const uintptr_t a0 = ((uintptr_t) q.blob + s) & ~(uintptr_t) 15;
        const uint32_t nch = e > s ? (uint32_t) (((uintptr_t) q.blob + e - a0 + 15) >> 4) : 0u;
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint32_t c = (uint32_t) j * 64 + (uint32_t) lane;
            if (c < nch && c < (uint32_t) (OD_SB / 16)) reinterpret_cast<uint4 *>(slice)[c] = ch[j];
        }
    };
    auto set_records = [&](const Offsets &f, uint64_t s, uint64_t e) {
        const uint32_t shift = (uint32_t) (((uintptr_t) q.blob + s) & 15u);
        const uint32_t s32 = (uint32_t) (s - f.base);
        const uint64_t span = (e - s) + shift;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t l = f.rec[k + 1] - f.rec[k]; // wraps to a huge value when not monotonic
            const uint32_t r = f.rec[k] - s32;
            lo[k] = r + shift;
            len[k] = l;
            if ((uint32_t) k >= f.nrec) {
                len[k] = 0; // past the end of the call
                continue;
            }
            if (l == 0 || l > GEN_MAX_RECORD || f.rec[k] < s32 || (uint64_t) r + shift + l > span
                || (uint64_t) r + shift + l + 20 > (uint64_t) OD_SB) {
                raise_error(p.ctrl, MGS_ERR_MALFORMED);
                len[k] = 0;
            }
        }
    };
    // Copy the previous tile's records (parity pr) from this wave's slice to their partitions.
    auto store_prev = [&](int pr) {
        const uint32_t mode = misc[4 + pr];
        if (mode == 2) return; // after the last round: nothing of that tile is consumed
        const uint8_t *slice = pr ? slice1 : slice0;
        const unsigned long long *gaddr = gaddr2 + pr * OS_MAXP;
        if (mode == 1) {
#pragma unroll
            for (int k = 0; k < K; k++)
                if (plen[k]) od_copy_out(gaddr[pd[k]] + ptoff[k], slice + plo[k], plen[k]);
            return;
        }
        const unsigned long long *gpos = gpos2 + pr * OS_MAXP;
        const uint32_t kfirst = misc[6 + pr], klast = misc[8 + pr];
#pragma unroll 1
        for (int k = 0; k < K; k++) {
            if (!plen[k]) continue;
            const uint32_t b = pd[k];
            const unsigned long long g = gpos[b] + ptoff[k]; // bytes of b before this record
            const unsigned long long cap = (unsigned long long) p.ctrl->head.part_cap[b];
            if ((int32_t) b == p.self) {
                const unsigned long long lbase = p.ctrl->head.part_base[b];
                if (g + plen[k] <= cap) od_copy_out(lbase + g, slice + plo[k], plen[k]);
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
            if (kk < p.max_rounds && g - bk + plen[k] <= cap)
                od_copy_out(p.ctrl->head.part_base[b] + (unsigned long long) kk * p.image_stride + (g - bk),
                            slice + plo[k], plen[k]);
        }
    };

    Offsets fn; // offsets of the tile after the current one, fetched one trip ahead of its bytes
    fn.base = 0;
    fn.nrec = 0;
#pragma unroll
    for (int k = 0; k <= K; k++) fn.rec[k] = 0;
    {
        Offsets f0;
        uint4 ch[CPT];
        if (tile < p.ntiles) {
            fetch_offsets(tile, f0);
            if (next < p.ntiles) fetch_offsets(next, fn);
            uint64_t s, e;
            wave_span(f0, s, e);
            fetch_chunks(s, e, ch);
            park_chunks(slice0, s, e, ch);
            set_records(f0, s, e);
        }
    }
    bool have_prev = false;
#pragma unroll
    for (int k = 0; k < K; k++) plo[k] = plen[k] = ptoff[k] = pd[k] = 0;

    while (tile < p.ntiles) {
        uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
        const uint8_t *sw = par ? slice1 : slice0;
        // ---- hash: destination of each of this lane's records
        uint32_t d[K], h[K];
        const uint32_t vlen = (q.lay.val_kind == MGS_VAL_FIXED) ? (uint32_t) q.lay.val_bytes : 0u;
#pragma unroll
        for (int k = 0; k < K; k++) {
            h[k] = 0;
            if (len[k] == 0) continue;
            const uint8_t *r = sw + lo[k];
            uint32_t klen = 0;
            bool ok;
            if (LM == 1) {
                klen = len[k] - vlen;
                ok = len[k] > vlen;
            }
            else if (LM == 2) {
                klen = (uint32_t) q.lay.key_bytes;
                ok = klen <= len[k];
            }
            else {
                ok = ov_key_len(q.lay, r, len[k], klen);
            }
            if (ok && klen <= 12) {
                // the whole key is lookup3's tail: three words, masked to the key's bytes
                const uint4 v = od_read16(r);
                const uint32_t m0 = klen >= 4 ? 0xFFFFFFFFu : ((1u << (8 * klen)) - 1u);
                const uint32_t m1 = klen >= 8 ? 0xFFFFFFFFu : (klen > 4 ? ((1u << (8 * (klen - 4))) - 1u) : 0u);
                const uint32_t m2 = klen >= 12 ? 0xFFFFFFFFu : (klen > 8 ? ((1u << (8 * (klen - 8))) - 1u) : 0u);
                if (LM == 1) {
                    // the key's first NUL must be its last byte (strlen + 1, src/serializer.h:131-136)
                    const uint32_t z0 = od_zero_bytes(v.x | ~m0), z1 = od_zero_bytes(v.y | ~m1),
                                   z2 = od_zero_bytes(v.z | ~m2);
                    const uint32_t fz = z0 ? ((uint32_t) __builtin_ctz(z0) >> 3)
                                           : (z1 ? 4u + ((uint32_t) __builtin_ctz(z1) >> 3)
                                                 : (z2 ? 8u + ((uint32_t) __builtin_ctz(z2) >> 3) : 99u));
                    ok = fz == klen - 1;
                }
                uint32_t a, b, c;
                a = b = c = 0xdeadbeefu + klen;
                a += v.x & m0;
                b += v.y & m1;
                c += v.z & m2;
                MGS_FINAL(a, b, c);
                h[k] = c;
            }
            else if (ok) {
                if (LM == 1) {
                    ok = r[klen - 1] == 0;
                    for (uint32_t j = 0; j + 1 < klen; j++) ok = ok && r[j] != 0;
                }
                h[k] = ov_hash_lds(sw, lo[k], klen);
            }
            if (!ok) {
                raise_error(p.ctrl, MGS_ERR_MALFORMED);
                len[k] = 0;
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

        // ---- rank: byte position of each record inside the wave's share of its destination.
        // Two destinations share one scan: 16-bit fields (a wave's bytes fit: one LDS slice).
        uint32_t pw[K], field[K];
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
                if ((d[k] >> 1) == b2) {
                    pw[k] = (run >> (16 * (d[k] & 1u))) & 0xffffu;
                    run += field[k];
                }
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
        __syncthreads(); // A: counts(tile) complete; the scan wave's tables of the previous tile are ready
        const uint32_t next2 = misc[2 + par];
        __builtin_amdgcn_s_setprio(1);

        uint32_t below = 0; // lane = destination: its bytes in the waves before this one
#pragma unroll
        for (int ww = 0; ww < WORKERS; ww++)
            if (ww < w) below += wc[ww * OS_MAXP + lane];
        // bytes of the next tile (its offsets arrived during the last trip): in flight during the store
        uint4 ch[CPT];
        uint64_t ns = 0, ne = 0;
        if (next < p.ntiles) {
            wave_span(fn, ns, ne);
            fetch_chunks(ns, ne, ch);
        }
        if (have_prev) store_prev(par ^ 1);
        // this tile's records wait one trip for their positions
#pragma unroll
        for (int k = 0; k < K; k++) {
            plo[k] = lo[k];
            plen[k] = len[k];
            pd[k] = len[k] ? d[k] : 0u;
            ptoff[k] = (uint32_t) __shfl((int) below, (int) (len[k] ? d[k] : 0u)) + pw[k];
        }
        // the other slice is free (this wave's own reads of it are done): stage the next tile
        if (next < p.ntiles) {
            park_chunks(par ? slice0 : slice1, ns, ne, ch);
            set_records(fn, ns, ne);
        }
        else {
#pragma unroll
            for (int k = 0; k < K; k++) len[k] = 0;
        }
        if (next2 < p.ntiles) fetch_offsets(next2, fn);
        __builtin_amdgcn_s_setprio(0);
        have_prev = true;
        tile = next;
        next = next2;
        par ^= 1;
    }
    __syncthreads(); // tables of the last tile are ready
    if (have_prev) store_prev(par ^ 1);
}

} // namespace mgs
#endif
