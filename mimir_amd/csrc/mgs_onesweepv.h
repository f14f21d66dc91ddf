// k_onesweepv: single-pass stable hash-partition of byte-granular records:
// variable width (NUL-terminated keys and/or values, located by a uint32
// offset array) and fixed widths other than 16 bytes; nranks <= 64.
//
// Every byte of a record is read from HBM once and written once.  A tile is
// T consecutive records, i.e. one contiguous span of the input; positions are
// counted in bytes.
//
//   stage     the tile's span is loaded with aligned 16-byte loads (whole-wave
//             contiguous accesses) and parked in LDS (`sin`);
//   hash      lane l owns K consecutive records: key length by the layout's
//             rule (src/serializer.h:418-421), lookup3 over the key bytes read
//             from LDS as unaligned dwords, dest = hash % nranks;
//   rank      emit order inside a tile is (wave, lane, k).  For every
//             destination b one wave scan over the lanes' byte totals for b
//             gives each lane its start inside the wave's share of b; inside a
//             lane the K records are accumulated in order.  Wave totals per
//             destination go to LDS; the last worker wave to finish publishes
//             the tile's per-destination byte counts (16-bit fields) for the
//             chained scan, before the barrier;
//   reorder   records are copied LDS -> LDS into an image grouped by
//             destination (`sout`), byte-granular, stable;
//   chain     the windowed chained scan of mgs_onesweep16w.h, run by a scan
//             wave, over byte counts; its first poll is in flight across the
//             workers' two staging barriers;
//   store     one trip later, the image is streamed out in 16-byte chunks
//             aligned to the *global* address of each partition run (unaligned
//             LDS reads, aligned full-line-friendly global stores); the partial
//             chunks at the two ends of a run are written byte by byte.  A run
//             is clipped at the partition's capacity; a clipped tile records its
//             index (round cut), k_osv_finalize re-walks that one tile.
//
// The kernel needs every tile's span to fit the LDS staging buffer.  For
// variable-width input k_osv_check tests that on the device first and raises
// Ctrl::head.fallback otherwise; the one-sweep kernels then return at once and
// the generic path, enqueued behind them, does the work.
#ifndef MGS_ONESWEEPV_H
#define MGS_ONESWEEPV_H

#include "mgs_generic.h"
#include "mgs_onesweep16w.h"

namespace mgs {

constexpr int OV_IN_BYTES = 35840; // LDS staging capacity for one tile's span (incl. alignment slack)

struct OvParams {
    const uint8_t *blob;
    const uint32_t *rec_off; // absolute record index -> byte offset in blob, or NULL (fixed width)
    uint32_t fixed_len;
    uint64_t first; // absolute index of record 0 of this call
    uint64_t n;     // records in this call
    mgs_layout lay;
    uint32_t P; // destinations
    // Chain geometry: os.P = P, plus one more column counting the records of the
    // local-bypass destination when there is one (os.self >= 0); os.nbits_pad and
    // os.lpt_log follow os.P.  Also self, fm, ntiles, tile_records, status, agg16,
    // dest_out, ctrl.
    OsParams os;
};

struct __attribute__((packed)) OvU16 { uint16_t v; };
struct __attribute__((packed)) OvU32 { uint32_t v; };
struct __attribute__((packed)) OvU64 { unsigned long long v; };
struct __attribute__((packed)) OvU128 { uint32_t x, y, z, w; };

// byte offset in blob of record i of this call (i <= n)
__device__ __forceinline__ uint64_t ov_offset(const OvParams &q, uint64_t i)
{
    return q.rec_off ? (uint64_t) q.rec_off[q.first + i] : (q.first + i) * (uint64_t) q.fixed_len;
}

// lookup3 hashlittle (src/hash.cpp:152-400, initval 0) over klen >= 1 bytes at s (LDS,
// any alignment; up to 11 bytes past the key may be read and are masked off).
__device__ __forceinline__ uint32_t ov_hash_lds(const uint8_t *s, uint32_t klen)
{
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + klen;
    uint32_t rem = klen;
    while (rem > 12) {
        a += reinterpret_cast<const OvU32 *>(s)->v;
        b += reinterpret_cast<const OvU32 *>(s + 4)->v;
        c += reinterpret_cast<const OvU32 *>(s + 8)->v;
        MGS_MIX(a, b, c);
        s += 12;
        rem -= 12;
    }
    const uint32_t w0 = reinterpret_cast<const OvU32 *>(s)->v;
    const uint32_t w1 = reinterpret_cast<const OvU32 *>(s + 4)->v;
    const uint32_t w2 = reinterpret_cast<const OvU32 *>(s + 8)->v;
    // word j keeps its first clamp(rem - 4j, 0, 4) bytes
    const uint32_t m0 = rem >= 4 ? 0xFFFFFFFFu : ((1u << (8 * rem)) - 1u);
    const uint32_t m1 = rem >= 8 ? 0xFFFFFFFFu : (rem > 4 ? ((1u << (8 * (rem - 4))) - 1u) : 0u);
    const uint32_t m2 = rem >= 12 ? 0xFFFFFFFFu : (rem > 8 ? ((1u << (8 * (rem - 8))) - 1u) : 0u);
    a += w0 & m0;
    b += w1 & m1;
    c += w2 & m2;
    MGS_FINAL(a, b, c);
    return c;
}

// Key length of one wire-form record held in LDS (same rules as gen_key_len).
__device__ __forceinline__ bool ov_key_len(const mgs_layout &lay, const uint8_t *r, uint32_t len, uint32_t &klen)
{
    if (lay.key_kind == MGS_KEY_FIXED) {
        klen = (uint32_t) lay.key_bytes;
        if (lay.val_kind == MGS_VAL_CSTR) return klen < len && r[len - 1] == 0;
        return klen <= len;
    }
    if (lay.val_kind != MGS_VAL_CSTR) {
        const uint32_t vlen = (lay.val_kind == MGS_VAL_FIXED) ? (uint32_t) lay.val_bytes : 0u;
        if (len < vlen + 1) return false;
        klen = len - vlen;
        return r[klen - 1] == 0;
    }
    uint32_t pos = 0;
    for (int s = 0; s < lay.key_count; s++) {
        while (pos < len && r[pos] != 0) pos++;
        if (pos >= len) return false;
        pos++;
    }
    klen = pos;
    return pos < len && r[len - 1] == 0;
}

// LDS -> LDS copy of len bytes, both sides at any alignment.
__device__ __forceinline__ void ov_copy_lds(uint8_t *dst, const uint8_t *src, uint32_t len)
{
    uint32_t j = 0;
    for (; j + 16 <= len; j += 16)
        *reinterpret_cast<OvU128 *>(dst + j) = *reinterpret_cast<const OvU128 *>(src + j);
    if (len & 8) {
        *reinterpret_cast<OvU64 *>(dst + j) = *reinterpret_cast<const OvU64 *>(src + j);
        j += 8;
    }
    if (len & 4) {
        *reinterpret_cast<OvU32 *>(dst + j) = *reinterpret_cast<const OvU32 *>(src + j);
        j += 4;
    }
    if (len & 2) {
        *reinterpret_cast<OvU16 *>(dst + j) = *reinterpret_cast<const OvU16 *>(src + j);
        j += 2;
    }
    if (len & 1) dst[j] = src[j];
}

// The workgroup barrier without the wait for outstanding memory operations
// that the compiler puts in front of the builtin: the scan wave crosses the
// workers' staging barriers with its poll still in flight.
__device__ __forceinline__ void ov_raw_barrier()
{
    asm volatile("s_barrier" ::: "memory");
}

// Does every tile's span fit the staging buffer?  One thread per tile.
__global__ void k_osv_check(OvParams q)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= q.os.ntiles) return;
    const uint64_t T = q.os.tile_records;
    const uint64_t i0 = (uint64_t) t * T, i1 = min(q.n, i0 + T);
    const uint64_t s = ov_offset(q, i0), e = ov_offset(q, i1);
    const uint64_t shift = ((uintptr_t) q.blob + s) & 15u;
    if (e < s || e - s + shift > (uint64_t) OV_IN_BYTES)
        __hip_atomic_store(&q.os.ctrl->head.fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int THREADS, int K, int VAR, int MINW, int LOOK>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweepv(OvParams q)
{
    const OsParams &p = q.os;
    constexpr int WORKERS = THREADS / 64 - 1, WT = WORKERS * 64, T = WT * K;
    constexpr int CPT = (OV_IN_BYTES / 16 + WT - 1) / WT; // 16-byte chunks per worker thread
    __shared__ __attribute__((aligned(16))) uint8_t sin[OV_IN_BYTES + 16];
    __shared__ __attribute__((aligned(16))) uint8_t sout[OV_IN_BYTES + 16];
    __shared__ uint32_t wcnt[2 * WORKERS * OS_MAXP]; // bytes per (wave, dest), two tiles
    __shared__ uint32_t wslot[WORKERS * OS_MAXP];    // first image byte of (wave, dest)
    __shared__ uint32_t dstart2[2 * OS_MAXP];        // first image byte of dest, two tiles
    __shared__ unsigned long long gaddr2[2 * OS_MAXP]; // global address of the run's first byte
    __shared__ uint32_t nbytes2[2 * OS_MAXP];          // bytes of the run to store (clipped)
    __shared__ uint32_t cstart2[2 * (OS_MAXP + 1)];    // first chunk of dest's run (+ total)
    __shared__ volatile uint32_t misc[4];              // [0] first tile, [1] second tile / tile after next
    __shared__ uint32_t arrive[2];

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
        OsWalkStats ws;
        const bool live = (uint32_t) lane < PC, is_dest = (uint32_t) lane < P;
        const unsigned long long my_base = is_dest ? p.ctrl->head.part_base[lane] : 0ull;
        const unsigned long long my_cap = is_dest ? (unsigned long long) p.ctrl->head.part_cap[lane] : 0ull;
        __syncthreads(); // the workers' staging barrier of the first tile
        while (tile < p.ntiles) {
            const uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
            __syncthreads(); // A: counts(tile) complete
            uint32_t tot = 0; // lane = destination, bytes
#pragma unroll
            for (int ww = 0; ww < WORKERS; ww++) tot += wc[ww * OS_MAXP + lane];
            unsigned long long acc = 0;
            if (tile == 0) {
                acc = is_dest ? (unsigned long long) p.ctrl->head.part_init[lane] : 0ull;
                if (live) os_store_status(p.status + lane, OS_FLAG_INCL | (acc + tot));
            }
            // the tile after next: drawn two trips ahead so that the workers can fetch a
            // tile's offsets one trip before its bytes
            if (lane == 0) misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
            for (int j = lane; j < WORKERS * OS_MAXP; j += 64) wcnt[(par ^ 1) * (WORKERS * OS_MAXP) + j] = 0;
            if (lane == 0) arrive[par ^ 1] = 0;
            __syncthreads(); // B
            const uint32_t next2 = misc[1];
            if (tile != 0) {
                // first poll in flight across the workers' two staging barriers
                acc = os_window_walk<VAR, LOOK>(p, tile, lane, &ws, [] {
                    ov_raw_barrier(); // C
                    ov_raw_barrier(); // D
                });
                if (live)
                    os_store_status(p.status + (uint64_t) tile * PC + lane,
                                    OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
            }
            else {
                ov_raw_barrier(); // C
                ov_raw_barrier(); // D
            }
            // what the workers need to store this tile's image one trip from now
            const unsigned long long avail = acc < my_cap ? my_cap - acc : 0ull;
            const uint32_t nb = is_dest ? (uint32_t) min((unsigned long long) tot, avail) : 0u;
            if (is_dest && nb < tot) {
                if ((int32_t) lane == p.self) raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                else atomicMin(&p.ctrl->head.cut_tile, tile);
            }
            const unsigned long long G = my_base + acc;
            const uint32_t nch = nb ? (uint32_t) (((G & 15ull) + nb + 15ull) >> 4) : 0u;
            const uint32_t cincl = wave_incl_scan_u32(nch, lane);
            gaddr2[par * OS_MAXP + lane] = G;
            nbytes2[par * OS_MAXP + lane] = nb;
            cstart2[par * (OS_MAXP + 1) + lane] = cincl - nch;
            if (lane == 63) cstart2[par * (OS_MAXP + 1) + OS_MAXP] = cincl;
            tile = next;
            next = next2;
            par ^= 1;
        }
        __syncthreads(); // matches the workers' last barrier
        if ((VAR & OSV_STATS) && lane == 0) ws.flush(p.ctrl);
        return;
    }

    // ------------------------------------------------------------------- workers
    const int wt = w * 64 + lane; // worker thread index: owns records wt*K .. wt*K+K-1 of a tile
    uint32_t lo[K], len[K];       // record k: first byte in sin, length (0 = no record)

    // Fetch the offsets of a tile's records: absolute byte offsets of records
    // wt*K .. wt*K+K (K+1 values), and the tile's span.
    auto fetch_offsets = [&](uint32_t t, uint64_t (&o)[K + 1], uint64_t &s, uint64_t &e) {
        const uint64_t i0 = (uint64_t) t * T, i1 = min(q.n, i0 + T);
        s = ov_offset(q, i0);
        e = ov_offset(q, i1);
#pragma unroll
        for (int k = 0; k <= K; k++) {
            const uint64_t i = min(i1, i0 + (uint64_t) wt * K + k);
            o[k] = ov_offset(q, i);
        }
    };
    // Issue the 16-byte loads of the span [s, e) (aligned down to 16 bytes).
    auto fetch_chunks = [&](uint64_t s, uint64_t e, uint4 (&ch)[CPT]) {
        const uintptr_t a0 = ((uintptr_t) q.blob + s) & ~(uintptr_t) 15;
        const uint32_t nch = (uint32_t) (((uintptr_t) q.blob + e - a0 + 15) >> 4);
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint32_t c = (uint32_t) j * WT + wt;
            ch[j] = (c < nch) ? reinterpret_cast<const uint4 *>(a0)[c] : make_uint4(0, 0, 0, 0);
        }
    };
    auto park_chunks = [&](uint64_t s, uint64_t e, const uint4 (&ch)[CPT]) {
        const uintptr_t a0 = ((uintptr_t) q.blob + s) & ~(uintptr_t) 15;
        const uint32_t nch = (uint32_t) (((uintptr_t) q.blob + e - a0 + 15) >> 4);
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const uint32_t c = (uint32_t) j * WT + wt;
            if (c < nch) reinterpret_cast<uint4 *>(sin)[c] = ch[j];
        }
    };
    auto set_records = [&](const uint64_t (&o)[K + 1], uint64_t s, uint64_t e) {
        const uint32_t shift = (uint32_t) (((uintptr_t) q.blob + s) & 15u);
        const uint32_t span = (uint32_t) (e - s) + shift;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint64_t l64 = o[k + 1] - o[k];
            const uint64_t r64 = o[k] - s + shift;
            lo[k] = (uint32_t) r64;
            len[k] = (uint32_t) l64;
            if (l64 == 0) continue; // past the end of the call
            if (l64 > GEN_MAX_RECORD || o[k] < s || r64 + l64 > span) {
                raise_error(p.ctrl, MGS_ERR_MALFORMED);
                len[k] = 0;
            }
        }
    };
    // Stream out the image of a tile (parity pr).
    auto store_image = [&](int pr) {
        const uint32_t *cstart = cstart2 + pr * (OS_MAXP + 1);
        const uint32_t total = cstart[OS_MAXP];
        for (uint32_t c = (uint32_t) wt; c < total; c += WT) {
            uint32_t b = 0;
#pragma unroll
            for (uint32_t step = OS_MAXP / 2; step > 0; step >>= 1)
                if (cstart[b + step] <= c) b += step;
            const unsigned long long G = gaddr2[pr * OS_MAXP + b];
            const uint32_t nb = nbytes2[pr * OS_MAXP + b];
            const uint8_t *src0 = sout + dstart2[pr * OS_MAXP + b]; // image byte of global byte G
            const unsigned long long g0 = (G & ~15ull) + ((unsigned long long) (c - cstart[b]) << 4);
            if (g0 >= G && g0 + 16 <= G + nb) {
                const OvU128 v = *reinterpret_cast<const OvU128 *>(src0 + (g0 - G));
                *reinterpret_cast<uint4 *>(g0) = make_uint4(v.x, v.y, v.z, v.w);
            }
            else {
                const unsigned long long ga = g0 > G ? g0 : G;
                const unsigned long long gb = g0 + 16 < G + nb ? g0 + 16 : G + nb;
                for (unsigned long long g = ga; g < gb; g++)
                    *reinterpret_cast<uint8_t *>(g) = src0[g - G];
            }
        }
    };

    {
        // first tile: offsets, then bytes, then park them
        uint64_t o[K + 1], s = 0, e = 0;
        uint4 ch[CPT];
        if (tile < p.ntiles) {
            fetch_offsets(tile, o, s, e);
            fetch_chunks(s, e, ch);
            park_chunks(s, e, ch);
            set_records(o, s, e);
        }
        __syncthreads();
    }
    bool have_prev = false;

    while (tile < p.ntiles) {
        uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
        // ---- hash: destination of each of this lane's records
        uint32_t d[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            d[k] = 0xFFu;
            if (len[k] != 0) {
                uint32_t klen = 0;
                if (!ov_key_len(q.lay, sin + lo[k], len[k], klen)) {
                    raise_error(p.ctrl, MGS_ERR_MALFORMED);
                    len[k] = 0;
                }
                else {
                    d[k] = fastmod(ov_hash_lds(sin + lo[k], klen), p.fm);
                    if (p.dest_out) p.dest_out[(uint64_t) tile * T + (uint64_t) wt * K + k] = (uint8_t) d[k];
                }
            }
        }
        // ---- rank: byte position of each record inside the wave's share of its destination
        uint32_t pw[K];
#pragma unroll
        for (int k = 0; k < K; k++) pw[k] = 0;
        for (uint32_t b = 0; b < P; b++) {
            uint32_t mine = 0;
#pragma unroll
            for (int k = 0; k < K; k++) mine += (d[k] == b) ? len[k] : 0u;
            const uint32_t incl = wave_incl_scan_u32(mine, lane);
            uint32_t run = incl - mine;
#pragma unroll
            for (int k = 0; k < K; k++) {
                if (d[k] == b) {
                    pw[k] = run;
                    run += len[k];
                }
            }
            if (lane == 63) wc[w * OS_MAXP + b] = incl;
        }
        if (p.self >= 0) { // chain column P: records of the local-bypass destination
            uint32_t mine = 0;
#pragma unroll
            for (int k = 0; k < K; k++) mine += (d[k] == (uint32_t) p.self) ? 1u : 0u;
            const uint32_t incl = wave_incl_scan_u32(mine, lane);
            if (lane == 63) wc[w * OS_MAXP + P] = incl;
        }
        {
            // the last worker wave to get here publishes the tile's byte counts
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uint32_t arrived = 0;
            if (lane == 0) arrived = atomicAdd(&arrive[par], 1u);
            arrived = __builtin_amdgcn_readfirstlane(arrived);
            if (arrived == (uint32_t) WORKERS - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                uint32_t tot_all = 0;
#pragma unroll
                for (int ww = 0; ww < WORKERS; ww++)
                    tot_all += ((volatile uint32_t *) wc)[ww * OS_MAXP + lane];
                os_publish_agg16(p, tile, lane, tot_all);
            }
        }
        __syncthreads(); // A: counts(tile) complete; tables of the previous tile are ready

        // offsets of the next tile: in flight during the store
        uint64_t o[K + 1], ns = 0, ne = 0;
        if (next < p.ntiles) fetch_offsets(next, o, ns, ne);

        uint32_t tot = 0, below = 0; // lane = destination
#pragma unroll
        for (int ww = 0; ww < WORKERS; ww++) {
            const uint32_t v = wc[ww * OS_MAXP + lane];
            tot += v;
            if (ww < w) below += v;
        }
        if ((uint32_t) lane >= P) tot = 0; // the record-count column takes no image space
        const uint32_t first_byte = wave_incl_scan_u32(tot, lane) - tot;
        wslot[w * OS_MAXP + lane] = first_byte + below;
        if (w == 0) dstart2[par * OS_MAXP + lane] = first_byte;
        if (have_prev) store_image(par ^ 1);
        __syncthreads(); // B: the image is free
        const uint32_t next2 = misc[1];

        // bytes of the next tile: in flight during the reorder
        uint4 ch[CPT];
        if (next < p.ntiles) fetch_chunks(ns, ne, ch);
#pragma unroll
        for (int k = 0; k < K; k++)
            if (len[k] != 0) ov_copy_lds(sout + wslot[w * OS_MAXP + d[k]] + pw[k], sin + lo[k], len[k]);
        __syncthreads(); // C: the staging buffer is free
        if (next < p.ntiles) {
            park_chunks(ns, ne, ch);
            set_records(o, ns, ne);
        }
        __syncthreads(); // D: the next tile is staged
        have_prev = true;
        tile = next;
        next = next2;
        par ^= 1;
    }
    __syncthreads(); // tables of the last tile are ready
    if (have_prev) store_image(par ^ 1);
}

// One wave.  send_offset[] / counts at the point where the round ended; when a
// tile was clipped, re-walk that tile record by record to find the first record
// that does not fit (src/collectiveshuffler.h:148-160).
__global__ __launch_bounds__(64) void k_osv_finalize(OvParams q)
{
    const OsParams &p = q.os;
    if (p.ctrl->head.fallback != 0) return;
    __shared__ volatile unsigned long long run[OS_MAXP];  // bytes of the cut tile kept so far, per dest
    __shared__ volatile unsigned long long lrun[1];       // local records of the cut tile kept so far
    const int lane = threadIdx.x;
    const uint32_t P = q.P, PC = p.P;
    mgs_round *res = &p.ctrl->result;
    const uint32_t cut = p.ctrl->head.cut_tile;
    const bool live = (uint32_t) lane < P;
    const uint32_t t = (cut == 0xFFFFFFFFu) ? p.ntiles : cut;

    // bytes in partition `lane` (lane P: local records) before tile t
    unsigned long long before = 0;
    if ((uint32_t) lane < PC)
        before = (t == 0) ? (live ? (unsigned long long) p.ctrl->head.part_init[lane] : 0ull)
                          : (p.status[(uint64_t)(t - 1) * PC + lane] & OS_VALUE_MASK);
    run[lane] = 0;
    if (lane == 0) lrun[0] = 0;
    unsigned long long e = q.n;
    if (cut != 0xFFFFFFFFu) {
        const uint64_t t0 = (uint64_t) cut * p.tile_records;
        bool found = false;
        for (uint32_t row = 0; row < p.tile_records / 64 && !found; row++) {
            const uint64_t i = t0 + (uint64_t) row * 64 + lane;
            bool valid = i < q.n;
            uint32_t d = 0, len = 0;
            if (valid) {
                const uint64_t o0 = ov_offset(q, i), o1 = ov_offset(q, i + 1);
                len = (uint32_t) (o1 - o0);
                uint32_t klen = 0;
                if (o1 <= o0 || o1 - o0 > GEN_MAX_RECORD || !gen_key_len(q.lay, q.blob + o0, len, klen))
                    valid = false; // reported by the pack kernel
                else
                    d = fastmod(hashlittle_bytes(q.blob + o0, klen), p.fm);
            }
            // byte position of every record of the row inside its destination's share
            unsigned long long pos = 0;
            uint64_t remaining = __ballot(valid);
            while (remaining) { // one trip per distinct destination of the row
                const int leader = __ffsll((unsigned long long) remaining) - 1;
                const uint32_t b = (uint32_t) __shfl((int) d, leader);
                const bool inb = valid && d == b;
                const uint32_t v = inb ? len : 0u;
                const uint32_t incl = wave_incl_scan_u32(v, lane);
                if (inb) pos = run[b] + (incl - v);
                remaining &= ~__ballot(inb);
            }
            bool over = false;
            if (valid && (int32_t) d != p.self) {
                const unsigned long long start =
                    (cut == 0) ? (unsigned long long) p.ctrl->head.part_init[d]
                               : (p.status[(uint64_t)(cut - 1) * PC + d] & OS_VALUE_MASK);
                const unsigned long long cap = (unsigned long long) p.ctrl->head.part_cap[d];
                over = start + pos + len > cap;
                if (over && len > cap) raise_error(p.ctrl, MGS_ERR_RECORD_SIZE);
            }
            const uint64_t om = __ballot(over);
            uint64_t keep = __ballot(valid);
            if (om) {
                const int fl = __ffsll((unsigned long long) om) - 1;
                e = t0 + (uint64_t) row * 64 + fl;
                keep &= (1ull << fl) - 1ull;
                found = true;
            }
            // add the kept records to the running totals, one destination at a time
            uint64_t rem2 = keep;
            while (rem2) {
                const int leader = __ffsll((unsigned long long) rem2) - 1;
                const uint32_t b = (uint32_t) __shfl((int) d, leader);
                const bool inb = ((keep >> lane) & 1ull) && d == b;
                const uint32_t v = inb ? len : 0u;
                const uint32_t incl = wave_incl_scan_u32(v, lane);
                const uint32_t total = (uint32_t) __shfl((int) incl, 63);
                const uint64_t mb = __ballot(inb);
                if (lane == leader) {
                    run[b] += total;
                    if ((int32_t) b == p.self) lrun[0] += (unsigned long long) __popcll(mb);
                }
                rem2 &= ~mb;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    const unsigned long long mytot = before + run[lane];
    if (live) {
        if ((int32_t) lane == p.self) {
            res->send_offset[lane] = 0;
            res->local_bytes = (int64_t) mytot;
        }
        else {
            res->send_offset[lane] = (int64_t) mytot;
        }
    }
    if (p.self >= 0 && (uint32_t) lane == P) {
        const unsigned long long lrec = before + lrun[0];
        res->next = q.first + e;
        res->local_records = lrec;
        res->packed_records = e - lrec;
    }
    if (p.self < 0 && lane == 0) {
        res->next = q.first + e;
        res->local_records = 0;
        res->packed_records = e;
        res->local_bytes = 0;
    }
}

} // namespace mgs
#endif
