// k_onesweepv: single-pass stable hash-partition of byte-granular records:
// variable width (NUL-terminated keys and/or values, located by a uint32
// offset array) and fixed widths other than 16 bytes; nranks <= 64.
//
// Every byte of a record is read from HBM once and written once.  A tile is
// T consecutive records, i.e. one contiguous span of the input; positions are
// counted in bytes.
//
//   stage     the tile's span is loaded with aligned 16-byte loads (whole-wave
//             contiguous accesses) and parked in LDS (`sin`); the offsets of
//             the tile after next and the bytes of the next tile are in flight
//             while this one is processed;
//   hash      lane l owns K consecutive records: key length by the layout's
//             rule (src/serializer.h:418-421), lookup3 over the key bytes read
//             from LDS, dest = hash % nranks;
//   rank      emit order inside a tile is (wave, lane, k).  Per pair of
//             destinations one wave scan (two 16-bit fields) over the lanes'
//             byte totals gives each lane its start inside the wave's share;
//             inside a lane the K records are accumulated in order.  Wave
//             totals per destination go to LDS; the last worker wave to finish
//             publishes the tile's per-destination byte counts (16-bit fields)
//             for the chained scan, before the barrier;
//   reorder   records are copied LDS -> LDS into an image grouped by
//             destination (`sout`), byte-granular, stable: the first 16 bytes of
//             every record by aligned dword reads and a byte shift, written as
//             pieces of 8/4/2/1 bytes;
//   chain     the windowed chained scan of mgs_onesweep16w.h, run by a scan
//             wave, over byte counts; its first poll is in flight across the
//             workers' two staging barriers;
//   store     one trip later, the image is streamed out in 16-byte chunks
//             aligned to the *global* address of each partition run (aligned
//             LDS dword reads + byte shift, aligned global stores); a wave takes 64
//             consecutive chunks of one destination at a time; the partial
//             chunks at the two ends of a run go out as naturally aligned
//             pieces.  A run is clipped at the partition's capacity; a clipped
//             tile records its index (round cut), k_osv_finalize re-walks that
//             one tile.
//
// The kernel needs every tile's span to fit the LDS staging buffer.  It tests that tile
// by tile (sanitize): a tile that does not fit is treated as empty and Ctrl::head.fallback is
// raised; the host then redoes the call with the generic kernels (mgs_pack_result).
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
    // checked build of k_onesweepd (debugging aid): every store must lie inside one of two
    // address ranges (send images, local stream); a store that does not is counted, not made
    uint64_t dbg_lo[2], dbg_hi[2];
    uint32_t dbg_flags; // timing experiments (wrong output): 1 = no fast-path stores, 2 = only the first store of a record
    uint32_t in_bytes;  // LDS staging capacity of the kernel instantiation that serves the call (k_onesweepv)
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

// 12 bytes of LDS at any alignment.  An LDS access that is not aligned to its size
// runs at a fifth of the speed, so read the four aligned dwords around them and
// shift by bytes.  Up to base[(off & ~3) + 15] is read.
__device__ __forceinline__ void ov_lds_read12(const uint8_t *base, uint32_t off, uint32_t &w0, uint32_t &w1,
                                              uint32_t &w2)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base + (off & ~3u));
    const uint32_t y0 = w[0], y1 = w[1], y2 = w[2], y3 = w[3];
    const uint32_t r = off & 3u;
    w0 = __builtin_amdgcn_alignbyte(y1, y0, r);
    w1 = __builtin_amdgcn_alignbyte(y2, y1, r);
    w2 = __builtin_amdgcn_alignbyte(y3, y2, r);
}

// lookup3 hashlittle (src/hash.cpp:152-400, initval 0) over klen >= 1 bytes at base + off
// (LDS, any alignment; up to 14 bytes past the key may be read and are masked off).
__device__ __forceinline__ uint32_t ov_hash_lds(const uint8_t *base, uint32_t off, uint32_t klen)
{
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + klen;
    uint32_t rem = klen;
    uint32_t w0, w1, w2;
    while (rem > 12) {
        ov_lds_read12(base, off, w0, w1, w2);
        a += w0;
        b += w1;
        c += w2;
        MGS_MIX(a, b, c);
        off += 12;
        rem -= 12;
    }
    ov_lds_read12(base, off, w0, w1, w2);
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
    return gen_key_len(lay, r, len, klen);
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

// 16 bytes of LDS at any alignment.  An LDS access that is not aligned to its
// size runs at a fifth of the speed, so read the five aligned dwords around them
// (dword reads at dword addresses are aligned to their size whatever the offset)
// and shift by bytes.  Up to base[(off & ~3) + 19] is read.
__device__ __forceinline__ uint4 ov_lds_read16(const uint8_t *base, uint32_t off)
{
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base + (off & ~3u));
    const uint32_t y0 = w[0], y1 = w[1], y2 = w[2], y3 = w[3], y4 = w[4];
    const uint32_t r = off & 3u;
    return make_uint4(__builtin_amdgcn_alignbyte(y1, y0, r), __builtin_amdgcn_alignbyte(y2, y1, r),
                      __builtin_amdgcn_alignbyte(y3, y2, r), __builtin_amdgcn_alignbyte(y4, y3, r));
}

// The workgroup barrier without the wait for outstanding memory operations
// that the compiler puts in front of the builtin: the scan wave crosses the
// workers' staging barriers with its poll still in flight.
__device__ __forceinline__ void ov_raw_barrier()
{
    asm volatile("s_barrier" ::: "memory");
}

// The round finder's view of a tile of byte-granular records (mgs_round_finder.h): the
// tile's span is parked in the staging buffer, key lengths and hashes are read from there.
struct FinderSrcV {
    const OvParams &q;
    uint8_t *sin;
    uint64_t s = 0;     // first byte of the staged span in blob
    uint32_t shift = 0; // the span starts `shift` bytes into the staging buffer
    uint64_t span = 0;
    __device__ void stage(uint32_t tile, int tid, int nthreads)
    {
        const uint64_t T = q.os.tile_records;
        const uint64_t i0 = (uint64_t) tile * T, i1 = min(q.n, i0 + T);
        s = ov_offset(q, i0);
        const uint64_t e = ov_offset(q, i1);
        const uintptr_t a0 = ((uintptr_t) q.blob + s) & ~(uintptr_t) 15;
        shift = (uint32_t) (((uintptr_t) q.blob + s) & 15u);
        span = (e >= s ? e - s : 0) + shift;
        if (span > (uint64_t) q.in_bytes) span = 0; // (the sweep sends such a call to the generic path)
        const uint32_t nch = (uint32_t) ((span + 15) >> 4);
        __syncthreads(); // the previous tile's bytes are no longer read
        for (uint32_t ch = (uint32_t) tid; ch < nch; ch += (uint32_t) nthreads)
            reinterpret_cast<uint4 *>(sin)[ch] = reinterpret_cast<const uint4 *>(a0)[ch];
        __syncthreads();
    }
    struct Pre {
        uint64_t o0, o1;
    };
    __device__ Pre fetch(uint32_t tile, uint32_t i, bool &valid) const
    {
        const uint64_t T = q.os.tile_records;
        const uint64_t idx = (uint64_t) tile * T + i;
        valid = i < T && idx < q.n;
        Pre f = {0, 0};
        if (valid) {
            f.o0 = ov_offset(q, idx);
            f.o1 = ov_offset(q, idx + 1);
        }
        return f;
    }
    __device__ void info(const Pre &f, uint32_t &d, uint32_t &u, bool &valid) const
    {
        d = 0;
        u = 0;
        const uint64_t o0 = f.o0, o1 = f.o1;
        const uint32_t len = (uint32_t) (o1 - o0);
        uint32_t klen = 0;
        const uint64_t r = o0 - s + shift;
        if (o1 <= o0 || o0 < s || o1 - o0 > GEN_MAX_RECORD || r + len > span || !gen_key_len(q.lay, sin + r, len, klen)) {
            valid = false; // reported by the sweep
            return;
        }
        d = fastmod(hashlittle_bytes(sin + r, klen), q.os.fm);
        u = len;
    }
};

// LM (layout mode) spares the per-record layout decisions: 1 = string key(s) and a
// fixed-width or absent value (key = the record minus the value, ends with a NUL);
// 2 = fixed-width key and a fixed-width or absent value; 0 = any layout.
// INB: bytes of the LDS staging buffer (and of the image).  The default holds a tile of 448 * K records
// of up to 80 / K bytes at two workgroups per compute unit; half of it, with K = 2, fits three.
template <int THREADS, int K, int VAR, int MINW, int LOOK, int LM, int INB = OV_IN_BYTES>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweepv(OvParams q)
{
    const OsParams &p = q.os;
    constexpr int WORKERS = THREADS / 64 - 1, WT = WORKERS * 64, T = WT * K;
    constexpr int CPT = (INB / 16 + WT - 1) / WT; // 16-byte chunks per worker thread
    __shared__ __attribute__((aligned(16))) uint8_t sin[INB + 32]; // reads run up to 19 bytes past a record
    __shared__ __attribute__((aligned(16))) uint8_t sout[INB + 16];
    __shared__ uint32_t wcnt[2 * WORKERS * OS_MAXP]; // bytes per (wave, dest), two tiles
    __shared__ uint32_t wslot[WORKERS * OS_MAXP];    // first image byte of (wave, dest)
    __shared__ uint32_t dstart2[2 * OS_MAXP];        // first image byte of dest, two tiles
    __shared__ unsigned long long gaddr2[2 * OS_MAXP]; // global address of the run's first byte
    __shared__ uint32_t nbytes2[2 * OS_MAXP];          // bytes of the run to store (clipped)
    __shared__ uint32_t cstart2[2 * (OS_MAXP + 1)];    // first store unit of dest's run (+ total)
    __shared__ volatile uint32_t misc[4];              // [0] first tile, [1] second tile / tile after next
    __shared__ uint32_t arrive[2];

    if (p.ctrl->head.fallback != 0) return; // the generic path handles this call

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t P = q.P;   // destinations
    const uint32_t PC = p.P;  // chain columns

    if (p.finder && blockIdx.x == 0) {
        // streams layout: this workgroup finds the rounds instead of sweeping
        typedef FinderLdsT<T / 64> FL;
        static_assert(sizeof(FL) <= sizeof(sout), "the finder's tables live in the image buffer");
        FinderSrcV src{q, sin};
        os_round_finder<FinderSrcV, THREADS / 64, T / 64>(p, P, src, reinterpret_cast<FL *>(sout), tid);
        return;
    }

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

    // The partial chunks at the two ends of the runs of a tile's image (at most two per
    // destination): one wave, lane = (destination, end), so that the divergent byte work runs
    // once per tile.  In the loop the scan wave does it, in the time it would otherwise spend
    // waiting at barrier B; after the loop, the last worker wave.
    auto store_partials = [&](int pr) {
        for (uint32_t it = (uint32_t) lane; it < 2u * P; it += 64u) {
            const uint32_t b = it >> 1;
            const unsigned long long G = gaddr2[pr * OS_MAXP + b];
            const uint32_t nb = nbytes2[pr * OS_MAXP + b];
            if (nb == 0) continue;
            const uint32_t nch = (uint32_t) (((G & 15ull) + nb + 15ull) >> 4);
            if ((it & 1u) && nch == 1) continue; // the only chunk is taken as the first
            const unsigned long long g0 = (G & ~15ull) + ((it & 1u) ? (unsigned long long) (nch - 1u) << 4 : 0ull);
            if (g0 >= G && g0 + 16 <= G + nb) continue; // a whole chunk: stored above
            // naturally aligned pieces, growing from the first byte, then shrinking towards the last
            const uint8_t *src0 = sout + dstart2[pr * OS_MAXP + b];
            unsigned long long ga = g0 > G ? g0 : G;
            const unsigned long long gb = g0 + 16 < G + nb ? g0 + 16 : G + nb;
            if ((ga & 1) && ga + 1 <= gb) {
                *reinterpret_cast<uint8_t *>(ga) = src0[ga - G];
                ga += 1;
            }
            if ((ga & 2) && ga + 2 <= gb) {
                *reinterpret_cast<uint16_t *>(ga) = reinterpret_cast<const OvU16 *>(src0 + (ga - G))->v;
                ga += 2;
            }
            if ((ga & 4) && ga + 4 <= gb) {
                *reinterpret_cast<uint32_t *>(ga) = reinterpret_cast<const OvU32 *>(src0 + (ga - G))->v;
                ga += 4;
            }
            if (ga + 8 <= gb) {
                *reinterpret_cast<unsigned long long *>(ga) = reinterpret_cast<const OvU64 *>(src0 + (ga - G))->v;
                ga += 8;
            }
            if (ga + 4 <= gb) {
                *reinterpret_cast<uint32_t *>(ga) = reinterpret_cast<const OvU32 *>(src0 + (ga - G))->v;
                ga += 4;
            }
            if (ga + 2 <= gb) {
                *reinterpret_cast<uint16_t *>(ga) = reinterpret_cast<const OvU16 *>(src0 + (ga - G))->v;
                ga += 2;
            }
            if (ga + 1 <= gb) *reinterpret_cast<uint8_t *>(ga) = src0[ga - G];
        }
        };

    // ------------------------------------------------------------------ scan wave
    if (w == WORKERS) {
        // every later tile waits for this wave's prefix: it goes first at the issue arbiter
        __builtin_amdgcn_s_setprio(3);
        OsWalkStats ws;
        const bool live = (uint32_t) lane < PC, is_dest = (uint32_t) lane < P;
        const unsigned long long my_base = is_dest ? p.ctrl->head.part_base[lane] : 0ull;
        const unsigned long long my_cap = is_dest ? (unsigned long long) p.ctrl->head.part_cap[lane] : 0ull;
        __syncthreads(); // the workers' staging barrier of the first tile
        bool scan_have_prev = false;
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
            if (scan_have_prev) store_partials(par ^ 1); // of the image the workers are storing now
            scan_have_prev = true;
            __syncthreads(); // B
            const uint32_t next2 = misc[1];
            if ((VAR & OSV_FAKE_CHAIN) && tile != 0) { // timing only: made-up positions, no scan
                ov_raw_barrier(); // C
                ov_raw_barrier(); // D
                acc = (unsigned long long) tile * 3600ull;
            }
            else if (tile != 0) {
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
            // work units of the store: (destination, block of 64 chunks of its run)
            const uint32_t nch = nb ? (uint32_t) (((G & 15ull) + nb + 15ull) >> 4) : 0u;
            const uint32_t units = (nch + 63u) >> 6;
            const uint32_t uincl = wave_incl_scan_u32(units, lane);
            gaddr2[par * OS_MAXP + lane] = G;
            nbytes2[par * OS_MAXP + lane] = nb;
            cstart2[par * (OS_MAXP + 1) + lane] = uincl - units;
            if (lane == 63) cstart2[par * (OS_MAXP + 1) + OS_MAXP] = uincl;
            tile = next;
            next = next2;
            par ^= 1;
        }
        __syncthreads(); // matches the workers' last barrier
        return;
    }

    // ------------------------------------------------------------------- workers
    const int wt = w * 64 + lane; // worker thread index: owns records wt*K .. wt*K+K-1 of a tile
    uint32_t lo[K], len[K];       // record k: first byte in sin, length (0 = no record)

    // Fetch the offsets of a tile's records: absolute byte offsets of records
    // wt*K .. wt*K+K (K+1 values), and the tile's span.
    // A zero the compiler cannot see through: an index that is the same in every lane
    // would be fetched with a scalar load, and scalar loads share their completion
    // counter with LDS, out of order -- every later wait for an LDS access would then
    // also wait out the round trip to memory.
    uint32_t vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    // Offsets of a tile's records as fetched: K+1 record boundaries of this lane plus the
    // span's two ends, as 32-bit values relative to `base` (the offset array's own values
    // with base 0, or multiples of the fixed width with base = the tile's first byte).
    // Nothing is computed from the loaded values here, so the loads stay in flight.
    struct Offsets {
        uint64_t base;
        uint32_t rec[K + 1], s, e;
        uint32_t nrec; // records of the tile
    };
    auto fetch_offsets = [&](uint32_t t, Offsets &f) {
        const uint64_t i0 = (uint64_t) t * T, i1 = min(q.n, i0 + T);
        f.nrec = (uint32_t) (i1 - i0);
        if (q.rec_off) {
            const uint32_t *ro = q.rec_off + q.first;
            f.base = 0;
            f.s = ro[i0 + vzero];
            f.e = ro[i1 + vzero];
#pragma unroll
            for (int k = 0; k <= K; k++) f.rec[k] = ro[min(i1, i0 + (uint64_t) wt * K + k)];
        }
        else {
            f.base = (q.first + i0) * (uint64_t) q.fixed_len;
            f.s = 0;
            f.e = (uint32_t) (i1 - i0) * q.fixed_len;
#pragma unroll
            for (int k = 0; k <= K; k++)
                f.rec[k] = (uint32_t) (min(i1, i0 + (uint64_t) wt * K + k) - i0) * q.fixed_len;
        }
    };
    // The tile's span must fit the staging buffer and lie inside the call's bytes.  One that
    // does not -- records too long for this kernel, or offsets that are not monotonic -- is
    // treated as empty (nothing of it is loaded) and Ctrl::head.fallback is raised: the host
    // then redoes the call with the kernels that take any record.
    const uint64_t blob_end = ov_offset(q, q.n);
    auto sanitize = [&](Offsets &f) {
        const uint32_t shift = (uint32_t) (((uintptr_t) q.blob + f.base + f.s) & 15u);
        const bool fits = f.e >= f.s && (uint64_t) (f.e - f.s) + shift <= (uint64_t) INB && f.base + f.e <= blob_end;
        if (!fits) {
            __hip_atomic_store(&p.ctrl->head.fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            f.s = f.e = 0;
            f.nrec = 0;
#pragma unroll
            for (int k = 0; k <= K; k++) f.rec[k] = 0;
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
    auto set_records = [&](const Offsets &f) {
        const uint32_t shift = (uint32_t) (((uintptr_t) q.blob + f.base + f.s) & 15u);
        const uint64_t span = (uint64_t) (f.e - f.s) + shift;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t l = f.rec[k + 1] - f.rec[k]; // wraps to a huge value when not monotonic
            const uint32_t r = f.rec[k] - f.s;
            lo[k] = r + shift;
            len[k] = l;
            if (l == 0) { // past the end of the call -- or an empty record, which no layout produces
                if ((uint32_t) (wt * K + k) < f.nrec) raise_error(p.ctrl, MGS_ERR_MALFORMED);
                continue;
            }
            if (l > GEN_MAX_RECORD || f.rec[k] < f.s || (uint64_t) r + shift + l > span) {
                raise_error(p.ctrl, MGS_ERR_MALFORMED);
                len[k] = 0;
            }
        }
    };
    // Stream out the image of a tile (parity pr).  A wave takes one unit at a time: 64
    // consecutive 16-byte chunks of one destination's run, so everything but the chunk
    // index is the same in all its lanes.
    auto store_image = [&](int pr, bool with_partials) {
        const uint32_t *ustart = cstart2 + pr * (OS_MAXP + 1);
        const uint32_t utotal = ustart[OS_MAXP];
        // lane = destination: the tables of its run, read once; a unit's entries then come out of
        // these registers by lane number, with no trip to LDS in the loop
        const uint32_t my_ustart = ustart[lane]; // non-decreasing, total from P on
        const unsigned long long my_G = gaddr2[pr * OS_MAXP + lane];
        const uint32_t my_nb = nbytes2[pr * OS_MAXP + lane], my_ds = dstart2[pr * OS_MAXP + lane];
        // one unit: 64 consecutive 16-byte chunks of one destination's run, a chunk per lane
        struct Unit {
            unsigned long long gbase; // global address of the unit's first chunk (16-byte aligned)
            uint32_t src;             // image byte that goes to gbase
            bool on;                  // this lane's chunk is a whole chunk of the run
        };
        auto unit = [&](uint32_t u) -> Unit {
            const uint32_t b = (uint32_t) __popcll(__ballot(my_ustart <= u)) - 1u; // last dest starting at or before u
            const uint32_t u0 = (uint32_t) __builtin_amdgcn_readlane((int) my_ustart, (int) b);
            const unsigned long long G =
                ((unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (my_G >> 32), (int) b) << 32)
                | (unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) my_G, (int) b);
            const uint32_t nb = (uint32_t) __builtin_amdgcn_readlane((int) my_nb, (int) b);
            const uint32_t ds = (uint32_t) __builtin_amdgcn_readlane((int) my_ds, (int) b); // image byte of global byte G
            const uint32_t a = (uint32_t) G & 15u; // the run starts `a` bytes into its first chunk
            const uint32_t off0 = (u - u0) << 10;  // the unit's first chunk, from the run's aligned start
            const uint32_t off = off0 + ((uint32_t) lane << 4);
            Unit r;
            r.gbase = (G - a) + off0;
            r.src = ds + off0 - a; // (wraps below zero only where `on` is false)
            r.on = off >= a && off + 16u <= a + nb; // whole chunks only; the partial ones follow below
            return r;
        };
        auto read_chunk = [&](const Unit &t) -> uint4 {
            const uint32_t at = t.on ? t.src + ((uint32_t) lane << 4) : 0u;
            return ov_lds_read16(sout, at);
        };
        auto write_chunk = [&](const Unit &t, const uint4 &val) {
            if (t.on) *reinterpret_cast<uint4 *>(t.gbase + ((unsigned long long) lane << 4)) = val;
        };
        // two units per trip: both chunks are read before either is stored
        uint32_t u = (uint32_t) w;
        for (; u + WORKERS < utotal; u += 2 * WORKERS) {
            const Unit t0 = unit(u), t1 = unit(u + WORKERS);
            const uint4 v0 = read_chunk(t0), v1 = read_chunk(t1);
            write_chunk(t0, v0);
            write_chunk(t1, v1);
        }
        if (u < utotal) {
            const Unit t0 = unit(u);
            write_chunk(t0, read_chunk(t0));
        }
        if (with_partials && w == WORKERS - 1) store_partials(pr);
    };

    // offsets of the tile after the current one: fetched one whole trip before its bytes,
    // so that neither of the two dependent round trips is waited for
    Offsets fn;
    fn.base = 0;
    fn.s = fn.e = 0;
    fn.nrec = 0;
    {
        // first tile: offsets, then bytes, then park them
        Offsets f0;
        uint4 ch[CPT];
        if (tile < p.ntiles) {
            fetch_offsets(tile, f0);
            if (next < p.ntiles) fetch_offsets(next, fn);
            sanitize(f0);
            fetch_chunks(f0.base + f0.s, f0.base + f0.e, ch);
            park_chunks(f0.base + f0.s, f0.base + f0.e, ch);
            set_records(f0);
        }
        __syncthreads();
    }
    bool have_prev = false;
    // diagnostic build: cycles of worker wave 0 per phase of the trip
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (VAR & OSV_STATS) tprev = __builtin_readcyclecounter();
#define OV_PHASE(i)                                                      \
    if (VAR & OSV_STATS) {                                               \
        const unsigned long long tnow = __builtin_readcyclecounter();    \
        ph[i] += tnow - tprev;                                           \
        tprev = tnow;                                                    \
    }

    while (tile < p.ntiles) {
        uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
        // ---- hash: destination of each of this lane's records
        uint32_t d[K], h[K];
        const uint32_t vlen = (q.lay.val_kind == MGS_VAL_FIXED) ? (uint32_t) q.lay.val_bytes : 0u;
#pragma unroll
        for (int k = 0; k < K; k++) {
            h[k] = 0;
            if (len[k] == 0) continue;
            const uint8_t *r = sin + lo[k];
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
                uint32_t w0, w1, w2;
                ov_lds_read12(sin, lo[k], w0, w1, w2);
                const uint32_t m0 = klen >= 4 ? 0xFFFFFFFFu : ((1u << (8 * klen)) - 1u);
                const uint32_t m1 = klen >= 8 ? 0xFFFFFFFFu : (klen > 4 ? ((1u << (8 * (klen - 4))) - 1u) : 0u);
                const uint32_t m2 = klen >= 12 ? 0xFFFFFFFFu : (klen > 8 ? ((1u << (8 * (klen - 8))) - 1u) : 0u);
                if (LM == 1) {
                    // the key's first NUL must be its last byte (strlen + 1, src/serializer.h:131-136):
                    // zero-byte flags of the three words with the bytes beyond the key set to 0xFF
                    const uint32_t t0 = w0 | ~m0, t1 = w1 | ~m1, t2 = w2 | ~m2;
                    const uint32_t z0 = (t0 - 0x01010101u) & ~t0 & 0x80808080u, z1 = (t1 - 0x01010101u) & ~t1 & 0x80808080u,
                                   z2 = (t2 - 0x01010101u) & ~t2 & 0x80808080u;
                    const uint32_t fz = z0 ? ((uint32_t) __builtin_ctz(z0) >> 3)
                                           : (z1 ? 4u + ((uint32_t) __builtin_ctz(z1) >> 3)
                                                 : (z2 ? 8u + ((uint32_t) __builtin_ctz(z2) >> 3) : 99u));
                    ok = fz == klen - 1;
                }
                uint32_t a, b, c;
                a = b = c = 0xdeadbeefu + klen;
                a += w0 & m0;
                b += w1 & m1;
                c += w2 & m2;
                MGS_FINAL(a, b, c);
                h[k] = c;
            }
            else if (ok) {
                if (LM == 1) {
                    ok = r[klen - 1] == 0;
                    for (uint32_t j = 0; j + 1 < klen; j++) ok = ok && r[j] != 0;
                }
                h[k] = ov_hash_lds(sin, lo[k], klen);
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
        OV_PHASE(0)
        // ---- rank: byte position of each record inside the wave's share of its destination.
        // Two destinations share one scan: 16-bit fields (a wave's bytes fit: they are
        // part of one staged span).
        uint32_t pw[K], field[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            pw[k] = 0;
            field[k] = len[k] << (16 * (d[k] & 1u)); // this record's bytes in its destination's field
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
            if (arrived == (uint32_t) WORKERS - 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                uint32_t tot_all = 0;
#pragma unroll
                for (int ww = 0; ww < WORKERS; ww++)
                    tot_all += ((volatile uint32_t *) wc)[ww * OS_MAXP + lane];
                os_publish_agg16(p, tile, lane, tot_all);
            }
        }
        OV_PHASE(1)
        __syncthreads(); // A: counts(tile) complete; tables of the previous tile are ready
        OV_PHASE(2)
        // the phases that issue memory operations go before the other waves' hashing
        __builtin_amdgcn_s_setprio(1);

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
        // bytes of the next tile (its offsets arrived during the last trip): in flight
        // during the store and the reorder
        uint4 ch[CPT];
        if (next < p.ntiles) {
            sanitize(fn);
            fetch_chunks(fn.base + fn.s, fn.base + fn.e, ch);
        }
        if (have_prev) store_image(par ^ 1, false);
        OV_PHASE(3)
        __syncthreads(); // B: the image is free
        OV_PHASE(4)
        const uint32_t next2 = misc[1];

        // offsets of the tile after the next: in flight during the reorder
        Offsets fn2;
        fn2.base = 0;
        fn2.s = fn2.e = 0;
        fn2.nrec = 0;
        if (next2 < p.ntiles) fetch_offsets(next2, fn2);
        {
            // reorder: all reads first (the first 16 bytes of every record, and where it
            // goes), then the writes -- pieces of 8/4/2/1 bytes out of registers
            uint4 head[K];
            uint32_t dst[K];
#pragma unroll
            for (int k = 0; k < K; k++) {
                head[k] = ov_lds_read16(sin, len[k] ? lo[k] : 0u);
                dst[k] = wslot[w * OS_MAXP + (len[k] ? d[k] : 0u)] + pw[k];
            }
            // n <= 16 bytes out of four registers to any address of the image
            auto put = [&](uint8_t *o, const uint4 &hd, uint32_t n) {
                if (n >= 16) {
                    *reinterpret_cast<OvU128 *>(o) = OvU128{hd.x, hd.y, hd.z, hd.w};
                    return;
                }
                uint32_t a0 = hd.x, a1 = hd.y, a2 = hd.z, a3 = hd.w;
                if (n & 8u) {
                    reinterpret_cast<OvU64 *>(o)->v = (unsigned long long) a0 | ((unsigned long long) a1 << 32);
                    a0 = a2;
                    a1 = a3;
                    o += 8;
                }
                if (n & 4u) {
                    reinterpret_cast<OvU32 *>(o)->v = a0;
                    a0 = a1;
                    o += 4;
                }
                if (n & 2u) {
                    reinterpret_cast<OvU16 *>(o)->v = (uint16_t) a0;
                    a0 >>= 16;
                    o += 2;
                }
                if (n & 1u) *o = (uint8_t) a0;
            };
#pragma unroll
            for (int k = 0; k < K; k++)
                if (len[k]) put(sout + dst[k], head[k], len[k]);
            // the rest of longer records: tails of a few bytes as unaligned pieces LDS -> LDS;
            // when a lane has more, bytes 16..31 through registers as above, all reads first
            bool more = false, much = false;
#pragma unroll
            for (int k = 0; k < K; k++) {
                more |= len[k] > 16;
                much |= len[k] > 22;
            }
            if (__any(much)) {
#pragma unroll
                for (int k = 0; k < K; k++) head[k] = ov_lds_read16(sin, len[k] > 16 ? lo[k] + 16 : 0u);
#pragma unroll
                for (int k = 0; k < K; k++)
                    if (len[k] > 16) put(sout + dst[k] + 16, head[k], len[k] - 16);
                for (int k = 0; k < K; k++) // not unrolled: rare
                    if (len[k] > 32) ov_copy_lds(sout + dst[k] + 32, sin + lo[k] + 32, len[k] - 32);
            }
            else if (__any(more)) {
                for (int k = 0; k < K; k++) // not unrolled
                    if (len[k] > 16) ov_copy_lds(sout + dst[k] + 16, sin + lo[k] + 16, len[k] - 16);
            }
        }
        OV_PHASE(5)
        __syncthreads(); // C: the staging buffer is free
        if (next < p.ntiles) {
            park_chunks(fn.base + fn.s, fn.base + fn.e, ch);
            set_records(fn);
        }
        fn = fn2;
        OV_PHASE(6)
        __builtin_amdgcn_s_setprio(0);
        __syncthreads(); // D: the next tile is staged
        OV_PHASE(7)
        have_prev = true;
        tile = next;
        next = next2;
        par ^= 1;
    }
    __syncthreads(); // tables of the last tile are ready
    if (have_prev) store_image(par ^ 1, true);
    if ((VAR & OSV_STATS) && wt == 0) {
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(&p.ctrl->head.stats[i], ph[i]);
    }
#undef OV_PHASE
}

// One wave.  send_offset[] / counts at the point where the round ended; when a
// tile was clipped, re-walk that tile record by record to find the first record
// that does not fit (src/collectiveshuffler.h:148-160).
#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
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
    if (p.clip_out) {
        if (lane == 0) p.clip_out[0] = e;
        if ((uint32_t) lane < P) p.clip_out[1 + lane] = mytot;
        else if ((uint32_t) lane < PC) p.clip_out[1 + lane] = before + lrun[0];
    }
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
#endif

} // namespace mgs
#endif
