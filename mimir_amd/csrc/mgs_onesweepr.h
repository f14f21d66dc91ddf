// k_onesweepr: single-pass stable hash-partition of byte-granular records whose
// bytes go from global memory straight into registers.
//
// Serves the layouts of k_onesweepv's modes 1 and 2 -- ONE string key with a
// fixed-width or absent value (wordcount, join), and fixed-width keys and values
// of any width up to 32 bytes (octree level passes) -- with fewer instructions per
// record: k_onesweepv parks a tile's span in LDS, reads every key from there to
// hash it, reads every record again to move it into the image and crosses four
// barriers per tile.  Here
//
//   load      lane l owns K consecutive records; each record's first 16 * NQ bytes
//             are fetched with NQ unaligned 16-byte loads from the record's own
//             address (neighbouring lanes read neighbouring bytes, so every line is
//             fetched from HBM once and the rest are hits in the vector cache); the
//             loads of tile i + 1 and the offsets of tile i + 2 are in flight while
//             tile i is processed.  No staging buffer, no staging barriers;
//   hash      lookup3 (src/hash.cpp:152-400) on the key words in registers, masked
//             to the key's length; a string key's NUL rule (strlen + 1,
//             src/serializer.h:131-136) from zero-byte flags of the same words;
//   rank      as in k_onesweepv: emit order in a tile is (wave, lane, k), one wave
//             scan per pair of destinations over the lanes' byte totals;
//   reorder   registers -> LDS image grouped by destination (pieces of 16/8/4/2/1
//             bytes; dwords when every record is a multiple of four bytes long);
//   chain     the windowed chained scan of mgs_onesweep16w.h over byte counts, run
//             by a scan wave;
//   store     one trip later, as k_onesweepv: 16-byte chunks aligned to the global
//             address of each partition run.
//
// Two barriers per tile.  A record longer than its registers (rare by the host's
// choice of NQ) gets its key hashed and its remaining bytes copied from global
// memory by its lane, and is counted in Ctrl::head.long_records so that the host can
// pick wider registers for the next call.  A tile whose bytes do not fit the image
// raises Ctrl::head.fallback; the host then redoes the call with other kernels.
#ifndef MGS_ONESWEEPR_H
#define MGS_ONESWEEPR_H

#include "mgs_onesweepv.h"

namespace mgs {

constexpr int OSR_THREADS = 512;
constexpr int OSR_LANES = (OSR_THREADS / 64 - 1) * 64; // worker lanes
constexpr int OSR_IMG1 = 29696;                        // image bytes, NQ = 1: 1792 records of 16 bytes + slack
constexpr int OSR_IMG2 = 58368;                        // image bytes, NQ = 2: 1792 records of 32 bytes + slack

// The round finder's view of a tile: records read where they lie in global memory.
struct FinderSrcG {
    const OvParams &q;
    __device__ void stage(uint32_t, int, int) {}
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
        const uint32_t len = (uint32_t) (f.o1 - f.o0);
        uint32_t klen = 0;
        if (f.o1 <= f.o0 || f.o1 - f.o0 > GEN_MAX_RECORD || !gen_key_len(q.lay, q.blob + f.o0, len, klen)) {
            valid = false; // reported by the sweep
            return;
        }
        d = fastmod(hashlittle_bytes(q.blob + f.o0, klen), q.os.fm);
        u = len;
    }
};

// mask of the first clamp(nb, 0, 4) bytes of a word
__device__ __forceinline__ uint32_t osr_byte_mask(int32_t nb)
{
    return nb >= 4 ? 0xFFFFFFFFu : (nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u));
}

// flags (bit 7 of a byte) of the zero bytes of t: exact for the lowest zero byte, and no
// flag at all when there is none
__device__ __forceinline__ uint32_t osr_zero_flags(uint32_t t)
{
    return (t - 0x01010101u) & ~t & 0x80808080u;
}

// lookup3 hashlittle (initval 0) of the first klen bytes (1 <= klen <= 16) of a record whose
// first 16 bytes are held as four little-endian words.  ktab: see the kernel.  CSTR: the key
// is a C string with its NUL -- `ok` is cleared unless the first zero byte of the key is its
// last byte.
template <bool CSTR>
__device__ __forceinline__ uint32_t osr_hash_key16(const uint32_t (&w)[4], uint32_t klen, const uint4 *ktab, bool &ok)
{
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + klen;
    if (klen <= 12) {
        const uint4 m = ktab[klen];
        const uint32_t x0 = w[0] & m.x, x1 = w[1] & m.y, x2 = w[2] & m.z;
        a += x0;
        b += x1;
        c += x2;
        MGS_FINAL(a, b, c);
        if (CSTR) {
            // bytes beyond the key count as non-zero; byte i of word j is bit 8 * i + j of Z
            const uint32_t Z = (osr_zero_flags(x0 | ~m.x) >> 7) | (osr_zero_flags(x1 | ~m.y) >> 6)
                               | (osr_zero_flags(x2 | ~m.z) >> 5);
            ok = ok && Z == m.w;
        }
    }
    else {
        a += w[0];
        b += w[1];
        c += w[2];
        MGS_MIX(a, b, c);
        const uint4 m = ktab[klen - 12u];
        const uint32_t x0 = w[3] & m.x;
        a += x0;
        MGS_FINAL(a, b, c);
        if (CSTR) {
            const uint32_t zfull = osr_zero_flags(w[0]) | osr_zero_flags(w[1]) | osr_zero_flags(w[2]);
            ok = ok && zfull == 0 && (osr_zero_flags(x0 | ~m.x) >> 7) == m.w;
        }
    }
    return c;
}

// THREADS = 64 * (workers + 1 scan wave); K records per worker lane; NQ 16-byte register
// quads per record; LM: 1 = one string key + fixed-width or absent value, 2 = fixed-width
// key + fixed-width or absent value; A4: every record length, partition base and initial
// fill is a multiple of four (fixed widths only), so image and partitions are moved as
// aligned dwords; IMG: bytes of the LDS image; PRE: the first quads of the next tile are
// requested before the store of the previous tile (16 more registers per 4 records) instead
// of at the end of the trip.
// Q1W: dwords of a record's bytes from 16 on that are held (NQ = 2): 4, or 1 for records of at
// most 20 bytes (wordcount: a word, its NUL and eight value bytes) -- fewer registers, which
// pays for requesting the next tile's first quads a phase early (PRE).
template <int THREADS, int K, int NQ, int VAR, int MINW, int LOOK, int LM, bool A4, int IMG, bool PRE, int Q1W = 4>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweepr(OvParams q)
{
    static_assert(LM == 1 || LM == 2, "layout modes 1 and 2");
    static_assert(!A4 || LM == 2, "dword moves need fixed widths");
    const OsParams &p = q.os;
    constexpr int WORKERS = THREADS / 64 - 1, WT = WORKERS * 64, T = WT * K;
    static_assert(Q1W == 4 || Q1W == 1, "the second part of a record is a quad or a dword");
    constexpr uint32_t RB = NQ > 1 ? 16u + 4u * Q1W : 16u; // bytes of a record held in registers
    __shared__ __attribute__((aligned(16))) uint8_t sout[IMG + 32]; // reads run up to 19 bytes past a run
    __shared__ uint32_t wcnt[2 * WORKERS * OS_MAXP];                // bytes per (wave, dest), two tiles
    __shared__ uint32_t wslot[WORKERS * OS_MAXP];                   // first image byte of (wave, dest)
    __shared__ uint32_t dstart2[2 * OS_MAXP];                       // first image byte of dest, two tiles
    __shared__ unsigned long long gaddr2[2 * OS_MAXP];              // global address of the run's first byte
    __shared__ uint32_t nbytes2[2 * OS_MAXP];                       // bytes of the run to store (clipped)
    __shared__ uint32_t cstart2[2 * (OS_MAXP + 1)];                 // first store unit of dest's run (+ total)
    __shared__ volatile uint32_t misc[4]; // [0] first tile, [1] second tile / tile after next
    __shared__ uint32_t arrive[2];
    __shared__ __attribute__((aligned(16))) uint4 ktab[16]; // key masks by length (see below)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t P = q.P;  // destinations
    const uint32_t PC = p.P; // chain columns

    if (p.finder && blockIdx.x == 0) {
        // streams layout: this workgroup finds the rounds instead of sweeping
        typedef FinderLdsT<T / 64> FL;
        static_assert(sizeof(FL) <= sizeof(sout), "the finder's tables live in the image buffer");
        FinderSrcG src{q};
        os_round_finder<FinderSrcG, THREADS / 64, T / 64>(p, P, src, reinterpret_cast<FL *>(sout), tid);
        return;
    }

    for (int j = tid; j < 2 * WORKERS * OS_MAXP; j += THREADS) wcnt[j] = 0;
    // key masks: ktab[r] = masks of the first r bytes of three words and the bit that the
    // NUL of an r-byte string sets in osr_hash_key16's zero-byte word
    if (tid < 13) {
        const int r = tid;
        ktab[r] = make_uint4(osr_byte_mask(r), osr_byte_mask(r - 4), osr_byte_mask(r - 8),
                             r ? 1u << (8u * ((uint32_t) (r - 1) & 3u) + ((uint32_t) (r - 1) >> 2)) : 0u);
    }
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
                acc = (unsigned long long) tile * 3600ull;
            }
            else if (tile != 0) {
                acc = os_window_walk<VAR, LOOK>(p, tile, lane, &ws);
                if (live)
                    os_store_status(p.status + (uint64_t) tile * PC + lane,
                                    OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
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
    // (tile numbers come out of LDS: say that they are the same in every lane, so that
    // everything derived from them is scalar work)
    tile = (uint32_t) __builtin_amdgcn_readfirstlane((int) tile);
    next = (uint32_t) __builtin_amdgcn_readfirstlane((int) next);
    const int wt = w * 64 + lane; // worker thread index: owns records wt*K .. wt*K+K-1 of a tile
    uint32_t len[K];              // record k: length (0 = no record)
    uint32_t roff[K];             // its first byte, relative to cur_base
    uint32_t q0[K][4];            // its bytes 0..15
    uint32_t q1[K][Q1W];          // its bytes from 16 on (NQ = 2), requested at the end of the trip before its own
    uint64_t cur_base = 0;        // byte of blob the current tile's offsets are relative to
    uint32_t cur_lim = 0;         // bytes from cur_base to the end of the call's records (clamped)
    uint32_t cur_end = 0;         // the current tile's last byte + 1, from cur_base

    // A zero the compiler cannot see through: an index that is the same in every lane
    // would be fetched with a scalar load, and scalar loads share their completion
    // counter with LDS, out of order -- every later wait for an LDS access would then
    // also wait out the round trip to memory.
    uint32_t vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    // first byte behind the call's records: no load may reach beyond it
    const uint64_t end_abs = q.rec_off ? (uint64_t) q.rec_off[q.first + q.n + vzero]
                                       : (q.first + q.n) * (uint64_t) q.fixed_len;

    // element k (not known at compile time) of a small register array: rare paths only
    // (each element passes through an empty asm: a plain chain of selects is turned back into
    // an indexed read of the array, which then lives in scratch memory)
    auto pick = [](const auto &a, int k) -> uint32_t {
        uint32_t r = 0;
#pragma unroll
        for (int j = 0; j < K; j++) {
            uint32_t v = a[j];
            asm volatile("" : "+v"(v));
            r = (j == k) ? v : r;
        }
        return r;
    };
    // up to 16 bytes from global memory, one at a time, as four little-endian words (rare paths)
    auto bytewise16 = [](const uint8_t *src, uint32_t nbytes, uint32_t (&t)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) t[j] = 0;
#pragma unroll 1
        for (uint32_t i = 0; i < nbytes; i++) {
            const uint32_t v = (uint32_t) src[i] << (8u * (i & 3u));
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((i >> 2) == (uint32_t) j) t[j] |= v;
        }
    };
    // Offsets of a tile's records as fetched: K+1 record boundaries of this lane plus the
    // tile's two ends, as 32-bit values relative to `base` (the offset array's own values
    // with base 0, or multiples of the fixed width with base = the tile's first byte), and
    // what is known before they arrive.
    struct Offsets {
        uint64_t base;
        uint32_t lim;  // bytes from base to the end of the call's records (clamped to 32 bits)
        uint32_t rec[K + 1], s, e;
        uint32_t nrec; // records of the tile
    };
    auto fetch_offsets = [&](uint32_t t, Offsets &f) {
        const uint64_t i0 = (uint64_t) t * T;
        const uint64_t left = q.n - i0;
        f.nrec = left < (uint64_t) T ? (uint32_t) left : (uint32_t) T;
        const uint32_t first = (uint32_t) (wt * K);
        if (q.rec_off) {
            const uint32_t *ro = q.rec_off + (q.first + i0); // the same in every lane
            f.base = 0;
            f.s = ro[vzero];
            f.e = ro[f.nrec + vzero];
#pragma unroll
            for (int k = 0; k <= K; k++) f.rec[k] = ro[min(f.nrec, first + (uint32_t) k)];
        }
        else {
            f.base = (q.first + i0) * (uint64_t) q.fixed_len;
            f.s = 0;
            f.e = f.nrec * q.fixed_len;
#pragma unroll
            for (int k = 0; k <= K; k++) f.rec[k] = min(f.nrec, first + (uint32_t) k) * q.fixed_len;
        }
        const uint64_t room = end_abs > f.base ? end_abs - f.base : 0ull;
        f.lim = room > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) room;
    };
    // May bytes [at, at + 16) from the tile's base be fetched with one 16-byte load?  They must end
    // inside the call's records.
    auto quad_ok = [](uint32_t at, uint32_t lim) -> bool { return lim >= 16u && at <= lim - 16u; };
    // Issue the loads of the first 16 bytes of a tile's records (nothing is computed from the
    // loaded values here).
    auto issue_q0 = [&](const Offsets &f, uint32_t (&r)[K][4]) {
        const uint8_t *tb = q.blob + f.base;
#pragma unroll
        for (int k = 0; k < K; k++) {
            // Every lane loads, from an address clamped into the call's records (or, when those
            // are shorter than one load, from the control block): a load under a condition is
            // followed by a select between its result and zero, and that select waits for the
            // load -- one round trip after the other.  What a clamped load fetches is never
            // looked at (the record's length is zero, or its bytes are fetched one by one in
            // settle).  The clamp keeps the lanes' addresses apart: one address for all of them
            // would be one cache line for the whole GPU.
            const uint8_t *src = f.lim >= 16u ? tb + min(f.rec[k], f.lim - 16u) : reinterpret_cast<const uint8_t *>(p.ctrl);
            const OvU128 v = *reinterpret_cast<const OvU128 *>(src);
            r[k][0] = v.x;
            r[k][1] = v.y;
            r[k][2] = v.z;
            r[k][3] = v.w;
        }
    };
    // Lengths and checks of a tile's records; the few records next to the end of the call
    // (no whole 16-byte load) are fetched byte by byte.  A record is what its offsets promise
    // when it is not empty and not longer than a record may be (offsets that are not monotonic
    // wrap to a huge length) and the lane's first and last boundary lie inside the tile's span;
    // boundaries past the end of the call all equal the tile's end (length zero: no record).
    auto settle = [&](const Offsets &f) {
        const bool fits = f.e >= f.s && f.e - f.s <= (uint32_t) IMG;
        cur_base = f.base;
        cur_lim = f.lim;
        cur_end = f.e;
        if (!fits) // the tile's bytes do not fit the image: the host redoes the call with other kernels
            __hip_atomic_store(&p.ctrl->head.fallback, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t first = (uint32_t) (wt * K);
        const uint32_t mine = f.nrec > first ? min(f.nrec - first, (uint32_t) K) : 0u; // records of this lane
        const bool inside = f.rec[0] >= f.s && f.rec[K] <= f.e;
        bool bad = mine != 0 && !inside;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint32_t l = f.rec[k + 1] - f.rec[k];
            const bool good = l - 1u < GEN_MAX_RECORD;
            roff[k] = f.rec[k];
            len[k] = (fits && inside && good) ? l : 0u;
            bad = bad || (!good && (uint32_t) k < mine);
        }
        if (bad) raise_error(p.ctrl, MGS_ERR_MALFORMED);
        if (__any(f.e + 16u > f.lim)) { // the tile ends next to the end of the call: its last record or two
            uint32_t bytewise = 0;
#pragma unroll
            for (int k = 0; k < K; k++)
                if (len[k] != 0 && !quad_ok(f.rec[k], f.lim)) bytewise |= 1u << k;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                if (!((bytewise >> k) & 1u)) continue;
                const uint32_t l = pick(len, k);
                uint32_t t[4];
                bytewise16(q.blob + f.base + pick(roff, k), l < 16u ? l : 16u, t);
#pragma unroll
                for (int kk = 0; kk < K; kk++) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (kk == k) q0[kk][j] = t[j];
                }
            }
        }
    };
    // Bytes 16..31 of the records that have them (after settle: the lengths are known).
    auto load_q1 = [&](uint32_t (&t)[K][Q1W]) {
        const uint8_t *tb = q.blob + cur_base;
        constexpr uint32_t LB = 4u * Q1W; // bytes of one load
#pragma unroll
        for (int k = 0; k < K; k++) {
            const uint8_t *src = cur_lim >= LB ? tb + min(roff[k] + 16u, cur_lim - LB) // (clamped: see issue_q0)
                                               : reinterpret_cast<const uint8_t *>(p.ctrl);
            if (Q1W == 4) {
                const OvU128 v = *reinterpret_cast<const OvU128 *>(src);
                t[k][0] = v.x;
                t[k][1 % Q1W] = v.y;
                t[k][2 % Q1W] = v.z;
                t[k][3 % Q1W] = v.w;
            }
            else {
                t[k][0] = reinterpret_cast<const OvU32 *>(src)->v;
            }
        }
        if (__any(cur_end + 16u > cur_lim)) { // the tile ends next to the end of the call
            uint32_t bytewise = 0;
#pragma unroll
            for (int k = 0; k < K; k++)
                if (len[k] > 16u && roff[k] + 16u + LB > cur_lim) bytewise |= 1u << k;
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                if (!((bytewise >> k) & 1u)) continue;
                const uint32_t l = pick(len, k) - 16u;
                uint32_t b16[4];
                bytewise16(tb + pick(roff, k) + 16u, l < LB ? l : LB, b16);
#pragma unroll
                for (int kk = 0; kk < K; kk++) {
#pragma unroll
                    for (int j = 0; j < Q1W; j++)
                        if (kk == k) t[kk][j] = b16[j];
                }
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
            if (A4) {
                const uint32_t *s4 = reinterpret_cast<const uint32_t *>(sout + at);
                return make_uint4(s4[0], s4[1], s4[2], s4[3]);
            }
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
    // n <= 16 bytes out of four registers to any address of the image.  Eight bytes and more go as
    // two pieces that may overlap -- the first eight and the last eight -- instead of 8 + 4 + 2 + 1.
    auto put = [&](uint8_t *o, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t n) {
        if (n >= 16) {
            *reinterpret_cast<OvU128 *>(o) = OvU128{a0, a1, a2, a3};
            return;
        }
        if (n >= 8) {
            const unsigned long long lo = (unsigned long long) a0 | ((unsigned long long) a1 << 32);
            const unsigned long long hi = (unsigned long long) a2 | ((unsigned long long) a3 << 32);
            const uint32_t sh = 8u * (n - 8u); // 0 .. 56
            reinterpret_cast<OvU64 *>(o)->v = lo;
            reinterpret_cast<OvU64 *>(o + (n - 8u))->v = sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
            return;
        }
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


    Offsets fn; // offsets of the tile after the current one: fetched one whole trip before its bytes
    fn.base = 0;
    fn.lim = 0;
    fn.s = fn.e = 0;
    fn.nrec = 0;
#pragma unroll
    for (int k = 0; k <= K; k++) fn.rec[k] = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        len[k] = 0;
        roff[k] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) q0[k][j] = 0;
#pragma unroll
        for (int j = 0; j < Q1W; j++) q1[k][j] = 0;
    }
    if (tile < p.ntiles) {
        Offsets f0;
        fetch_offsets(tile, f0);
        if (next < p.ntiles) fetch_offsets(next, fn);
        issue_q0(f0, q0);
        settle(f0);
        if (NQ > 1) load_q1(q1);
    }
    bool have_prev = false;
    // diagnostic build: cycles of worker wave 0 per phase of the trip
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    if (VAR & OSV_STATS) tprev = __builtin_readcyclecounter();
#define OSR_PHASE(i)                                                     \
    if (VAR & OSV_STATS) {                                               \
        const unsigned long long tnow = __builtin_readcyclecounter();    \
        ph[i] += tnow - tprev;                                           \
        tprev = tnow;                                                    \
    }
    const uint32_t vlen = (q.lay.val_kind == MGS_VAL_FIXED) ? (uint32_t) q.lay.val_bytes : 0u;
    const uint32_t fixed_klen = (uint32_t) q.lay.key_bytes; // LM == 2

    while (tile < p.ntiles) {
        uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
        // ---- hash: destination of each of this lane's records
        uint32_t d[K];
        uint32_t longkey = 0; // string keys of more than 12 bytes: hashed below, one at a time
        bool bad = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (LM == 1) {
                // straight-line code: a record that does not exist (length 0) or whose key is not
                // 1..12 bytes long computes on zero masks and is sorted out by the flags
                const uint32_t klen = len[k] - vlen;
                const bool have = len[k] != 0, nokey = have && len[k] <= vlen;
                const bool shortk = klen - 1u < 12u && len[k] > vlen;
                const uint4 m = ktab[shortk ? klen : 0u];
                const uint32_t x0 = q0[k][0] & m.x, x1 = q0[k][1] & m.y, x2 = q0[k][2] & m.z;
                uint32_t a, b, c;
                a = b = c = 0xdeadbeefu + klen;
                a += x0;
                b += x1;
                c += x2;
                MGS_FINAL(a, b, c);
                // the key's NUL rule: bytes beyond the key count as non-zero; byte i of word j is bit 8 * i + j of Z
                const uint32_t Z = (osr_zero_flags(x0 | ~m.x) >> 7) | (osr_zero_flags(x1 | ~m.y) >> 6)
                                   | (osr_zero_flags(x2 | ~m.z) >> 5);
                const bool valid = shortk && Z == m.w;
                const bool lk = have && !nokey && !shortk;
                bad = bad || nokey || (have && shortk && !valid);
                d[k] = valid ? (p.fm.pow2 ? fastmod_pow2(c, p.fm) : fastmod_magic(c, p.fm)) : 0xFFu;
                len[k] = (valid || lk) ? len[k] : 0u;
                if (lk) longkey |= 1u << k;
            }
            else {
                d[k] = 0xFFu;
                bool ok = fixed_klen <= len[k] && fixed_klen <= 16u; // (the same for every record)
                uint32_t hk = 0;
                if (ok) hk = osr_hash_key16<false>(q0[k], fixed_klen, ktab, ok);
                if (len[k] != 0 && !ok) bad = true;
                if (len[k] != 0 && ok) d[k] = p.fm.pow2 ? fastmod_pow2(hk, p.fm) : fastmod_magic(hk, p.fm);
                else len[k] = 0;
            }
        }
        if (LM == 1 && __any(longkey != 0)) {
#pragma unroll 1
            for (int k = 0; k < K; k++) {
                if (!((longkey >> k) & 1u)) continue;
                const uint32_t klen = pick(len, k) - vlen;
                bool ok = true;
                uint32_t hk;
                if (klen <= 16u) { // still inside the first quad
                    uint32_t w4[4];
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        uint32_t col[K];
#pragma unroll
                        for (int kk = 0; kk < K; kk++) col[kk] = q0[kk][j];
                        w4[j] = pick(col, k);
                    }
                    hk = osr_hash_key16<true>(w4, klen, ktab, ok);
                }
                else { // from global memory, byte by byte
                    const uint8_t *r = q.blob + cur_base + pick(roff, k);
                    ok = r[klen - 1] == 0;
#pragma unroll 1
                    for (uint32_t j = 0; j + 1 < klen; j++) ok = ok && r[j] != 0;
                    hk = hashlittle_bytes(r, klen);
                }
                const uint32_t dk = p.fm.pow2 ? fastmod_pow2(hk, p.fm) : fastmod_magic(hk, p.fm);
                bad = bad || !ok;
#pragma unroll
                for (int kk = 0; kk < K; kk++) {
                    if (kk != k) continue;
                    d[kk] = ok ? dk : 0xFFu;
                    if (!ok) len[kk] = 0;
                }
            }
        }
        if (bad) raise_error(p.ctrl, MGS_ERR_MALFORMED);
        if (p.dest_out) {
#pragma unroll
            for (int k = 0; k < K; k++)
                if (len[k]) p.dest_out[(uint64_t) tile * T + (uint64_t) wt * K + k] = (uint8_t) d[k];
        }
        OSR_PHASE(0)
        // ---- rank: byte position of each record inside the wave's share of its destination.
        // Two destinations share one scan: 16-bit fields (a wave's bytes fit: the tile's
        // bytes fit the image).
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
        OSR_PHASE(1)
        __syncthreads(); // A: counts(tile) complete; tables of the previous tile are ready
        OSR_PHASE(2)
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
        // the first quads of the next tile's records (its offsets arrived during the last
        // trip): in flight during the store and the reorder
        uint32_t nx[K][4];
        if (PRE) issue_q0(fn, nx); // (offsets of a tile that does not exist are all zero: nothing is loaded)
        if (have_prev) store_image(par ^ 1, false);
        OSR_PHASE(3)
        __syncthreads(); // B: the image is free
        OSR_PHASE(4)
        const uint32_t next2 = (uint32_t) __builtin_amdgcn_readfirstlane((int) misc[1]);

        // offsets of the tile after the next: in flight during the reorder
        Offsets fn2;
        fn2.base = 0;
        fn2.lim = 0;
        fn2.s = fn2.e = 0;
        fn2.nrec = 0;
#pragma unroll
        for (int k = 0; k <= K; k++) fn2.rec[k] = 0;
        if (next2 < p.ntiles) fetch_offsets(next2, fn2);
        {
            // reorder: where every record goes (reads), then the writes out of registers
            uint32_t dst[K];
#pragma unroll
            for (int k = 0; k < K; k++) dst[k] = wslot[w * OS_MAXP + (len[k] ? d[k] : 0u)] + pw[k];
            bool more = false;
#pragma unroll
            for (int k = 0; k < K; k++) more |= len[k] > 16u;
            if (A4) {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (len[k] == 0) continue;
                    uint32_t *o = reinterpret_cast<uint32_t *>(sout + dst[k]);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((uint32_t) (4 * j) < len[k]) o[j] = q0[k][j];
                }
                if (NQ > 1 && __any(more)) {
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        uint32_t *o = reinterpret_cast<uint32_t *>(sout + dst[k]);
#pragma unroll
                        for (int j = 0; j < Q1W; j++)
                            if ((uint32_t) (16 + 4 * j) < len[k]) o[4 + j] = q1[k][j];
                    }
                }
            }
            else {
#pragma unroll
                for (int k = 0; k < K; k++) {
                    if (len[k] == 0) continue;
                    put(sout + dst[k], q0[k][0], q0[k][1], q0[k][2], q0[k][3], len[k]);
                }
                if (NQ > 1 && __any(more)) {
#pragma unroll
                    for (int k = 0; k < K; k++) {
                        if (len[k] <= 16) continue;
                        const uint32_t n1 = len[k] - 16u < 4u * Q1W ? len[k] - 16u : 4u * Q1W; // (the rest: below)
                        if (n1 < 8u) {
                            // one piece: the eight bytes that end with byte 16 + n1 of the record
                            const unsigned long long lo = (unsigned long long) q0[k][2] | ((unsigned long long) q0[k][3] << 32);
                            const unsigned long long hi =
                                (unsigned long long) q1[k][0] | ((unsigned long long) q1[k][1 % Q1W] << 32);
                            const uint32_t sh = 8u * n1; // 8 .. 56
                            reinterpret_cast<OvU64 *>(sout + dst[k] + 8u + n1)->v = (lo >> sh) | (hi << (64u - sh));
                        }
                        else {
                            put(sout + dst[k] + 16, q1[k][0], q1[k][1 % Q1W], q1[k][2 % Q1W], q1[k][3 % Q1W], n1);
                        }
                    }
                }
            }
            // records longer than their registers: the rest from global memory, byte by byte
            bool longer = false;
#pragma unroll
            for (int k = 0; k < K; k++) longer |= len[k] > RB;
            if (__any(longer)) {
                uint32_t nlong = 0;
#pragma unroll 1
                for (int k = 0; k < K; k++) {
                    const uint32_t l = pick(len, k);
                    if (l <= RB) continue;
                    nlong++;
                    const uint8_t *src = q.blob + cur_base + pick(roff, k);
                    uint8_t *o = sout + pick(dst, k);
#pragma unroll 1
                    for (uint32_t j = RB; j < l; j++) o[j] = src[j];
                }
                if (nlong) atomicAdd(&p.ctrl->head.long_records, nlong);
            }
        }
        // the next tile's records become the current ones
        OSR_PHASE(5)
        if (PRE) {
#pragma unroll
            for (int k = 0; k < K; k++) {
#pragma unroll
                for (int j = 0; j < 4; j++) q0[k][j] = nx[k][j];
            }
        }
        else {
            issue_q0(fn, q0); // the registers are free again; the other waves' work covers the round trip
        }
        settle(fn); // (a tile that does not exist: every length becomes zero)
        if (NQ > 1) {
            bool more2 = false;
#pragma unroll
            for (int k = 0; k < K; k++) more2 |= len[k] > 16u;
            if (__any(more2)) load_q1(q1);
        }
        fn = fn2;
        OSR_PHASE(6)
        __builtin_amdgcn_s_setprio(0);
        have_prev = true;
        tile = next;
        next = next2;
        par ^= 1;
    }
    __syncthreads(); // tables of the last tile are ready
    if (have_prev) store_image(par ^ 1, true);
    // (OSV_NO_PREFETCH has no other meaning here: with it the last worker wave reports)
    if ((VAR & OSV_STATS) && wt == ((VAR & OSV_NO_PREFETCH) ? WT - 64 : 0)) {
#pragma unroll
        for (int i = 0; i < 8; i++) atomicAdd(&p.ctrl->head.stats[i], ph[i]);
    }
#undef OSR_PHASE
}

} // namespace mgs
#endif
