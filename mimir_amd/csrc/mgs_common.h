// Shared host/device definitions for the gfx950 shuffle-transform kernels.
#ifndef MGS_COMMON_H
#define MGS_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mimir_gpu_shuffle.h"

namespace mgs {

// ---------------------------------------------------------------------------
// lookup3 (Bob Jenkins, public domain) as used by Mimir: src/hash.cpp:14
// (rot), :60-80 (mix), :107-123 (final).  All arithmetic is u32 wrap-around.
// ---------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int k)
{
    return (x << k) | (x >> (32 - k));
}

#define MGS_MIX(a, b, c)                 \
    {                                    \
        a -= c; a ^= mgs::rotl32(c, 4);  c += b; \
        b -= a; b ^= mgs::rotl32(a, 6);  a += c; \
        c -= b; c ^= mgs::rotl32(b, 8);  b += a; \
        a -= c; a ^= mgs::rotl32(c, 16); c += b; \
        b -= a; b ^= mgs::rotl32(a, 19); a += c; \
        c -= b; c ^= mgs::rotl32(b, 4);  b += a; \
    }

#define MGS_FINAL(a, b, c)               \
    {                                    \
        c ^= b; c -= mgs::rotl32(b, 14); \
        a ^= c; a -= mgs::rotl32(c, 11); \
        b ^= a; b -= mgs::rotl32(a, 25); \
        c ^= b; c -= mgs::rotl32(b, 16); \
        a ^= c; a -= mgs::rotl32(c, 4);  \
        b ^= a; b -= mgs::rotl32(a, 14); \
        c ^= b; c -= mgs::rotl32(b, 24); \
    }

// hashlittle(key, len, 0) over bytes fetched one at a time (src/hash.cpp:
// 344-400).  Never reads past key+len, whatever the alignment.
template <typename BytePtr>
__host__ __device__ inline uint32_t hashlittle_bytes(BytePtr k, uint32_t length)
{
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + length;
    while (length > 12) {
        a += (uint32_t) k[0] | ((uint32_t) k[1] << 8) | ((uint32_t) k[2] << 16)
             | ((uint32_t) k[3] << 24);
        b += (uint32_t) k[4] | ((uint32_t) k[5] << 8) | ((uint32_t) k[6] << 16)
             | ((uint32_t) k[7] << 24);
        c += (uint32_t) k[8] | ((uint32_t) k[9] << 8) | ((uint32_t) k[10] << 16)
             | ((uint32_t) k[11] << 24);
        MGS_MIX(a, b, c);
        length -= 12;
        k += 12;
    }
    if (length == 0) return c; // src/hash.cpp:394-395
    uint32_t t0 = 0, t1 = 0, t2 = 0;
    for (uint32_t i = 0; i < length; i++) {
        const uint32_t v = (uint32_t) k[i] << (8 * (i & 3));
        if (i < 4) t0 |= v;
        else if (i < 8) t1 |= v;
        else t2 |= v;
    }
    a += t0;
    b += t1;
    c += t2;
    MGS_FINAL(a, b, c);
    return c;
}

// hashlittle of the first `klen` (1..16) bytes of a 16-byte record held as
// four little-endian words.
__host__ __device__ __forceinline__ uint32_t hash_rec16(uint32_t w0, uint32_t w1,
                                                        uint32_t w2, uint32_t w3,
                                                        uint32_t klen)
{
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + klen;
    if (klen > 12) {
        a += w0; b += w1; c += w2;
        MGS_MIX(a, b, c);
        const uint32_t r = klen - 12; // 1..4 tail bytes, all in w3
        a += (r >= 4) ? w3 : (w3 & ((1u << (8 * r)) - 1u));
    }
    else {
        // tail of 1..12 bytes: whole words, then one masked word
        const uint32_t full = klen >> 2, r = klen & 3;
        const uint32_t m = (1u << (8 * r)) - 1u; // 0 when r == 0
        a += (full >= 1) ? w0 : (w0 & m);
        b += (full >= 2) ? w1 : ((full == 1) ? (w1 & m) : 0u);
        c += (full >= 3) ? w2 : ((full == 2) ? (w2 & m) : 0u);
    }
    MGS_FINAL(a, b, c);
    return c;
}

// ---------------------------------------------------------------------------
// dest = hash % (uint32_t) nranks (src/baseshuffler.h:277) without a divide:
// round-up multiply-high division (exact for every 32-bit numerator), then
// n - q*d.  Power-of-two divisors use a mask.
// ---------------------------------------------------------------------------
struct FastMod {
    uint32_t d;
    uint32_t magic;
    uint32_t shift;
    uint32_t pow2; // 1: d is a power of two
};

inline FastMod make_fastmod(uint32_t d)
{
    FastMod f;
    f.d = d;
    f.magic = 0;
    f.shift = 0;
    f.pow2 = ((d & (d - 1)) == 0) ? 1u : 0u;
    if (!f.pow2) {
        // L = floor(log2 d); magic = floor(2^(33+L... see below) ; the 33-bit
        // reciprocal m' = 2^32 + magic = ceil(2^(32+L+1) / d), and
        // q = (((n - hi) >> 1) + hi) >> L with hi = mulhi(magic, n).
        uint32_t L = 31 - (uint32_t) __builtin_clz(d);
        const unsigned __int128 one = 1;
        const unsigned __int128 num = one << (33 + L);
        const unsigned __int128 mp = (num + d - 1) / d; // ceil
        f.magic = (uint32_t)(mp - (one << 32));
        f.shift = L;
    }
    return f;
}

// The two cases of fastmod as separate functions, for callers that hoist the
// (uniform) choice between them out of their inner loop.
__host__ __device__ __forceinline__ uint32_t fastmod_pow2(uint32_t n, const FastMod &f)
{
    return n & (f.d - 1);
}

__host__ __device__ __forceinline__ uint32_t fastmod_magic(uint32_t n, const FastMod &f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t hi = __umulhi(f.magic, n);
#else
    const uint32_t hi = (uint32_t)(((uint64_t) f.magic * (uint64_t) n) >> 32);
#endif
    const uint32_t q = (((n - hi) >> 1) + hi) >> f.shift;
    return n - q * f.d;
}

__host__ __device__ __forceinline__ uint32_t fastmod(uint32_t n, const FastMod &f)
{
    if (f.pow2) return n & (f.d - 1);
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t hi = __umulhi(f.magic, n);
#else
    const uint32_t hi = (uint32_t)(((uint64_t) f.magic * (uint64_t) n) >> 32);
#endif
    const uint32_t q = (((n - hi) >> 1) + hi) >> f.shift;
    return n - q * f.d;
}

// ---------------------------------------------------------------------------
// Device-side control block shared by the kernels of one pack call.
// ---------------------------------------------------------------------------
struct CtrlHead {
    unsigned long long cut_index; // generic path: first record that did not fit
    uint32_t cut_tile;            // one-sweep path: first tile holding such a record
    uint32_t ticket;              // one-sweep path: next tile to hand out
    int32_t error;                // first mgs_status raised on the device
    uint32_t fallback;            // byte-granular one-sweep path: 1 = a tile does not fit LDS, run the generic path
    uint32_t rounds_cut;          // round-cutting kernels: round cuts published so far (= index of the last round begun)
    uint32_t f_cuts;              // streams layout: round cuts found by the round finder
    uint32_t long_records;        // k_onesweepr: records longer than the registers that hold a record
    uint64_t part_base[MGS_MAX_RANKS]; // device address of each partition's byte 0
    int64_t part_cap[MGS_MAX_RANKS];   // capacity in bytes
    int64_t part_init[MGS_MAX_RANKS];  // bytes already there
    int64_t round_base0[MGS_MAX_RANKS]; // streams layout: byte of each stream at which the open round began
    unsigned long long stats[8];       // diagnostic builds only (see OSV_STATS)
    unsigned long long fstats[4];      // round finder: its wall time (100 MHz ticks), cycles searching / hashing / walking
};

struct Ctrl {
    CtrlHead head;
    mgs_round result;
};

__device__ __forceinline__ void raise_error(Ctrl *ctrl, int code)
{
    atomicCAS(&ctrl->head.error, 0, code);
}

// 64-lane helpers -----------------------------------------------------------
__device__ __forceinline__ uint32_t lanes_below(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t) mask, 0u));
}

// Inclusive prefix sum over the 64 lanes of a wave with data-parallel-primitive
// moves (no LDS traffic): shifts by 1, 2, 4, 8 inside each row of 16 lanes,
// then the last lane of row 0/2 into row 1/3, then lane 31 into rows 2-3.
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane)
{
    (void) lane;
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x111, 0xf, 0xf, false); // row_shr:1
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x112, 0xf, 0xf, false); // row_shr:2
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x114, 0xf, 0xf, false); // row_shr:4
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x118, 0xf, 0xf, false); // row_shr:8
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x142, 0xa, 0xf, false); // row_bcast:15
    v += (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, 0x143, 0xc, 0xf, false); // row_bcast:31
    return v;
}

} // namespace mgs
#endif
