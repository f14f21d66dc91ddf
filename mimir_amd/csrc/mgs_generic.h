// Generic three-pass path: any record layout, any nranks <= MGS_MAX_RANKS.
//
//   k_gen_hist     one lane per record: key length, hashlittle, dest; per-tile
//                  byte histogram (one column per destination + a record count
//                  for the local-bypass destination)
//   k_gen_scan     exclusive scan of the tile histograms per destination
//   k_gen_scatter  stable scatter: each record goes to
//                  partition[dest] + (bytes of earlier records with that dest)
//   k_gen_finalize send_offset[] / counts at the point where the round ended
//
// Order inside a tile is (wave, row, lane): wave w of a tile owns the
// GEN_ITEMS*64 consecutive records starting at tile*GEN_TILE + w*GEN_ITEMS*64,
// so index order == (wave, row, lane) order and the scatter is stable.
#ifndef MGS_GENERIC_H
#define MGS_GENERIC_H

#include "mgs_common.h"

namespace mgs {

constexpr int GEN_THREADS = 256;
constexpr int GEN_WAVES = GEN_THREADS / 64;
constexpr int GEN_ITEMS = 8;
constexpr int GEN_TILE = GEN_THREADS * GEN_ITEMS; // 2048 records
constexpr uint32_t GEN_INVALID = 0xFFFFFFFFu;
constexpr uint32_t GEN_MAX_RECORD = 1u << 20; // MAX_RECORD_SIZE, src/config.cpp

struct GenParams {
    const uint8_t *blob;
    const uint32_t *rec_off; // absolute record index -> byte offset, or NULL
    uint32_t fixed_len;
    uint64_t first; // absolute index of record 0 of this call
    uint64_t n;     // records in this call
    mgs_layout lay;
    uint32_t P;
    int32_t self;
    FastMod fm;
    uint32_t ntiles;
    uint32_t *tile_hist; // [ntiles][P+1]
    uint64_t *tile_base; // [ntiles+1][P+1]
    uint8_t *dest;       // [n] scratch (or the caller's d_dest_out)
    uint32_t gate;       // 1: run only when Ctrl::head.fallback is raised (enqueued behind the one-sweep path)
    Ctrl *ctrl;
};

__device__ __forceinline__ void gen_span(const GenParams &p, uint64_t i,
                                         uint64_t &off, uint32_t &len)
{
    if (p.rec_off) {
        const uint32_t a = p.rec_off[p.first + i];
        const uint32_t b = p.rec_off[p.first + i + 1];
        off = a;
        len = b - a;
    }
    else {
        off = (p.first + i) * (uint64_t) p.fixed_len;
        len = p.fixed_len;
    }
}

// Key length of one wire-form record (Serializer::get_key_bytes,
// src/serializer.h:418-421): sizeof(K)*keycount, or the bytes of key_count C
// strings including their NULs.  Returns false when the record cannot be a
// serialisation of the layout.
__device__ inline bool gen_key_len(const mgs_layout &lay, const uint8_t *r,
                                   uint32_t len, uint32_t &klen)
{
    // the bytes of `count` C strings with their NULs, starting at r[from]; ~0u: they do not end inside the record
    auto strings = [&](uint32_t from, int count) -> uint32_t {
        uint32_t pos = from;
        for (int s = 0; s < (count > 0 ? count : 1); s++) {
            while (pos < len && r[pos] != 0) pos++;
            if (pos >= len) return ~0u;
            pos++;
        }
        return pos - from;
    };
    if (lay.key_kind == MGS_KEY_FIXED) klen = (uint32_t) lay.key_bytes;
    else klen = strings(0, lay.key_count);
    if (klen == ~0u || klen > len) return false;
    uint32_t vlen = 0;
    if (lay.val_kind == MGS_VAL_FIXED) vlen = (uint32_t) lay.val_bytes;
    else if (lay.val_kind == MGS_VAL_CSTR) vlen = strings(klen, lay.val_count);
    // key and value are the whole record: a string key's first NUL(s) end it, nothing
    // may follow the value
    return vlen != ~0u && klen + vlen == len;
}

#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
__global__ __launch_bounds__(GEN_THREADS) void k_gen_hist(GenParams p)
{
    if (p.gate && p.ctrl->head.fallback == 0) return;
    __shared__ uint32_t hist[MGS_MAX_RANKS + 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t cols = p.P + 1;
    for (uint32_t b = tid; b < cols; b += GEN_THREADS) hist[b] = 0;
    __syncthreads();

#pragma unroll 1
    for (int k = 0; k < GEN_ITEMS; k++) {
        const uint64_t i = (uint64_t) tile * GEN_TILE + w * (GEN_ITEMS * 64) + k * 64 + lane;
        if (i >= p.n) continue;
        uint64_t off;
        uint32_t len, klen = 0;
        gen_span(p, i, off, len);
        const uint8_t *r = p.blob + off;
        uint32_t d = 0;
        if (len == 0 || len > GEN_MAX_RECORD || !gen_key_len(p.lay, r, len, klen)) {
            raise_error(p.ctrl, MGS_ERR_MALFORMED);
            len = 0;
        }
        else {
            d = fastmod(hashlittle_bytes(r, klen), p.fm);
        }
        p.dest[i] = (uint8_t) d;
        if (len) {
            atomicAdd(&hist[d], len);
            if ((int32_t) d == p.self) atomicAdd(&hist[p.P], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = tid; b < cols; b += GEN_THREADS)
        p.tile_hist[(uint64_t) tile * cols + b] = hist[b];
}
#endif

// One block.  Column b < P of tile_base starts at part_init[b]; column P
// (local-bypass record count) starts at 0.  Row ntiles receives the totals.
#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
__global__ __launch_bounds__(1024) void k_gen_scan(GenParams p)
{
    if (p.gate && p.ctrl->head.fallback == 0) return;
    __shared__ unsigned long long csum[1024];
    const uint32_t cols = p.P + 1;
    const uint32_t C = 1024 / cols; // >= 3
    const uint32_t tid = threadIdx.x;
    const uint32_t c = tid / cols, b = tid % cols;
    const uint32_t T = p.ntiles;
    const uint32_t Lc = (T + C - 1) / C;
    const bool active = c < C;
    const uint32_t t0 = active ? min(T, c * Lc) : T;
    const uint32_t t1 = active ? min(T, t0 + Lc) : T;

    unsigned long long sum = 0;
#pragma unroll 8
    for (uint32_t t = t0; t < t1; t++) sum += p.tile_hist[(uint64_t) t * cols + b];
    csum[tid] = sum;
    __syncthreads();
    if (!active) return;
    unsigned long long run = (b < p.P) ? (unsigned long long) p.ctrl->head.part_init[b] : 0ull;
    for (uint32_t cc = 0; cc < c; cc++) run += csum[cc * cols + b];
#pragma unroll 8
    for (uint32_t t = t0; t < t1; t++) {
        p.tile_base[(uint64_t) t * cols + b] = run;
        run += p.tile_hist[(uint64_t) t * cols + b];
    }
    // exactly one chunk per column ends at T and is either non-empty or chunk 0
    if (t1 == T && (t0 < T || c == 0)) p.tile_base[(uint64_t) T * cols + b] = run;
}
#endif

__device__ __forceinline__ void copy_record(uint8_t *dst, const uint8_t *src, uint32_t len)
{
    const uintptr_t mix = (uintptr_t) dst | (uintptr_t) src | (uintptr_t) len;
    if ((mix & 15) == 0) {
        for (uint32_t j = 0; j < len; j += 16)
            *reinterpret_cast<uint4 *>(dst + j) = *reinterpret_cast<const uint4 *>(src + j);
    }
    else if ((mix & 3) == 0) {
        for (uint32_t j = 0; j < len; j += 4)
            *reinterpret_cast<uint32_t *>(dst + j) = *reinterpret_cast<const uint32_t *>(src + j);
    }
    else {
        for (uint32_t j = 0; j < len; j++) dst[j] = src[j];
    }
}

#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
__global__ __launch_bounds__(GEN_THREADS) void k_gen_scatter(GenParams p)
{
    if (p.gate && p.ctrl->head.fallback == 0) return;
    __shared__ uint32_t wtot[GEN_WAVES][MGS_MAX_RANKS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint32_t tile = blockIdx.x;
    const uint32_t cols = p.P + 1;
    for (uint32_t j = tid; j < GEN_WAVES * MGS_MAX_RANKS; j += GEN_THREADS)
        (&wtot[0][0])[j] = 0;
    __syncthreads();

    uint64_t off[GEN_ITEMS];
    uint32_t len[GEN_ITEMS];
    uint32_t d[GEN_ITEMS];
#pragma unroll
    for (int k = 0; k < GEN_ITEMS; k++) {
        const uint64_t i = (uint64_t) tile * GEN_TILE + w * (GEN_ITEMS * 64) + k * 64 + lane;
        d[k] = GEN_INVALID;
        len[k] = 0;
        off[k] = 0;
        if (i < p.n) {
            gen_span(p, i, off[k], len[k]);
            if (len[k] != 0 && len[k] <= GEN_MAX_RECORD) {
                d[k] = p.dest[i];
                atomicAdd(&wtot[w][d[k]], len[k]);
            }
        }
    }
    __syncthreads();
    // per destination: exclusive prefix over the waves of this tile
    for (uint32_t b = tid; b < p.P; b += GEN_THREADS) {
        uint32_t run = 0;
        for (int ww = 0; ww < GEN_WAVES; ww++) {
            const uint32_t t = wtot[ww][b];
            wtot[ww][b] = run;
            run += t;
        }
    }
    __syncthreads();

    volatile uint32_t *wr = &wtot[w][0]; // this wave's running byte position per destination
#pragma unroll 1
    for (int k = 0; k < GEN_ITEMS; k++) {
        const bool valid = d[k] != GEN_INVALID;
        uint32_t myoff = 0;
        uint64_t remaining = __ballot(valid);
        while (remaining) { // wave-uniform: one trip per distinct destination in the row
            const int leader = __ffsll((unsigned long long) remaining) - 1;
            const uint32_t b = __shfl(d[k], leader, 64);
            const bool inb = valid && d[k] == b;
            const uint64_t mask = __ballot(inb);
            const uint32_t v = inb ? len[k] : 0u;
            const uint32_t incl = wave_incl_scan_u32(v, lane);
            const uint32_t total = __shfl(incl, 63, 64);
            const uint32_t base = wr[b];
            if (inb) myoff = base + incl - v;
            if (lane == leader) wr[b] = base + total;
            remaining &= ~mask;
        }
        if (valid) {
            const uint64_t i = (uint64_t) tile * GEN_TILE + w * (GEN_ITEMS * 64) + k * 64 + lane;
            const uint32_t b = d[k];
            const uint64_t pos = p.tile_base[(uint64_t) tile * cols + b] + myoff;
            const int64_t cap = p.ctrl->head.part_cap[b];
            if ((int64_t)(pos + len[k]) > cap) {
                if ((int32_t) b == p.self)
                    raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                else if ((int64_t) len[k] > cap)
                    raise_error(p.ctrl, MGS_ERR_RECORD_SIZE);
                else
                    atomicMin(&p.ctrl->head.cut_index, (unsigned long long) i);
            }
            else {
                copy_record(reinterpret_cast<uint8_t *>(p.ctrl->head.part_base[b]) + pos,
                            p.blob + off[k], len[k]);
            }
        }
    }
}
#endif

// One block.  Everything before the cut index counts; the rest is ignored.
#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
__global__ __launch_bounds__(GEN_THREADS) void k_gen_finalize(GenParams p)
{
    if (p.gate && p.ctrl->head.fallback == 0) return;
    __shared__ unsigned long long part[MGS_MAX_RANKS + 1];
    const uint32_t tid = threadIdx.x;
    const uint32_t cols = p.P + 1;
    unsigned long long e = p.ctrl->head.cut_index;
    if (e > p.n) e = p.n;
    const uint64_t te = e / GEN_TILE;
    for (uint32_t b = tid; b < cols; b += GEN_THREADS) part[b] = 0;
    __syncthreads();
    for (uint64_t i = te * GEN_TILE + tid; i < e; i += GEN_THREADS) {
        uint64_t off;
        uint32_t len;
        gen_span(p, i, off, len);
        if (len == 0 || len > GEN_MAX_RECORD) continue;
        const uint32_t d = p.dest[i];
        atomicAdd(&part[d], (unsigned long long) len);
        if ((int32_t) d == p.self) atomicAdd(&part[p.P], 1ull);
    }
    __syncthreads();
    mgs_round *res = &p.ctrl->result;
    for (uint32_t b = tid; b < p.P; b += GEN_THREADS) {
        const unsigned long long tot = p.tile_base[te * cols + b] + part[b];
        if ((int32_t) b == p.self) {
            res->send_offset[b] = 0;
            res->local_bytes = (int64_t) tot;
        }
        else {
            res->send_offset[b] = (int64_t) tot;
        }
    }
    if (tid == 0) {
        const unsigned long long lrec = p.tile_base[te * cols + p.P] + part[p.P];
        res->next = p.first + e;
        res->local_records = lrec;
        res->packed_records = e - lrec;
        if (p.self < 0) res->local_bytes = 0;
    }
}
#endif

} // namespace mgs
#endif
