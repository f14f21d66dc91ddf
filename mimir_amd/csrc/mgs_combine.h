// Send side with a combiner: one round of CombineCollectiveShuffler::write
// (src/combinecollectiveshuffler.h:63-180) for an 8-byte value and a declared
// operation (sum / min / max of uint64; the reference calls a user callback,
// :139 -- wordcount's is the sum, examples/wordcount.cpp:96-100).
//
// What the round leaves: a record whose destination is this rank goes to the
// local stream as it is (:67-83); of the other records only the first with each
// key is written to its partition (:94-135), in emit order, and every later one
// is combined into it in place (:137-151); the round ends before the first NEW
// key that does not fit its partition (:122-127), after which the keys are
// forgotten (:237).
//
// On the device, over records [start, n_end):
//   k_grp_insert   (mgs_group.h) the hash table: slot of every record, the slot
//                  holds the first record with the key; destination per record
//   k_cmb_cols     kept = local or first with its key (the first leaves its index
//                  beside the slot); per tile of 2048 records the kept bytes per
//                  destination (+ two count columns)
//   k_cmb_colscan  exclusive sums of every column over the tiles
//   k_cmb_place    byte position of every kept record in its partition / the
//                  local stream (stable: emit order); the first remote record
//                  that does not fit marks the end of the round
//   k_cmb_reduce   every combined record before that end: atomic op on the
//                  accumulator of its key's first record
//   k_cmb_emit     kept records before the end -> partitions (key + accumulator)
//                  and local stream
//   k_cmb_final    send_offset[] / counts at the end of the round
// Keys that first occur after the end of the round cost table work for nothing;
// the caller feeds batches of about a round (as with mgs_pack_device).
#ifndef MGS_COMBINE_H
#define MGS_COMBINE_H

#include "mgs_group.h"

namespace mgs {

constexpr int CMB_MAXCOL = OS_MAXP + 2; // destinations + local records + new keys

struct CmbParams {
    GrpParams g;           // records [start, n_end) as records 0 .. g.n-1; table, slots, tiles, first_bits (= kept)
    uint64_t start;        // index of record 0 in the caller's stream
    uint32_t P;            // destinations
    int32_t self;          // this rank, or -1
    int64_t buf_size;
    int64_t local_cap;
    int op;                // MGS_COMBINE_*
    uint32_t *col;         // [ntiles + 1][ncol] kept bytes per (tile, destination), count columns last
    uint32_t ncol;         // P + 2
    unsigned long long *col_x; // [ntiles + 1][ncol] the same before the tile (+ totals)
    uint32_t *pos;         // [n] byte position of a kept record
    unsigned long long *acc; // [n] combined value, at the first record of a key
    unsigned long long *end; // [1] first record not in the round (g.n when all fit)
    uint8_t *image;        // send image
    uint8_t *local_out;
    mgs_round *round;      // out
};

__device__ __forceinline__ uint32_t cmb_len(const GrpParams &g, uint64_t i)
{
    return (uint32_t) (grp_offset(g, i + 1) - grp_offset(g, i));
}

// kept bit of every record and the tile's column sums.
__global__ __launch_bounds__(GRP_THREADS) void k_cmb_cols(CmbParams p)
{
    __shared__ uint32_t sum[CMB_MAXCOL];
    const GrpParams &g = p.g;
    const uint64_t t0 = (uint64_t) blockIdx.x * GRP_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (uint32_t c = threadIdx.x; c < p.ncol; c += GRP_THREADS) sum[c] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        const uint64_t i = t0 + (uint64_t) k * GRP_THREADS + threadIdx.x;
        bool kept = false;
        if (i < g.n) {
            const uint32_t d = g.dest_out[i];
            const bool local = (int32_t) d == p.self;
            const uint32_t slot = g.slot_or_rank[i];
            const bool first = (uint32_t) g.table[slot] == grp_place(g, i);
            kept = local || first;
            if (first && !local) g.rank_table[slot] = (uint32_t) i; // where the key's accumulator is
            if (kept) {
                atomicAdd(&sum[d], cmb_len(g, i));
                atomicAdd(&sum[p.P + (local ? 0u : 1u)], 1u);
            }
        }
        const uint64_t bits = __ballot(kept);
        if (lane == 0) g.first_bits[(t0 + (uint64_t) k * GRP_THREADS) / 64 + w] = bits;
    }
    __syncthreads();
    for (uint32_t c = threadIdx.x; c < p.ncol; c += GRP_THREADS) p.col[(uint64_t) blockIdx.x * p.ncol + c] = sum[c];
}

// One workgroup: exclusive sums of every column over the tiles; totals at [ntiles].
__global__ __launch_bounds__(1024) void k_cmb_colscan(CmbParams p)
{
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t nt = p.g.ntiles;
    for (uint32_t c = 0; c < p.ncol; c++) {
        if (threadIdx.x == 0) carry = 0;
        __syncthreads();
        for (uint32_t t0 = 0; t0 < nt; t0 += 1024) {
            const uint32_t t = t0 + threadIdx.x;
            const unsigned long long v = t < nt ? p.col[(uint64_t) t * p.ncol + c] : 0ull;
            // inclusive sum over the wave, 64-bit
            unsigned long long incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned long long up = __shfl_up(incl, o);
                if (lane >= o) incl += up;
            }
            if (lane == 63) wsum[w] = incl;
            __syncthreads();
            unsigned long long before = carry;
            for (int ww = 0; ww < w; ww++) before += wsum[ww];
            if (t < nt) p.col_x[(uint64_t) t * p.ncol + c] = before + incl - v;
            __syncthreads();
            if (threadIdx.x == 1023) carry = before + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) p.col_x[(uint64_t) nt * p.ncol + c] = carry;
        __syncthreads();
    }
    if (threadIdx.x == 0) *p.end = p.g.n;
}

// Positions.  Thread t of a tile owns GRP_ITEMS consecutive records; per
// destination one exclusive sum over the workgroup of the threads' kept bytes.
__global__ __launch_bounds__(GRP_THREADS) void k_cmb_place(CmbParams p)
{
    __shared__ uint32_t wsum[GRP_THREADS / 64];
    const GrpParams &g = p.g;
    const uint64_t t0 = (uint64_t) blockIdx.x * GRP_TILE + (uint64_t) threadIdx.x * GRP_ITEMS;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t d[GRP_ITEMS], len[GRP_ITEMS];
    bool kept[GRP_ITEMS];
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        const uint64_t i = t0 + k;
        kept[k] = i < g.n && ((g.first_bits[i / 64] >> (i & 63)) & 1ull);
        d[k] = kept[k] ? g.dest_out[i] : 0xFFFFFFFFu;
        len[k] = kept[k] ? cmb_len(g, i) : 0u;
    }
    for (uint32_t b = 0; b < p.P; b++) {
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < GRP_ITEMS; k++) mine += d[k] == b ? len[k] : 0u;
        const uint32_t incl = wave_incl_scan_u32(mine, lane);
        __syncthreads(); // wsum free
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        if (__syncthreads_or(mine != 0) == 0) continue; // nothing for this destination in the tile
        unsigned long long run = p.col_x[(uint64_t) blockIdx.x * p.ncol + b] + (incl - mine);
        for (int ww = 0; ww < w; ww++) run += wsum[ww];
#pragma unroll
        for (int k = 0; k < GRP_ITEMS; k++) {
            if (d[k] != b) continue;
            const uint64_t i = t0 + k;
            p.pos[i] = (uint32_t) run;
            const bool local = (int32_t) b == p.self;
            if (!local) {
                if (len[k] > (uint64_t) p.buf_size) raise_error(g.ctrl, MGS_ERR_RECORD_SIZE); // :86-88
                if (run + len[k] > (unsigned long long) p.buf_size) atomicMin(p.end, (unsigned long long) i);
                unsigned long long v; // the accumulator starts with the first record's own value
                const uint8_t *src = g.blob + grp_offset(g, i + 1) - 8;
                v = reinterpret_cast<const GrpU64 *>(src)->v;
                p.acc[i] = v;
            }
            run += len[k];
        }
    }
}

__device__ __forceinline__ unsigned long long cmb_apply(int op, unsigned long long a, unsigned long long b)
{
    return op == MGS_COMBINE_SUM_U64 ? a + b : op == MGS_COMBINE_MIN_U64 ? (a < b ? a : b) : (a > b ? a : b);
}

__device__ __forceinline__ void cmb_atomic(int op, unsigned long long *at, unsigned long long v)
{
    if (op == MGS_COMBINE_SUM_U64) atomicAdd(at, v);
    else if (op == MGS_COMBINE_MIN_U64) atomicMin(at, v);
    else atomicMax(at, v);
}

// Skewed keys would send millions of atomics to one address, and atomics on one
// address are served one after the other.  So a workgroup takes a contiguous
// stretch of the records and combines in LDS first: a small table keyed by the
// accumulator's index (a few probes; what finds no place goes to memory
// directly), flushed with one atomic per entry.  In front of that a wave
// combines the lanes that hit the same accumulator (up to CMB_PEEL distinct ones
// per step; the frequent ones are met early because they hold many lanes).
constexpr int CMB_PEEL = 4;
constexpr int CMB_LDS_LOG = 11, CMB_LDS_SLOTS = 1 << CMB_LDS_LOG, CMB_LDS_PROBES = 4;

__device__ __forceinline__ void cmb_lds_atomic(int op, unsigned long long *at, unsigned long long v)
{
    if (op == MGS_COMBINE_SUM_U64) atomicAdd(at, v);
    else if (op == MGS_COMBINE_MIN_U64) atomicMin(at, v);
    else atomicMax(at, v);
}

__global__ __launch_bounds__(GRP_THREADS) void k_cmb_reduce(CmbParams p)
{
    __shared__ uint32_t skey[CMB_LDS_SLOTS];
    __shared__ unsigned long long sval[CMB_LDS_SLOTS];
    const GrpParams &g = p.g;
    const uint64_t end = *p.end;
    const int lane = threadIdx.x & 63;
    const unsigned long long neutral = p.op == MGS_COMBINE_MIN_U64 ? ~0ull : 0ull;
    for (int e = threadIdx.x; e < CMB_LDS_SLOTS; e += GRP_THREADS) {
        skey[e] = GRP_EMPTY;
        sval[e] = neutral;
    }
    __syncthreads();
    // this workgroup's stretch, a multiple of the workgroup size
    const uint64_t per = ((end + gridDim.x - 1) / gridDim.x + GRP_THREADS - 1) / GRP_THREADS * GRP_THREADS;
    const uint64_t s0 = (uint64_t) blockIdx.x * per, s1 = min(end, s0 + per);
    for (uint64_t i0 = s0; i0 < s1; i0 += GRP_THREADS) { // whole waves stay together
        const uint64_t i = i0 + threadIdx.x;
        // kept records are written, not combined
        bool todo = i < s1 && !((g.first_bits[i / 64] >> (i & 63)) & 1ull);
        uint32_t first = 0;
        unsigned long long v = neutral;
        if (todo) {
            first = g.rank_table[g.slot_or_rank[i]];
            v = reinterpret_cast<const GrpU64 *>(g.blob + grp_offset(g, i + 1) - 8)->v;
        }
        bool speaker = false; // this lane carries a combined value to the table
        for (int it = 0; it < CMB_PEEL; it++) {
            const uint64_t rest = __ballot(todo);
            if (rest == 0) break;
            const int leader = __ffsll((unsigned long long) rest) - 1;
            const uint32_t f = (uint32_t) __shfl((int) first, leader);
            const bool in = todo && first == f;
            unsigned long long r = in ? v : neutral;
            for (int o = 1; o < 64; o <<= 1) r = cmb_apply(p.op, r, __shfl_xor(r, o));
            if (lane == leader) {
                v = r;
                speaker = true;
            }
            todo = todo && !in;
        }
        if (todo || speaker) {
            uint32_t s = (first * 0x9E3779B1u) >> (32 - CMB_LDS_LOG);
            bool placed = false;
            for (int pr = 0; pr < CMB_LDS_PROBES && !placed; pr++) {
                const uint32_t cur = atomicCAS(&skey[s], GRP_EMPTY, first);
                if (cur == GRP_EMPTY || cur == first) {
                    cmb_lds_atomic(p.op, &sval[s], v);
                    placed = true;
                }
                s = (s + 1) & (CMB_LDS_SLOTS - 1);
            }
            if (!placed) cmb_atomic(p.op, p.acc + first, v);
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < CMB_LDS_SLOTS; e += GRP_THREADS)
        if (skey[e] != GRP_EMPTY) cmb_atomic(p.op, p.acc + skey[e], sval[e]);
}

__global__ __launch_bounds__(GRP_THREADS) void k_cmb_emit(CmbParams p)
{
    const GrpParams &g = p.g;
    if (g.ctrl->head.error != 0) return;
    const uint64_t end = *p.end;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < end; i += stride) {
        if (!((g.first_bits[i / 64] >> (i & 63)) & 1ull)) continue;
        const uint32_t d = g.dest_out[i];
        const uint64_t o0 = grp_offset(g, i);
        const uint32_t len = cmb_len(g, i);
        const uint8_t *src = g.blob + o0;
        const bool local = (int32_t) d == p.self;
        if (local) {
            if ((uint64_t) p.pos[i] + len > (uint64_t) p.local_cap) {
                raise_error(g.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                continue;
            }
            uint8_t *dst = p.local_out + p.pos[i];
            for (uint32_t b = 0; b < len; b++) dst[b] = src[b];
            continue;
        }
        uint8_t *dst = p.image + (uint64_t) d * (uint64_t) p.buf_size + p.pos[i];
        const uint32_t klen = len - 8;
        for (uint32_t b = 0; b < klen; b++) dst[b] = src[b];
        reinterpret_cast<GrpU64 *>(dst + klen)->v = p.acc[i];
    }
}

// One workgroup: the columns summed up to the end of the round.
__global__ __launch_bounds__(GRP_THREADS) void k_cmb_final(CmbParams p)
{
    __shared__ unsigned long long sum[CMB_MAXCOL];
    const GrpParams &g = p.g;
    const uint64_t end = *p.end;
    const uint32_t te = (uint32_t) (end / GRP_TILE); // tiles before it are whole
    for (uint32_t c = threadIdx.x; c < p.ncol; c += GRP_THREADS) sum[c] = p.col_x[(uint64_t) te * p.ncol + c];
    __syncthreads();
    for (uint64_t i = (uint64_t) te * GRP_TILE + threadIdx.x; i < end; i += GRP_THREADS) {
        if (!((g.first_bits[i / 64] >> (i & 63)) & 1ull)) continue;
        const uint32_t d = g.dest_out[i];
        atomicAdd(&sum[d], (unsigned long long) cmb_len(g, i));
        atomicAdd(&sum[p.P + ((int32_t) d == p.self ? 0u : 1u)], 1ull);
    }
    __syncthreads();
    if (threadIdx.x < p.P) {
        const bool local = (int32_t) threadIdx.x == p.self;
        p.round->send_offset[threadIdx.x] = local ? 0 : (int64_t) sum[threadIdx.x];
        if (local) p.round->local_bytes = (int64_t) sum[threadIdx.x];
    }
    if (threadIdx.x == 0) {
        p.round->next = p.start + end;
        p.round->local_records = sum[p.P];
        p.round->packed_records = sum[p.P + 1];
        if (p.self < 0) p.round->local_bytes = 0;
    }
}

} // namespace mgs
#endif
