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

constexpr int OS_THREADS = 512;
constexpr int OS_WAVES = OS_THREADS / 64;
constexpr int OS_ITEMS = 8;
constexpr int OS_TILE = OS_THREADS * OS_ITEMS; // 4096 records = 64 KiB
constexpr int OS_MAXP = 64;
constexpr uint64_t OS_FLAG_AGG = 1ull << 62;  // value = this tile's count
constexpr uint64_t OS_FLAG_INCL = 2ull << 62; // value = count of tiles 0..t (+ initial fill)
constexpr uint64_t OS_VALUE_MASK = (1ull << 62) - 1;
constexpr uint32_t OS_SPIN_LIMIT = 1u << 20; // polls with nothing new before giving up
constexpr int OS_LOOK = 4;                   // predecessor entries fetched per poll

// LDS carve-up (bytes): one array, 16-byte aligned pieces
constexpr int OS_LDS_REC = OS_TILE * 16;                  // tile image
constexpr int OS_LDS_CNT = OS_WAVES * OS_MAXP * 4;        // per (wave,dest) counts
constexpr int OS_LDS_SLOT = OS_WAVES * OS_MAXP * 4;       // per (wave,dest) running slot
constexpr int OS_LDS_BINSTART = (OS_MAXP + 4) * 4;        // dest -> first slot (+ total)
constexpr int OS_LDS_GPOS = OS_MAXP * 8;                  // dest -> global record position
constexpr int OS_LDS_PBASE = OS_MAXP * 8;                 // dest -> partition address
constexpr int OS_LDS_PCAP = OS_MAXP * 8;                  // dest -> capacity in records
constexpr int OS_LDS_MISC = 16;
constexpr int OS_LDS_BYTES = OS_LDS_REC + OS_LDS_CNT + OS_LDS_SLOT + OS_LDS_BINSTART
                             + OS_LDS_GPOS + OS_LDS_PBASE + OS_LDS_PCAP + OS_LDS_MISC;

struct OsParams {
    const uint4 *rec; // record 0 of this call
    uint64_t n;       // records in this call
    uint64_t first;   // absolute index of record 0 (for the result only)
    uint32_t P;
    int32_t self;
    uint32_t klen;  // key bytes (1..16)
    uint32_t nbits; // ceil(log2(P)), >= 1
    FastMod fm;
    uint32_t ntiles;
    unsigned long long *status; // [ntiles][P]
    uint8_t *dest_out;          // optional
    Ctrl *ctrl;
};

__device__ __forceinline__ unsigned long long os_load_status(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void os_store_status(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(OS_THREADS, 4) void k_onesweep16(OsParams p)
{
    __shared__ __attribute__((aligned(16))) uint8_t smem[OS_LDS_BYTES];
    uint4 *srec = reinterpret_cast<uint4 *>(smem);
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + OS_LDS_REC);
    volatile uint32_t *wslot = reinterpret_cast<uint32_t *>(smem + OS_LDS_REC + OS_LDS_CNT);
    uint32_t *binstart =
        reinterpret_cast<uint32_t *>(smem + OS_LDS_REC + OS_LDS_CNT + OS_LDS_SLOT);
    unsigned long long *gpos = reinterpret_cast<unsigned long long *>(
        smem + OS_LDS_REC + OS_LDS_CNT + OS_LDS_SLOT + OS_LDS_BINSTART);
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

    for (;;) {
        if (tid == 0) misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
        wcnt[tid] = 0; // OS_WAVES*OS_MAXP == OS_THREADS entries
        __syncthreads();
        const uint32_t tile = misc[0];
        if (tile >= p.ntiles) break;

        const uint64_t idx0 = (uint64_t) tile * OS_TILE + (uint32_t) w * (OS_ITEMS * 64) + lane;

        // ---- load: 8 independent 16-byte loads per lane --------------------
        uint4 r[OS_ITEMS];
#pragma unroll
        for (int k = 0; k < OS_ITEMS; k++) {
            const uint64_t i = idx0 + (uint32_t) k * 64;
            r[k] = (i < p.n) ? p.rec[i] : make_uint4(0, 0, 0, 0);
        }

        // ---- hash, destination, rank within row -----------------------------
        uint32_t meta[OS_ITEMS]; // dest | rank<<8 | count<<16 | valid<<31
#pragma unroll
        for (int k = 0; k < OS_ITEMS; k++) {
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
            if (valid && rank == 0) atomicAdd(&wcnt[w * OS_MAXP + d], cnt);
            meta[k] = d | (rank << 8) | (cnt << 16) | (valid ? 0x80000000u : 0u);
        }
        __syncthreads(); // S1: wcnt complete

        // ---- per-destination totals, slot bases (every wave, redundantly) ---
        uint32_t tot = 0, below = 0; // lane = destination
#pragma unroll
        for (int ww = 0; ww < OS_WAVES; ww++) {
            const uint32_t v = wcnt[ww * OS_MAXP + lane];
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

        // ---- reorder into the LDS image -------------------------------------
#pragma unroll
        for (int k = 0; k < OS_ITEMS; k++) {
            const uint32_t d = meta[k] & 0xffu, rank = (meta[k] >> 8) & 0xffu;
            const uint32_t cnt = (meta[k] >> 16) & 0xffu;
            const bool valid = (meta[k] >> 31) != 0;
            if (valid) {
                const uint32_t base = wslot[w * OS_MAXP + d];
                if (rank == 0) wslot[w * OS_MAXP + d] = base + cnt;
                srec[base + rank] = r[k];
            }
        }

        // ---- chained scan: where does this tile start in every partition? ---
        if (w == 0 && tile != 0) {
            const bool live = (uint32_t) lane < P;
            unsigned long long excl = 0;
            int64_t look = (int64_t) tile - 1;
            bool done = !live;
            uint32_t idle = 0;
            while (__any(!done)) {
                unsigned long long v[OS_LOOK];
#pragma unroll
                for (int j = 0; j < OS_LOOK; j++) {
                    const int64_t t = look - j;
                    v[j] = (!done && t >= 0) ? os_load_status(p.status + (uint64_t) t * P + lane)
                                             : 0ull;
                }
                if (!done) {
                    int used = 0;
#pragma unroll
                    for (int j = 0; j < OS_LOOK; j++) {
                        if (used != j) continue;       // stopped at an earlier entry
                        const unsigned long long f = v[j] >> 62;
                        if (f == 0 || look - j < 0) continue; // not published yet
                        excl += v[j] & OS_VALUE_MASK;
                        used = j + 1;
                        if (f == 2) {
                            done = true;
                            break;
                        }
                    }
                    look -= used;
                    if (!done) {
                        if (used == 0) {
                            // Bounded wait: give up after OS_SPIN_LIMIT empty polls, and
                            // leave at once when any other tile has already given up.
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
                            __builtin_amdgcn_s_sleep(4);
                        }
                        else {
                            idle = 0;
                        }
                    }
                }
            }
            if (live) {
                os_store_status(p.status + (uint64_t) tile * P + lane,
                                OS_FLAG_INCL | ((excl + tot) & OS_VALUE_MASK));
                gpos[lane] = excl;
            }
        }
        __syncthreads(); // S2: LDS image and gpos complete

        // ---- stream the image out, one partition run after another ----------
        const uint32_t total = binstart[64];
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < OS_ITEMS; k++) {
            const uint32_t s = (uint32_t) k * OS_THREADS + tid;
            if (s < total) {
                uint32_t b = 0;
                for (uint32_t step = 1u << (p.nbits - 1); step > 0; step >>= 1) {
                    const uint32_t nb = b + step;
                    if (nb < P && binstart[nb] <= s) b = nb;
                }
                const unsigned long long pos = gpos[b] + (s - binstart[b]);
                if (pos < pcap[b]) {
                    reinterpret_cast<uint4 *>(pbase[b])[pos] = srec[s];
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
        // the next trip's first barrier keeps the image intact until every lane is done with it
    }
}

// One wave.  Computes send_offset[] and the counters at the end of the round:
// either everything (no record overflowed) or everything before the first
// record that did not fit, which lies in the lowest tile that reported one.
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
        const uint64_t t0 = (uint64_t) cut * OS_TILE;
        bool found = false;
        for (int row = 0; row < OS_TILE / 64 && !found; row++) {
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

} // namespace mgs
#endif
