// Single-pass stable hash-partition of 16-byte records, direct-scatter form:
// no LDS image of the tile.  Each lane computes the exact position of its
// records (global start of the tile in the partition + records of earlier
// waves + records of earlier rows/lanes of its wave) and stores them straight
// from registers.  Within one wave-wide store the lanes of one destination
// write consecutive 16-byte slots, and consecutive rows / waves / tiles
// continue the same runs, so the partial cache lines meet again in L2 before
// they are written back.  LDS holds only the small per-destination tables,
// which leaves room for many workgroups per CU to cover each other's waits.
#ifndef MGS_ONESWEEP16D_H
#define MGS_ONESWEEP16D_H

#include "mgs_onesweep16s.h"

namespace mgs {

template <int THREADS, int ITEMS, int VAR, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweep16d(OsParams p)
{
    constexpr int WAVES = THREADS / 64, TILE = THREADS * ITEMS;
    __shared__ __attribute__((aligned(16))) uint32_t wcnt[2 * WAVES * OS_MAXP];
    __shared__ volatile uint32_t wslot[WAVES * OS_MAXP];
    __shared__ unsigned long long gpos[OS_MAXP];
    __shared__ unsigned long long pbase[OS_MAXP];
    __shared__ unsigned long long pcap[OS_MAXP];
    __shared__ volatile uint32_t misc[4];

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
        if (tid == THREADS - 64) misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
        uint32_t *wc = wcnt + par * (WAVES * OS_MAXP);
        const uint64_t idx0 = (uint64_t) tile * TILE + (uint32_t) w * (ITEMS * 64) + lane;

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
        __syncthreads(); // S1: counts complete, misc[1] set
        const uint32_t next = misc[1];

        uint32_t tot = 0, below = 0; // lane = destination
#pragma unroll
        for (int ww = 0; ww < WAVES; ww++) {
            const uint32_t v = wc[ww * OS_MAXP + lane];
            tot += v;
            if (ww < w) below += v;
        }
        wslot[w * OS_MAXP + lane] = below;
        if (w == 0 && (uint32_t) lane < P) {
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
        for (int j = tid; j < WAVES * OS_MAXP; j += THREADS)
            wcnt[(par ^ 1) * (WAVES * OS_MAXP) + j] = 0;

        // position of every record inside its destination's share of the tile
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t d = meta[k] & 0xffu, rank = (meta[k] >> 8) & 0xffu;
            const uint32_t cnt = (meta[k] >> 16) & 0xffu;
            const bool valid = (meta[k] >> 31) != 0;
            uint32_t off = 0;
            if (valid) {
                const uint32_t base = wslot[w * OS_MAXP + d];
                if (rank == 0) wslot[w * OS_MAXP + d] = base + cnt;
                off = base + rank;
            }
            meta[k] = d | (off << 8) | (valid ? 0x80000000u : 0u); // off < 2^20
        }

        if (w == 0 && tile != 0) {
            unsigned long long acc;
            if (VAR & OSV_FAKE_CHAIN) acc = (unsigned long long) tile * (TILE / P);
            else acc = os_chain_walk<VAR>(p, tile, lane);
            if ((uint32_t) lane < P) {
                if (!(VAR & OSV_FAKE_CHAIN))
                    os_store_status(p.status + (uint64_t) tile * P + lane,
                                    OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
                gpos[lane] = acc;
            }
        }
        __syncthreads(); // S2: gpos ready

        bool overflow = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t d = meta[k] & 0xffu, off = (meta[k] >> 8) & 0xfffffu;
            if ((meta[k] >> 31) != 0) {
                const unsigned long long pos = gpos[d] + off;
                if (pos < pcap[d]) {
                    reinterpret_cast<uint4 *>(pbase[d])[pos] = r[k];
                }
                else if ((int32_t) d == p.self) {
                    raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                }
                else {
                    overflow = true;
                }
            }
        }
        if (overflow) atomicMin(&p.ctrl->head.cut_tile, tile);

        if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        tile = next;
        par ^= 1;
        // gpos of this trip is read before the next S1, and written again only after it
    }
}

} // namespace mgs
#endif
