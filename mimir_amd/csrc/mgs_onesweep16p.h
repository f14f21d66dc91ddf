// k_onesweep16p: the LDS-image one-sweep kernel with the scan wave's code and
// the record-moving waves' code in two separate loops.
//
// Same algorithm and the same two barriers per tile as k_onesweep16s (see
// mgs_onesweep16s.h); the only difference is structural: the scan wave never
// executes the workers' loop body and vice versa, so the registers of a wide
// chained-scan poll (LOOK status loads in flight) and the registers holding a
// tile of records are never live in the same code and both fit under 128.
// Both loops execute exactly one barrier A and one barrier B per tile plus the
// prologue and epilogue barriers, so the workgroup barrier counts match.
#ifndef MGS_ONESWEEP16P_H
#define MGS_ONESWEEP16P_H

#include "mgs_onesweep16s.h"

namespace mgs {

template <int THREADS, int ITEMS, int VAR, int MINW, int LOOK>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweep16p(OsParams p)
{
    using G = OssGeom<THREADS, ITEMS>;
    constexpr int WORKERS = G::WORKERS, TILE = G::TILE;
    constexpr int WTHREADS = WORKERS * 64;
    constexpr int CNT = WORKERS * OS_MAXP;
    __shared__ __attribute__((aligned(16))) uint8_t smem[G::BYTES];
    uint4 *srec = reinterpret_cast<uint4 *>(smem);
    uint32_t *wcnt = reinterpret_cast<uint32_t *>(smem + G::REC);
    volatile uint32_t *wslot = reinterpret_cast<uint32_t *>(smem + G::REC + G::CNT);
    uint32_t *binstart2 = reinterpret_cast<uint32_t *>(smem + G::REC + G::CNT + G::SLOT);
    unsigned long long *gpos2 = reinterpret_cast<unsigned long long *>(
        smem + G::REC + G::CNT + G::SLOT + G::BINSTART);
    unsigned long long *pbase = gpos2 + 2 * OS_MAXP;
    unsigned long long *pcap = pbase + OS_MAXP;
    volatile uint32_t *misc = reinterpret_cast<uint32_t *>(pcap + OS_MAXP);

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr bool SPEC = (VAR & OSV_SPEC8) != 0;
    const uint32_t P = SPEC ? 8u : p.P;
    const uint32_t nbits = SPEC ? 3u : p.nbits;
    const uint32_t klen = SPEC ? 8u : p.klen;

    if (tid < OS_MAXP) {
        const bool live = (uint32_t) tid < P;
        pbase[tid] = live ? p.ctrl->head.part_base[tid] : 0ull;
        pcap[tid] = live ? (unsigned long long) (p.ctrl->head.part_cap[tid] >> 4) : 0ull;
    }
    for (int j = tid; j < 2 * CNT; j += THREADS) wcnt[j] = 0;
    if (tid == 0) misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
    __syncthreads();
    uint32_t tile = misc[0];
    int par = 0;

    // ------------------------------------------------------------------ scan wave
    if (w == WORKERS) {
        OsWalkStats ws;
        while (tile < p.ntiles) {
            const uint32_t *wc = wcnt + par * CNT;
            __syncthreads(); // A: counts(tile) complete
            uint32_t tot = 0; // lane = destination
#pragma unroll
            for (int ww = 0; ww < WORKERS; ww++) tot += wc[ww * OS_MAXP + lane];
            unsigned long long *gp = gpos2 + par * OS_MAXP;
            if ((uint32_t) lane < P) { // publish this tile's counts before anything else
                unsigned long long *st = p.status + (uint64_t) tile * P + lane;
                if (tile == 0) {
                    const unsigned long long start =
                        (unsigned long long) (p.ctrl->head.part_init[lane] >> 4);
                    gp[lane] = start;
                    os_store_status(st, OS_FLAG_INCL | (start + tot));
                }
                else {
                    os_store_status(st, OS_FLAG_AGG | tot);
                }
            }
            // draw the next tile now: a short time between a tile's ticket and the
            // publication of its counts keeps neighbouring tiles close in time
            if (lane == 0) misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
            // the other copy of the counts was last read before the previous barrier B
            for (int j = lane; j < CNT; j += 64) wcnt[(par ^ 1) * CNT + j] = 0;
            __syncthreads(); // B
            const uint32_t next = misc[1];
            if (tile != 0) {
                unsigned long long acc;
                if (VAR & OSV_FAKE_CHAIN) acc = (unsigned long long) tile * (TILE / P);
                else acc = os_chain_walk<VAR | OSV_WIDE_FIRST, LOOK>(p, tile, lane, &ws);
                if ((uint32_t) lane < P) {
                    if (!(VAR & OSV_FAKE_CHAIN))
                        os_store_status(p.status + (uint64_t) tile * P + lane,
                                        OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
                    gp[lane] = acc;
                }
            }
            tile = next;
            par ^= 1;
        }
        __syncthreads(); // matches the workers' last barrier
        if ((VAR & OSV_STATS) && lane == 0) ws.flush(p.ctrl);
        return;
    }

    // ------------------------------------------------------------------- workers
    uint32_t prev = 0xFFFFFFFFu; // tile whose image waits in LDS to be stored
    uint4 r[ITEMS];
    if (tile < p.ntiles) os_load_tile<ITEMS, VAR>(p, tile, TILE, w, lane, r);

    // Store the LDS image of a tile: consecutive lanes write consecutive
    // 16-byte slots of one partition run.
    auto store_image = [&](uint32_t t, int pr) {
        const uint32_t *binstart = binstart2 + pr * (OS_MAXP + 4);
        const unsigned long long *gpos = gpos2 + pr * OS_MAXP;
        const uint32_t total = binstart[64];
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t s = (uint32_t) k * WTHREADS + tid;
            if (s < total) {
                uint32_t b = 0;
                for (uint32_t step = 1u << (nbits - 1); step > 0; step >>= 1) {
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
        if (overflow) atomicMin(&p.ctrl->head.cut_tile, t);
    };

    while (tile < p.ntiles) {
        uint32_t *wc = wcnt + par * CNT;
        uint32_t meta[ITEMS]; // dest | rank<<8 | count<<16 | valid<<31
        const uint64_t idx0 = (uint64_t) tile * TILE + (uint32_t) w * (ITEMS * 64) + lane;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint64_t i = idx0 + (uint32_t) k * 64;
            const bool valid = i < p.n;
            const uint32_t h = hash_rec16(r[k].x, r[k].y, r[k].z, r[k].w, klen);
            const uint32_t d = SPEC ? (h & 7u) : fastmod(h, p.fm);
            if (!SPEC && p.dest_out && valid) p.dest_out[i] = (uint8_t) d;
            uint64_t m = __ballot(valid);
#pragma unroll
            for (uint32_t bit = 0; bit < nbits; bit++) {
                const bool one = (d >> bit) & 1u;
                const uint64_t bk = __ballot(one);
                m &= one ? bk : ~bk;
            }
            const uint32_t rank = lanes_below(m);
            const uint32_t cnt = (uint32_t) __popcll(m);
            if (valid && rank == 0) atomicAdd(&wc[w * OS_MAXP + d], cnt);
            meta[k] = d | (rank << 8) | (cnt << 16) | (valid ? 0x80000000u : 0u);
        }
        __syncthreads(); // A: counts(tile) complete; image(prev) and position(prev) ready

        uint32_t tot = 0, below = 0; // lane = destination
#pragma unroll
        for (int ww = 0; ww < WORKERS; ww++) {
            const uint32_t v = wc[ww * OS_MAXP + lane];
            tot += v;
            if (ww < w) below += v;
        }
        const uint32_t incl = wave_incl_scan_u32(tot, lane);
        wslot[w * OS_MAXP + lane] = incl - tot + below;
        if (w == 0) {
            uint32_t *binstart = binstart2 + par * (OS_MAXP + 4);
            binstart[lane] = incl - tot;
            if (lane == 63) binstart[64] = incl;
        }
        if (prev != 0xFFFFFFFFu) store_image(prev, par ^ 1);
        __syncthreads(); // B: every slot of image(prev) has been read, the image is free
        const uint32_t next = misc[1];

#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t d = meta[k] & 0xffu, rank = (meta[k] >> 8) & 0xffu;
            const uint32_t cnt = (meta[k] >> 16) & 0xffu;
            if ((meta[k] >> 31) != 0) {
                const uint32_t base = wslot[w * OS_MAXP + d];
                if (rank == 0) wslot[w * OS_MAXP + d] = base + cnt;
                srec[base + rank] = r[k];
            }
        }
        // the registers are free again: fetch the next tile
        if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        prev = tile;
        tile = next;
        par ^= 1;
    }
    __syncthreads(); // image(prev) and position(prev) ready
    if (prev != 0xFFFFFFFFu) store_image(prev, par ^ 1);
}

} // namespace mgs
#endif
