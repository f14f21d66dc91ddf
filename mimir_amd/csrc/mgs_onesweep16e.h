// Single-pass stable hash-partition of 16-byte fixed-width records, nranks <= 64:
// direct scatter from registers + a dedicated scan wave + stores deferred by
// one tile.  Production kernel for 16-byte layouts.
//
//   tile     WORKERS*64*ITEMS consecutive records.  Worker wave w owns ITEMS
//            rows of 64 consecutive records: every global load is one fully
//            coalesced 1 KiB access (global_load_dwordx4).
//   hash     lookup3 on the key words in registers, dest = hash % nranks.
//   rank     per row, a match-any over the destination bits (log2(P) ballots)
//            gives each lane the lanes sharing its destination: rank in row =
//            popcount below, row count = popcount.  Row counts are summed per
//            (wave, dest) in LDS; an exclusive scan over the waves and a
//            running count per wave turn them into each record's position
//            inside its destination's share of the tile.
//   chain    the workgroup's last wave moves no records.  It publishes the
//            tile's per-destination counts in a global status array, walks
//            back over the predecessors' entries (aggregate / inclusive-prefix
//            words: one naturally aligned 8-byte device-scope store each, the
//            data is the flag) up to an inclusive prefix, publishes its own
//            inclusive prefix and leaves the tile's start position per
//            destination in LDS.  Tiles are handed out by an atomic ticket, so
//            every predecessor of a running tile is itself running or done.
//   store    each lane stores its records straight from registers to
//            partition[dest] + start + position: within one wave-wide store the
//            lanes of a destination write consecutive 16-byte slots, and the
//            following rows / waves / tiles continue those runs, so partial
//            cache lines are completed in L2 before write-back.
//
// The workers never wait for the chain: they keep TWO tiles in registers and
// store tile i-1 while the scan wave resolves tile i,
//
//     workers                                   scan wave
//     hash(i)                                   ticket(i+1)
//     ---- barrier A(i): counts(i) complete, start(i-1) in LDS ----
//     positions(i); store(i-1); load(i+1)       publish counts(i); walk back;
//                                               publish prefix(i); start(i) -> LDS
//
// and two workgroups per CU cover each other's load latency.
#ifndef MGS_ONESWEEP16E_H
#define MGS_ONESWEEP16E_H

#include "mgs_onesweep16s.h"

namespace mgs {

constexpr int OSE_LOOKMAX = 16; // status loads of a widened poll

template <int THREADS, int ITEMS, int VAR, int MINW>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweep16e(OsParams p)
{
    constexpr int WORKERS = THREADS / 64 - 1, TILE = WORKERS * 64 * ITEMS;
    constexpr int CNT = WORKERS * OS_MAXP;
    __shared__ __attribute__((aligned(16))) uint32_t wcnt[3 * CNT]; // counts of tiles i, i+1, i+2
    __shared__ volatile uint32_t wslot[CNT];                        // running position per (wave, dest)
    __shared__ unsigned long long gpos2[2 * OS_MAXP];               // tile start per dest, two tiles
    __shared__ unsigned long long pbase[OS_MAXP];
    __shared__ unsigned long long pcap[OS_MAXP];
    __shared__ volatile uint32_t misc[4];

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const bool scanner = (w == WORKERS);
    const uint32_t P = p.P;

    if (tid < OS_MAXP) {
        const bool live = (uint32_t) tid < P;
        pbase[tid] = live ? p.ctrl->head.part_base[tid] : 0ull;
        pcap[tid] = live ? (unsigned long long) (p.ctrl->head.part_cap[tid] >> 4) : 0ull;
    }
    for (int j = tid; j < 3 * CNT; j += THREADS) wcnt[j] = 0;
    if (tid == 0) misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
    __syncthreads();

    uint32_t tile = misc[0];
    uint32_t it = 0; // trip counter: selects the LDS copies

    // ------------------------------------------------------------------ scan wave
    if (scanner) {
        while (tile < p.ntiles) {
            const uint32_t *wc = wcnt + (it % 3u) * CNT;
            volatile uint32_t *nextslot = misc + 1 + (it & 1u);
            if (lane == 0) *nextslot = atomicAdd(&p.ctrl->head.ticket, 1u);
            __syncthreads(); // A: counts(tile) complete
            const uint32_t next = *nextslot;
            uint32_t tot = 0; // lane = destination
#pragma unroll
            for (int ww = 0; ww < WORKERS; ww++) tot += wc[ww * OS_MAXP + lane];
            unsigned long long *gp = gpos2 + (it & 1u) * OS_MAXP;
            if (tile == 0) {
                if ((uint32_t) lane < P) {
                    const unsigned long long start =
                        (unsigned long long) (p.ctrl->head.part_init[lane] >> 4);
                    gp[lane] = start;
                    os_store_status(p.status + lane, OS_FLAG_INCL | (start + tot));
                }
            }
            else if ((uint32_t) lane < P) {
                os_store_status(p.status + (uint64_t) tile * P + lane, OS_FLAG_AGG | tot);
            }
            // the copy used two trips from now was last read before this barrier A
            for (int j = lane; j < CNT; j += 64) wcnt[((it + 2u) % 3u) * CNT + j] = 0;
            if (tile != 0) {
                unsigned long long acc;
                if (VAR & OSV_FAKE_CHAIN) acc = (unsigned long long) tile * (TILE / P);
                else acc = os_chain_walk<VAR, OSE_LOOKMAX>(p, tile, lane);
                if ((uint32_t) lane < P) {
                    if (!(VAR & OSV_FAKE_CHAIN))
                        os_store_status(p.status + (uint64_t) tile * P + lane,
                                        OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
                    gp[lane] = acc;
                }
            }
            tile = next;
            it++;
        }
        __syncthreads(); // matches the workers' last barrier
        return;
    }

    // ------------------------------------------------------------------- workers
    uint32_t prev = 0xFFFFFFFFu; // tile that waits in registers for its start position
    uint4 rX[ITEMS], rY[ITEMS];
    uint32_t mX[ITEMS], mY[ITEMS]; // dest | position<<8 | valid<<31
    if (tile < p.ntiles) os_load_tile<ITEMS, VAR>(p, tile, TILE, w, lane, rX);

    auto store_tile = [&](uint32_t t, const unsigned long long *gpos, uint4 (&r)[ITEMS],
                          uint32_t (&m)[ITEMS]) {
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t d = m[k] & 0xffu, off = (m[k] >> 8) & 0xfffffu;
            if ((m[k] >> 31) != 0) {
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
        if (overflow) atomicMin(&p.ctrl->head.cut_tile, t);
    };

    // One trip: `rc/mc` hold the current tile, `rp/mp` the previous one.
    auto trip = [&](uint4 (&rc)[ITEMS], uint32_t (&mc)[ITEMS], uint4 (&rp)[ITEMS],
                    uint32_t (&mp)[ITEMS]) {
        uint32_t *wc = wcnt + (it % 3u) * CNT;
        volatile uint32_t *nextslot = misc + 1 + (it & 1u);
        const uint64_t idx0 = (uint64_t) tile * TILE + (uint32_t) w * (ITEMS * 64) + lane;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint64_t i = idx0 + (uint32_t) k * 64;
            const bool valid = i < p.n;
            const uint32_t h = hash_rec16(rc[k].x, rc[k].y, rc[k].z, rc[k].w, p.klen);
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
            mc[k] = d | (rank << 8) | (cnt << 16) | (valid ? 0x80000000u : 0u);
        }
        __syncthreads(); // A: counts(tile) complete; start(prev) is in LDS
        const uint32_t next = *nextslot;

        uint32_t below = 0; // lane = destination: records of earlier waves of this tile
#pragma unroll
        for (int ww = 0; ww < WORKERS; ww++) {
            const uint32_t v = wc[ww * OS_MAXP + lane];
            if (ww < w) below += v;
        }
        wslot[w * OS_MAXP + lane] = below;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t d = mc[k] & 0xffu, rank = (mc[k] >> 8) & 0xffu;
            const uint32_t cnt = (mc[k] >> 16) & 0xffu;
            const bool valid = (mc[k] >> 31) != 0;
            uint32_t off = 0;
            if (valid) {
                const uint32_t base = wslot[w * OS_MAXP + d];
                if (rank == 0) wslot[w * OS_MAXP + d] = base + cnt;
                off = base + rank;
            }
            mc[k] = d | (off << 8) | (valid ? 0x80000000u : 0u);
        }
        if (prev != 0xFFFFFFFFu) store_tile(prev, gpos2 + ((it & 1u) ^ 1u) * OS_MAXP, rp, mp);
        if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, rp);
        prev = tile;
        tile = next;
        it++;
    };

    bool pending_in_x = false;
    while (tile < p.ntiles) {
        trip(rX, mX, rY, mY);
        pending_in_x = true;
        if (!(tile < p.ntiles)) break;
        trip(rY, mY, rX, mX);
        pending_in_x = false;
    }
    __syncthreads(); // start(prev) is in LDS
    if (prev != 0xFFFFFFFFu) {
        const unsigned long long *gp = gpos2 + ((it & 1u) ^ 1u) * OS_MAXP;
        if (pending_in_x) store_tile(prev, gp, rX, mX);
        else store_tile(prev, gp, rY, mY);
    }
}

} // namespace mgs
#endif
