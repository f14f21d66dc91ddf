// k_onesweep16f: one-sweep stable hash-partition of 16-byte records with
// direct scatter, a dedicated scan wave and stores deferred by one tile, the
// waiting tile parked in LDS.
//
//   workers (WORKERS waves)                    scan wave
//   hash(i) in registers                       ticket(i+1)
//   ---- barrier A(i): counts(i) complete; start(i-1) is in LDS ----
//   positions(i)                               publish counts(i)
//   per row: read parked record(i-1) -> store  walk back (LOOK loads in flight)
//            to partition[dest]+start+pos;     publish prefix(i); start(i) -> LDS
//            park record(i) in the same slot
//   load(i+1)
//
// Each lane parks and reads back only its own slots (slot = row*WTHREADS+tid,
// consecutive lanes 16 bytes apart: conflict-free b128 accesses), so the
// parking needs no barrier; one barrier per tile in all.  Stores go straight
// to the record's final position: the lanes of one destination write
// consecutive 16-byte slots within a wave-wide store, and the following rows /
// waves / tiles continue those runs, so partial lines meet again in L2.
#ifndef MGS_ONESWEEP16F_H
#define MGS_ONESWEEP16F_H

#include "mgs_onesweep16w.h"

namespace mgs {

template <int THREADS, int ITEMS, int VAR, int MINW, int LOOK>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweep16f(OsParams p)
{
    constexpr int WORKERS = THREADS / 64 - 1, TILE = WORKERS * 64 * ITEMS;
    constexpr int WTHREADS = WORKERS * 64;
    constexpr int CNT = WORKERS * OS_MAXP + 64; // per (wave, dest) counts, then the arrival counter
    constexpr int ARRIVE = WORKERS * OS_MAXP;
    __shared__ __attribute__((aligned(16))) uint4 park[TILE];       // tile i-1, in load order
    __shared__ __attribute__((aligned(16))) uint32_t wcnt[3 * CNT]; // counts of tiles i, i+1, i+2
    __shared__ volatile uint32_t wslot[WORKERS * OS_MAXP];                        // running position per (wave, dest)
    __shared__ unsigned long long gpos2[2 * OS_MAXP];               // tile start per dest, two tiles
    __shared__ unsigned long long pbase[OS_MAXP];
    __shared__ unsigned long long pcap[OS_MAXP];
    __shared__ volatile uint32_t misc[8]; // [0] first tile, [2+j] next tile of trip parity j, [4+j] its trip tag

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr bool S_NB = (VAR & OSV_NB3) != 0, S_KL = (VAR & OSV_KLEN8) != 0;
    const uint32_t P = S_NB ? 8u : p.P;
    const uint32_t nbits = S_NB ? 3u : p.nbits;
    const uint32_t klen = S_KL ? 8u : p.klen;

    if (tid < OS_MAXP) {
        const bool live = (uint32_t) tid < P;
        pbase[tid] = live ? p.ctrl->head.part_base[tid] : 0ull;
        pcap[tid] = live ? (unsigned long long) (p.ctrl->head.part_cap[tid] >> 4) : 0ull;
    }
    for (int j = tid; j < 3 * CNT; j += THREADS) wcnt[j] = 0;
    if (tid == 0) {
        misc[4] = 0;
        misc[5] = 0;
        misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
    }
    __syncthreads();

    uint32_t tile = misc[0];
    uint32_t it = 0; // trip counter: selects the LDS copies

    // ------------------------------------------------------------------ scan wave
    if (w == WORKERS) {
        OsWalkStats ws;
        while (tile < p.ntiles) {
            const uint32_t *wc = wcnt + (it % 3u) * CNT;
            __syncthreads(); // A: counts(tile) complete
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
            else if (!(VAR & (OSV_EARLY_AGG | OSV_WINDOW)) && (uint32_t) lane < P) {
                os_store_status(p.status + (uint64_t) tile * P + lane, OS_FLAG_AGG | tot);
            }
            if (VAR & OSV_WINDOW) os_publish_agg16(p, tile, lane, tot);
            // Draw the next tile at a fixed point of the workers' trip (right after barrier
            // A), so that every tile's counts are published about one trip after its
            // ticket: neighbouring tiles then publish close together and no walk waits long
            // for a late predecessor.  The workers pick it up from LDS (tagged by trip).
            uint32_t next = 0;
            if (lane == 0) {
                next = atomicAdd(&p.ctrl->head.ticket, 1u);
                misc[2 + (it & 1u)] = next;
                misc[4 + (it & 1u)] = it + 1u;
            }
            next = __builtin_amdgcn_readfirstlane(next);
            // the copy used two trips from now was last read before this barrier A
            for (int j = lane; j < CNT; j += 64) wcnt[((it + 2u) % 3u) * CNT + j] = 0;
            if (tile != 0) {
                unsigned long long acc;
                if (VAR & OSV_FAKE_CHAIN) acc = (unsigned long long) tile * (TILE / P);
                else {
                    // The nearest predecessors published their counts only moments ago and
                    // the stores take a while to become visible: a poll issued at once would
                    // stop at the first of them.  Waiting first makes the first poll count.
                    constexpr int DELAY = (VAR >> OSV_DELAY_SHIFT) & 3;
                    if (DELAY == 1) __builtin_amdgcn_s_sleep(24);
                    if (DELAY == 2) __builtin_amdgcn_s_sleep(48);
                    if (DELAY == 3) __builtin_amdgcn_s_sleep(80);
                    if (VAR & OSV_WINDOW) acc = os_window_walk<VAR, LOOK>(p, tile, lane, &ws);
                    else acc = os_chain_walk<VAR | OSV_WIDE_FIRST, LOOK>(p, tile, lane, &ws);
                }
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
        if ((VAR & OSV_STATS) && lane == 0) ws.flush(p.ctrl);
        return;
    }

    // ------------------------------------------------------------------- workers
    uint32_t prev = 0xFFFFFFFFu; // tile parked in LDS, waiting for its start positions
    uint4 r[ITEMS];
    uint32_t mc[ITEMS], mp[ITEMS]; // dest | position<<8 | valid<<31 of the current / parked tile
#pragma unroll
    for (int k = 0; k < ITEMS; k++) mp[k] = 0;
    if (tile < p.ntiles) os_load_tile<ITEMS, VAR>(p, tile, TILE, w, lane, r);
    bool overflow_prev = false;

    while (tile < p.ntiles) {
        uint32_t *wc = wcnt + (it % 3u) * CNT;
        const uint64_t idx0 = (uint64_t) tile * TILE + (uint32_t) w * (ITEMS * 64) + lane;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            const uint64_t i = idx0 + (uint32_t) k * 64;
            const bool valid = i < p.n;
            const uint32_t h = hash_rec16(r[k].x, r[k].y, r[k].z, r[k].w, klen);
            const uint32_t d = S_NB ? (h & 7u) : fastmod(h, p.fm);
            if (p.dest_out && valid) p.dest_out[i] = (uint8_t) d;
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
            mc[k] = d | (rank << 8) | (cnt << 16) | (valid ? 0x80000000u : 0u);
        }
        if (VAR & OSV_EARLY_AGG) {
            // The last worker wave to finish hashing publishes the tile's counts, before
            // barrier A: a scan wave that is late (a long walk for the previous tile) then
            // delays only its own workgroup, not every later tile waiting for these counts.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            uint32_t arrived = 0;
            if (lane == 0) arrived = atomicAdd(&wc[ARRIVE], 1u);
            arrived = __builtin_amdgcn_readfirstlane(arrived);
            if (arrived == (uint32_t) WORKERS - 1u && tile != 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                uint32_t tot = 0;
#pragma unroll
                for (int ww = 0; ww < WORKERS; ww++)
                    tot += ((volatile uint32_t *) wc)[ww * OS_MAXP + lane];
                if ((uint32_t) lane < P)
                    os_store_status(p.status + (uint64_t) tile * P + lane, OS_FLAG_AGG | tot);
            }
        }
        __syncthreads(); // A: counts(tile) complete; start(prev) is in LDS

        uint32_t below = 0; // lane = destination: records of earlier waves of this tile
#pragma unroll
        for (int ww = 0; ww < WORKERS; ww++) {
            const uint32_t v = wc[ww * OS_MAXP + lane];
            if (ww < w) below += v;
        }
        wslot[w * OS_MAXP + lane] = below;
        const unsigned long long *gpos = gpos2 + ((it & 1u) ^ 1u) * OS_MAXP; // of the parked tile
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            // position of record k of the current tile inside its destination's share
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
            // store the parked record of the same slot, then park the current one there
            const int slot = k * WTHREADS + tid;
            if ((mp[k] >> 31) != 0) {
                const uint32_t dp = mp[k] & 0xffu, offp = (mp[k] >> 8) & 0xfffffu;
                const unsigned long long pos = gpos[dp] + offp;
                if (pos < pcap[dp]) {
                    reinterpret_cast<uint4 *>(pbase[dp])[pos] = park[slot];
                }
                else if ((int32_t) dp == p.self) {
                    raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                }
                else {
                    overflow = true;
                }
            }
            park[slot] = r[k];
            mp[k] = mc[k];
        }
        if (overflow) atomicMin(&p.ctrl->head.cut_tile, prev);
        // the scan wave drew the next tile right after barrier A
        while (misc[4 + (it & 1u)] != it + 1u) __builtin_amdgcn_s_sleep(1);
        const uint32_t next = misc[2 + (it & 1u)];
        if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        prev = tile;
        tile = next;
        it++;
    }
    __syncthreads(); // start(prev) is in LDS
    if (prev != 0xFFFFFFFFu) {
        const unsigned long long *gpos = gpos2 + ((it & 1u) ^ 1u) * OS_MAXP;
        bool overflow = false;
#pragma unroll
        for (int k = 0; k < ITEMS; k++) {
            if ((mp[k] >> 31) != 0) {
                const uint32_t dp = mp[k] & 0xffu, offp = (mp[k] >> 8) & 0xfffffu;
                const unsigned long long pos = gpos[dp] + offp;
                if (pos < pcap[dp]) {
                    reinterpret_cast<uint4 *>(pbase[dp])[pos] = park[k * WTHREADS + tid];
                }
                else if ((int32_t) dp == p.self) {
                    raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                }
                else {
                    overflow = true;
                }
            }
        }
        if (overflow) atomicMin(&p.ctrl->head.cut_tile, prev);
    }
    (void) overflow_prev;
}

} // namespace mgs
#endif
