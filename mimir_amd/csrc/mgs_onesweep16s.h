// Single-pass stable hash-partition of 16-byte records with a dedicated scan
// wave (the production kernel for 16-byte fixed-width layouts, nranks <= 64).
//
// Same data flow as described in mgs_onesweep16.h -- coalesced 16-byte loads,
// lookup3 in registers, match-any ranking, LDS image grouped by destination,
// chained scan over a global status array, coalesced stores -- but arranged so
// that no record-moving wave ever waits for the chained scan:
//
//   * the workgroup's LAST wave moves no records.  Per tile it draws the next
//     ticket, publishes the tile's per-destination counts, walks the
//     predecessors' status words and publishes the inclusive prefix;
//   * the other waves store tile i-1 out of LDS while that walk for tile i is
//     in flight, and only need its result one trip later:
//
//       worker waves                          scan wave
//       hash(i)                               ticket(i+1)
//       ---- barrier A(i): counts(i) complete, image(i-1)+position(i-1) ready
//       slot bases(i); store(i-1)             publish aggregate(i)
//       ---- barrier B(i): image free again
//       reorder(i) -> LDS; prefetch(i+1)      walk back, publish inclusive(i),
//       (wait for loads) hash(i+1)              position(i) -> LDS
//       ---- barrier A(i+1)
//
// The round trip of the walk (a few microseconds under full streaming load)
// overlaps the workers' wait for their own prefetched records.
#ifndef MGS_ONESWEEP16S_H
#define MGS_ONESWEEP16S_H

#include "mgs_onesweep16.h"

namespace mgs {

template <int THREADS, int ITEMS>
struct OssGeom {
    static constexpr int WORKERS = THREADS / 64 - 1; // record-moving waves
    static constexpr int TILE = WORKERS * 64 * ITEMS;
    // LDS carve-up (bytes): one array, 16-byte aligned pieces
    static constexpr int REC = TILE * 16;                  // tile image
    static constexpr int CNT = 2 * WORKERS * OS_MAXP * 4;  // per (wave,dest) counts, two copies
    static constexpr int SLOT = WORKERS * OS_MAXP * 4;     // per (wave,dest) running slot
    static constexpr int BINSTART = 2 * (OS_MAXP + 4) * 4; // dest -> first slot (+ total), two copies
    static constexpr int GPOS = 2 * OS_MAXP * 8;           // dest -> global record position, two copies
    static constexpr int PBASE = OS_MAXP * 8;              // dest -> partition address
    static constexpr int PCAP = OS_MAXP * 8;               // dest -> capacity in records
    static constexpr int MISC = 16;
    static constexpr int BYTES = REC + CNT + SLOT + BINSTART + GPOS + PBASE + PCAP + MISC;
};

// The chained scan for `tile` (> 0), run by one whole wave.  Returns, in lanes
// b < P, the number of records of destination b in all earlier tiles plus the
// initial fill.  Lane = (j, b): predecessor slot j of a load, destination b;
// one poll fetches OS_LOOK loads x (64 >> nbits_pad) predecessors and every
// decision is wave-uniform, so the walk is a scalar loop.
struct OsWalkStats { // per scan wave, flushed to Ctrl::stats once at the end of the kernel
    unsigned long long tiles = 0, polls = 0, idle = 0, depth = 0, cycles = 0;
    unsigned long long max_depth = 0, max_cycles = 0, max_polls = 0;
    // windowed scan only: first polls that did not finish, by cause
    unsigned long long fail_near = 0, fail_far = 0, fail_base = 0;
    bool windowed = false;
    __device__ void flush(Ctrl *ctrl) const
    {
        if (windowed) {
            atomicAdd(&ctrl->head.stats[0], tiles);
            atomicAdd(&ctrl->head.stats[1], polls);
            atomicAdd(&ctrl->head.stats[2], idle);
            atomicAdd(&ctrl->head.stats[3], depth);
            atomicAdd(&ctrl->head.stats[4], cycles);
            atomicAdd(&ctrl->head.stats[5], fail_near);
            atomicAdd(&ctrl->head.stats[6], fail_far);
            atomicAdd(&ctrl->head.stats[7], fail_base);
            return;
        }
        atomicAdd(&ctrl->head.stats[0], tiles);
        atomicAdd(&ctrl->head.stats[1], polls);
        atomicAdd(&ctrl->head.stats[2], idle);
        atomicAdd(&ctrl->head.stats[3], depth);
        atomicAdd(&ctrl->head.stats[4], cycles);
        atomicMax(&ctrl->head.stats[5], max_depth);
        atomicMax(&ctrl->head.stats[6], max_cycles);
        atomicMax(&ctrl->head.stats[7], max_polls);
    }
};

template <int VAR, int LOOKMAX = OS_LOOK>
__device__ __forceinline__ unsigned long long os_chain_walk(const OsParams &p, uint32_t tile, int lane,
                                                            OsWalkStats *ws = nullptr)
{
    const uint32_t P = p.P;
    const uint32_t Pp = 1u << p.nbits_pad;   // P rounded up to a power of two
    const uint32_t tpw = 64u >> p.nbits_pad; // predecessors per load
    const uint32_t j = (uint32_t) lane >> p.nbits_pad, b = (uint32_t) lane & (Pp - 1);
    const bool bin_ok = b < P;
    const uint64_t gmask = (Pp == 64) ? ~0ull : ((1ull << Pp) - 1ull);
    const uint32_t gshift = j << p.nbits_pad;
    unsigned long long acc = 0;
    int64_t look = (int64_t) tile - 1; // nearest predecessor not yet accounted for
    bool done = false;
    uint32_t idle = 0;
    uint32_t st_polls = 0, st_idle = 0;
    unsigned long long st_t0 = 0;
    if (VAR & OSV_STATS) st_t0 = __builtin_readcyclecounter();
    int nq = (LOOKMAX < OS_LOOK0) ? LOOKMAX : ((VAR & OSV_WIDE_FIRST) ? LOOKMAX : OS_LOOK); // loads in this poll
    while (!done) {
        unsigned long long v[LOOKMAX];
#pragma unroll
        for (int q = 0; q < LOOKMAX; q++) {
            const int64_t t = look - (int64_t)(q * tpw + j);
            v[q] = (q < nq && bin_ok && t >= 0) ? os_load_status(p.status + (uint64_t) t * P + b)
                                                : OS_FLAG_INCL; // before tile 0 / padding lane
        }
        uint32_t consumed = 0;
        bool stop = false;
#pragma unroll
        for (int q = 0; q < LOOKMAX; q++) {
            if (stop || q >= nq) continue;
            const unsigned long long f = v[q] >> 62;
            const int64_t t = look - (int64_t)(q * tpw + j);
            const bool pad = !bin_ok;
            // state of predecessor (q, j): all of its destinations must agree
            const uint64_t m_agg = __ballot(f == 1 || pad);
            const uint64_t m_inc = __ballot(f == 2 || pad);
            const bool g_agg = ((m_agg >> gshift) & gmask) == gmask;
            const bool g_inc = ((m_inc >> gshift) & gmask) == gmask;
            const uint64_t t_inc = __ballot(g_inc);
            const uint64_t t_wait = __ballot(!g_agg && !g_inc);
            const uint32_t ji =
                t_inc ? (uint32_t)(__ffsll((unsigned long long) t_inc) - 1) >> p.nbits_pad : 64u;
            const uint32_t jn =
                t_wait ? (uint32_t)(__ffsll((unsigned long long) t_wait) - 1) >> p.nbits_pad : 64u;
            const unsigned long long val = (bin_ok && t >= 0) ? (v[q] & OS_VALUE_MASK) : 0ull;
            if (ji < jn) { // reached an inclusive prefix: finished
                if (j <= ji) acc += val;
                done = true;
                stop = true;
            }
            else if (jn < 64u) { // predecessor jn of this load is not published yet
                if (j < jn) acc += val;
                consumed += jn;
                stop = true;
            }
            else { // tpw aggregates
                acc += val;
                consumed += tpw;
            }
        }
        look -= consumed;
        nq = (!stop || (VAR & OSV_WIDE_FIRST)) ? LOOKMAX : OS_LOOK;
        if (VAR & OSV_STATS) {
            st_polls++;
            if (consumed == 0 && !done) st_idle++;
        }
        if (!done) {
            if (consumed == 0) {
                // Bounded wait: give up after OS_SPIN_LIMIT empty polls, and leave at
                // once when any other tile has already given up.
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
                __builtin_amdgcn_s_sleep(2);
            }
            else {
                idle = 0;
            }
        }
    }
    if ((VAR & OSV_STATS) && ws) {
        const unsigned long long dt = __builtin_readcyclecounter() - st_t0;
        const unsigned long long depth = (unsigned long long) ((int64_t) tile - 1 - look);
        ws->tiles += 1;
        ws->polls += st_polls;
        ws->idle += st_idle;
        ws->depth += depth;
        ws->cycles += dt;
        ws->max_depth = depth > ws->max_depth ? depth : ws->max_depth;
        ws->max_cycles = dt > ws->max_cycles ? dt : ws->max_cycles;
        ws->max_polls = st_polls > ws->max_polls ? st_polls : ws->max_polls;
    }
    // fold the per-slot partial sums: lanes with equal b, strides Pp..32
    for (uint32_t o = Pp; o < 64u; o <<= 1) {
        const uint32_t lo = __shfl_xor((uint32_t) acc, (int) o, 64);
        const uint32_t hi = __shfl_xor((uint32_t)(acc >> 32), (int) o, 64);
        acc += ((unsigned long long) hi << 32) | lo;
    }
    return acc;
}

template <int THREADS, int ITEMS, int VAR, int MINW, int LOOK = OS_LOOK>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweep16s(OsParams p)
{
    using G = OssGeom<THREADS, ITEMS>;
    constexpr int WORKERS = G::WORKERS, TILE = G::TILE;
    constexpr int WTHREADS = WORKERS * 64;
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
    const bool scanner = (w == WORKERS);
    const uint32_t P = p.P;

    if (tid < OS_MAXP) {
        const bool live = (uint32_t) tid < P;
        pbase[tid] = live ? p.ctrl->head.part_base[tid] : 0ull;
        pcap[tid] = live ? (unsigned long long) (p.ctrl->head.part_cap[tid] >> 4) : 0ull;
    }
    for (int j = tid; j < 2 * WORKERS * OS_MAXP; j += THREADS) wcnt[j] = 0;
    if (tid == 0) {
        misc[2] = 0;
        misc[3] = 0;
        misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
    }
    __syncthreads();
    uint32_t tile = misc[0];
    uint32_t prev = 0xFFFFFFFFu; // tile whose image waits in LDS to be stored
    uint4 r[ITEMS];
    if (!scanner && tile < p.ntiles) os_load_tile<ITEMS, VAR>(p, tile, TILE, w, lane, r);
    int par = 0;
    OsWalkStats ws;

    // Store the LDS image of the previous tile: consecutive lanes write
    // consecutive 16-byte slots of one partition run.
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
                for (uint32_t step = 1u << (p.nbits - 1); step > 0; step >>= 1) {
                    const uint32_t nb = b + step;
                    if (nb < P && binstart[nb] <= s) b = nb;
                }
                const unsigned long long pos = gpos[b] + (s - binstart[b]);
                if (pos < pcap[b]) {
                    uint4 *dst = reinterpret_cast<uint4 *>(pbase[b]) + pos;
                    const uint4 v = srec[s];
                    if (VAR & OSV_NT_STORE) {
                        unsigned long long *q = reinterpret_cast<unsigned long long *>(dst);
                        __builtin_nontemporal_store(((unsigned long long) v.y << 32) | v.x, q);
                        __builtin_nontemporal_store(((unsigned long long) v.w << 32) | v.z, q + 1);
                    }
                    else {
                        *dst = v;
                    }
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
        uint32_t *wc = wcnt + par * (WORKERS * OS_MAXP);
        uint32_t meta[ITEMS]; // dest | rank<<8 | count<<16 | valid<<31
        if (scanner) {
            if (!(VAR & OSV_LATE_TICKET)) {
                if (lane == 0) misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
            }
        }
        else {
            // ---- hash, destination, rank within row ---------------------------
            const uint64_t idx0 = (uint64_t) tile * TILE + (uint32_t) w * (ITEMS * 64) + lane;
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
            if (VAR & OSV_EARLY_AGG) {
                // The worker wave that finishes last publishes the tile's counts at once,
                // whether or not the scan wave is still busy with the previous tile.
                uint32_t old = 0;
                if (lane == 0) old = atomicAdd(const_cast<uint32_t *>(&misc[2 + par]), 1u);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old == (uint32_t)(WORKERS - 1) && tile != 0) {
                    uint32_t t2 = 0;
#pragma unroll
                    for (int ww = 0; ww < WORKERS; ww++) t2 += wc[ww * OS_MAXP + lane];
                    if ((uint32_t) lane < P)
                        os_store_status(p.status + (uint64_t) tile * P + lane, OS_FLAG_AGG | t2);
                }
            }
        }
        __syncthreads(); // A: counts(tile) complete; image(prev) and position(prev) ready
        uint32_t next = 0;
        if (!(VAR & OSV_LATE_TICKET)) next = misc[1];

        // ---- per-destination totals (lane = destination) ----------------------
        uint32_t tot = 0, below = 0;
#pragma unroll
        for (int ww = 0; ww < WORKERS; ww++) {
            const uint32_t v = wc[ww * OS_MAXP + lane];
            tot += v;
            if (ww < w) below += v;
        }
        if (scanner) {
            // publish this tile's counts before anything else
            if ((uint32_t) lane < P) {
                unsigned long long *st = p.status + (uint64_t) tile * P + lane;
                if (tile == 0) {
                    const unsigned long long start =
                        (unsigned long long) (p.ctrl->head.part_init[lane] >> 4);
                    gpos2[par * OS_MAXP + lane] = start;
                    os_store_status(st, OS_FLAG_INCL | (start + tot));
                }
                else if (!(VAR & OSV_EARLY_AGG)) {
                    os_store_status(st, OS_FLAG_AGG | tot);
                }
            }
            // the other copy of the counts was last read before the previous barrier B
            for (int j = lane; j < WORKERS * OS_MAXP; j += 64)
                wcnt[(par ^ 1) * (WORKERS * OS_MAXP) + j] = 0;
            if (lane == 0) misc[2 + (par ^ 1)] = 0;
            // Drawing the next tile only now keeps the time between a tile's ticket and
            // the publication of its counts short (less skew between neighbouring tiles).
            if ((VAR & OSV_LATE_TICKET) && lane == 0) misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
        }
        else {
            const uint32_t incl = wave_incl_scan_u32(tot, lane);
            wslot[w * OS_MAXP + lane] = incl - tot + below;
            if (w == 0) {
                uint32_t *binstart = binstart2 + par * (OS_MAXP + 4);
                binstart[lane] = incl - tot;
                if (lane == 63) binstart[64] = incl;
            }
            if (prev != 0xFFFFFFFFu) store_image(prev, par ^ 1);
        }
        __syncthreads(); // B: every slot of image(prev) has been read, the image is free
        if (VAR & OSV_LATE_TICKET) next = misc[1];

        if (scanner) {
            if (tile != 0) {
                unsigned long long acc;
                if (VAR & OSV_FAKE_CHAIN) acc = (unsigned long long) tile * (TILE / P);
                else acc = os_chain_walk<VAR, LOOK>(p, tile, lane, &ws);
                if ((uint32_t) lane < P) {
                    if (!(VAR & OSV_FAKE_CHAIN))
                        os_store_status(p.status + (uint64_t) tile * P + lane,
                                        OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
                    gpos2[par * OS_MAXP + lane] = acc;
                }
            }
        }
        else {
            // ---- reorder into the LDS image -----------------------------------
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const uint32_t d = meta[k] & 0xffu, rank = (meta[k] >> 8) & 0xffu;
                const uint32_t cnt = (meta[k] >> 16) & 0xffu;
                const bool valid = (meta[k] >> 31) != 0;
                if (valid) {
                    const uint32_t base = wslot[w * OS_MAXP + d];
                    if (rank == 0) wslot[w * OS_MAXP + d] = base + cnt;
                    srec[base + rank] = r[k];
                }
            }
            // ---- the registers are free again: fetch the next tile ------------
            if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        }
        prev = tile;
        tile = next;
        par ^= 1;
    }
    __syncthreads(); // image(prev) and position(prev) ready
    if (!scanner && prev != 0xFFFFFFFFu) store_image(prev, par ^ 1);
    if ((VAR & OSV_STATS) && scanner && lane == 0) ws.flush(p.ctrl);
}

} // namespace mgs
#endif
