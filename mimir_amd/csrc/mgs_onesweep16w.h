// k_onesweep16q: the LDS-image one-sweep kernel with a dedicated scan wave
// and a windowed chained scan.
//
// The plain chained scan (k_onesweep16) walks back from tile i-1 until it
// meets a tile that has published its inclusive prefix; one poll covers 128
// predecessors.  When tiles are handed out faster than (window / poll round
// trip) the nearest inclusive prefix lies outside the window, the walk needs a
// second dependent round trip, finishes later, publishes later, and the walks
// behind it get deeper still: the scan, not HBM, then sets the pace.
//
// Here a tile publishes two things in two arrays:
//   agg16[tile]   its per-destination counts as 16-bit fields (count + 1,
//                 0 = not published yet), four destinations per 8-byte word,
//                 written once at barrier A and never changed;
//   status[tile]  its inclusive prefix (64-bit per destination), written when
//                 its own scan is done.
// One poll of the scan loads the counts of the `window` nearest predecessors
// (LOOK wave-wide loads; 384 tiles for 8 destinations, half the bytes of the
// 64-bit entries) AND, in the same breath, the inclusive prefixes of several
// candidate base tiles at distances step, 2*step, ... (one more load: lane =
// (candidate, destination)).  The nearest candidate that is published, with
// all counts nearer than it published, finishes the scan in this one round
// trip: prefix(i) = inclusive(base) + sum of counts of base+1 .. i-1.  If no
// candidate qualifies the published part of the window is accumulated and the
// window moves back, so the scan stays correct at any depth.
#ifndef MGS_ONESWEEP16W_H
#define MGS_ONESWEEP16W_H

#include "mgs_onesweep16.h"
#include "mgs_round_finder.h"

namespace mgs {

struct OsWalkStats { // per scan wave, flushed to Ctrl::stats once at the end of the kernel
    unsigned long long tiles = 0, polls = 0, idle = 0, depth = 0, cycles = 0;
    unsigned long long max_depth = 0, max_cycles = 0, max_polls = 0;
    // windowed scan only: first polls that did not finish, by cause
    unsigned long long fail_near = 0, fail_far = 0, fail_base = 0;
    // rounds: cycles spent waiting for a round's table entry / finding cuts, and how often
    unsigned long long rt_wait_cycles = 0, rt_waits = 0, cut_cycles = 0;
    bool windowed = false;
    __device__ void flush(Ctrl *ctrl) const
    {
        if (windowed) {
            atomicAdd(&ctrl->head.stats[0], tiles);
            atomicAdd(&ctrl->head.stats[1], polls);
            atomicAdd(&ctrl->head.stats[2], idle);
            atomicAdd(&ctrl->head.stats[3], depth);
            atomicAdd(&ctrl->head.stats[4], cycles);
            if (rt_waits || cut_cycles) { // a call with round cuts reports those instead
                atomicAdd(&ctrl->head.stats[5], rt_wait_cycles);
                atomicAdd(&ctrl->head.stats[6], rt_waits);
                atomicAdd(&ctrl->head.stats[7], cut_cycles);
                return;
            }
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


constexpr unsigned long long OS_AGG16_ONES = 0x0001000100010001ull;
constexpr unsigned long long OS_AGG16_EVEN = 0x0000ffff0000ffffull;

// Publish the counts of `tile` (lane = destination holds `tot`).
__device__ __forceinline__ void os_publish_agg16(const OsParams &p, uint32_t tile, int lane, uint32_t tot)
{
    const uint32_t lpt = 1u << p.lpt_log;
    unsigned long long word = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t c = (uint32_t) __shfl((int) tot, (lane * 4 + k) & 63);
        word |= (unsigned long long) ((c + 1u) & 0xffffu) << (16 * k);
    }
    if ((uint32_t) lane < lpt) os_store_status(p.agg16 + (uint64_t) tile * lpt + lane, word);
}

// The windowed chained scan for `tile` (> 0), run by one whole wave.  Returns,
// in lanes b < P, the number of records of destination b in all earlier tiles
// plus the initial fill.
struct OsNoop {
    __device__ void operator()() const {}
};

// `between` runs once, after the first poll's loads have been issued and before
// their results are used (a caller with other duties while the poll is in flight).
template <int VAR, int LOOK, class Between = OsNoop>
__device__ __forceinline__ unsigned long long os_window_walk(const OsParams &p, uint32_t tile, int lane,
                                                             OsWalkStats *ws = nullptr,
                                                             Between between = Between())
{
    const uint32_t P = p.P;
    // counts: lane = (j, g): predecessor slot j of a load, word g of its row
    const uint32_t lptl = p.lpt_log, lpt = 1u << lptl, tpl = 64u >> lptl;
    const uint32_t j = (uint32_t) lane >> lptl, g = (uint32_t) lane & (lpt - 1);
    const uint32_t window = (uint32_t) LOOK * tpl;
    // candidates: lane = (c, b): candidate c, destination b
    const uint32_t Pp = 1u << p.nbits_pad;
    const uint32_t c = (uint32_t) lane >> p.nbits_pad, b = (uint32_t) lane & (Pp - 1);
    const uint32_t ncand = 64u >> p.nbits_pad;
    const uint32_t step = window / ncand;
    const uint32_t dist = (c + 1) * step; // candidate base = look - dist, counts 0..dist-1 are needed
    const bool bin_ok = b < P;
    const uint64_t gmask = (Pp == 64) ? ~0ull : ((1ull << Pp) - 1ull);
    const uint32_t gshift = c << p.nbits_pad;

    unsigned long long acc_lo = 0, acc_hi = 0; // 32-bit fields: dests 4g+0 | 4g+2, 4g+1 | 4g+3
    unsigned long long base = 0;
    int64_t look = (int64_t) tile - 1; // nearest predecessor not yet accounted for
    uint32_t cstar = 0;
    bool done = false, first_poll = true;
    uint32_t idle = 0, st_polls = 0, st_idle = 0;
    unsigned long long st_t0 = 0;
    if (VAR & OSV_STATS) st_t0 = __builtin_readcyclecounter();

    auto entry_ok = [](unsigned long long x) {
        return (x & 0xffffull) && (x & 0xffff0000ull) && (x & 0xffff00000000ull)
               && (x & 0xffff000000000000ull);
    };
    while (!done) {
        unsigned long long v[LOOK];
        // the far part of the window and the candidate bases first ...
#pragma unroll
        for (int q = 1; q < LOOK; q++) {
            const int64_t t = look - (int64_t)((uint32_t) q * tpl + j);
            v[q] = (t >= 0) ? os_load_status(p.agg16 + (uint64_t) t * lpt + g) : OS_AGG16_ONES;
        }
        const int64_t tb = look - (int64_t) dist;
        unsigned long long cv;
        if (!bin_ok) cv = OS_FLAG_INCL;
        else if (tb >= 0) cv = os_load_status(p.status + (uint64_t) tb * P + b);
        else cv = OS_FLAG_INCL | (unsigned long long) (p.ctrl->head.part_init[b] >> p.unit_shift);
        // ... the nearest predecessors last: they published their counts only moments
        // ago, and every bit of time makes it likelier that this load sees them
        const int64_t t0 = look - (int64_t) j;
        v[0] = (t0 >= 0) ? os_load_status(p.agg16 + (uint64_t) t0 * lpt + g) : OS_AGG16_ONES;
        if (first_poll) {
            between();
            first_poll = false;
        }

        // first predecessor beyond the nearest load whose counts are not published yet
        uint32_t far_invalid = window;
#pragma unroll
        for (int q = LOOK - 1; q >= 1; q--) {
            const uint64_t inv = ~__ballot(entry_ok(v[q]));
            if (inv) far_invalid = (uint32_t) q * tpl + ((uint32_t)(__ffsll((unsigned long long) inv) - 1) >> lptl);
        }
        const uint64_t m_inc = __ballot((cv >> 62) == 2);
        const bool g_ok = ((m_inc >> gshift) & gmask) == gmask;
        uint32_t take; // predecessors accounted for by this poll
        for (;;) {
            const uint64_t inv0 = ~__ballot(entry_ok(v[0]));
            const uint32_t near_invalid =
                inv0 ? ((uint32_t)(__ffsll((unsigned long long) inv0) - 1) >> lptl) : tpl;
            const uint32_t first_invalid = near_invalid < tpl ? near_invalid : far_invalid;
            // nearest candidate that is published and has all nearer counts
            const uint64_t usable = __ballot(g_ok && dist <= first_invalid);
            if (usable) {
                cstar = (uint32_t)(__ffsll((unsigned long long) usable) - 1) >> p.nbits_pad;
                take = (cstar + 1) * step;
                done = true;
                break;
            }
            take = first_invalid;
            if ((VAR & OSV_STATS) && ws && st_polls == 0) {
                ws->windowed = true;
                if (near_invalid < tpl) ws->fail_near++;
                else if (far_invalid < window) ws->fail_far++;
                else ws->fail_base++;
            }
            // Only some of the nearest predecessors are missing and a candidate is ready
            // behind them: poll just that one load again, everything else stays valid.
            const bool near_only = near_invalid < tpl && __ballot(g_ok && dist <= far_invalid) != 0;
            if (!near_only) break;
            if (VAR & OSV_STATS) {
                st_polls++;
                st_idle++;
            }
            if (++idle > OS_SPIN_LIMIT
                || ((idle & 255u) == 0
                    && __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0))
                break; // the outer loop's bounded wait deals with it
            __builtin_amdgcn_s_sleep(1);
            v[0] = (t0 >= 0) ? os_load_status(p.agg16 + (uint64_t) t0 * lpt + g) : OS_AGG16_ONES;
        }
#pragma unroll
        for (int q = 0; q < LOOK; q++) {
            if ((uint32_t) q * tpl + j < take) {
                const unsigned long long u = v[q] - OS_AGG16_ONES;
                acc_lo += u & OS_AGG16_EVEN;
                acc_hi += (u >> 16) & OS_AGG16_EVEN;
            }
        }
        look -= take;
        if (done) base = cv & OS_VALUE_MASK;
        if (VAR & OSV_STATS) {
            st_polls++;
            if (take == 0) st_idle++;
        }
        if (!done) {
            if (take == 0) {
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
    // add up the lanes that hold the same word of different predecessors
    for (uint32_t s = lpt; s < 64u; s <<= 1) {
        acc_lo += (unsigned long long) __shfl_xor((long long) acc_lo, (int) s);
        acc_hi += (unsigned long long) __shfl_xor((long long) acc_hi, (int) s);
    }
    // lane b: its field of word b/4, and the base prefix of candidate cstar
    const int src = (int) ((uint32_t) lane >> 2) & (int) (lpt - 1);
    const unsigned long long lo = (unsigned long long) __shfl((long long) acc_lo, src);
    const unsigned long long hi = (unsigned long long) __shfl((long long) acc_hi, src);
    const unsigned long long pick = (lane & 1) ? hi : lo;
    const unsigned long long sum = (lane & 2) ? (pick >> 32) : (pick & 0xffffffffull);
    const unsigned long long bv =
        (unsigned long long) __shfl((long long) base, (int) ((cstar << p.nbits_pad) | b));
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
    return bv + sum;
}

// Hash the ITEMS rows a wave holds of a tile, find every record's destination,
// its rank among the records of the same destination in its row and that
// group's size (a match-any over the destination bits: NB wave ballots), and
// add the group sizes to the wave's per-destination counts in LDS.  The two
// wave-uniform choices (power-of-two modulus, key longer than 12 bytes) are
// template parameters so that the loop body is one straight run of code.
template <int ITEMS, int NB, bool POW2, bool LONGKEY>
__device__ __forceinline__ void os_hash_rows(const OsParams &p, const uint4 (&r)[ITEMS], uint64_t idx0,
                                             uint32_t *wave_counts, uint32_t (&meta)[ITEMS])
{
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool valid = idx0 + (uint32_t) k * 64 < p.n;
        uint32_t a, b, c;
        a = b = c = 0xdeadbeefu + p.klen;
        if (LONGKEY) {
            a += r[k].x; b += r[k].y; c += r[k].z;
            MGS_MIX(a, b, c);
            a += r[k].w & p.kmask[3];
        }
        else {
            a += r[k].x & p.kmask[0]; b += r[k].y & p.kmask[1]; c += r[k].z & p.kmask[2];
        }
        MGS_FINAL(a, b, c);
        const uint32_t d = POW2 ? fastmod_pow2(c, p.fm) : fastmod_magic(c, p.fm);
        uint64_t m = __ballot(valid);
#pragma unroll
        for (uint32_t bit = 0; bit < (uint32_t) NB; bit++) {
            const bool one = (d >> bit) & 1u;
            const uint64_t bk = __ballot(one);
            m &= one ? bk : ~bk;
        }
        const uint32_t rank = lanes_below(m);
        const uint32_t cnt = (uint32_t) __popcll(m);
        if (valid && rank == 0) atomicAdd(&wave_counts[d], cnt);
        meta[k] = d | (rank << 8) | (cnt << 16) | (valid ? 0x80000000u : 0u);
    }
}

// The same for kernels built with OSV_DIGIT: a record's destination is the bit field
// (word 0 >> p.klen) & p.kmask[0].  With the partitions laid end to end in the order of the
// field's values -- part_base from a histogram -- one launch is one pass of a stable
// least-significant-digit radix sort of 16-byte records whose first word is the sort key
// (KV -> KMV, mgs_group.h).
template <int ITEMS, int NB>
__device__ __forceinline__ void os_digit_rows(const OsParams &p, const uint4 (&r)[ITEMS], uint64_t idx0,
                                              uint32_t *wave_counts, uint32_t (&meta)[ITEMS])
{
#pragma unroll
    for (int k = 0; k < ITEMS; k++) {
        const bool valid = idx0 + (uint32_t) k * 64 < p.n;
        const uint32_t d = (r[k].x >> p.klen) & p.kmask[0];
        uint64_t m = __ballot(valid);
#pragma unroll
        for (uint32_t bit = 0; bit < (uint32_t) NB; bit++) {
            const bool one = (d >> bit) & 1u;
            const uint64_t bk = __ballot(one);
            m &= one ? bk : ~bk;
        }
        const uint32_t rank = lanes_below(m);
        const uint32_t cnt = (uint32_t) __popcll(m);
        if (valid && rank == 0) atomicAdd(&wave_counts[d], cnt);
        meta[k] = d | (rank << 8) | (cnt << 16) | (valid ? 0x80000000u : 0u);
    }
}

// ---- rounds ------------------------------------------------------------------------
// A round (src/collectiveshuffler.h:148-160) ends before the first record that does
// not fit its partition; every partition then starts again from empty.  In terms of
// the chained scan's running totals C_b(r) (units of column b before record r, never
// reset): with B_k the totals at the first record of round k, record r of round k
// goes to position C_b(r) - B_k[b] of image k, and the round is cut at the first r
// with C_b(r) + size(r) - B_k[b] > capacity, i.e. C_b(r+1) > B_k[b] + cap[b] =: lim.
// Totals only grow, so a tile knows from its own exclusive prefix `acc` alone whether
// a cut lies before it (acc[b] > lim[b] for some b) and, with its counts, whether one
// lies inside it (acc + tot > lim).  The tile that holds a cut finds the exact record,
// publishes the new round's B in the round table and later tiles pick it up there.

// Wait for entry k of the round table (lane = column) and return its word.
__device__ __forceinline__ unsigned long long os_rt_wait(const OsParams &p, uint32_t k, int lane, bool live)
{
    unsigned long long v = OS_RT_VALID;
    uint32_t idle = 0;
    for (;;) {
        if (live) v = os_load_status(p.rt + (uint64_t) k * p.P + lane);
        if (__ballot((v & OS_RT_VALID) == 0) == 0) break;
        if (++idle > OS_SPIN_LIMIT) {
            raise_error(p.ctrl, MGS_ERR_TIMEOUT);
            break;
        }
        if ((idle & 255u) == 0
            && __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
            break;
        __builtin_amdgcn_s_sleep(1);
    }
    return v & ~OS_RT_VALID;
}

// LDS the scan wave uses while it looks for round cuts inside its tile.
struct OsCutLds {
    volatile unsigned long long pos[OS_MAXP]; // lane = column: units before the current row
    volatile unsigned long long lim[OS_MAXP]; // lane = column: B + capacity of the current round (~0: never cut)
    volatile uint32_t cnt[OS_MAXP];           // scratch: units of a column in a row before a cut
};

// The tile `tile` (16-byte records) starts in round k_cur and holds at least one cut.
// Finds every cut of the tile in emit order, publishes the rounds they begin and
// leaves k_cur / base / lim at the state after the tile.  One wave; lane = column for
// acc / tot / base / lim / my_cap, lane = record inside the row walk.  `wc` holds the
// tile's counts per (worker wave, destination): the first cut is looked for only from
// the first wave segment that overflows.
template <int NB, int SEG>
__device__ __forceinline__ void os_find_cuts16(const OsParams &p, uint32_t tile, int nseg, int lane,
                                            unsigned long long acc, uint32_t tot, const uint32_t *wc,
                                            const unsigned long long *sde, bool is_part, unsigned long long my_cap,
                                            uint32_t &k_cur, unsigned long long &base, unsigned long long &lim,
                                            unsigned long long &first_nb, OsCutLds *x)
{
    static_assert(SEG == 8, "a lane's destinations of one wave segment are one 8-byte word");
    const uint32_t P = p.P;
    const uint32_t k_in = k_cur;
    uint32_t c = 0, wst = (uint32_t) nseg;
    for (int ww = 0; ww < nseg; ww++) {
        const uint32_t v = wc[ww * OS_MAXP + lane];
        if (is_part && wst == (uint32_t) nseg && acc + c + v > lim) wst = (uint32_t) ww;
        c += v;
    }
    uint32_t wstar = (uint32_t) nseg;
    for (int ww = nseg - 1; ww >= 0; ww--)
        if (__ballot(wst == (uint32_t) ww)) wstar = (uint32_t) ww;
    uint32_t run = 0;
    for (int ww = 0; ww < nseg; ww++)
        if ((uint32_t) ww < wstar) run += wc[ww * OS_MAXP + lane];
    x->pos[lane] = acc + run;
    x->lim[lane] = is_part ? lim : ~0ull;
    __builtin_amdgcn_wave_barrier();
    const uint64_t t0 = (uint64_t) tile * p.tile_records;
    bool stop = false;
    for (int seg = (int) wstar; seg < nseg && !stop; seg++) {
        // the destinations of this lane's records in the segment's SEG rows (byte j = row j,
        // 0xFF = no record), left in LDS by the worker wave that hashed them
        const unsigned long long dq = sde[seg * 64 + lane];
        const uint32_t dpack[2] = {(uint32_t) dq, (uint32_t) (dq >> 32)};
        // Fast-forward: rows in which no partition overflows hold no cut; find the first row
        // that does (per overflowing destination: rows counted with wave ballots) and add the
        // rows before it to the running totals in one go.
        const unsigned long long left = x->lim[lane] - x->pos[lane]; // lane = column: units that still fit
        uint64_t cand = __ballot(is_part && acc + tot > lim);
        int row0 = SEG;
        while (cand) {
            const int b = __ffsll((unsigned long long) cand) - 1;
            cand &= cand - 1;
            const unsigned long long room = (unsigned long long) __shfl((long long) left, b);
            unsigned long long cum = 0;
            int rb = SEG;
#pragma unroll
            for (int j = 0; j < SEG; j++) {
                cum += (unsigned long long) __popcll(__ballot(((dpack[j >> 2] >> (8 * (j & 3))) & 0xffu) == (uint32_t) b));
                if (cum > room && rb == SEG) rb = j;
            }
            row0 = rb < row0 ? rb : row0;
        }
#pragma unroll
        for (int j = 0; j < SEG; j++) {
            const uint32_t dj = (dpack[j >> 2] >> (8 * (j & 3))) & 0xffu;
            if (j < row0 && dj != 0xffu) atomicAdd(const_cast<unsigned long long *>(&x->pos[dj]), 1ull);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int j = row0; j < SEG; j++) {
            if (stop) break;
            const int row = seg * SEG + j;
            const uint32_t d = ((j < 4 ? dpack[0] : dpack[1]) >> (8 * (j & 3))) & 0xffu;
            const bool valid = d != 0xffu;
            uint64_t m = __ballot(valid);
#pragma unroll
            for (uint32_t bit = 0; bit < (uint32_t) NB; bit++) {
                const bool one = (d >> bit) & 1u;
                const uint64_t bk = __ballot(one);
                m &= one ? bk : ~bk;
            }
            const uint32_t rank = lanes_below(m);
            uint64_t open = ~0ull; // lanes not yet given to a finished round
            for (;;) {
                const bool over = valid && ((open >> lane) & 1ull) && x->pos[d] + rank >= x->lim[d];
                const uint64_t om = __ballot(over);
                if (!om) break;
                const int fl = __ffsll((unsigned long long) om) - 1;
                const uint64_t below = (1ull << fl) - 1ull;
                x->cnt[lane] = 0;
                __builtin_amdgcn_wave_barrier();
                if (valid && lane < fl && rank == 0) x->cnt[d] = (uint32_t) __popcll(m & below);
                __builtin_amdgcn_wave_barrier();
                // the round that begins with record (row, fl)
                k_cur++;
                const unsigned long long nb = x->pos[lane] + x->cnt[lane];
                if (k_cur == k_in + 1) first_nb = nb;
                if ((uint32_t) lane < P) os_store_status(p.rt + (uint64_t) k_cur * P + lane, OS_RT_VALID | nb);
                if (lane == 0) {
                    os_store_status(p.rt_start + k_cur, OS_RT_VALID | (t0 + (uint32_t) row * 64 + (uint32_t) fl));
                    atomicMax(&p.ctrl->head.rounds_cut, k_cur);
                }
                if (is_part) {
                    base = nb;
                    lim = nb + my_cap;
                    x->lim[lane] = lim;
                }
                __builtin_amdgcn_wave_barrier();
                open &= ~below;
                // a record that does not fit an empty partition (src/collectiveshuffler.h:156-159)
                const bool never = valid && lane == fl && x->pos[d] + rank >= x->lim[d];
                if (__ballot(never)) {
                    raise_error(p.ctrl, MGS_ERR_RECORD_SIZE);
                    stop = true;
                    break;
                }
                if (k_cur >= p.max_rounds) {
                    stop = true; // the end marker: nothing from here on is consumed
                    break;
                }
            }
            if (valid && rank == 0) x->pos[d] += (unsigned long long) __popcll(m);
            __builtin_amdgcn_wave_barrier();
            if (!__ballot(is_part && acc + tot > lim)) stop = true; // no cut in what is left of the tile
        }
    }
}

// NB: number of destination bits (ceil(log2(P)), at least 1) as a constant, so
// that the match-any loop and the search for a slot's destination unroll.
//
// Per tile the scan wave leaves in LDS, for the workers' deferred store:
//   addr[b]  = address of partition b's byte 0 + 16 * (position of the tile's
//              first record of b in the partition - first slot of b in the
//              image), so that slot s of the image goes to addr[dest(s)] + 16*s;
//   fits     = every record of the tile lies below its partition's capacity.
// A tile that fits is stored with one table read per record; any other tile
// takes the checked path (capacity test per record, round cut, local overflow).
template <int THREADS, int ITEMS, int VAR, int MINW, int LOOK, int NB>
__global__ __launch_bounds__(THREADS, MINW) void k_onesweep16q(OsParams p)
{
    constexpr int WORKERS = THREADS / 64 - 1, TILE = WORKERS * 64 * ITEMS;
    constexpr int WTHREADS = WORKERS * 64;
    constexpr int CNT = WORKERS * OS_MAXP;
    static_assert(TILE < 65535, "a tile's count must fit a 16-bit field");
    __shared__ __attribute__((aligned(16))) uint4 srec[TILE];   // tile image, grouped by destination
    __shared__ uint8_t sdest[TILE];                             // destination of every image slot
    __shared__ uint32_t wcnt[2 * CNT];                          // per (wave, dest) counts, two tiles
    __shared__ volatile uint32_t wslot[CNT];                    // running slot per (wave, dest)
    __shared__ uint32_t binstart2[2 * (OS_MAXP + 4)];           // dest -> first slot (+ total), two tiles
    __shared__ unsigned long long gpos2[2 * OS_MAXP];           // dest -> position of the tile, two tiles
    __shared__ unsigned long long addr2[2 * OS_MAXP];           // see above, two tiles
    __shared__ unsigned long long pbase[OS_MAXP];
    __shared__ unsigned long long pcap[OS_MAXP];
    __shared__ volatile uint32_t misc[8]; // [0] first tile, [1] next tile, [2+par] store mode,
                                          // [4+par] / [6+par] first / last round of the tile
    __shared__ uint32_t arrive[2];        // worker waves done hashing, per tile parity
    __shared__ OsCutLds cutlds;           // scan wave only, tiles that hold a round cut
    // destination of every record in emit order, two tiles: entry (wave, lane) holds the lane's
    // ITEMS records, byte k = row k (0xFF: no record); read only for tiles that hold a round cut
    __shared__ unsigned long long sde[2 * WTHREADS];
    __shared__ unsigned long long rb2[2 * 2 * OS_MAXP]; // per parity: totals at the first record of the tile's
                                                        // first round and of the round after it (checked path)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    constexpr bool S_ND = (VAR & OSV_NODEST) != 0;
    const uint32_t P = p.P;
    constexpr uint32_t nbits = NB;

    if (p.finder && blockIdx.x == 0) {
        // streams layout: this workgroup finds the rounds instead of sweeping
        typedef FinderLdsT<TILE / 64> FL;
        static_assert(sizeof(FL) <= sizeof(srec), "the finder's tables live in the tile image");
        FinderSrc16 src{p};
        os_round_finder<FinderSrc16, THREADS / 64, TILE / 64, ITEMS>(p, P, src, reinterpret_cast<FL *>(srec), tid);
        return;
    }

    if (tid < OS_MAXP) {
        const bool live = (uint32_t) tid < P;
        pbase[tid] = live ? p.ctrl->head.part_base[tid] : 0ull;
        pcap[tid] = live ? (unsigned long long) (p.ctrl->head.part_cap[tid] >> 4) : 0ull;
    }
    for (int j = tid; j < 2 * CNT; j += THREADS) wcnt[j] = 0;
    if (tid == 0) {
        arrive[0] = 0;
        arrive[1] = 0;
        misc[0] = atomicAdd(&p.ctrl->head.ticket, 1u);
    }
    __syncthreads();
    uint32_t tile = misc[0];
    int par = 0;

    // ------------------------------------------------------------------ scan wave
    if (w == WORKERS) {
        // every later tile waits for this wave's prefix: it goes first at the issue arbiter
        __builtin_amdgcn_s_setprio(3);
        OsWalkStats ws;
        const unsigned long long my_base = pbase[lane], my_cap = pcap[lane];
        // round state, lane = destination: the round the previous tile ended in, the running
        // totals at its first record and the totals at which its partitions are full
        const bool is_part = (uint32_t) lane < P && lane != p.self, is_self = lane == p.self;
        uint32_t k_cur = 0;
        unsigned long long base = 0, lim = my_cap;
        while (tile < p.ntiles) {
            const uint32_t *wc = wcnt + par * CNT;
            __syncthreads(); // A: counts(tile) complete
            uint32_t tot = 0; // lane = destination
#pragma unroll
            for (int ww = 0; ww < WORKERS; ww++) tot += wc[ww * OS_MAXP + lane];
            if (!(VAR & OSV_EARLY_AGG)) os_publish_agg16(p, tile, lane, tot); // before anything else
            unsigned long long acc = 0;
            if (tile == 0) {
                acc = ((uint32_t) lane < P) ? (unsigned long long) (p.ctrl->head.part_init[lane] >> 4) : 0ull;
                if ((uint32_t) lane < P) os_store_status(p.status + lane, OS_FLAG_INCL | (acc + tot));
            }
            // draw the next tile now: a short time between a tile's ticket and the
            // publication of its counts keeps neighbouring tiles close in time
            if (lane == 0) misc[1] = atomicAdd(&p.ctrl->head.ticket, 1u);
            // the other copy of the counts was last read before the previous barrier B
            for (int j = lane; j < CNT; j += 64) wcnt[(par ^ 1) * CNT + j] = 0;
            if (lane == 0) arrive[par ^ 1] = 0;
            const uint32_t first_slot = wave_incl_scan_u32(tot, lane) - tot;
            __syncthreads(); // B
            const uint32_t next = misc[1];
            // for the round finder: the units per (worker wave, column) -- a worker wave's records
            // are 512 consecutive ones.  Stored before the look-back, so that they are out when the
            // tile's prefix is (the finder waits for them otherwise); the counts stay in LDS until
            // this wave clears them one tile later
            if (p.seg && (uint32_t) lane < P) {
#pragma unroll
                for (int ww = 0; ww < WORKERS; ww++)
                    p.seg[((uint64_t) tile * WORKERS + ww) * P + lane] = (uint16_t) wc[ww * OS_MAXP + lane];
            }
            if (tile != 0) {
                if (VAR & OSV_FAKE_CHAIN) acc = (unsigned long long) tile * (TILE / P);
                else acc = os_window_walk<VAR, LOOK>(p, tile, lane, &ws);
                if (!(VAR & OSV_FAKE_CHAIN) && (uint32_t) lane < P)
                    os_store_status(p.status + (uint64_t) tile * P + lane,
                                    OS_FLAG_INCL | ((acc + tot) & OS_VALUE_MASK));
            }
            // Which round does the tile start in?  Totals beyond a partition's limit mean that
            // a cut lies before this tile: its round is in the table, or will be shortly.
            if (!(VAR & OSV_FAKE_CHAIN) && k_cur < p.max_rounds) {
                while (__ballot(is_part && acc > lim)) {
                    k_cur++;
                    unsigned long long st0 = 0;
                    if (VAR & OSV_STATS) st0 = __builtin_readcyclecounter();
                    const unsigned long long nb = os_rt_wait(p, k_cur, lane, (uint32_t) lane < P);
                    if (VAR & OSV_STATS) {
                        ws.rt_wait_cycles += __builtin_readcyclecounter() - st0;
                        ws.rt_waits++;
                    }
                    if (is_part) {
                        base = nb;
                        lim = nb + my_cap;
                    }
                    if (k_cur >= p.max_rounds
                        || __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                        break;
                }
            }
            // store mode: 1 = the tile lies inside one round and below every capacity (one
            // table read per record), 0 = checked path (round cuts inside the tile, or the
            // local stream runs full), 2 = the tile lies after the last round: not consumed
            const uint32_t kfirst = k_cur;
            uint32_t mode = 1;
            if (VAR & OSV_FAKE_CHAIN) mode = 1;
            else if (k_cur >= p.max_rounds) mode = 2;
            else if (__ballot(is_part && acc + tot > lim)) {
                unsigned long long first_nb = 0;
                rb2[(par * 2 + 0) * OS_MAXP + lane] = base;
                unsigned long long st0 = 0;
                if (VAR & OSV_STATS) st0 = __builtin_readcyclecounter();
                os_find_cuts16<NB, ITEMS>(p, tile, WORKERS, lane, acc, tot, wc, sde + par * WTHREADS, is_part, my_cap,
                                          k_cur, base, lim, first_nb, &cutlds);
                if (VAR & OSV_STATS) ws.cut_cycles += __builtin_readcyclecounter() - st0;
                rb2[(par * 2 + 1) * OS_MAXP + lane] = first_nb;
                mode = 0;
            }
            else if (__ballot(is_self && acc + tot > my_cap)) {
                rb2[(par * 2 + 0) * OS_MAXP + lane] = base;
                mode = 0;
            }
            gpos2[par * OS_MAXP + lane] = acc;
            addr2[par * OS_MAXP + lane] =
                my_base + (is_part ? (unsigned long long) kfirst * p.image_stride : 0ull) + ((acc - base - first_slot) << 4);
            if (lane == 0) {
                misc[2 + par] = mode;
                misc[4 + par] = kfirst;
                misc[6 + par] = k_cur;
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
        const uint32_t total = binstart2[pr * (OS_MAXP + 4) + 64];
        const uint32_t mode = misc[2 + pr];
        if (mode == 2) return; // after the last round: nothing of this tile is consumed
        if (mode == 1) {
            const unsigned long long *addr = addr2 + pr * OS_MAXP;
#pragma unroll
            for (int k = 0; k < ITEMS; k++) {
                const uint32_t s = (uint32_t) k * WTHREADS + tid;
                if (s < total) {
                    if (VAR & OSV_NT_STORE) {
                        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                        const uint4 v = srec[s];
                        __builtin_nontemporal_store(v4u{v.x, v.y, v.z, v.w},
                                                    reinterpret_cast<v4u *>(addr[sdest[s]] + ((unsigned long long) s << 4)));
                    }
                    else {
                        *reinterpret_cast<uint4 *>(addr[sdest[s]] + ((unsigned long long) s << 4)) = srec[s];
                    }
                }
            }
            return;
        }
        const uint32_t *binstart = binstart2 + pr * (OS_MAXP + 4);
        const unsigned long long *gpos = gpos2 + pr * OS_MAXP;
        const uint32_t kfirst = misc[4 + pr], klast = misc[6 + pr];
        (void) t;
#pragma unroll 1
        for (int k = 0; k < ITEMS; k++) {
            const uint32_t s = (uint32_t) k * WTHREADS + tid;
            if (s < total) {
                const uint32_t b = sdest[s];
                const unsigned long long g = gpos[b] + (s - binstart[b]); // units of b before this record
                if ((int32_t) b == p.self) {
                    if (g < pcap[b]) reinterpret_cast<uint4 *>(pbase[b])[g] = srec[s];
                    else raise_error(p.ctrl, MGS_ERR_LOCAL_OVERFLOW);
                }
                else {
                    // the record's round: the last one of the tile's rounds that begins at or before
                    // it.  The first two are at hand in LDS (nearly every cut tile holds one cut);
                    // a tile with more reads the round table.
                    uint32_t kk = kfirst;
                    unsigned long long bk = rb2[(pr * 2 + 0) * OS_MAXP + b];
                    if (kk < klast && rb2[(pr * 2 + 1) * OS_MAXP + b] <= g) {
                        bk = rb2[(pr * 2 + 1) * OS_MAXP + b];
                        kk++;
                        while (kk < klast) {
                            const unsigned long long bn = os_load_status(p.rt + (uint64_t) (kk + 1) * P + b) & ~OS_RT_VALID;
                            if (bn > g) break;
                            bk = bn;
                            kk++;
                        }
                    }
                    if (kk < p.max_rounds && g - bk < pcap[b])
                        reinterpret_cast<uint4 *>(pbase[b] + (unsigned long long) kk * p.image_stride)[g - bk] = srec[s];
                }
            }
        }
    };

    while (tile < p.ntiles) {
        uint32_t *wc = wcnt + par * CNT;
        uint32_t meta[ITEMS]; // dest | rank<<8 | count<<16 | valid<<31
        const uint64_t idx0 = (uint64_t) tile * TILE + (uint32_t) w * (ITEMS * 64) + lane;
        if (VAR & OSV_DIGIT) {
            os_digit_rows<ITEMS, NB>(p, r, idx0, wc + w * OS_MAXP, meta);
        }
        else if (p.fm.pow2) {
            if (p.klen > 12) os_hash_rows<ITEMS, NB, true, true>(p, r, idx0, wc + w * OS_MAXP, meta);
            else os_hash_rows<ITEMS, NB, true, false>(p, r, idx0, wc + w * OS_MAXP, meta);
        }
        else {
            if (p.klen > 12) os_hash_rows<ITEMS, NB, false, true>(p, r, idx0, wc + w * OS_MAXP, meta);
            else os_hash_rows<ITEMS, NB, false, false>(p, r, idx0, wc + w * OS_MAXP, meta);
        }
        if (!S_ND && p.dest_out) {
#pragma unroll
            for (int k = 0; k < ITEMS; k++)
                if ((meta[k] >> 31) != 0) p.dest_out[idx0 + (uint32_t) k * 64] = (uint8_t) (meta[k] & 0xffu);
        }
        {
            static_assert(ITEMS == 8, "a lane's destinations are packed into one 8-byte word");
            uint32_t dlo = 0, dhi = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                dlo |= ((meta[k] >> 31) ? (meta[k] & 0xffu) : 0xffu) << (8 * k);
                dhi |= ((meta[k + 4] >> 31) ? (meta[k + 4] & 0xffu) : 0xffu) << (8 * k);
            }
            sde[par * WTHREADS + tid] = ((unsigned long long) dhi << 32) | dlo;
        }
        if (VAR & OSV_EARLY_AGG) {
            // The last worker wave to finish hashing publishes the tile's counts, before
            // barrier A: a scan wave that arrives late at the barrier (a long scan for the
            // previous tile) then holds back only its own workgroup, not every later tile
            // whose scan needs these counts.
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
        __syncthreads(); // A: counts(tile) complete; image(prev) and position(prev) ready
        // the phases that issue memory operations go before the other waves' hashing: the
        // stores of the previous image, the reorder that frees the registers, the next loads
        __builtin_amdgcn_s_setprio(1);

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
                sdest[base + rank] = (uint8_t) d;
            }
        }
        // the registers are free again: fetch the next tile
        if (next < p.ntiles) os_load_tile<ITEMS, VAR>(p, next, TILE, w, lane, r);
        __builtin_amdgcn_s_setprio(0);
        prev = tile;
        tile = next;
        par ^= 1;
    }
    __syncthreads(); // image(prev) and position(prev) ready
    if (prev != 0xFFFFFFFFu) store_image(prev, par ^ 1);
    (void) nbits;
}

// Result table of a round-cutting call, 8-byte words, R = max_rounds:
//   [0] rounds begun (<= R)   [1] first record not consumed (absolute index)
//   [2] records packed into send images   [3] local-bypass records   [4] their bytes
//   [8 + k]          index of the first record after round k (absolute)
//   [8 + R + k]      local-bypass records emitted during round k
//   [8 + 2R + k]     their bytes
//   [8 + 3R + k*P + b]  bytes of partition b in image k (send_offset of that round)
constexpr int OS_RTAB_HEAD = 8;
__host__ __device__ inline size_t os_rtab_words(uint32_t max_rounds, uint32_t P)
{
    return (size_t) OS_RTAB_HEAD + (size_t) max_rounds * (3 + (size_t) P);
}

// Columns of the chained scan: PC of them, the first P count partition units of
// 1 << unit_shift bytes; lrec_col counts local-bypass records (for 16-byte records the
// local partition's own column, else an extra column).
#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
__global__ __launch_bounds__(256) void k_os_rounds_final(OsParams p, uint32_t P, uint32_t lrec_col,
                                                         unsigned long long *tab)
{
    const uint32_t PC = p.P, R = p.max_rounds;
    const uint32_t cuts = p.ctrl->head.rounds_cut;
    const uint32_t n_rounds = cuts + 1 < R ? cuts + 1 : R;
    const bool all = cuts < R; // every record was consumed
    auto begin_of = [&](uint32_t k, uint32_t col) -> unsigned long long { // totals at the first record of round k
        return k ? (p.rt[(uint64_t) k * PC + col] & ~OS_RT_VALID) : 0ull;
    };
    auto end_of = [&](uint32_t k, uint32_t col) -> unsigned long long { // totals at the end of round k
        if (k + 1 <= cuts) return p.rt[(uint64_t) (k + 1) * PC + col] & ~OS_RT_VALID;
        return p.status[(uint64_t) (p.ntiles - 1) * PC + col] & OS_VALUE_MASK;
    };
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (uint64_t t = gtid; t < (uint64_t) n_rounds * (P + 1); t += gsz) {
        const uint32_t k = (uint32_t) (t / (P + 1)), b = (uint32_t) (t % (P + 1));
        if (b == P) {
            const unsigned long long nx = (k + 1 <= cuts) ? (p.rt_start[k + 1] & ~OS_RT_VALID) : p.n;
            tab[OS_RTAB_HEAD + k] = p.first + nx;
            unsigned long long lrec = 0, lbytes = 0;
            if (p.self >= 0) {
                lrec = end_of(k, lrec_col) - begin_of(k, lrec_col);
                lbytes = (end_of(k, (uint32_t) p.self) - begin_of(k, (uint32_t) p.self)) << p.unit_shift;
            }
            tab[OS_RTAB_HEAD + R + k] = lrec;
            tab[OS_RTAB_HEAD + 2 * (uint64_t) R + k] = lbytes;
        }
        else {
            const unsigned long long u = ((int32_t) b == p.self) ? 0ull : end_of(k, b) - begin_of(k, b);
            tab[OS_RTAB_HEAD + 3 * (uint64_t) R + (uint64_t) k * P + b] = u << p.unit_shift;
        }
    }
    if (gtid == 0) {
        const unsigned long long nx = all ? p.n : (p.rt_start[R] & ~OS_RT_VALID);
        unsigned long long lrec = 0, lbytes = 0;
        if (p.self >= 0) {
            lrec = end_of(n_rounds - 1, lrec_col);
            lbytes = end_of(n_rounds - 1, (uint32_t) p.self) << p.unit_shift;
        }
        tab[0] = n_rounds;
        tab[1] = p.first + nx;
        tab[2] = nx - lrec;
        tab[3] = lrec;
        tab[4] = lbytes;
        if (p.clip_out) p.clip_out[0] = nx;
    }
    if (p.clip_out && gtid < PC) p.clip_out[1 + gtid] = end_of(n_rounds - 1, gtid);
}
#endif

} // namespace mgs
#endif
