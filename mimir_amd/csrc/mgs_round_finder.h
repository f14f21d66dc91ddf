// Streams layout: where do the reference's rounds begin?
//
// CollectiveShuffler::write (src/collectiveshuffler.h:131-166) ends a round before the
// first record that does not fit its partition of buf_size bytes, exchanges, and starts
// every partition again from empty.  A sweep that has to know the current round before
// it may store a record stalls at every cut: the tiles behind the cut wait for the one
// tile that holds it.  In the streams layout the sweep does not know about rounds at
// all.  It packs the whole batch as ONE round whose "partitions" are the caller's
// destination streams (as long as the caller likes), and the rounds are views into the
// streams: with C_b(r) the units of column b before record r (the chained scan's running
// totals, never reset) and B_k[b] = C_b(first record of round k), round k's partition b
// is bytes [B_k[b], B_{k+1}[b]) of stream b -- byte for byte what the reference's send
// buffer holds for that partition in its k-th exchange.
//
// What is left is to find the B_k.  Workgroup 0 of the sweep's grid does that instead of
// sweeping: it trails the chained scan, reading the inclusive prefixes the scan waves
// publish (OsParams::status).  Round k is cut at the first record r (destination b, not
// the local rank) with C_b(r) + size(r) > B_k[b] + capacity.  Totals only grow, so the
// tile that holds the cut is the first one whose inclusive prefix exceeds a limit: a
// monotone predicate, found by probing FR_LOOK * (64 / columns) tiles per step at a
// spacing that follows the previous round's length.  That one tile is then hashed again
// (all waves of the workgroup, a row of 64 records each) and walked for the exact
// record; B_{k+1} and the record's index go to the round table.  The sweep never waits
// for the finder, so the two cannot deadlock whatever the dispatch order; the finder
// waits for the sweep with a bounded spin and leaves when the sweep has raised an error.
#ifndef MGS_ROUND_FINDER_H
#define MGS_ROUND_FINDER_H

#include "mgs_onesweep16.h"

namespace mgs {

constexpr int FR_LOOK = 8;                 // status loads in flight per probe step
constexpr int FR_MAX_ROWS = 56;            // rows of 64 records in the largest tile (3584 records)
constexpr int FR_MAX_TILE = FR_MAX_ROWS * 64;
constexpr uint32_t FR_EXIT = 0xFFFFFFFFu;

template <int ROWS> // rows of 64 records in a tile
struct FinderLdsT {
    uint32_t rowcnt[ROWS][OS_MAXP];  // units per (row of the tile, column)
    uint32_t units[ROWS * 64];       // units of every record of the tile (0: no record)
    uint8_t dest[ROWS * 64];         // its destination (0xFF: no record)
    unsigned long long cnt[OS_MAXP]; // units per column of a row's lanes before a cut
    volatile uint32_t cmd[4];        // [0] tile to examine next, or FR_EXIT
};

// 16-byte records straight from global memory.
// A source hands out a record in two steps, so that a wave can have the loads of all its
// rows in flight before it looks at any of them: fetch (loads only), then info.
struct FinderSrc16 {
    const OsParams &p;
    typedef uint4 Pre;
    __device__ void stage(uint32_t, int, int) {}
    __device__ Pre fetch(uint32_t tile, uint32_t i, bool &valid) const
    {
        const uint64_t idx = (uint64_t) tile * p.tile_records + i;
        valid = i < p.tile_records && idx < p.n;
        return valid ? p.rec[idx] : make_uint4(0, 0, 0, 0);
    }
    __device__ void info(const Pre &r, uint32_t &d, uint32_t &u, bool &valid) const
    {
        d = valid ? fastmod(hash_rec16(r.x, r.y, r.z, r.w, p.klen), p.fm) : 0u;
        u = 1;
    }
};

// Wave 0's search for the tile that holds the next cut.  lane = column for `lim`; for the
// probes lane = (slot j of a load, column c).
struct FrSearch {
    uint32_t PC, npad, Pp, tpl, M, j, c, gshift;
    uint64_t gmask;
    bool col_ok, c_part;
    int64_t lo;       // tiles 0 .. lo hold no cut of the current round
    int64_t last_cut; // tile of the previous cut
    int64_t pred;     // where the next cut is expected (the previous round's length on), or -1
    uint64_t S;       // spacing of the probes, in tiles
    bool finished;

    __device__ __forceinline__ void init(const OsParams &p, uint32_t P, int lane)
    {
        PC = p.P;
        npad = p.nbits_pad;
        Pp = 1u << npad;
        tpl = 64u >> npad;
        M = (uint32_t) FR_LOOK * tpl;
        j = (uint32_t) lane >> npad;
        c = (uint32_t) lane & (Pp - 1u);
        col_ok = c < PC;
        c_part = c < P && (int32_t) c != p.self;
        gmask = (Pp == 64) ? ~0ull : ((1ull << Pp) - 1ull);
        gshift = j << npad;
        lo = -1;
        last_cut = 0;
        pred = -1;
        S = 16;
        finished = false;
    }

    // The first tile whose inclusive prefix exceeds a limit (lim: lane = column), with that
    // prefix in hi_val (lane = column); FR_EXIT when there is none, or the sweep has failed.
    __device__ __forceinline__ uint32_t find(const OsParams &p, unsigned long long lim, int lane, unsigned long long &hi_val)
    {
        if (finished) return FR_EXIT;
        const unsigned long long lim_c = (unsigned long long) __shfl((long long) lim, (int) c);
        int64_t hi = (int64_t) p.ntiles; // first tile known to hold a cut (ntiles: none known)
        // probes at pb + (slot + 1) * S.  Rounds of one batch are about equally long: when the last
        // one was longer than a step's reach the first step looks at M consecutive tiles around the
        // place where the cut is expected, which nearly always holds it
        int64_t pb = lo;
        if (pred >= 0 && pred - (int64_t) (M / 2) > lo) {
            pb = pred - (int64_t) (M / 2);
            S = 1;
        }
        uint32_t idle = 0;
        while (hi - lo > 1) {
            const int64_t top = hi - 1; // probes stay inside (lo, hi)
            unsigned long long v[FR_LOOK];
#pragma unroll
            for (int q = 0; q < FR_LOOK; q++) {
                int64_t t = pb + (int64_t) (((uint64_t) q * tpl + j + 1u) * S);
                t = t < top ? t : top;
                v[q] = col_ok ? os_load_status(p.status + (uint64_t) t * PC + c) : OS_FLAG_INCL;
            }
            // the first probed tile that is not known to lie before the cut
            bool hit = false, hit_cross = false;
            uint32_t slot = 0;
            unsigned long long hv = 0;
#pragma unroll
            for (int q = 0; q < FR_LOOK; q++) {
                const bool pub = (v[q] >> 62) == 2;
                const bool crossing = pub && c_part && (v[q] & OS_VALUE_MASK) > lim_c;
                const uint64_t m_pub = __ballot(pub), m_cr = __ballot(crossing);
                const bool s_pub = ((m_pub >> gshift) & gmask) == gmask;
                const bool s_cr = ((m_cr >> gshift) & gmask) != 0;
                const uint64_t nbm = __ballot(!s_pub || s_cr);
                if (nbm && !hit) {
                    const uint32_t fj = (uint32_t) (__ffsll((unsigned long long) nbm) - 1) >> npad;
                    hit = true;
                    slot = (uint32_t) q * tpl + fj;
                    hit_cross = ((__ballot(s_pub && s_cr) >> (fj << npad)) & 1ull) != 0;
                    hv = (unsigned long long) __shfl((long long) v[q], (int) ((fj << npad) | ((uint32_t) lane & (Pp - 1u))));
                }
            }
            if (!hit) { // every probe lies before the cut
                int64_t t = pb + (int64_t) ((uint64_t) M * S);
                lo = t < top ? t : top;
                pb = lo;
                if (S < (1ull << 24)) S *= 2;
                idle = 0;
                continue;
            }
            int64_t t_f = pb + (int64_t) (((uint64_t) slot + 1u) * S);
            t_f = t_f < top ? t_f : top;
            int64_t t_prev = pb + (int64_t) ((uint64_t) slot * S);
            t_prev = t_prev < top ? t_prev : top;
            if (t_prev >= t_f) t_prev = lo; // clamped probes of one tile: nothing new is known
            if (hit_cross) {
                hi = t_f;
                hi_val = hv & OS_VALUE_MASK;
                if (slot) lo = t_prev > lo ? t_prev : lo; // (slot 0: tile pb itself was not probed)
                pb = lo;
                const uint64_t gap = (uint64_t) (hi - lo - 1);
                S = (gap + M - 1) / M;
                if (S == 0) S = 1;
                idle = 0;
                continue;
            }
            // not published yet: the sweep has not got there
            if (slot && t_prev > lo) {
                lo = t_prev;
                pb = lo;
                idle = 0;
            }
            if (++idle > OS_SPIN_LIMIT) {
                raise_error(p.ctrl, MGS_ERR_TIMEOUT);
                finished = true;
                break;
            }
            if ((idle & 63u) == 0
                && __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                finished = true;
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        if (!finished && hi < (int64_t) p.ntiles) return (uint32_t) hi;
        return FR_EXIT;
    }

    // The tile's cuts have been found: the next search starts behind it; its probes span 1.25
    // times the length of the round just found.
    __device__ __forceinline__ void next_round(uint32_t tile)
    {
        const uint64_t len = (uint64_t) ((int64_t) tile - last_cut) + 1u;
        S = (len + len / 4 + M - 1) / M;
        if (S == 0) S = 1;
        pred = len > (uint64_t) M ? (int64_t) tile + (int64_t) len - 1 : -1;
        last_cut = (int64_t) tile;
        lo = (int64_t) tile;
    }
};

// Destinations and units of rows of a tile, units per (row, column): wave `wrel` of NWS
// takes rows row0 + wrel, row0 + wrel + NWS, ... below row1.
template <class Src, int NWS, int NROWS, int MAXROWS>
__device__ __forceinline__ void fr_stage_rows(const OsParams &p, uint32_t P, const Src &src, FinderLdsT<MAXROWS> *x,
                                              uint32_t tile, uint32_t row0, uint32_t row1, uint32_t wrel, int lane,
                                              bool extra_col)
{
    constexpr int RPW = (NROWS + NWS - 1) / NWS; // row1 - row0 <= NROWS
    typename Src::Pre pre[RPW];
    bool pv[RPW];
#pragma unroll
    for (int kk = 0; kk < RPW; kk++) {
        const uint32_t r = row0 + wrel + (uint32_t) kk * NWS;
        pv[kk] = false;
        if (r < row1) pre[kk] = src.fetch(tile, r * 64u + (uint32_t) lane, pv[kk]);
    }
#pragma unroll
    for (int kk = 0; kk < RPW; kk++) {
        const uint32_t r = row0 + wrel + (uint32_t) kk * NWS;
        if (r >= row1) continue;
        x->rowcnt[r][lane] = 0;
        __builtin_amdgcn_wave_barrier();
        const uint32_t i = r * 64u + (uint32_t) lane;
        uint32_t d = 0, u = 0;
        bool valid = pv[kk];
        if (valid) src.info(pre[kk], d, u, valid);
        x->dest[i] = valid ? (uint8_t) d : (uint8_t) 0xFF;
        x->units[i] = valid ? u : 0u;
        if (valid) {
            atomicAdd(&x->rowcnt[r][d], u);
            if (extra_col && (int32_t) d == p.self) atomicAdd(&x->rowcnt[r][P], 1u);
        }
    }
}

// Wave 0, lane = column: every cut in rows [r0, rows) of the staged tile, in emit order.
// run: units before row r0; hi_val: units at the end of row rows - 1; lim: the current
// round's limits; k: cuts found so far.  Leaves lim / k at the state after the rows (and, with
// want_end, run at the units at their end); false when the search is over (error, or the caller's tables are full).
template <int MAXROWS>
__device__ __forceinline__ bool fr_walk(const OsParams &p, uint32_t P, FinderLdsT<MAXROWS> *x, uint32_t tile, uint32_t r0,
                                        uint32_t rows, int lane, bool live, bool is_part, bool extra_col,
                                        unsigned long long &run, unsigned long long hi_val, unsigned long long &lim,
                                        uint32_t &k, bool want_end)
{
    const uint32_t PC = p.P;
    for (;;) {
        // Columns that pass their limit before the rows end (hi_val: the totals at their end).
        // For each (nearly always one) the rows' counts are scanned with lane = row: the
        // first row at whose end the column is past its limit; the cut lies in the earliest.
        uint64_t cc = __ballot(is_part && hi_val > lim);
        if (!cc) break; // no (further) cut in these rows
        uint32_t rs = rows;
        const bool my_row = (uint32_t) lane >= r0 && (uint32_t) lane < rows;
        while (cc) {
            const int b = __ffsll((unsigned long long) cc) - 1;
            cc &= cc - 1;
            const uint32_t inc = wave_incl_scan_u32(my_row ? x->rowcnt[lane][b] : 0u, lane);
            const unsigned long long run_b = (unsigned long long) __shfl((long long) run, b);
            const unsigned long long lim_b = (unsigned long long) __shfl((long long) lim, b);
            const uint64_t cr = __ballot(my_row && run_b + inc > lim_b);
            if (cr) {
                const uint32_t r = (uint32_t) (__ffsll((unsigned long long) cr) - 1);
                rs = r < rs ? r : rs;
            }
        }
        if (rs == rows) break; // cannot happen: the totals said otherwise
        // totals at the start of row rs
        unsigned long long acc = run;
#pragma unroll 8
        for (uint32_t r = r0; r < rs; r++) acc += x->rowcnt[r][lane];
        run = acc;
        const uint32_t i = rs * 64u + (uint32_t) lane;
        const uint32_t d = x->dest[i], u = x->units[i];
        const bool act = d != 0xFFu;
        // Only a column that passes its limit in this row can hold the cut (nearly always one
        // column): per such column, the running units of its records up to and including each
        const uint32_t cn_rs = x->rowcnt[rs][lane];
        uint64_t cand = __ballot(is_part && run + cn_rs > lim);
        bool over = false;
        while (cand) {
            const int b = __ffsll((unsigned long long) cand) - 1;
            cand &= cand - 1;
            const bool inb = act && d == (uint32_t) b;
            const uint32_t inc = wave_incl_scan_u32(inb ? u : 0u, lane);
            const unsigned long long run_b = (unsigned long long) __shfl((long long) run, b);
            const unsigned long long lim_b = (unsigned long long) __shfl((long long) lim, b);
            if (inb && run_b + inc > lim_b) over = true;
        }
        const uint64_t om = __ballot(over);
        if (!om) { // cannot happen: the row's totals said otherwise
            r0 = rs + 1;
            continue;
        }
        const int fl = __ffsll((unsigned long long) om) - 1;
        // totals at record (rs, fl), the first of the next round
        x->cnt[lane] = 0;
        __builtin_amdgcn_wave_barrier();
        if (act && lane < fl) {
            atomicAdd(&x->cnt[d], (unsigned long long) u);
            if (extra_col && (int32_t) d == p.self) atomicAdd(&x->cnt[P], 1ull);
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long before = x->cnt[lane];
        const unsigned long long nb = run + before;
        k++;
        if (live) p.f_rt[(uint64_t) k * PC + lane] = OS_RT_VALID | nb;
        if (lane == 0)
            p.f_rt_start[k] = OS_RT_VALID | ((uint64_t) tile * p.tile_records + (uint64_t) rs * 64u + (uint32_t) fl);
        // a record that does not fit an empty partition (src/collectiveshuffler.h:156-159)
        const uint32_t uf = (uint32_t) __shfl((int) u, fl);
        if ((unsigned long long) uf > p.f_cap) {
            raise_error(p.ctrl, MGS_ERR_RECORD_SIZE);
            return false;
        }
        if (k >= p.f_max_rounds) return false; // the end marker: the caller's tables are full
        if (is_part) lim = nb + p.f_cap;
        // the rest of the row belongs to the new round
        if (lane < fl) {
            x->dest[i] = 0xFF;
            x->units[i] = 0;
        }
        x->rowcnt[rs][lane] -= (uint32_t) before;
        __builtin_amdgcn_wave_barrier();
        run = nb;
        r0 = rs;
    }
    if (want_end) { // the totals at the end of the rows
        unsigned long long acc = run;
#pragma unroll 8
        for (uint32_t r = r0; r < rows; r++) acc += x->rowcnt[r][lane];
        run = acc;
    }
    return true;
}

// NW waves in the workgroup, tiles of at most MAXROWS rows of 64 records.  SEGROWS != 0: the
// sweep leaves its units per worker wave of SEGROWS rows in p.seg, and only the segments that
// hold a cut are hashed again, not the whole tile.
template <class Src, int NW, int MAXROWS, int SEGROWS = 0>
__device__ __forceinline__ void os_round_finder(const OsParams &p, uint32_t P, Src &src, FinderLdsT<MAXROWS> *x, int tid)
{
    constexpr int nthreads = NW * 64;
    static_assert(MAXROWS <= FR_MAX_ROWS, "the finder's tables hold FR_MAX_ROWS rows");
    static_assert(SEGROWS == 0 || (MAXROWS % SEGROWS == 0 && MAXROWS / SEGROWS <= 8), "segments of a tile");
    constexpr int NSEG = SEGROWS ? MAXROWS / SEGROWS : 1;
    const int lane = tid & 63, w = tid >> 6;
    const uint32_t PC = p.P; // columns of the chained scan: P destinations (+ the local record count)
    const uint32_t rows = (p.tile_records + 63u) / 64u;
    const bool extra_col = PC > P; // column P counts the records of the local rank

    // ---- wave 0: the state of the search, lane = column
    const bool live = (uint32_t) lane < PC;
    const bool is_part = (uint32_t) lane < P && lane != p.self;
    unsigned long long lim = ~0ull;
    if (is_part) lim = (unsigned long long) (p.ctrl->head.round_base0[lane] >> p.unit_shift) + p.f_cap;
    uint32_t k = 0; // cuts found
    FrSearch fs;
    fs.init(p, P, lane);
    const unsigned long long wall0 = wall_clock64();
    unsigned long long cy_search = 0, cy_hash = 0, cy_walk = 0;

    if constexpr (SEGROWS == 0) {
        for (;;) {
            unsigned long long hi_val = 0; // lane = column: inclusive prefix of the tile that holds the cut
            unsigned long long cy0 = __builtin_readcyclecounter();
            if (w == 0) {
                const uint32_t found = fs.find(p, lim, lane, hi_val);
                if (lane == 0) x->cmd[0] = found;
            }
            __syncthreads();
            const uint32_t tile = x->cmd[0];
            cy_search += __builtin_readcyclecounter() - cy0;
            if (tile == FR_EXIT) break;
            cy0 = __builtin_readcyclecounter();
            // ---- every wave: destinations and units of the tile's records, units per (row, column)
            src.stage(tile, tid, nthreads);
            fr_stage_rows<Src, NW, MAXROWS, MAXROWS>(p, P, src, x, tile, 0, rows, (uint32_t) w, lane, extra_col);
            __syncthreads();
            cy_hash += __builtin_readcyclecounter() - cy0;
            cy0 = __builtin_readcyclecounter();
            // ---- wave 0: every cut of the tile, in emit order
            if (w == 0) {
                unsigned long long tot = 0;
#pragma unroll 8
                for (uint32_t r = 0; r < rows; r++) tot += x->rowcnt[r][lane];
                unsigned long long run = hi_val - tot; // lane = column: units before the tile
                if (!fr_walk<MAXROWS>(p, P, x, tile, 0, rows, lane, live, is_part, extra_col, run, hi_val, lim, k, false))
                    fs.finished = true;
                fs.next_round(tile);
                cy_walk += __builtin_readcyclecounter() - cy0;
            }
            __syncthreads(); // the tile's tables are free again
        }
    }
    else {
        // ---- segments: wave 0 reads the units per (segment, column) of the tile that holds the cut,
        // lane = column, and names the segments that hold cuts one after the other; every wave
        // hashes a row of the named segment
        unsigned long long run = 0; // wave 0, lane = column: units before the next segment to look at
        uint32_t sc[NSEG];          // wave 0, lane = column: units of the tile's segments
        uint32_t s_next = 0;
        // wave 0: the first segment from s_next on at whose end a column is past its limit -> cmd[1]
        auto name_segment = [&]() {
            uint32_t sfound = FR_EXIT;
            if (!fs.finished) {
                unsigned long long at = run;
#pragma unroll
                for (int sg = 0; sg < NSEG; sg++) {
                    if ((uint32_t) sg < s_next) continue;
                    at += sc[sg];
                    if (sfound == FR_EXIT && __any(is_part && at > lim)) sfound = (uint32_t) sg;
                    if (sfound == FR_EXIT) run = at; // units before the segment that holds the cut
                }
            }
            if (lane == 0) x->cmd[1] = sfound;
        };
        for (;;) {
            unsigned long long hi_val = 0; // lane = column: inclusive prefix of the tile that holds the cut
            unsigned long long cy0 = __builtin_readcyclecounter();
            if (w == 0) {
                // while the search's loads are under way: the counts of the tiles around the place where
                // the cut is expected, lane = (tile, column) as in the probes -- one trip to memory less
                // when the cut is there (rounds of a batch are about equally long)
                uint16_t sv[NSEG];
                int64_t sbase = -1;
                if (fs.pred >= 0) {
                    sbase = fs.pred - (int64_t) (fs.tpl / 2);
                    if (sbase < 0) sbase = 0;
                    int64_t t = sbase + (int64_t) fs.j;
                    if (t >= (int64_t) p.ntiles) t = (int64_t) p.ntiles - 1;
#pragma unroll
                    for (int sg = 0; sg < NSEG; sg++)
                        sv[sg] = fs.col_ok ? __hip_atomic_load(p.seg + ((uint64_t) t * NSEG + sg) * PC + fs.c, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT)
                                           : (uint16_t) 0;
                }
                const uint32_t found = fs.find(p, lim, lane, hi_val);
                if (found != FR_EXIT) {
                    bool have = false;
                    if (sbase >= 0 && (int64_t) found >= sbase && (int64_t) found < sbase + (int64_t) fs.tpl
                        && sbase + (int64_t) fs.tpl <= (int64_t) p.ntiles) {
                        const int from = (int) ((((uint32_t) ((int64_t) found - sbase)) << fs.npad) | ((uint32_t) lane & (fs.Pp - 1u)));
                        bool missing = false;
#pragma unroll
                        for (int sg = 0; sg < NSEG; sg++) {
                            const uint32_t v = (uint32_t) __shfl((int) sv[sg], from);
                            sc[sg] = live ? v : 0u;
                            missing = missing || (live && v == 0xFFFFu);
                        }
                        have = !__any(missing);
                    }
                    uint32_t idle = 0;
                    for (; !have;) {
                        bool missing = false;
#pragma unroll
                        for (int sg = 0; sg < NSEG; sg++) {
                            const uint16_t v = live ? __hip_atomic_load(p.seg + ((uint64_t) found * NSEG + sg) * PC + lane,
                                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                    : (uint16_t) 0;
                            sc[sg] = v;
                            missing = missing || v == 0xFFFFu;
                        }
                        if (!__any(missing)) break;
                        // the tile's prefix is out but its counts per worker wave are not yet
                        if (++idle > OS_SPIN_LIMIT) {
                            raise_error(p.ctrl, MGS_ERR_TIMEOUT);
                            fs.finished = true;
                            break;
                        }
                        if ((idle & 63u) == 0
                            && __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                            fs.finished = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    unsigned long long tot = 0;
#pragma unroll
                    for (int sg = 0; sg < NSEG; sg++) tot += sc[sg];
                    run = hi_val - tot;
                    s_next = 0;
                    name_segment();
                }
                if (lane == 0) x->cmd[0] = found;
            }
            __syncthreads();
            const uint32_t tile = x->cmd[0];
            cy_search += __builtin_readcyclecounter() - cy0;
            if (tile == FR_EXIT) break;
            for (;;) {
                const uint32_t sgm = x->cmd[1]; // (written again only behind the next barrier)
                if (sgm == FR_EXIT) break;
                cy0 = __builtin_readcyclecounter();
                const uint32_t row0 = sgm * (uint32_t) SEGROWS;
                const uint32_t row1 = min(rows, row0 + (uint32_t) SEGROWS);
                fr_stage_rows<Src, NW, (SEGROWS ? SEGROWS : 1), MAXROWS>(p, P, src, x, tile, row0, row1, (uint32_t) w, lane,
                                                                         extra_col);
                __syncthreads();
                cy_hash += __builtin_readcyclecounter() - cy0;
                if (w == 0) {
                    cy0 = __builtin_readcyclecounter();
                    uint32_t in_seg = 0;
#pragma unroll
                    for (int sg = 0; sg < NSEG; sg++)
                        if ((uint32_t) sg == sgm) in_seg = sc[sg];
                    const unsigned long long seg_end = run + in_seg;
                    if (!fr_walk<MAXROWS>(p, P, x, tile, row0, row1, lane, live, is_part, extra_col, run, seg_end, lim, k, true))
                        fs.finished = true;
                    s_next = sgm + 1;
                    name_segment();
                    cy_walk += __builtin_readcyclecounter() - cy0;
                }
                __syncthreads(); // the rows' tables are free again, the next segment is named
            }
            if (w == 0) fs.next_round(tile);
        }
    }
    if (w == 0 && lane == 0) {
        p.ctrl->head.f_cuts = k;
        p.ctrl->head.fstats[0] = wall_clock64() - wall0;
        p.ctrl->head.fstats[1] = cy_search;
        p.ctrl->head.fstats[2] = cy_hash;
        p.ctrl->head.fstats[3] = cy_walk;
    }
}

// Result table of a streams-layout call: the table of a round-cutting call (see
// os_rtab_words) followed by
//   [8 + 3R + R*P + k*P + b]  byte of stream b at which round k's partition begins
__host__ __device__ inline size_t os_stab_words(uint32_t max_rounds, uint32_t P)
{
    return (size_t) 8 + (size_t) max_rounds * (3 + 2 * (size_t) P);
}

// After the sweep and the kernel that finished its one round (p.clip_out: the records that
// were packed and the totals at that point): which of the finder's rounds were packed, and
// the per-round arrays.  A round that begins at or after the first record that was not
// packed is not reported; the last reported round is open.
#ifndef MGS_TEMPLATES_ONLY // a translation unit that only instantiates kernel templates skips the plain kernels
__global__ __launch_bounds__(256) void k_os_streams_final(OsParams p, uint32_t P, uint32_t lrec_col,
                                                          unsigned long long *tab)
{
    const uint32_t PC = p.P, R = p.f_max_rounds;
    const uint32_t K = p.ctrl->head.f_cuts < R ? p.ctrl->head.f_cuts : R;
    const unsigned long long clip = p.clip_out[0];
    // rounds that begin before record `clip` (starts grow with k)
    uint32_t lo = 0, hi = K; // invariant: rounds 1..lo begin before clip, rounds hi+1..K do not
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) / 2;
        if ((p.f_rt_start[mid] & ~OS_RT_VALID) < clip) lo = mid;
        else hi = mid - 1;
    }
    const uint32_t m = lo + 1;
    const bool marker = m > R; // round R begins before the clip: the tables are full, it is the end marker
    const uint32_t n_rounds = marker ? R : m;
    const unsigned long long nx = marker ? (p.f_rt_start[R] & ~OS_RT_VALID) : clip;
    auto begin_of = [&](uint32_t k, uint32_t col) -> unsigned long long {
        if (k) return p.f_rt[(uint64_t) k * PC + col] & ~OS_RT_VALID;
        return (col < P && (int32_t) col != p.self) ? (unsigned long long) (p.ctrl->head.round_base0[col] >> p.unit_shift)
                                                    : 0ull;
    };
    auto end_of = [&](uint32_t k, uint32_t col) -> unsigned long long {
        if (k + 1 < m) return p.f_rt[(uint64_t) (k + 1) * PC + col] & ~OS_RT_VALID;
        return p.clip_out[1 + col];
    };
    const size_t base_at = (size_t) 8 + 3 * (size_t) R + (size_t) R * P;
    const uint32_t gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (uint64_t t = gtid; t < (uint64_t) n_rounds * (P + 1); t += gsz) {
        const uint32_t k = (uint32_t) (t / (P + 1)), b = (uint32_t) (t % (P + 1));
        if (b == P) {
            const unsigned long long e = (k + 1 < n_rounds || marker) ? (p.f_rt_start[k + 1] & ~OS_RT_VALID) : clip;
            tab[8 + k] = p.first + e;
            unsigned long long lrec = 0, lbytes = 0;
            if (p.self >= 0) {
                lrec = end_of(k, lrec_col) - begin_of(k, lrec_col);
                lbytes = (end_of(k, (uint32_t) p.self) - begin_of(k, (uint32_t) p.self)) << p.unit_shift;
            }
            tab[8 + R + k] = lrec;
            tab[8 + 2 * (uint64_t) R + k] = lbytes;
        }
        else {
            const bool self = (int32_t) b == p.self;
            const unsigned long long u = self ? 0ull : end_of(k, b) - begin_of(k, b);
            tab[8 + 3 * (uint64_t) R + (uint64_t) k * P + b] = u << p.unit_shift;
            tab[base_at + (uint64_t) k * P + b] = self ? 0ull : begin_of(k, b) << p.unit_shift;
        }
    }
    if (gtid == 0) {
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
    }
}
#endif

} // namespace mgs
#endif
