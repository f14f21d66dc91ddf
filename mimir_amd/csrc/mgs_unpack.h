// Receive side: record boundaries of a wire-form stream on the device, and the
// page cuts of the container that stores it.
//
// What it replaces: CollectiveShuffler::save_data walks the receive buffer
// record by record (kv_from_bytes: strlen for a string key, sizeof for fixed
// parts; src/collectiveshuffler.h:563-585, src/serializer.h:376-403) and
// KVContainer::write appends every record to the current 64 MiB page, opening
// a new page when a record does not fit (src/kvcontainer.h:133-160).  Pages
// hold whole records back to back, so a page is a contiguous slice of the
// stream: all that has to be computed is where the records start and at
// which records new pages begin.
//
// Records of a string key followed by a V-byte value cannot be found by
// looking for NULs alone (value bytes may be zero), the parse is a chain:
// after a key's NUL skip V bytes, the next key starts there.  The chain is cut
// twice: the stream into chunks, one wave each, and a chunk into 64 pieces, one
// lane each, so that the walk over the NUL bits is vector work (as wave-uniform
// scalar work it is bound by the one scalar unit of a compute unit).
//
//   k_unp_spec   a lane parses its piece for every state it could be entered in
//                (x = 0..V bytes still to skip, or the middle of a key).  Entry
//                states that have led to the same state are stepped once, and
//                once all have met (after a record or two they have) the lane
//                goes on with one parse.  The 64 piece tables (entry state ->
//                exit state, records) are kept for k_unp_emit and composed into
//                the chunk's table.
//                The chain through the chunks is cut into runs of 64 chunks: the
//                workgroup that finishes a run last composes the run's table, and
//                the one that finishes the last run follows state 0 through the
//                run tables -> entry state and first record index of every run.
//   k_unp_emit   follows its run's entry state through the chunks before its own,
//                walks the chunk's entry state through the piece tables, then
//                every lane parses its piece again from its true entry state and
//                writes the record start offsets.
//   k_unp_pages  one wave: greedy page cuts by a search in the offsets.
//
// Reading: per 64-byte step the wave fetches 64 rows (the lanes' current blocks)
// as 16 rows x 64 contiguous bytes per load instruction, turns every 16 bytes
// into a 16-bit NUL mask where they land, and hands the masks to their rows'
// lanes through LDS (2 bytes per 16 instead of the bytes themselves).
#ifndef MGS_UNPACK_H
#define MGS_UNPACK_H

#include "mgs_common.h"

namespace mgs {

constexpr int UNP_MAX_V = 62;        // value bytes handled: V + 2 entry states
constexpr int UNP_MAX_CHUNKS = 4096; // one wave each: four per SIMD on a full device
constexpr int UNP_PIECES = 64;       // pieces of a chunk, one lane each
constexpr uint32_t UNP_RUN = 64;     // chunks whose tables are composed into one run table

struct UnpParams {
    const uint8_t *stream;
    uint64_t nbytes;
    uint32_t V;           // bytes to skip after a key's NUL
    uint32_t chunk_bytes; // 64 pieces
    uint32_t piece_bytes; // multiple of 128 * UNP_AHEAD, below 65536 (record counts of a piece are 16-bit)
    uint32_t nchunks;
    uint32_t *table;      // [nchunks][V+2]: exit state << 24 | records, per entry state
    uint8_t *piece_exit;  // [nchunks][64][V+2]
    uint16_t *piece_count; // [nchunks][64][V+2]
    unsigned long long *masks; // [nchunks][piece_bytes / 128][64][2]: NUL masks of every pair of 64-byte blocks of
                               // every piece, left by k_unp_spec for k_unp_emit (an eighth of the stream, so
                               // that it is read once); 16-byte aligned
    uint32_t nruns;       // runs of UNP_RUN chunks
    uint32_t *run_exit;   // [nruns][V+2] exit state of a run per entry state
    uint64_t *run_count;  // [nruns][V+2] records of a run per entry state
    uint32_t *run_entry;  // [nruns] entry state of each run
    uint64_t *run_first;  // [nruns] index of the run's first record start
    uint32_t *sync;       // zeroed per call: [0] runs composed, [2 + r] chunks of run r parsed
    uint32_t *rec_off;    // out: rec_off[i] = start of record i; rec_off[n] = nbytes
    uint64_t rec_cap;     // entries available in rec_off
    Ctrl *ctrl;
    // pages
    int64_t page_bytes;
    int64_t first_fill;
    uint64_t *cuts;       // out: [2*max_cuts] (first_record, first_byte) of every new page
    uint32_t max_cuts;
    uint64_t *result;     // out: [0] records, [1] number of cuts, [2] the first status raised (k_unp_pages)
};

struct __attribute__((packed)) UnpU128 { uint32_t x, y, z, w; };

// The lane's 16 bytes p.stream[at .. at+16), fetched without ever reading past the
// stream: a piece that sticks out is fetched 16 bytes back from the end instead
// (unp_m16 shifts it into place).  One load and nothing else: with a second,
// byte-wise path in here for streams under 16 bytes the compiler waited for every
// load of the parse loop right where it was issued (the counters of both paths are
// merged where they join).  Such streams take k_unp_tiny.
// EDGE = false: the caller knows that at + 16 <= nbytes.
template <bool EDGE = true>
__device__ __forceinline__ uint4 unp_load16(const UnpParams &p, uint64_t at)
{
    const uint64_t a = EDGE ? min(at, p.nbytes - 16) : at;
    const UnpU128 v = *reinterpret_cast<const UnpU128 *>(p.stream + a);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Bit i of the result: byte i of the lane's 16 bytes at p.stream[at .. at+16) is a
// NUL (bytes past the end of the stream count as non-NUL).  v = unp_load16(p, at).
template <bool EDGE = true>
__device__ __forceinline__ uint32_t unp_m16(const UnpParams &p, uint64_t at, const uint4 &v)
{
    auto nibble = [](uint32_t w) { // one bit per zero byte, lowest address first
        const uint32_t t = ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu); // 0x80 per zero byte
        return (((t >> 7) * 0x01020408u) >> 24) & 0xFu;
    };
    const uint32_t m = nibble(v.x) | (nibble(v.y) << 4) | (nibble(v.z) << 8) | (nibble(v.w) << 12);
    if (!EDGE) return m;
    if (at >= p.nbytes) return 0;
    if (at + 16 > p.nbytes) return m >> (uint32_t) (at - (p.nbytes - 16));
    return m;
}

// Blocks 2s and 2s+1 of all 64 pieces of the chunk at c0: load j (0..7) of a lane covers 16
// bytes of row 8j + lane/8, so one load instruction reads 8 rows x 128 contiguous bytes.
// (With 64 bytes of a row per instruction -- one block at a time -- the kernel ran at 3.6 TB/s
// whatever was done to the parse or to the number of loads in flight.)
struct UnpRows {
    uint4 raw[8];
};

__device__ __forceinline__ uint64_t unp_row_at(const UnpParams &p, uint64_t c0, uint32_t s, int j, int lane)
{
    return c0 + (uint64_t) (8 * j + (lane >> 3)) * p.piece_bytes + 128u * s + 16u * (uint32_t) (lane & 7);
}

template <bool EDGE>
__device__ __forceinline__ void unp_rows_load(const UnpParams &p, uint64_t c0, uint32_t s, int lane, UnpRows &r)
{
#pragma unroll
    for (int j = 0; j < 8; j++) r.raw[j] = unp_load16<EDGE>(p, unp_row_at(p, c0, s, j, lane));
}

// The masks of the rows just loaded go to LDS (512 x 16 bit); the lane's own two
// 64-bit masks come back.
template <bool EDGE>
__device__ __forceinline__ void unp_rows_mask(const UnpParams &p, uint64_t c0, uint32_t s, int lane, const UnpRows &r,
                                              uint16_t *smask, uint64_t &nul0, uint64_t &nul1)
{
#pragma unroll
    for (int j = 0; j < 8; j++)
        smask[64 * j + lane] = (uint16_t) unp_m16<EDGE>(p, unp_row_at(p, c0, s, j, lane), r.raw[j]);
    __syncthreads();
    const uint4 m = reinterpret_cast<const uint4 *>(smask)[lane];
    __syncthreads();
    nul0 = (uint64_t) m.x | ((uint64_t) m.y << 32);
    nul1 = (uint64_t) m.z | ((uint64_t) m.w << 32);
}

// One 64-byte block for one state.  state: bytes to skip before the next key
// starts (0..V), or UNP_INKEY when the block is entered in the middle of a key.
// put(pos) is called wherever a key begins inside the `valid` stream bytes of
// the block.  Returns
// the state at the block's end (for a partial last block: the bytes still
// owed, 0 = clean end; for valid = 0 the state itself).
constexpr uint32_t UNP_INKEY = 0x80000000u;

template <class Put>
__device__ __forceinline__ uint32_t unp_step(uint64_t nul, uint32_t state, uint32_t V, uint32_t valid, Put &&put)
{
    uint32_t pos;
    if (state & UNP_INKEY) {
        if (nul == 0) return UNP_INKEY;
        pos = (uint32_t) __ffsll((unsigned long long) nul) /* index of the NUL + 1 */ + V;
    }
    else {
        pos = state;
    }
    while (pos < valid) {
        put(pos); // a key begins here
        const uint64_t mm = nul & (~0ull << pos);
        if (mm == 0) return UNP_INKEY; // the key runs on into the next block
        pos = (uint32_t) __ffsll((unsigned long long) mm) + V;
    }
    return pos - valid;
}

// Words handed from one workgroup of a launch to another go past the caches of the compute
// units' own die in both directions (unp_post, unp_peek), so no fence has to write a whole
// cache back -- with hundreds of megabytes of masks on their way out that costs 0.2 ms a launch.
template <class T>
__device__ __forceinline__ void unp_post(T *ptr, T v)
{
    __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <class T>
__device__ __forceinline__ T unp_peek(const T *ptr)
{
    return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wave: true in every lane for the workgroup that arrives last of `expect` at *counter.
// What the workgroup posted before has landed when it is counted.
__device__ __forceinline__ bool unp_arrive_last(uint32_t *counter, uint32_t expect, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the posts have been acknowledged
    uint32_t last = 0;
    if (lane == 0) last = atomicAdd(counter, 1u) == expect - 1 ? 1u : 0u;
    return __builtin_amdgcn_readfirstlane((int) last) != 0;
}

// A chunk's table is a function from entry state to (exit state, records) and the
// chain through the chunks is their composition.  Lane x follows entry state x
// through the chunks of run `run` (a row of the table per chunk in registers, lane
// x holding entry x; a step is one shuffle) and leaves the run's table.
constexpr int UNP_BATCH = 32;
#ifndef UNP_AHEAD
#define UNP_AHEAD 2
#endif

__device__ __forceinline__ void unp_compose_run(const UnpParams &p, uint32_t run, int lane)
{
    const uint32_t S = p.V + 2, x = min((uint32_t) lane, S - 1);
    const uint32_t c0 = run * UNP_RUN, len = min(UNP_RUN, p.nchunks - c0);
    uint32_t cur = x;
    uint64_t cnt = 0;
    for (uint32_t b = 0; b < len; b += UNP_BATCH) {
        uint32_t row[UNP_BATCH];
#pragma unroll
        for (int j = 0; j < UNP_BATCH; j++) row[j] = unp_peek(&p.table[(uint64_t) min(c0 + b + j, p.nchunks - 1) * S + x]);
#pragma unroll
        for (int j = 0; j < UNP_BATCH; j++) {
            if (b + j < len) {
                const uint32_t t = (uint32_t) __shfl((int) row[j], (int) cur);
                cnt += t & 0xFFFFFFu;
                cur = t >> 24;
            }
        }
    }
    if ((uint32_t) lane < S) {
        unp_post(&p.run_exit[(uint64_t) run * S + lane], cur);
        unp_post(&p.run_count[(uint64_t) run * S + lane], cnt);
    }
}

// State 0 (the stream starts at a record boundary) through the run tables: entry
// state and first record of every run, the number of records, and whether the
// stream ends where a record ends.
__device__ __forceinline__ void unp_run_prefix(const UnpParams &p, int lane)
{
    const uint32_t S = p.V + 2, x = min((uint32_t) lane, S - 1);
    uint32_t state = 0;
    uint64_t total = 0;
    for (uint32_t r0 = 0; r0 < p.nruns; r0 += UNP_BATCH) {
        uint32_t ex[UNP_BATCH];
        uint64_t cn[UNP_BATCH];
#pragma unroll
        for (int j = 0; j < UNP_BATCH; j++) {
            const uint64_t at = (uint64_t) min(r0 + j, p.nruns - 1) * S + x;
            ex[j] = unp_peek(&p.run_exit[at]);
            cn[j] = unp_peek(&p.run_count[at]);
        }
        uint32_t my_entry = 0;
        uint64_t my_first = 0;
#pragma unroll
        for (int j = 0; j < UNP_BATCH; j++) {
            if (lane == j) {
                my_entry = state;
                my_first = total;
            }
            if (r0 + j < p.nruns) {
                const uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) cn[j], (int) state);
                const uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (cn[j] >> 32), (int) state);
                total += ((uint64_t) hi << 32) | lo;
                state = (uint32_t) __builtin_amdgcn_readlane((int) ex[j], (int) state);
            }
        }
        if (lane < UNP_BATCH && r0 + lane < p.nruns) {
            p.run_entry[r0 + lane] = my_entry;
            p.run_first[r0 + lane] = my_first;
        }
    }
    if (lane == 0) {
        p.result[0] = total;
        // the stream must end at a record boundary: nothing owed, not inside a key
        if (state != 0) raise_error(p.ctrl, MGS_ERR_MALFORMED);
        else if (total + 1 > p.rec_cap) raise_error(p.ctrl, MGS_ERR_ARG);
    }
}

// The blocks of k_unp_spec's chunk c.  EDGE = false: the chunk and 16 bytes behind it are inside the stream.
template <bool EDGE>
__device__ __forceinline__ void unp_spec_blocks(const UnpParams &p, uint32_t c, int lane, uint32_t *sst, uint32_t *scn,
                                                uint16_t *smask, bool &merged, uint32_t &state, uint32_t &dn)
{
    const uint32_t V = p.V, S = V + 2;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint64_t a0 = c0 + (uint64_t) lane * p.piece_bytes;
    const uint64_t a1 = min(p.nbytes, a0 + p.piece_bytes); // a1 <= a0: the piece is empty
    const uint32_t nb = p.piece_bytes / 64;
    // UNP_AHEAD steps of two blocks in flight, each in registers of its own (no moves: a move would wait
    // for its load).  Every load and store of the loop is issued on every trip -- past the last step the
    // last step again -- so that the wait for a step's rows can be a count that leaves the younger
    // requests in flight: with a load under a condition the compiler waits for everything.
    const uint32_t ns = nb / 2; // a multiple of UNP_AHEAD: pieces are multiples of 128 * UNP_AHEAD bytes
    UnpRows ring[UNP_AHEAD];
#pragma unroll
    for (uint32_t j = 0; j < UNP_AHEAD; j++) {
        unp_rows_load<EDGE>(p, c0, j, lane, ring[j]);
        __builtin_amdgcn_sched_barrier(0); // step by step, as the loop requests them: the waits count on the order
    }
    auto block = [&](uint32_t b, uint64_t nul) {
        const uint64_t boff = a0 + 64u * b;
        if (EDGE && boff >= a1) return;
        const uint32_t valid = EDGE ? (uint32_t) min((uint64_t) 64, a1 - boff) : 64u;
        if (merged) {
            state = unp_step(nul, state, V, valid, [&](uint32_t) { dn++; });
            return;
        }
        uint32_t prev_in = 0xFFFFFFFFu, out = 0, dnx = 0, first = 0;
        bool same = true;
        for (uint32_t x = 0; x < S; x++) {
            const uint32_t st = sst[x * 65 + lane];
            if (st != prev_in) {
                dnx = 0;
                out = unp_step(nul, st, V, valid, [&](uint32_t) { dnx++; });
                prev_in = st;
            }
            sst[x * 65 + lane] = out;
            scn[x * 65 + lane] += dnx;
            if (x == 0) first = out;
            else same = same && out == first;
        }
        if (same) {
            merged = true;
            state = first;
        }
    };
    for (uint32_t s0 = 0; s0 < ns; s0 += UNP_AHEAD) {
#pragma unroll
        for (uint32_t j = 0; j < UNP_AHEAD; j++) {
            const uint32_t st = s0 + j;
            uint64_t nul0, nul1;
            unp_rows_mask<EDGE>(p, c0, st, lane, ring[j], smask, nul0, nul1);
            __builtin_amdgcn_sched_barrier(0);
            unp_rows_load<EDGE>(p, c0, min(st + UNP_AHEAD, ns - 1), lane, ring[j]); // in flight during the next parses
            __builtin_amdgcn_sched_barrier(0);
            // a lane's two masks side by side, one KiB row per step: k_unp_emit parses from these
            reinterpret_cast<ulonglong2 *>(p.masks)[((uint64_t) c * ns + st) * 64 + lane] = make_ulonglong2(nul0, nul1);
            block(2 * st, nul0);
            block(2 * st + 1, nul1);
        }
    }
}

// States as table indices: 0..V = bytes to skip, V+1 = inside a key.
// Dynamic LDS: states [S][65], counts [S][65] (a lane's column; the pitch of 65
// keeps the composition's reads apart), then the 512 sixteen-byte masks of a step.
__global__ __launch_bounds__(64) void k_unp_spec(UnpParams p)
{
    extern __shared__ uint32_t unp_lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x, V = p.V, S = V + 2;
    uint32_t *sst = unp_lds, *scn = unp_lds + S * 65;
    uint16_t *smask = reinterpret_cast<uint16_t *>(unp_lds + ((2 * S * 65 + 3) & ~3u)); // read 16 bytes at a time
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    for (uint32_t x = 0; x < S; x++) {
        sst[x * 65 + lane] = (x == V + 1) ? UNP_INKEY : x;
        scn[x * 65 + lane] = 0;
    }
    bool merged = false; // every entry state of this lane has led to the same state
    uint32_t state = 0, dn = 0;
    // a chunk with 16 bytes of stream behind it needs no care at its end
    if (c0 + p.chunk_bytes + 16 <= p.nbytes) unp_spec_blocks<false>(p, c, lane, sst, scn, smask, merged, state, dn);
    else unp_spec_blocks<true>(p, c, lane, sst, scn, smask, merged, state, dn);
    for (uint32_t x = 0; x < S; x++) {
        const uint32_t e = merged ? state : sst[x * 65 + lane];
        sst[x * 65 + lane] = (e & UNP_INKEY) ? V + 1 : e; // at most V bytes are ever owed after a NUL
        if (merged) scn[x * 65 + lane] += dn;
    }
    __syncthreads();
    for (uint32_t i = lane; i < UNP_PIECES * S; i += 64) {
        const uint32_t piece = i / S, x = i - piece * S;
        p.piece_exit[(uint64_t) c * UNP_PIECES * S + i] = (uint8_t) sst[x * 65 + piece];
        p.piece_count[(uint64_t) c * UNP_PIECES * S + i] = (uint16_t) scn[x * 65 + piece];
    }
    if ((uint32_t) lane < S) { // the chunk's table: entry state `lane` through the 64 pieces
        uint32_t cur = lane, cnt = 0;
        for (int piece = 0; piece < UNP_PIECES; piece++) {
            cnt += scn[cur * 65 + piece];
            cur = sst[cur * 65 + piece];
        }
        unp_post(&p.table[(uint64_t) c * S + lane], (cur << 24) | cnt); // a chunk is at most 2^21 bytes
    }
    // the last workgroup of a run composes the run, the last of those chains the runs
    const uint32_t run = c / UNP_RUN;
    if (!unp_arrive_last(&p.sync[2 + run], min(UNP_RUN, p.nchunks - run * UNP_RUN), lane)) return;
    unp_compose_run(p, run, lane);
    if (!unp_arrive_last(&p.sync[0], p.nruns, lane)) return;
    unp_run_prefix(p, lane);
}

// A lane's record starts are consecutive entries of rec_off, a few per block, and
// the 64 lanes write to 64 places.  Stored one by one every store instruction
// touches 64 lines for 4 bytes each (0.79 ms per 6e7 records); stored by the lane
// as the four 16-byte pieces of a 64-byte group it still is 64 lines per
// instruction (0.11 ms of the kernel's 0.21).  So a lane collects its entries in a
// ring in LDS, and after every block the wave stores the groups that are complete
// together: four lanes per group, one 16-byte piece each, sixteen groups per store
// instruction, every group one 64-byte request.  Groups are 64-byte groups of
// rec_off counted from a 64-byte boundary at or below it; the entries of the
// group a lane shares with its neighbour (the first and the last of its piece) go
// one by one.
constexpr uint32_t UNP_RING = 24;  // entries a lane keeps: a group of 16 and what a block adds before the wave stores
constexpr int UNP_SLOT_PITCH = 28; // dwords per lane: 16-byte reads of 8 lanes at this pitch fall into different banks
constexpr int UNP_LIST_DWORDS = 3 * 64;

// Dwords of the piece tables in k_unp_emit's LDS (room for the list that takes their place).
__host__ __device__ inline uint32_t unp_emit_table_dwords(uint32_t S)
{
    const uint32_t t = (uint32_t) UNP_PIECES * S;
    return ((t > (uint32_t) UNP_LIST_DWORDS ? t : (uint32_t) UNP_LIST_DWORDS) + 3u) & ~3u;
}

struct UnpSink {
    uint32_t *rings;   // lane 0's ring in LDS; a lane's own is at UNP_SLOT_PITCH * lane
    uint32_t *ring;    // this lane's
    uint32_t *list;    // [64][3]: lane << 8 | ring position of a group to store, its first entry (two words)
    uint32_t *out;     // rec_off moved back to a 64-byte boundary
    uint64_t v;        // next entry to write, counted from `out`
    uint64_t v_first;  // first entry of this lane
    uint64_t v_done;   // a multiple of 16: the entries below it are stored (or another lane's)
    uint32_t rp, rp_done; // ring positions of v and v_done

    __device__ __forceinline__ void start(uint64_t first)
    {
        v = v_first = first;
        v_done = first & ~(uint64_t) 15;
        rp_done = 0;
        rp = (uint32_t) (v - v_done);
    }
    __device__ __forceinline__ void put(uint32_t value)
    {
        if (v - v_done >= UNP_RING) store_own(); // more than 8 records in a block: values of fewer than 7 bytes only
        ring[rp] = value;
        rp = (rp + 1 == UNP_RING) ? 0 : rp + 1;
        v++;
    }
    __device__ __forceinline__ void advance_done()
    {
        v_done += 16;
        rp_done = (rp_done + 16 >= UNP_RING) ? rp_done + 16 - UNP_RING : rp_done + 16;
    }
    __device__ __forceinline__ void store_own() // the oldest group, entry by entry
    {
        for (uint32_t e = 0; e < 16; e++) {
            const uint32_t at = rp_done + e;
            if (v_done + e >= v_first) out[v_done + e] = ring[at >= UNP_RING ? at - UNP_RING : at];
        }
        advance_done();
    }
    // The whole wave: every complete group of every lane.
    __device__ __forceinline__ void store_ready(int lane)
    {
        for (;;) {
            bool pend = v - v_done >= 16;
            if (pend && v_done < v_first) { // shared with the neighbour
                store_own();
                pend = v - v_done >= 16;
            }
            const uint64_t m = __ballot(pend);
            if (m == 0) break;
            if (pend) {
                const uint32_t r = lanes_below(m);
                list[3 * r] = ((uint32_t) lane << 8) | rp_done;
                list[3 * r + 1] = (uint32_t) v_done;
                list[3 * r + 2] = (uint32_t) (v_done >> 32);
            }
            asm volatile("" ::: "memory"); // one wave: its LDS accesses complete in order
            const uint32_t n = (uint32_t) __popcll(m);
            for (uint32_t e0 = 0; e0 < n; e0 += 16) {
                const uint32_t e = e0 + ((uint32_t) lane >> 2);
                if (e < n) {
                    const uint32_t w = list[3 * e];
                    const uint64_t first = (uint64_t) list[3 * e + 1] | ((uint64_t) list[3 * e + 2] << 32);
                    uint32_t at = (w & 0xFFu) + 4u * ((uint32_t) lane & 3u);
                    if (at >= UNP_RING) at -= UNP_RING;
                    const uint4 piece = *reinterpret_cast<const uint4 *>(rings + UNP_SLOT_PITCH * (w >> 8) + at);
                    *reinterpret_cast<uint4 *>(out + first + 4u * ((uint32_t) lane & 3u)) = piece;
                }
            }
            asm volatile("" ::: "memory");
            if (pend) advance_done();
        }
    }
    __device__ __forceinline__ void finish() // what is left: less than a group, or the groups of a wave that stopped early
    {
        for (uint64_t i = max(v_done, v_first); i < v; i++) {
            const uint32_t at = rp_done + (uint32_t) (i - v_done);
            out[i] = ring[at >= UNP_RING ? at - UNP_RING : at];
        }
    }
};

// Dynamic LDS: the chunk's piece tables as (exit << 16 | count) [64][S] -- the list of groups to
// store once they have been walked -- and the lanes' rings.
__global__ __launch_bounds__(64) void k_unp_emit(UnpParams p)
{
    extern __shared__ uint32_t unp_lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x, V = p.V, S = V + 2;
    if (p.ctrl->head.error != 0) return;
    uint32_t *stab = unp_lds;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint64_t a0 = c0 + (uint64_t) lane * p.piece_bytes;
    const uint64_t a1 = min(p.nbytes, a0 + p.piece_bytes);
    const uint32_t nb = p.piece_bytes / 64;
    // the NUL masks k_unp_spec left (the stream itself is not read again), a few blocks ahead
    constexpr uint32_t AHEAD = 4;
    const uint32_t ns = nb / 2; // pieces are multiples of 256 bytes: a pair of blocks per step, steps in pairs
    const ulonglong2 *mrow = reinterpret_cast<const ulonglong2 *>(p.masks) + (uint64_t) c * ns * 64 + lane;
    ulonglong2 mq[AHEAD / 2];
#pragma unroll
    for (uint32_t j = 0; j < AHEAD / 2; j++) mq[j] = mrow[(uint64_t) j * 64]; // ns is a multiple of AHEAD / 2
    for (uint32_t i = lane; i < UNP_PIECES * S; i += 64)
        stab[i] = ((uint32_t) p.piece_exit[(uint64_t) c * UNP_PIECES * S + i] << 16)
                  | p.piece_count[(uint64_t) c * UNP_PIECES * S + i];
    __syncthreads();
    // the run's entry state through the chunks before this one (lane x holds entry x of a chunk's row)
    const uint32_t run = c / UNP_RUN, before = c - run * UNP_RUN;
    uint32_t entry = (uint32_t) __builtin_amdgcn_readfirstlane((int) p.run_entry[run]);
    uint64_t first = p.run_first[run];
    for (uint32_t b = 0; b < before; b += UNP_BATCH) {
        uint32_t row[UNP_BATCH];
#pragma unroll
        for (int j = 0; j < UNP_BATCH; j++)
            row[j] = p.table[(uint64_t) min(run * UNP_RUN + b + j, p.nchunks - 1) * S + min((uint32_t) lane, S - 1)];
#pragma unroll
        for (int j = 0; j < UNP_BATCH; j++) {
            if (b + j < before) {
                const uint32_t t = (uint32_t) __builtin_amdgcn_readlane((int) row[j], (int) entry);
                first += t & 0xFFFFFFu;
                entry = t >> 24;
            }
        }
    }
    // the chunk's entry state through the pieces, the same walk in every lane
    uint32_t cur = entry, cnt = 0, mine = 0, base = 0;
    for (int piece = 0; piece < UNP_PIECES; piece++) {
        if (piece == lane) {
            mine = cur;
            base = cnt;
        }
        const uint32_t t = stab[piece * S + cur];
        cnt += t & 0xFFFFu;
        cur = t >> 16;
    }
    uint32_t state = (mine == V + 1) ? UNP_INKEY : mine;
    const uint32_t back = (uint32_t) ((reinterpret_cast<uintptr_t>(p.rec_off) >> 2) & 15u);
    UnpSink sink;
    sink.list = unp_lds;
    sink.rings = unp_lds + unp_emit_table_dwords(S);
    sink.ring = sink.rings + UNP_SLOT_PITCH * lane;
    sink.out = p.rec_off - back;
    sink.start(first + base + back);
    for (uint32_t s0 = 0; s0 < ns; s0 += AHEAD / 2) {
#pragma unroll
        for (uint32_t j = 0; j < AHEAD / 2; j++) {
            const uint32_t st = s0 + j;
            const ulonglong2 pair = mq[j];
            mq[j] = mrow[(uint64_t) min(st + AHEAD / 2, ns - 1) * 64]; // in flight during the parse; issued on every trip
#pragma unroll
            for (uint32_t hb = 0; hb < 2; hb++) {
                const uint64_t nul = hb ? pair.y : pair.x;
                const uint64_t boff = a0 + 64u * (2 * st + hb);
                if (boff < a1) {
                    const uint32_t valid = (uint32_t) min((uint64_t) 64, a1 - boff);
                    state = unp_step(nul, state, V, valid, [&](uint32_t pos) { sink.put((uint32_t) boff + pos); });
                }
                sink.store_ready(lane);
            }
        }
    }
    sink.finish();
    if (c == p.nchunks - 1 && lane == 0) p.rec_off[p.result[0]] = (uint32_t) p.nbytes;
}

// A stream of fewer than 16 bytes: one lane walks it, first to count, then to write.
__global__ void k_unp_tiny(UnpParams p)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    uint64_t n = 0;
    for (int pass = 0; pass < 2; pass++) {
        uint64_t pos = 0;
        n = 0;
        while (pos < p.nbytes) {
            if (pass) p.rec_off[n] = (uint32_t) pos;
            n++;
            while (pos < p.nbytes && p.stream[pos] != 0) pos++;
            pos = (pos < p.nbytes) ? pos + 1 + p.V : p.nbytes + 1; // a key that does not end
        }
        if (pos != p.nbytes) { // the stream must end at a record boundary
            raise_error(p.ctrl, MGS_ERR_MALFORMED);
            return;
        }
        if (n + 1 > p.rec_cap) {
            raise_error(p.ctrl, MGS_ERR_ARG);
            return;
        }
    }
    p.rec_off[n] = (uint32_t) p.nbytes;
    p.result[0] = n;
}

// Fixed-width records: the offsets are multiples of the width.
__global__ void k_unp_fixed(UnpParams p, uint32_t width)
{
    const uint64_t n = p.nbytes / width;
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride)
        p.rec_off[i] = (uint32_t) (i * width);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.result[0] = n;
        if (p.nbytes % width != 0) raise_error(p.ctrl, MGS_ERR_MALFORMED);
    }
}

// KVContainer::write, src/kvcontainer.h:133-160: a record goes to the current
// page if it fits exactly or with room to spare, else a new page is opened.
// One wave: the last record that still fits is found by a search that probes 64
// places at a time.
__device__ __forceinline__ void unp_pages_walk(const UnpParams &p, int lane)
{
    auto off = [&](uint64_t i) { return (uint64_t) p.rec_off[i]; };
    const uint64_t n = p.result[0];
    const uint64_t S = (uint64_t) p.page_bytes;
    uint64_t room = S - (uint64_t) p.first_fill; // bytes left in the current page
    uint64_t i = 0, ncuts = 0;
    uint64_t base = n ? off(0) : 0; // rec_off[i]
    while (i < n) {
        // largest j in [i, n] with rec_off[j] - rec_off[i] <= room
        uint64_t lo = i, hi = n;
        {
            // first the neighbourhood of where the cut falls for records of the mean length
            const uint64_t est = min(n, i + min(room, p.nbytes) * n / p.nbytes);
            const uint64_t w = 2048;
            const uint64_t q0 = est > i + w ? est - w : i;
            const uint64_t probe = min(hi, q0 + (uint64_t) lane * (2 * w / 64));
            const bool fits = off(probe) - base <= room;
            const uint64_t k = (uint64_t) __popcll(__ballot(fits)); // the offsets ascend: the first k probes fit
            if (k > 0) lo = min(hi, q0 + (k - 1) * (2 * w / 64));
            if (k < 64) hi = min(hi, q0 + k * (2 * w / 64)) - 1; // probe k does not fit
        }
        while (lo < hi) {
            const uint64_t step = (hi - lo + 63) / 64;
            const uint64_t probe = min(hi, lo + (uint64_t) (lane + 1) * step);
            const bool fits = off(probe) - base <= room;
            const uint64_t k = (uint64_t) __popcll(__ballot(fits));
            const uint64_t nlo = min(hi, lo + k * step);
            hi = min(hi, lo + (k + 1) * step - 1);
            lo = nlo;
        }
        if (lo == n) break;
        if (lo == i && room == S) { // does not fit an empty page (:147-149)
            if (lane == 0) raise_error(p.ctrl, MGS_ERR_RECORD_SIZE);
            break;
        }
        base = off(lo);
        if (ncuts < p.max_cuts && lane == 0) {
            p.cuts[2 * ncuts] = lo;
            p.cuts[2 * ncuts + 1] = base;
        }
        ncuts++;
        i = lo;
        room = S;
    }
    if (lane == 0) p.result[1] = ncuts;
}

__global__ __launch_bounds__(64) void k_unp_pages(UnpParams p)
{
    if (blockIdx.x != 0) return;
    if (p.ctrl->head.error == 0) unp_pages_walk(p, threadIdx.x);
    // the call's status travels with its result: one copy back to the host
    if (threadIdx.x == 0)
        p.result[2] = (uint64_t) (uint32_t) __hip_atomic_load(&p.ctrl->head.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

} // namespace mgs
#endif
