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
// after a key's NUL skip V bytes, the next key starts there.  The chain is
// broken into chunks, one wave each:
//
//   k_unp_spec   the chunk is parsed for every state it could be entered in
//                (x = 0..V bytes still to skip, or the middle of a key); lane x
//                keeps state x.  Per 64-byte block the wave makes a 64-bit NUL
//                mask (one byte per lane, ballot) and steps the states one
//                after the other as wave-uniform scalar work -- states that
//                have met (after a few records they all have) are stepped
//                once, so the cost is that of one parse.  (Letting every lane
//                walk its own bits in a divergent loop gave wrong counts now
//                and then on gfx950; the uniform form does not.)
//                Out: exit state and record count per entry state.
//   k_unp_chain  one wave follows the chain through the chunks -> entry state
//                and first record index of every chunk.
//   k_unp_emit   every wave parses its chunk again, now from its true entry
//                state, and writes the record start offsets.
//   k_unp_pages  one thread: greedy page cuts by binary search in the offsets.
#ifndef MGS_UNPACK_H
#define MGS_UNPACK_H

#include "mgs_common.h"

namespace mgs {

constexpr int UNP_MAX_V = 62;        // value bytes handled: V + 2 entry states, one lane each
constexpr int UNP_MAX_CHUNKS = 4096; // one wave each: four per SIMD on a full device

struct UnpParams {
    const uint8_t *stream;
    uint64_t nbytes;
    uint32_t V;           // bytes to skip after a key's NUL
    uint32_t chunk_bytes; // multiple of 1024
    uint32_t nchunks;
    uint8_t *exit_state;  // [nchunks][V+1]
    uint32_t *count;      // [nchunks][V+1]
    uint32_t *entry;      // [nchunks] entry state of each chunk
    uint64_t *first_rec;  // [nchunks+1] index of the chunk's first record start (+ total)
    uint32_t *rec_off;    // out: rec_off[i] = start of record i; rec_off[n] = nbytes
    uint64_t rec_cap;     // entries available in rec_off
    Ctrl *ctrl;
    // pages
    int64_t page_bytes;
    int64_t first_fill;
    uint64_t *cuts;       // out: [2*max_cuts] (first_record, first_byte) of every new page
    uint32_t max_cuts;
    uint64_t *result;     // out: [0] records, [1] number of cuts
};

struct __attribute__((packed)) UnpU128 { uint32_t x, y, z, w; };

// The lane's 16 bytes p.stream[at .. at+16), fetched without ever reading past the
// stream: a piece that sticks out is fetched 16 bytes back from the end instead
// (unp_m16 shifts it into place).  Streams shorter than 16 bytes go byte by byte.
__device__ __forceinline__ uint4 unp_load16(const UnpParams &p, uint64_t at)
{
    if (p.nbytes < 16) {
        uint32_t w[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (uint32_t i = 0; i < 16 && at + i < p.nbytes; i++)
            w[i >> 2] = (w[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((uint32_t) p.stream[at + i] << (8 * (i & 3)));
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    const uint64_t a = min(at, p.nbytes - 16);
    const UnpU128 v = *reinterpret_cast<const UnpU128 *>(p.stream + a);
    return make_uint4(v.x, v.y, v.z, v.w);
}

// Bit i of the result: byte i of the lane's 16 bytes at p.stream[at .. at+16) is a
// NUL (bytes past the end of the stream count as non-NUL).  v = unp_load16(p, at).
__device__ __forceinline__ uint32_t unp_m16(const UnpParams &p, uint64_t at, const uint4 &v)
{
    auto nibble = [](uint32_t w) { // one bit per zero byte, lowest address first
        const uint32_t t = ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu); // 0x80 per zero byte
        return (((t >> 7) * 0x01020408u) >> 24) & 0xFu;
    };
    const uint32_t m = nibble(v.x) | (nibble(v.y) << 4) | (nibble(v.z) << 8) | (nibble(v.w) << 12);
    if (at >= p.nbytes) return 0;
    if (p.nbytes >= 16 && at + 16 > p.nbytes) return m >> (uint32_t) (at - (p.nbytes - 16));
    return m;
}

// The NUL mask of 64-byte block b (0..15) of the kilobyte whose 16-byte pieces the
// lanes hold in m16: lanes 4b..4b+3.
template <int B>
__device__ __forceinline__ uint64_t unp_block_mask(uint32_t m16)
{
    const uint32_t a = (uint32_t) __builtin_amdgcn_readlane((int) m16, 4 * B);
    const uint32_t b = (uint32_t) __builtin_amdgcn_readlane((int) m16, 4 * B + 1);
    const uint32_t c = (uint32_t) __builtin_amdgcn_readlane((int) m16, 4 * B + 2);
    const uint32_t d = (uint32_t) __builtin_amdgcn_readlane((int) m16, 4 * B + 3);
    return (uint64_t) (a | (b << 16)) | ((uint64_t) (c | (d << 16)) << 32);
}

// One 64-byte block for one state.  state: bytes to skip before the next key
// starts (0..V), or UNP_INKEY when the block is entered in the middle of a key.
// A record start is noted (bit in `starts`, n++) wherever a key begins inside
// the `valid` stream bytes of the block.  Returns the state at the block's end
// (for a partial last block: the bytes still owed, 0 = clean end).
constexpr uint32_t UNP_INKEY = 0x80000000u;

__device__ __forceinline__ uint32_t unp_step(uint64_t nul, uint32_t state, uint32_t V, uint32_t valid,
                                             uint64_t &starts, uint32_t &n)
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
        starts |= 1ull << pos; // a key begins here
        n++;
        const uint64_t mm = nul & (~0ull << pos);
        if (mm == 0) return UNP_INKEY; // the key runs on into the next block
        pos = (uint32_t) __ffsll((unsigned long long) mm) + V;
    }
    return pos - valid;
}

// States: 0..V = bytes to skip; "inside a key" is one more state.  Lane x keeps
// entry state x (x <= V), lane V+1 "inside a key".  A wave reads its chunk a
// kilobyte at a time (16 bytes per lane, the next kilobyte in flight while this
// one is parsed).
template <int B>
__device__ __forceinline__ void unp_spec_block(const UnpParams &p, uint32_t m16, uint64_t off, uint64_t c1,
                                               int lane, uint32_t S, uint32_t &state, uint32_t &n, bool &merged)
{
    const uint64_t boff = off + 64u * B;
    if (boff >= c1) return;
    const uint64_t nul = unp_block_mask<B>(m16);
    const uint32_t valid = (uint32_t) min((uint64_t) 64, c1 - boff);
    if (merged) { // every entry state has led to the same state: one parse serves all
        const uint32_t st = (uint32_t) __builtin_amdgcn_readfirstlane((int) state);
        uint64_t starts = 0;
        uint32_t dn = 0;
        state = unp_step(nul, st, p.V, valid, starts, dn);
        n += dn;
        return;
    }
    // every state in turn, as wave-uniform scalar work
    uint32_t prev_in = 0xFFFFFFFFu, prev_out = 0, prev_n = 0;
    for (uint32_t x = 0; x < S; x++) {
        const uint32_t st = (uint32_t) __builtin_amdgcn_readlane((int) state, (int) x);
        if (st != prev_in) {
            uint64_t starts = 0;
            prev_n = 0;
            prev_out = unp_step(nul, st, p.V, valid, starts, prev_n);
            prev_in = st;
        }
        if ((uint32_t) lane == x) {
            state = prev_out;
            n += prev_n;
        }
    }
    const uint32_t s0 = (uint32_t) __builtin_amdgcn_readfirstlane((int) state);
    merged = __ballot((uint32_t) lane < S && state != s0) == 0;
}

__global__ __launch_bounds__(64) void k_unp_spec(UnpParams p)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint64_t c1 = min(p.nbytes, c0 + p.chunk_bytes);
    const uint32_t S = p.V + 2; // number of states
    uint32_t state = ((uint32_t) lane == p.V + 1) ? UNP_INKEY : (uint32_t) lane;
    uint32_t n = 0;
    bool merged = false;
    uint4 raw = unp_load16(p, c0 + 16u * lane);
    for (uint64_t off = c0; off < c1; off += 1024) {
        const uint32_t cur = unp_m16(p, off + 16u * lane, raw);
        if (off + 1024 < c1) raw = unp_load16(p, off + 1024 + 16u * lane); // in flight during the parse
        unp_spec_block<0>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<1>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<2>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<3>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<4>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<5>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<6>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<7>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<8>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<9>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<10>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<11>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<12>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<13>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<14>(p, cur, off, c1, lane, S, state, n, merged);
        unp_spec_block<15>(p, cur, off, c1, lane, S, state, n, merged);
    }
    if ((uint32_t) lane < S) {
        // exit state as an index: 0..V, or V+1 for "inside a key" (at most V bytes are
        // ever owed after a NUL)
        const uint32_t e = (state & UNP_INKEY) ? p.V + 1 : state;
        p.exit_state[(uint64_t) c * S + lane] = (uint8_t) e;
        p.count[(uint64_t) c * S + lane] = n;
    }
}

// One wave follows the chain through the chunks: lane x holds the table row entry
// (exit state, count) for state x of the chunk at hand; the rows of 16 chunks are
// fetched at once (their addresses do not depend on the chain), a step is two
// register reads, and the 16 results leave with one store.
__global__ __launch_bounds__(64) void k_unp_chain(UnpParams p)
{
    const int lane = threadIdx.x;
    const uint32_t S = p.V + 2;
    constexpr int BATCH = 16;
    uint32_t state = 0; // the stream starts at a record boundary
    uint64_t total = 0;
    const uint32_t x = min((uint32_t) lane, S - 1);
    auto fetch = [&](uint32_t c0, uint32_t (&ex)[BATCH], uint32_t (&cn)[BATCH]) {
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const uint32_t c = min(c0 + j, p.nchunks - 1);
            ex[j] = p.exit_state[(uint64_t) c * S + x];
            cn[j] = p.count[(uint64_t) c * S + x];
        }
    };
    uint32_t ex[BATCH], cn[BATCH], ex2[BATCH], cn2[BATCH];
    fetch(0, ex, cn);
    for (uint32_t c0 = 0; c0 < p.nchunks; c0 += BATCH) {
        if (c0 + BATCH < p.nchunks) fetch(c0 + BATCH, ex2, cn2); // in flight during this batch
        uint32_t my_entry = 0;
        uint64_t my_first = 0;
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            if (lane == j) {
                my_entry = state;
                my_first = total;
            }
            if (c0 + j < p.nchunks) {
                total += (uint32_t) __builtin_amdgcn_readlane((int) cn[j], (int) state);
                state = (uint32_t) __builtin_amdgcn_readlane((int) ex[j], (int) state);
            }
        }
        if (lane < BATCH && c0 + lane < p.nchunks) {
            p.entry[c0 + lane] = my_entry;
            p.first_rec[c0 + lane] = my_first;
        }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            ex[j] = ex2[j];
            cn[j] = cn2[j];
        }
    }
    if (lane == 0) {
        p.first_rec[p.nchunks] = total;
        p.result[0] = total;
        // the stream must end at a record boundary: nothing owed, not inside a key
        if (state != 0) raise_error(p.ctrl, MGS_ERR_MALFORMED);
        else if (total + 1 > p.rec_cap) raise_error(p.ctrl, MGS_ERR_ARG);
    }
}

template <int B>
__device__ __forceinline__ void unp_emit_block(const UnpParams &p, uint32_t m16, uint64_t off, uint64_t c1,
                                               int lane, uint32_t &state, uint64_t &idx)
{
    const uint64_t boff = off + 64u * B;
    if (boff >= c1) return;
    const uint64_t nul = unp_block_mask<B>(m16);
    const uint32_t valid = (uint32_t) min((uint64_t) 64, c1 - boff);
    uint64_t starts = 0;
    uint32_t n = 0;
    state = unp_step(nul, state, p.V, valid, starts, n); // the same in every lane
    if ((starts >> lane) & 1ull)
        p.rec_off[idx + (uint32_t) __popcll(starts & ((1ull << lane) - 1ull))] = (uint32_t) (boff + lane);
    idx += n;
}

__global__ __launch_bounds__(64) void k_unp_emit(UnpParams p)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (p.ctrl->head.error != 0) return;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint64_t c1 = min(p.nbytes, c0 + p.chunk_bytes);
    const uint32_t e = p.entry[c];
    uint32_t state = (e == p.V + 1) ? UNP_INKEY : e;
    uint64_t idx = p.first_rec[c];
    uint4 raw = unp_load16(p, c0 + 16u * lane);
    for (uint64_t off = c0; off < c1; off += 1024) {
        const uint32_t cur = unp_m16(p, off + 16u * lane, raw);
        if (off + 1024 < c1) raw = unp_load16(p, off + 1024 + 16u * lane); // in flight during the parse
        unp_emit_block<0>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<1>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<2>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<3>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<4>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<5>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<6>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<7>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<8>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<9>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<10>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<11>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<12>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<13>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<14>(p, cur, off, c1, lane, state, idx);
        unp_emit_block<15>(p, cur, off, c1, lane, state, idx);
    }
    if (c == p.nchunks - 1 && lane == 0) p.rec_off[p.first_rec[p.nchunks]] = (uint32_t) p.nbytes;
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
__global__ void k_unp_pages(UnpParams p)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (p.ctrl->head.error != 0) {
        p.result[1] = 0;
        return;
    }
    const uint64_t n = p.result[0];
    const uint64_t S = (uint64_t) p.page_bytes;
    uint64_t room = S - (uint64_t) p.first_fill; // bytes left in the current page
    uint64_t i = 0, ncuts = 0;
    while (i < n) {
        // largest j in (i, n] with rec_off[j] - rec_off[i] <= room
        const uint64_t base = p.rec_off[i];
        uint64_t lo = i, hi = n;
        while (lo < hi) {
            const uint64_t mid = (lo + hi + 1) >> 1;
            if ((uint64_t) p.rec_off[mid] - base <= room) lo = mid;
            else hi = mid - 1;
        }
        if (lo == n) break;
        if (lo == i && room == S) { // does not fit an empty page (:147-149)
            raise_error(p.ctrl, MGS_ERR_RECORD_SIZE);
            break;
        }
        if (ncuts < p.max_cuts) {
            p.cuts[2 * ncuts] = lo;
            p.cuts[2 * ncuts + 1] = p.rec_off[lo];
        }
        ncuts++;
        i = lo;
        room = S;
    }
    p.result[1] = ncuts;
}

} // namespace mgs
#endif
