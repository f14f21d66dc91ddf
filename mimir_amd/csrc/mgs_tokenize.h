// Map side of wordcount on the device: text -> <word, count> records in wire form.
//
// What it replaces: FileReader::read hands the map callback one line at a time
// (FileParser::to_line cuts at '\n', src/fileparser.h:21-41; the reader then
// advances by strlen + 1, src/filereader.h:138-183, so a NUL byte ends a line as
// well), and the callback splits the line at spaces with strtok_r and writes
// <word, 1> for every word shorter than 1024 bytes (examples/wordcount.cpp:
// 66-82).  Put together: a word is a maximal run of bytes other than ' ', '\n'
// and NUL; words of 1024 bytes or more are dropped; every other word becomes the
// record  word bytes, NUL, 8-byte value  (src/serializer.h:157-168, :63-83), in
// text order.
//
// A word byte goes to the output at (kept word bytes before it) + 9 * (kept
// words closed before it), so all that is needed are three prefix sums over the
// text: kept word bytes, words closed, words started (the record's index).  The
// text is cut like the receive stream of mgs_unpack.h: chunks, one wave each, 64
// pieces per chunk, one lane each; a lane walks its piece 64 bytes at a time on
// a 64-bit mask of word bytes.
//
//   k_tok_count  per piece: kept word bytes, words closed, words started;
//                exclusive sums over the pieces of a chunk, chunk totals.
//   k_tok_scan   one wave: exclusive sums over the chunks, capacity checks.
//   k_tok_emit   every lane walks its piece again and streams its output: word
//                bytes as they come, NUL + value where a word closes; the record
//                offsets where words start.  Both streams are collected in LDS
//                in 64-byte groups of the output and leave with 16-byte stores.
//
// The 1024-byte rule is the only thing that needs more than the byte before a
// block: whether a word is dropped must be known where it starts.  A word that
// starts and closes inside one block or the next is short.  Only where a word is
// open at the end of a block and the whole next block consists of word bytes
// (or where a piece begins inside a word) is its length measured by walking the
// text itself (tok_word_is_long) -- the same answer in whichever lane asks.
#ifndef MGS_TOKENIZE_H
#define MGS_TOKENIZE_H

#include "mgs_unpack.h"

namespace mgs {

constexpr uint32_t TOK_MAX_WORD = 1024; // strlen(word) < 1024, examples/wordcount.cpp:72
constexpr int TOK_ROW_PITCH = 80;       // bytes per lane in the LDS copies of a block / an output group

struct TokParams {
    const uint8_t *text;
    uint64_t nbytes;
    uint32_t chunk_bytes; // 64 pieces
    uint32_t piece_bytes; // multiple of 64
    uint32_t nchunks;     // covers nbytes + 1 bytes: the end of the text closes an open word
    uint32_t *piece_x;    // [nchunks][3][64] kept word bytes / closed / started before the piece, in its chunk
    uint32_t *chunk_tot;  // [nchunks][3]
    uint64_t *chunk_x;    // [nchunks][3] the same before the chunk, in the text
    uint8_t *out;         // records
    uint64_t out_cap;
    uint32_t *rec_off;    // rec_off[i] = first byte of record i, rec_off[n] = total bytes
    uint64_t rec_cap;
    uint64_t value;       // the 8 value bytes of every record
    Ctrl *ctrl;
    uint64_t *result;     // [0] words, [1] bytes
};

__device__ __forceinline__ bool tok_is_word_byte(uint8_t b)
{
    return b != 0 && b != ' ' && b != '\n';
}

// Is the word that covers text[at] TOK_MAX_WORD bytes or longer?
__device__ __noinline__ bool tok_word_is_long(const TokParams &p, uint64_t at)
{
    uint32_t len = 1;
    for (uint64_t i = at; i > 0 && len < TOK_MAX_WORD && tok_is_word_byte(p.text[i - 1]); i--) len++;
    for (uint64_t i = at + 1; i < p.nbytes && len < TOK_MAX_WORD && tok_is_word_byte(p.text[i]); i++) len++;
    return len >= TOK_MAX_WORD;
}

// The lane's 16 bytes of text at `at` (see unp_load16) and their word-byte mask:
// bit i set = text[at + i] is a word byte; bytes past the end are not.
__device__ __forceinline__ uint4 tok_load16(const TokParams &p, uint64_t at)
{
    if (p.nbytes < 16) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (uint32_t i = 0; i < 16 && at + i < p.nbytes; i++) w[i >> 2] |= (uint32_t) p.text[at + i] << (8 * (i & 3));
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
    const uint64_t a = min(at, p.nbytes - 16);
    const UnpU128 v = *reinterpret_cast<const UnpU128 *>(p.text + a);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t tok_m16(const TokParams &p, uint64_t at, const uint4 &v)
{
    auto zero = [](uint32_t w) { return ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu); }; // 0x80 per zero byte
    auto delim = [&](uint32_t w) { // one bit per delimiter byte, lowest address first
        const uint32_t t = zero(w) | zero(w ^ 0x20202020u) | zero(w ^ 0x0a0a0a0au);
        return (((t >> 7) * 0x01020408u) >> 24) & 0xFu;
    };
    const uint32_t m = ~(delim(v.x) | (delim(v.y) << 4) | (delim(v.z) << 8) | (delim(v.w) << 12)) & 0xFFFFu;
    if (at >= p.nbytes) return 0;
    if (p.nbytes < 16) return m; // tok_load16 left zeros past the end
    if (at + 16 > p.nbytes) return m >> (uint32_t) (at - (p.nbytes - 16));
    return m;
}

// The bytes tok_m16 speaks of, in place (a piece that sticks out of the text was
// fetched 16 bytes back from its end).
__device__ __forceinline__ uint4 tok_in_place(const TokParams &p, uint64_t at, const uint4 &v)
{
    if (p.nbytes < 16 || at + 16 <= p.nbytes || at >= p.nbytes) return v;
    const uint32_t sh = (uint32_t) (at - (p.nbytes - 16)); // 1..15 bytes
    const uint32_t w[8] = {v.x, v.y, v.z, v.w, 0, 0, 0, 0};
    uint32_t r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 7; j++) {
            if ((int) (sh >> 2) + i == j) {
                lo = w[j];
                hi = w[j + 1];
            }
        }
        r[i] = __builtin_amdgcn_alignbyte(hi, lo, sh & 3u);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}

__device__ __forceinline__ uint64_t tok_row_at(const TokParams &p, uint64_t c0, uint32_t b, int j, int lane)
{
    return c0 + (uint64_t) (16 * j + (lane >> 2)) * p.piece_bytes + 64u * b + 16u * (uint32_t) (lane & 3);
}

__device__ __forceinline__ void tok_rows_load(const TokParams &p, uint64_t c0, uint32_t b, int lane, UnpRows &r)
{
#pragma unroll
    for (int j = 0; j < 4; j++) r.raw[j] = tok_load16(p, tok_row_at(p, c0, b, j, lane));
}

// Word masks of the rows just loaded -> every lane's own 64-bit mask (through
// LDS).  With `bytes`, the bytes of that (other) block b_bytes are handed to
// their lanes on the way: row r at rows_lds + r * TOK_ROW_PITCH.
__device__ __forceinline__ uint64_t tok_rows_mask(const TokParams &p, uint64_t c0, uint32_t b, int lane,
                                                  const UnpRows &r, uint16_t *smask,
                                                  const UnpRows *bytes = nullptr, uint32_t b_bytes = 0,
                                                  uint8_t *rows_lds = nullptr)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        smask[64 * j + lane] = (uint16_t) tok_m16(p, tok_row_at(p, c0, b, j, lane), r.raw[j]);
        if (bytes)
            *reinterpret_cast<uint4 *>(rows_lds + (16 * j + (lane >> 2)) * TOK_ROW_PITCH + 16 * (lane & 3)) =
                tok_in_place(p, tok_row_at(p, c0, b_bytes, j, lane), bytes->raw[j]);
    }
    __syncthreads();
    const uint64_t w = reinterpret_cast<const uint64_t *>(smask)[lane];
    __syncthreads();
    return w;
}

// Parser state of a lane between blocks.
struct TokState {
    bool open;     // the byte before the block belongs to a word
    bool dropping; // ... that is being dropped
};

// What a block holds, given the state it is entered in.
struct TokBlock {
    uint64_t keep;   // word bytes of kept words
    uint64_t starts; // first bytes of kept words
    uint64_t closes; // positions right after the last byte of a kept word
};

// pos0 = text position of the block's first byte; W, Wnext = word masks of the
// block and the one after it.
__device__ __forceinline__ TokBlock tok_block(const TokParams &p, uint64_t W, uint64_t Wnext, uint64_t pos0,
                                              TokState &st)
{
    const uint64_t prev = (W << 1) | (uint64_t) st.open;
    TokBlock r;
    r.keep = W;
    r.starts = W & ~prev;
    r.closes = ~W & prev;
    if (st.open && st.dropping) { // the word the block is entered in is not kept
        const uint32_t lead = (~W == 0) ? 64u : (uint32_t) __ffsll((unsigned long long) ~W) - 1u;
        const uint64_t m = lead >= 64 ? ~0ull : ((1ull << lead) - 1ull);
        r.keep &= ~m;
        if (lead < 64) r.closes &= ~(1ull << lead);
    }
    bool drop_last = st.open && st.dropping && ~W == 0;
    if ((W >> 63) && ~W != 0) { // a word is open at the end of the block
        const uint32_t ts = 64u - (uint32_t) __clzll((long long) ~W); // its first byte in this block
        const bool starts_here = ts > 0 || !st.open;
        if (starts_here) {
            drop_last = (~Wnext == 0) && tok_word_is_long(p, pos0 + ts);
            if (drop_last) {
                r.keep &= (1ull << ts) - 1ull;
                r.starts &= ~(1ull << ts);
            }
        }
    }
    else if ((W >> 63) && !st.open) { // the whole block is the beginning of a word
        drop_last = (~Wnext == 0) && tok_word_is_long(p, pos0);
        if (drop_last) {
            r.keep = 0;
            r.starts = 0;
        }
    }
    st.open = (W >> 63) != 0;
    st.dropping = st.open && drop_last;
    return r;
}

// State at the beginning of a piece.
__device__ __forceinline__ TokState tok_enter(const TokParams &p, uint64_t a0)
{
    TokState st;
    st.open = a0 > 0 && a0 <= p.nbytes && tok_is_word_byte(p.text[a0 - 1]);
    st.dropping = st.open && tok_word_is_long(p, a0 - 1);
    return st;
}

// Dynamic LDS: 256 block masks.
__global__ __launch_bounds__(64) void k_tok_count(TokParams p)
{
    extern __shared__ uint32_t tok_lds[];
    uint16_t *smask = reinterpret_cast<uint16_t *>(tok_lds);
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint64_t a0 = c0 + (uint64_t) lane * p.piece_bytes;
    const uint32_t nb = p.piece_bytes / 64;
    TokState st = tok_enter(p, a0);
    uint32_t nbytes = 0, nclose = 0, nstart = 0;
    UnpRows rows;
    tok_rows_load(p, c0, 0, lane, rows);
    uint64_t W = tok_rows_mask(p, c0, 0, lane, rows, smask);
    tok_rows_load(p, c0, 1, lane, rows);
    for (uint32_t b = 0; b < nb; b++) {
        const UnpRows nxt = rows;
        if (b + 2 <= nb) tok_rows_load(p, c0, b + 2, lane, rows); // in flight during the walk
        const uint64_t Wnext = tok_rows_mask(p, c0, b + 1, lane, nxt, smask);
        const TokBlock blk = tok_block(p, W, Wnext, a0 + 64u * b, st);
        nbytes += (uint32_t) __popcll(blk.keep);
        nclose += (uint32_t) __popcll(blk.closes);
        nstart += (uint32_t) __popcll(blk.starts);
        W = Wnext;
    }
    const uint32_t ib = wave_incl_scan_u32(nbytes, lane), ic = wave_incl_scan_u32(nclose, lane),
                   is = wave_incl_scan_u32(nstart, lane);
    uint32_t *px = p.piece_x + (uint64_t) c * 3 * 64;
    px[lane] = ib - nbytes;
    px[64 + lane] = ic - nclose;
    px[128 + lane] = is - nstart;
    if (lane == 63) {
        p.chunk_tot[3 * c] = ib;
        p.chunk_tot[3 * c + 1] = ic;
        p.chunk_tot[3 * c + 2] = is;
    }
}

__global__ __launch_bounds__(64) void k_tok_scan(TokParams p)
{
    const int lane = threadIdx.x;
    uint64_t base[3] = {0, 0, 0};
    for (uint32_t c0 = 0; c0 < p.nchunks; c0 += 64) {
        const uint32_t c = c0 + lane;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const uint32_t v = c < p.nchunks ? p.chunk_tot[3 * c + k] : 0u; // a chunk is at most 2^21 bytes: sums of 64 fit
            const uint32_t incl = wave_incl_scan_u32(v, lane);
            if (c < p.nchunks) p.chunk_x[3 * (uint64_t) c + k] = base[k] + (incl - v);
            base[k] += (uint32_t) __shfl((int) incl, 63);
        }
    }
    if (lane == 0) {
        const uint64_t words = base[1], bytes = base[0] + 9 * base[1];
        p.result[0] = words;
        p.result[1] = bytes;
        // every kept word starts once and closes once
        if (base[1] != base[2]) raise_error(p.ctrl, MGS_ERR_STATE);
        else if (bytes > p.out_cap || words + 1 > p.rec_cap || bytes >= (1ull << 32)) raise_error(p.ctrl, MGS_ERR_ARG);
        else p.rec_off[words] = (uint32_t) bytes;
    }
}

// A lane's output bytes.  They are gathered eight at a time in a register and
// collected in LDS in groups of 64 that are 64-byte groups of the output
// (counted from a 16-byte boundary at or below `out`); a complete group leaves
// with four 16-byte stores, the bytes of a group shared with a neighbour one by
// one.
struct TokSink {
    uint8_t *slot;
    uint8_t *out;     // moved back to a 16-byte boundary
    uint64_t v;       // next byte to write, counted from `out`
    uint64_t v_first; // first byte of this lane
    uint64_t acc;     // the (v & 7) bytes of the current 8-byte unit that are there already

    // the k low bytes of x (1 <= k <= 8, the bytes above them zero)
    __device__ __forceinline__ void append(uint64_t x, uint32_t k)
    {
        const uint32_t fill = (uint32_t) v & 7u;
        acc |= x << (8 * fill);
        if (fill + k >= 8) {
            const uint32_t unit = (uint32_t) v & 56u;
            const uint64_t g = v & ~(uint64_t) 63;
            *reinterpret_cast<uint64_t *>(slot + unit) = acc;
            acc = fill ? x >> (8 * (8 - fill)) : 0ull;
            v += k;
            if (unit == 56u) flush_group(g); // the group's last unit
        }
        else {
            v += k;
        }
    }
    __device__ __forceinline__ void flush_group(uint64_t g)
    {
        if (g >= v_first) {
            const uint4 *src = reinterpret_cast<const uint4 *>(slot);
            uint4 *dst = reinterpret_cast<uint4 *>(out + g);
#pragma unroll
            for (int i = 0; i < 4; i++) dst[i] = src[i];
        }
        else {
            store_part(g, (uint32_t) v_first & 63u, 64u);
        }
    }
    // bytes [lo, hi) of the group at g: single bytes up to an 8-byte boundary, 8-byte
    // units, single bytes again
    __device__ __forceinline__ void store_part(uint64_t g, uint32_t lo, uint32_t hi)
    {
        uint32_t e = lo;
        for (; e < hi && (e & 7u); e++) out[g + e] = slot[e];
        for (; e + 8 <= hi; e += 8)
            *reinterpret_cast<uint64_t *>(out + g + e) = *reinterpret_cast<const uint64_t *>(slot + e);
        for (; e < hi; e++) out[g + e] = slot[e];
    }
    __device__ __forceinline__ void finish()
    {
        if ((uint32_t) v & 7u) *reinterpret_cast<uint64_t *>(slot + ((uint32_t) v & 56u)) = acc;
        const uint64_t g = v & ~(uint64_t) 63;
        const uint64_t lo = max(g, v_first);
        if (lo < v) store_part(g, (uint32_t) (lo - g), (uint32_t) (v - g));
    }
};

// Eight bytes of a lane's block in LDS from byte `off` on (row: 8-byte aligned;
// bytes up to off + 15 are read).
__device__ __forceinline__ uint64_t tok_row_get8(const uint8_t *row, uint32_t off)
{
    const uint64_t *q = reinterpret_cast<const uint64_t *>(row + (off & ~7u));
    const uint64_t lo = q[0], hi = q[1];
    const uint32_t sh = 8u * (off & 7u);
    return sh ? (lo >> sh) | (hi << (64u - sh)) : lo;
}

// Dynamic LDS: 256 block masks, the 64 lanes' blocks, the lanes' output groups,
// the lanes' offset groups.
__global__ __launch_bounds__(64) void k_tok_emit(TokParams p)
{
    extern __shared__ uint32_t tok_lds[];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (p.ctrl->head.error != 0) return;
    uint16_t *smask = reinterpret_cast<uint16_t *>(tok_lds);
    uint8_t *rows_lds = reinterpret_cast<uint8_t *>(tok_lds + 128);
    uint8_t *bytes_lds = rows_lds + 64 * TOK_ROW_PITCH;
    uint32_t *offs_lds = reinterpret_cast<uint32_t *>(bytes_lds + 64 * TOK_ROW_PITCH);
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint64_t a0 = c0 + (uint64_t) lane * p.piece_bytes;
    const uint32_t nb = p.piece_bytes / 64;
    TokState st = tok_enter(p, a0);
    const uint32_t *px = p.piece_x + (uint64_t) c * 3 * 64;
    const uint64_t *cx = p.chunk_x + 3 * (uint64_t) c;
    const uint64_t closed_before = cx[1] + px[64 + lane];
    TokSink sink;
    const uint32_t back = (uint32_t) (reinterpret_cast<uintptr_t>(p.out) & 15u);
    sink.slot = bytes_lds + TOK_ROW_PITCH * lane;
    sink.out = p.out - back;
    sink.v = sink.v_first = cx[0] + px[lane] + 9 * closed_before + back;
    sink.acc = 0;
    UnpSink offs;
    const uint32_t oback = (uint32_t) ((reinterpret_cast<uintptr_t>(p.rec_off) >> 2) & 3u);
    offs.slot = offs_lds + UNP_SLOT_PITCH * lane;
    offs.out = p.rec_off - oback;
    offs.v = offs.v_first = cx[2] + px[128 + lane] + oback;
    const uint8_t *mine = rows_lds + lane * TOK_ROW_PITCH;
    const uint64_t tail0 = p.value << 8, tail1 = p.value >> 56; // NUL + value: 8 bytes, then 1

    UnpRows cur, nxt, ahead;
    tok_rows_load(p, c0, 0, lane, cur);
    uint64_t W = tok_rows_mask(p, c0, 0, lane, cur, smask);
    tok_rows_load(p, c0, 1, lane, nxt);
    for (uint32_t b = 0; b < nb; b++) {
        if (b + 2 <= nb) tok_rows_load(p, c0, b + 2, lane, ahead); // in flight during the walk
        // this block's bytes to its lanes, the next block's masks
        const uint64_t Wnext = tok_rows_mask(p, c0, b + 1, lane, nxt, smask, &cur, b, rows_lds);
        const TokBlock blk = tok_block(p, W, Wnext, a0 + 64u * b, st);
        // A word carried into the block may end right at its first byte; every other end
        // of a word follows a run of kept bytes of this block.  So: that one first, then
        // run by run (a word that crosses a block boundary is two runs), each with the
        // end of its word if it is here -- the same steps in every lane.
        if (blk.closes & 1ull) {
            sink.append(tail0, 8); // NUL + value (src/serializer.h:63-83, :157-168)
            sink.append(tail1, 1);
        }
        uint64_t runs = blk.keep & ~(blk.keep << 1);
        while (runs) {
            const uint32_t i = (uint32_t) __ffsll((unsigned long long) runs) - 1u;
            runs &= runs - 1;
            if ((blk.starts >> i) & 1ull) offs.put((uint32_t) (sink.v - back));
            const uint64_t rest = ~(blk.keep >> i);
            const uint32_t len = rest ? (uint32_t) __ffsll((unsigned long long) rest) - 1u : 64u - i;
            const uint64_t x = tok_row_get8(mine, i);
            sink.append(len < 8 ? x & ((1ull << (8 * len)) - 1ull) : x, min(8u, len));
            for (uint32_t j = 8; j < len; j += 8) { // rare: words of more than 8 bytes
                const uint32_t k = min(8u, len - j);
                const uint64_t y = tok_row_get8(mine, i + j);
                sink.append(k < 8 ? y & ((1ull << (8 * k)) - 1ull) : y, k);
            }
            if (i + len < 64 && ((blk.closes >> (i + len)) & 1ull)) {
                sink.append(tail0, 8);
                sink.append(tail1, 1);
            }
        }
        W = Wnext;
        cur = nxt;
        nxt = ahead;
    }
    sink.finish();
    offs.finish();
}

} // namespace mgs
#endif
