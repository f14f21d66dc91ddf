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
// text is cut into chunks, one wave each; a wave streams through its chunk 1 KiB
// at a time, lane = 16 consecutive bytes: one coalesced 16-byte load per lane, a
// 16-bit mask of word bytes, and the three counts of a step are one packed wave
// scan.
//
//   k_tok_count   per chunk: kept word bytes, words closed, words started.
//   k_tok_scan    one workgroup: exclusive sums over the chunks, capacity checks.
//   k_tok_stream  the same walk again; every lane writes its runs of word bytes,
//                 the NUL + value after a word that closes in its bytes and the
//                 offsets of the words that start there into an LDS image of the
//                 step's output, which leaves with aligned 16-byte stores.
//
// The 1024-byte rule needs the length of a word where it starts.  A word that
// starts at byte o of a step and runs to the step's end is 1024 - o bytes there
// plus the leading run of word bytes of the next step, whose masks are at hand
// one step ahead; if that whole step is word bytes the word is long in any
// case.  So a wave carries two flags from step to step -- a word is open, it is
// being dropped -- and measures a word on the text itself (tok_word_is_long)
// only where its chunk begins inside one.
#ifndef MGS_TOKENIZE_H
#define MGS_TOKENIZE_H

#include "mgs_unpack.h"

namespace mgs {

constexpr uint32_t TOK_MAX_WORD = 1024; // strlen(word) < 1024, examples/wordcount.cpp:72
constexpr int TOK_CHUNKS = 4096;        // chunks (one wave each) a large text is cut into: four waves per SIMD
constexpr int TOKS_STEP = 1024;         // text bytes per step of a wave
constexpr int TOKS_OBUF = 16 + 9 * 512 + 1024 + 32; // output of a step at most: every byte kept, a word closed every other byte
constexpr int TOKS_RBUF = 4 + 512 + 8;              // record offsets of a step

struct TokParams {
    const uint8_t *text;
    uint64_t nbytes;
    uint32_t chunk_bytes; // multiple of 4 * TOKS_STEP
    uint32_t nchunks;     // covers nbytes + 1 bytes: the end of the text closes an open word
    uint32_t *chunk_tot;  // [nchunks][3] kept word bytes / words closed / words started in the chunk
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

// The lane's 16 bytes of text at `at` and their word-byte mask: bit i set = text[at + i]
// is a word byte; bytes past the end are not.  The text has at least 16 bytes (shorter ones
// go through k_tok_tiny): a lane that sticks out of it fetches the text's last 16 bytes.
// Nothing but this one load on purpose -- a second path with loads of its own would make
// every wait for memory in the callers' loops a wait for all of it.
__device__ __forceinline__ uint4 tok_load16(const TokParams &p, uint64_t at)
{
    const uint64_t a = min(at, p.nbytes - 16);
    const UnpU128 v = *reinterpret_cast<const UnpU128 *>(p.text + a);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t tok_m16(const TokParams &p, uint64_t at, const uint4 &v)
{
    // 0x80 per delimiter byte (0x00, 0x0a, 0x20).  A delimiter has no bit outside 0x2a, and of
    // the eight bytes that have none (bits 1, 3, 5 free) the three delimiters are those with
    // bit 1 == bit 3 and not (bit 1 and bit 5): one zero-byte test instead of three.
    auto delim = [](uint32_t w) {
        const uint32_t c = w & 0xd5d5d5d5u;
        const uint32_t cand = ~(((c & 0x7f7f7f7fu) + 0x7f7f7f7fu) | c | 0x7f7f7f7fu); // 0x80 per byte inside 0x2a
        const uint32_t bad = ((w ^ (w >> 2)) | (w & (w >> 4))) & 0x02020202u;          // bit 1: not a delimiter after all
        return cand & ~(bad << 6);
    };
    // one mask bit per byte, lowest address first: a dot product of the flag bytes with powers
    // of two (flags are 0x80: the sums come out times 128)
    const uint32_t lo = __builtin_amdgcn_udot4(delim(v.y), 0x80402010u, __builtin_amdgcn_udot4(delim(v.x), 0x08040201u, 0u, false), false);
    const uint32_t hi = __builtin_amdgcn_udot4(delim(v.w), 0x80402010u, __builtin_amdgcn_udot4(delim(v.z), 0x08040201u, 0u, false), false);
    const uint32_t m = ~((lo >> 7) | ((hi >> 7) << 8)) & 0xFFFFu;
    if (at >= p.nbytes) return 0;
    if (at + 16 > p.nbytes) return m >> (uint32_t) (at - (p.nbytes - 16));
    return m;
}

// The bytes tok_m16 speaks of, in place (a piece that sticks out of the text was
// fetched 16 bytes back from its end).
__device__ __forceinline__ uint4 tok_in_place(const TokParams &p, uint64_t at, const uint4 &v)
{
    if (at + 16 <= p.nbytes || at >= p.nbytes) return v;
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

// A text of fewer than 16 bytes: one lane does everything.
__global__ __launch_bounds__(64) void k_tok_tiny(TokParams p)
{
    if (threadIdx.x != 0) return;
    uint64_t words = 0, bytes = 0;
    for (uint64_t i = 0; i < p.nbytes; i++) {
        if (!tok_is_word_byte(p.text[i])) continue;
        bytes++;
        if (i + 1 == p.nbytes || !tok_is_word_byte(p.text[i + 1])) words++;
    }
    bytes += 9 * words;
    p.result[0] = words;
    p.result[1] = bytes;
    if (bytes > p.out_cap || words + 1 > p.rec_cap) {
        raise_error(p.ctrl, MGS_ERR_ARG);
        return;
    }
    uint64_t v = 0, n = 0;
    bool open = false;
    for (uint64_t i = 0; i <= p.nbytes; i++) {
        const bool w = i < p.nbytes && tok_is_word_byte(p.text[i]);
        if (w) {
            if (!open) p.rec_off[n++] = (uint32_t) v;
            p.out[v++] = p.text[i];
        }
        else if (open) { // NUL + value (src/serializer.h:63-83, :157-168)
            p.out[v++] = 0;
            for (int k = 0; k < 8; k++) p.out[v++] = (uint8_t) (p.value >> (8 * k));
        }
        open = w;
    }
    p.rec_off[n] = (uint32_t) v;
}

// State of the walk between steps.
struct TokState {
    bool open;     // the byte before the step belongs to a word
    bool dropping; // ... that is being dropped
};

// State at the beginning of a chunk.
__device__ __forceinline__ TokState tok_enter(const TokParams &p, uint64_t a0)
{
    TokState st;
    st.open = a0 > 0 && a0 <= p.nbytes && tok_is_word_byte(p.text[a0 - 1]);
    st.dropping = st.open && tok_word_is_long(p, a0 - 1);
    return st;
}

// leading run of word bytes of a step, from the lanes' 16-bit masks (0..1024)
__device__ __forceinline__ uint32_t toks_lead(uint32_t m)
{
    const uint64_t notfull = __ballot(m != 0xFFFFu);
    if (!notfull) return TOKS_STEP;
    const int l = __ffsll((unsigned long long) notfull) - 1;
    const uint32_t ml = (uint32_t) __builtin_amdgcn_readlane((int) m, l);
    return 16u * (uint32_t) l + (uint32_t) __builtin_ctz(~ml | 0x10000u);
}

// What a lane's 16 bytes hold, given the state the step is entered in.
struct ToksLane {
    uint32_t keep;   // word bytes of kept words
    uint32_t starts; // first bytes of kept words
    uint32_t closes; // positions right after the last byte of a kept word
};

// m, m_n = the lanes' word masks of this step and the next; st: in = before the step,
// out = after it.
__device__ __forceinline__ ToksLane toks_step(uint32_t m, uint32_t m_n, TokState &st, int lane)
{
    const bool open_in = st.open, dropping_in = st.dropping;
    // the byte before the lane's 16 (a word that ends with a lane's last byte closes in the next lane)
    const uint32_t up = (uint32_t) __shfl_up((int) m, 1);
    const uint32_t pb = lane ? (up >> 15) & 1u : (open_in ? 1u : 0u);
    // the step's words that are dropped: the one it is entered in, the one open at its end
    const uint32_t lead = toks_lead(m);
    const bool open_out = (((uint32_t) __builtin_amdgcn_readlane((int) m, 63) >> 15) & 1u) != 0;
    uint32_t drop_from = TOKS_STEP; // bytes from here on belong to a dropped word
    bool dropping_out = false;
    if (open_out) {
        if (lead == TOKS_STEP) {
            dropping_out = open_in ? dropping_in : true; // a word that fills a whole step is long
            if (dropping_out) drop_from = 0;
        }
        else {
            // first byte of the trailing run
            const uint64_t notfull = __ballot(m != 0xFFFFu);
            const int l = 63 - __builtin_clzll((unsigned long long) notfull);
            const uint32_t ml = (uint32_t) __builtin_amdgcn_readlane((int) m, l);
            const uint32_t o = 16u * (uint32_t) l + (32u - (uint32_t) __builtin_clz(~ml & 0xFFFFu));
            dropping_out = toks_lead(m_n) >= o; // 1024 - o bytes here + the next step's leading run >= 1024
            if (dropping_out) drop_from = o;
        }
    }
    const bool drop_in = open_in && dropping_in; // the step's first `lead` bytes belong to a dropped word
    const uint32_t prevm = ((m << 1) | pb) & 0xFFFFu;
    ToksLane r;
    r.keep = m;
    r.starts = m & ~prevm;
    r.closes = ~m & prevm & 0xFFFFu;
    const uint32_t b0 = 16u * (uint32_t) lane; // the lane's first byte in the step
    if (drop_in && lead > b0) {
        const uint32_t k = lead - b0; // the lane's first k bytes are dropped (k may pass 16)
        const uint32_t mk = k >= 16 ? 0xFFFFu : ((1u << k) - 1u);
        r.keep &= ~mk;
        if (k < 16) r.closes &= ~(1u << k); // the dropped word's end is no record's end
    }
    else if (drop_in && lead == b0) {
        r.closes &= ~1u;
    }
    if (drop_from < b0 + 16u) {
        const uint32_t k = drop_from > b0 ? drop_from - b0 : 0u;
        const uint32_t mk = ~((1u << k) - 1u) & 0xFFFFu;
        r.keep &= ~mk;
        r.starts &= ~mk;
    }
    st.open = open_out;
    st.dropping = dropping_out;
    return r;
}

__global__ __launch_bounds__(64) void k_tok_count(TokParams p)
{
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint32_t nsteps = p.chunk_bytes / TOKS_STEP;
    auto at_of = [&](uint32_t s) { return c0 + (uint64_t) s * TOKS_STEP + 16u * (uint32_t) lane; };
    TokState st = tok_enter(p, c0);
    uint32_t nbytes = 0, nclose = 0, nstart = 0;
    // Three steps' bytes in flight in a ring of four registers (a ring of eight was timed: no
    // faster, the kernel is bound by its vector instructions, not by the loads).  The loop is
    // unrolled by the ring size so that no loaded value is moved (a move would wait for its
    // load); a step's mask is made when it is next in line.
    uint4 R[4];
    R[0] = tok_load16(p, at_of(0));
#pragma unroll
    for (int i = 1; i < 4; i++) R[i] = tok_load16(p, at_of((uint32_t) i));
    uint32_t m = tok_m16(p, at_of(0), R[0]);
    for (uint32_t s0 = 0; s0 < nsteps; s0 += 4) { // chunk_bytes is a multiple of 4 steps
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t s = s0 + (uint32_t) i;
            R[i] = tok_load16(p, at_of(s + 4)); // past the chunk: the next chunk's bytes, or nothing
            const uint32_t m_n = tok_m16(p, at_of(s + 1), R[(i + 1) & 3]);
            const ToksLane x = toks_step(m, m_n, st, lane);
            nbytes += (uint32_t) __popc(x.keep);
            nclose += (uint32_t) __popc(x.closes);
            nstart += (uint32_t) __popc(x.starts);
            m = m_n;
        }
    }
    const uint32_t ib = wave_incl_scan_u32(nbytes, lane), ic = wave_incl_scan_u32(nclose, lane),
                   is = wave_incl_scan_u32(nstart, lane);
    if (lane == 63) {
        p.chunk_tot[3 * c] = ib;
        p.chunk_tot[3 * c + 1] = ic;
        p.chunk_tot[3 * c + 2] = is;
    }
}

// One workgroup: thread t owns a run of consecutive chunks.
__global__ __launch_bounds__(1024) void k_tok_scan(TokParams p)
{
    __shared__ uint64_t sh[3][1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (p.nchunks + 1023u) / 1024u;
    const uint32_t lo = min(t * per, p.nchunks), hi = min(lo + per, p.nchunks);
    const uint32_t *__restrict__ tot = p.chunk_tot;
    uint64_t *__restrict__ before = p.chunk_x;
    // a thread's chunks eight at a time, so that the loads of a batch are in flight together
    constexpr uint32_t TB = 8;
    uint64_t sum[3] = {0, 0, 0};
    for (uint32_t c = lo; c < hi; c += TB) {
        uint32_t v[TB][3];
#pragma unroll
        for (uint32_t i = 0; i < TB; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) v[i][k] = c + i < hi ? tot[3 * (c + i) + k] : 0u;
#pragma unroll
        for (uint32_t i = 0; i < TB; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) sum[k] += v[i][k];
    }
#pragma unroll
    for (int k = 0; k < 3; k++) sh[k][t] = sum[k];
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint64_t add[3];
#pragma unroll
        for (int k = 0; k < 3; k++) add[k] = t >= d ? sh[k][t - d] : 0ull;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; k++) sh[k][t] += add[k];
        __syncthreads();
    }
    uint64_t base[3];
#pragma unroll
    for (int k = 0; k < 3; k++) base[k] = sh[k][t] - sum[k];
    for (uint32_t c = lo; c < hi; c += TB) {
        uint32_t v[TB][3];
#pragma unroll
        for (uint32_t i = 0; i < TB; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) v[i][k] = c + i < hi ? tot[3 * (c + i) + k] : 0u;
#pragma unroll
        for (uint32_t i = 0; i < TB; i++) {
            if (c + i < hi) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    before[3 * (uint64_t) (c + i) + k] = base[k];
                    base[k] += v[i][k];
                }
            }
        }
    }
    if (t == 1023) {
        const uint64_t words = sh[1][1023], bytes = sh[0][1023] + 9 * sh[1][1023];
        p.result[0] = words;
        p.result[1] = bytes;
        // every kept word starts once and closes once
        if (sh[1][1023] != sh[2][1023]) raise_error(p.ctrl, MGS_ERR_STATE);
        else if (bytes > p.out_cap || words + 1 > p.rec_cap || bytes >= (1ull << 32)) raise_error(p.ctrl, MGS_ERR_ARG);
        else p.rec_off[words] = (uint32_t) bytes;
    }
}

struct __attribute__((packed)) TokU16 { uint16_t v; };
struct __attribute__((packed)) TokU32 { uint32_t v; };
struct __attribute__((packed)) TokU64 { unsigned long long v; };

// len (1..8) low bytes of x to LDS at any alignment
__device__ __forceinline__ void toks_put(uint8_t *o, unsigned long long x, uint32_t len)
{
    if (len >= 8) {
        reinterpret_cast<TokU64 *>(o)->v = x;
        return;
    }
    if (len & 4u) {
        reinterpret_cast<TokU32 *>(o)->v = (uint32_t) x;
        x >>= 32;
        o += 4;
    }
    if (len & 2u) {
        reinterpret_cast<TokU16 *>(o)->v = (uint16_t) x;
        x >>= 16;
        o += 2;
    }
    if (len & 1u) *o = (uint8_t) x;
}

// bytes i .. i+7 of the 16 bytes (lo, hi); bytes past the 16th are zero
__device__ __forceinline__ unsigned long long toks_get8(unsigned long long lo, unsigned long long hi, uint32_t i)
{
    if (i >= 8) return hi >> (8u * (i - 8u));
    return i ? (lo >> (8u * i)) | (hi << (64u - 8u * i)) : lo;
}

__global__ __launch_bounds__(64) void k_tok_stream(TokParams p)
{
    __shared__ __attribute__((aligned(16))) uint8_t obuf[TOKS_OBUF];
    __shared__ __attribute__((aligned(16))) uint32_t rbuf[TOKS_RBUF];
    const int lane = threadIdx.x;
    const uint32_t c = blockIdx.x;
    if (p.ctrl->head.error != 0) return;
    const uint64_t c0 = (uint64_t) c * p.chunk_bytes;
    const uint32_t nsteps = p.chunk_bytes / TOKS_STEP;
    const uint64_t *cx = p.chunk_x + 3 * (uint64_t) c;
    const unsigned long long tail0 = p.value << 8, tail1 = p.value >> 56; // NUL + value: 8 bytes, then 1

    // positions count from 16-byte boundaries at or below the outputs
    const uint32_t back = (uint32_t) (reinterpret_cast<uintptr_t>(p.out) & 15u);
    uint8_t *const gout = p.out - back;
    const uint32_t oback = (uint32_t) ((reinterpret_cast<uintptr_t>(p.rec_off) >> 2) & 3u);
    uint32_t *const goff = p.rec_off - oback;
    uint64_t A = cx[0], B = cx[1], S = cx[2]; // kept bytes / closed / started words before the step
    const uint64_t v_first = A + 9 * B + back, r_first = S + oback;
    uint64_t fl = v_first & ~(uint64_t) 15, rfl = r_first & ~(uint64_t) 3; // what lies below is stored (or not ours)

    auto at_of = [&](uint32_t s) { return c0 + (uint64_t) s * TOKS_STEP + 16u * (uint32_t) lane; };
    TokState st = tok_enter(p, c0);
    // three steps' bytes in flight in a ring of four registers (the loop is unrolled by four so
    // that no loaded value is moved: a move would wait for its load); a step's mask is made
    // when it is next in line
    uint4 R0 = tok_load16(p, at_of(0)), R1 = tok_load16(p, at_of(1)), R2 = tok_load16(p, at_of(2)), R3;
    uint32_t m = tok_m16(p, at_of(0), R0);
    auto step = [&](uint32_t s, const uint4 &cur, const uint4 &nxt, uint4 &fill) {
        fill = tok_load16(p, at_of(s + 3)); // past the chunk: the next chunk's bytes, or nothing
        const uint32_t m_n = tok_m16(p, at_of(s + 1), nxt);
        const uint4 bytes = tok_in_place(p, at_of(s), cur);
        const ToksLane x = toks_step(m, m_n, st, lane);
        const uint32_t keep = x.keep, starts = x.starts, closes = x.closes;
        // where the lane's output goes: one scan of three packed counts
        const uint32_t kb = (uint32_t) __popc(keep), cl = (uint32_t) __popc(closes), sn = (uint32_t) __popc(starts);
        const uint32_t packed = kb | (cl << 11) | (sn << 21);
        const uint32_t incl = wave_incl_scan_u32(packed, lane);
        const uint32_t excl = incl - packed;
        const uint32_t tot = (uint32_t) __builtin_amdgcn_readlane((int) incl, 63);
        const uint32_t tA = tot & 0x7FFu, tB = (tot >> 11) & 0x3FFu, tS = tot >> 21;
        // positions inside the LDS images: obuf[0] is output byte fl, rbuf[0] is offset rfl
        const uint32_t vbase = (uint32_t) (A + 9 * B + back - fl), rbase = (uint32_t) (S + oback - rfl); // < 16, < 4
        const uint32_t opos = (uint32_t) (fl - back); // obuf[0] as a position in the records
        uint32_t v = vbase + (excl & 0x7FFu) + 9u * ((excl >> 11) & 0x3FFu); // next output byte of this lane
        uint32_t r = rbase + (excl >> 21);                                    // next record offset of this lane
        // A word carried into the lane's bytes may end right at their first byte; every other
        // end of a word follows a run of kept bytes of this lane.  So: that one first, then
        // run by run, each with the end of its word if it is here.
        const unsigned long long lo = (unsigned long long) bytes.x | ((unsigned long long) bytes.y << 32);
        const unsigned long long hi = (unsigned long long) bytes.z | ((unsigned long long) bytes.w << 32);
        if (closes & 1u) { // NUL + value (src/serializer.h:63-83, :157-168)
            toks_put(obuf + v, tail0, 8);
            obuf[v + 8] = (uint8_t) tail1;
            v += 9;
        }
        uint32_t runs = keep & ~(keep << 1);
        while (runs) {
            const uint32_t i = (uint32_t) __builtin_ctz(runs);
            runs &= runs - 1;
            if ((starts >> i) & 1u) rbuf[r++] = opos + v;
            const uint32_t rest = ~(keep >> i);
            const uint32_t len = (uint32_t) __builtin_ctz(rest); // keep has 16 bits: the run ends inside them
            toks_put(obuf + v, toks_get8(lo, hi, i), min(8u, len));
            if (len > 8) toks_put(obuf + v + 8, toks_get8(lo, hi, i + 8), len - 8);
            v += len;
            if (i + len < 16 && ((closes >> (i + len)) & 1u)) {
                toks_put(obuf + v, tail0, 8);
                obuf[v + 8] = (uint8_t) tail1;
                v += 9;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // The step's output: whole 16-byte units leave, the rest is carried to the next step.
        // Always the same number of (predicated) store instructions, so that the waits for
        // the text loads can leave the stores in flight.
        A += tA;
        B += tB;
        S += tS;
        const uint32_t top = (vbase + tA + 9u * tB) & ~15u, rtop = (rbase + tS) & ~3u;
        const bool head = fl < v_first, rhead = rfl < r_first; // the chunk's first unit is shared with the chunk before
        uint8_t *const gdst = gout + fl;
        uint32_t *const rdst = goff + rfl;
#pragma unroll
        for (int j = 0; j < (TOKS_OBUF + 1023) / 1024; j++) {
            const uint32_t o = 16u * ((uint32_t) lane + 64u * (uint32_t) j);
            if (o < top && !(head && o == 0))
                *reinterpret_cast<uint4 *>(gdst + o) = *reinterpret_cast<const uint4 *>(obuf + o);
        }
#pragma unroll
        for (int j = 0; j < (TOKS_RBUF + 255) / 256; j++) {
            const uint32_t o = 4u * ((uint32_t) lane + 64u * (uint32_t) j);
            if (o < rtop && !(rhead && o == 0))
                *reinterpret_cast<uint4 *>(rdst + o) = *reinterpret_cast<const uint4 *>(rbuf + o);
        }
        if (head && top > 0 && lane < 16 && fl + (uint32_t) lane >= v_first) gdst[lane] = obuf[lane];
        if (rhead && rtop > 0 && lane < 4 && rfl + (uint32_t) lane >= r_first) rdst[lane] = rbuf[lane];
        __builtin_amdgcn_wave_barrier();
        if (top > 0) {
            uint4 carry = make_uint4(0, 0, 0, 0);
            if (lane == 0) carry = *reinterpret_cast<const uint4 *>(obuf + top);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) *reinterpret_cast<uint4 *>(obuf) = carry;
            fl += top;
        }
        if (rtop > 0) {
            uint4 carry = make_uint4(0, 0, 0, 0);
            if (lane == 0) carry = *reinterpret_cast<const uint4 *>(rbuf + rtop);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) *reinterpret_cast<uint4 *>(rbuf) = carry;
            rfl += rtop;
        }
        __builtin_amdgcn_wave_barrier();
        m = m_n;
    };
    for (uint32_t s = 0; s < nsteps; s += 4) { // chunk_bytes is a multiple of 4 steps
        step(s, R0, R1, R3);
        step(s + 1, R1, R2, R0);
        step(s + 2, R2, R3, R1);
        step(s + 3, R3, R0, R2);
    }
    // what is left below a 16-byte boundary belongs to this chunk up to its end
    {
        const uint64_t v_end = A + 9 * B + back, r_end = S + oback;
        const uint64_t lo_v = fl > v_first ? fl : v_first;
        for (uint64_t g = lo_v + (uint32_t) lane; g < v_end; g += 64) gout[g] = obuf[g - fl];
        const uint64_t lo_r = rfl > r_first ? rfl : r_first;
        for (uint64_t g = lo_r + (uint32_t) lane; g < r_end; g += 64) goff[g] = rbuf[g - rfl];
    }
}

} // namespace mgs
#endif
