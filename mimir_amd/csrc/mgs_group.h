// Reduce side: KV -> KMV (group the records of a rank by key) on the device.
//
// What it replaces: KMVContainer::convert (src/kmvcontainer.h:74-150) reads
// the KV container twice.  Pass 1 looks every key up in a chained hash table
// (HashBucket::findEntry / insertEntry, src/hashbucket.h:108-126, :185-290: a
// new key is appended to the table's entry buffers, so HashBucket::next,
// :301-342, walks the unique keys in the order of their first occurrence) and
// adds up the value bytes per key; the unique keys are then given consecutive
// slices of value buffers in that order (:98-130: a buffer is closed with the
// entry that takes it to DATA_PAGE_SIZE or beyond); pass 2 appends every
// record's value to its key's slice, in stream order (:132-144).  The reduce
// callback then sees the unique keys in first-occurrence order, each with its
// values in stream order (src/mimircontext.h:397-417).
//
// So the result is a stable group-by with groups ordered by first occurrence:
//   group_first[g]  index of the first record with the g-th unique key
//   group_start[g]  number of records in groups before g (values are fixed
//                   width: value j of the output belongs to record sorted[j])
//   values          the records' values, grouped
//   cuts            the groups at which a new value buffer begins
//
//   k_grp_insert  open-addressing table: a slot holds the key's hash and the place
//                 (byte offset or index) of the first record seen so far with
//                 that key -- keys are compared in the stream itself, and only
//                 when the hashes agree; slot of every record
//   k_grp_count / k_grp_scan / k_grp_rank
//                 a record is a first occurrence iff its slot holds its own
//                 place; prefix sum over those flags = the group's rank, left in
//                 a second table by the first record
//   k_grp_keys    rank of every record's group (the sort key), left with the record's
//                 index and the first 8 bytes of its value as a 16-byte sort record;
//                 histograms of the ranks' digits
//   (sort)        stable least-significant-digit radix sort of those records on the
//                 bits the ranks need: passes of at most 6 bits, each pass one launch
//                 of the 16-byte one-sweep partition kernel with the digit as the
//                 destination (mgs_onesweep16w.h, OSV_DIGIT) and the partitions laid
//                 end to end by k_grp_sort_setup
//   k_grp_emit    values in sorted order (out of the sort records when a value is at
//                 most 8 bytes long: no gather); group boundaries
//   k_grp_pages   value-buffer cuts, 64 probes per search step
#ifndef MGS_GROUP_H
#define MGS_GROUP_H

#include "mgs_generic.h"

namespace mgs {

constexpr uint32_t GRP_EMPTY = 0xFFFFFFFFu;
constexpr unsigned long long GRP_EMPTY64 = ~0ull;
constexpr int GRP_THREADS = 256, GRP_ITEMS = 8, GRP_TILE = GRP_THREADS * GRP_ITEMS;
constexpr int GRP_SORT_MAX_PASSES = 6, GRP_SORT_BINS = 64;

// How ranks in [0, groups) are sorted: npass passes from the least significant bits up,
// pass i on width[i] <= 6 bits at shift[i] (the widths differ by at most one).
struct GrpSortPlan {
    int npass;
    uint32_t shift[GRP_SORT_MAX_PASSES], width[GRP_SORT_MAX_PASSES];
};

__host__ __device__ inline GrpSortPlan grp_sort_plan(uint64_t groups)
{
    GrpSortPlan s;
    uint32_t bits = 1;
    while (bits < 32 && (1ull << bits) < groups) bits++;
    s.npass = (int) ((bits + 5) / 6);
    const uint32_t base = bits / (uint32_t) s.npass, extra = bits % (uint32_t) s.npass;
    uint32_t sh = 0;
    for (int i = 0; i < GRP_SORT_MAX_PASSES; i++) {
        s.width[i] = i < s.npass ? base + ((uint32_t) i < extra ? 1u : 0u) : 0u;
        s.shift[i] = sh;
        sh += s.width[i];
    }
    return s;
}

struct GrpParams {
    const uint8_t *blob;
    const uint32_t *rec_off; // record index -> byte offset, or NULL (fixed width)
    uint32_t fixed_len;
    uint32_t n;              // records
    mgs_layout lay;
    uint32_t vbytes;         // value bytes per record (0 = no value)
    // [slots * table_stride] per slot: hash << 32 | place of the first record with a key (grp_place), or
    // GRP_EMPTY64; with table_stride == 2 followed by the key's first 8 bytes (zero-padded; GRP_EMPTY64 =
    // not written yet), so that most keys are compared without a second, dependent read into the stream
    unsigned long long *table;
    uint32_t table_stride;   // 1 or 2
    uint32_t *rank_table;    // [slots] rank of the key's group
    uint32_t slot_mask, slot_shift;
    uint32_t *slot_or_rank;  // [n] slot of record i (| cand_bit, see there)
    // 0x80000000 or 0: with the bit, k_grp_insert marks in slot_or_rank the records that may be the
    // first with their key (they opened their slot or lowered its place); a record that met an
    // earlier one is never the first, so k_grp_count looks only the marked ones up again
    uint32_t cand_bit;
    // sort records {rank of the group, record index, first 8 value bytes}: pass 0 reads rec_a and
    // writes rec_b, pass 1 the other way round, ...; `sorted` is where the last pass left them
    uint4 *rec_a, *rec_b;
    const uint4 *sorted;
    uint32_t *hist;          // [GRP_SORT_MAX_PASSES][GRP_SORT_BINS] records per digit value and pass
    Ctrl *sort_ctrl;         // [GRP_SORT_MAX_PASSES] control blocks of the passes
    uint32_t *tile_count;    // [ntiles + 1] first occurrences per tile, then exclusive sums (+ total)
    uint64_t *first_bits;    // [ntiles * GRP_TILE / 64] bit i: record i is the first with its key
    uint32_t ntiles;
    uint32_t *group_first;   // out [groups]
    uint32_t *group_start;   // out [groups + 1]
    uint64_t group_cap;
    uint8_t *values;         // out [n * vbytes]
    int64_t page_bytes;
    uint64_t *cuts;          // out [max_cuts]
    uint32_t max_cuts;
    uint64_t *result;        // [0] groups, [1] cuts
    Ctrl *ctrl;
    uint8_t *dest_out;       // optional [n]: hashlittle(key) % fm.d of every record (the combiner needs it)
    FastMod fm;
};

struct __attribute__((packed)) GrpU64 { unsigned long long v; };

__device__ __forceinline__ uint64_t grp_offset(const GrpParams &p, uint64_t i)
{
    return p.rec_off ? (uint64_t) p.rec_off[i] : i * (uint64_t) p.fixed_len;
}

// Where a record's key is, as the table keeps it: the record's byte offset
// (variable width: the key is found without another look-up) or its index
// (fixed width).  Ascends with the record index either way.
__device__ __forceinline__ uint32_t grp_place(const GrpParams &p, uint64_t i)
{
    return p.rec_off ? p.rec_off[i] : (uint32_t) i;
}

__device__ __forceinline__ const uint8_t *grp_key_at(const GrpParams &p, uint32_t place)
{
    return p.blob + (p.rec_off ? (uint64_t) place : (uint64_t) place * p.fixed_len);
}

// Key of record i: first byte and length (false: the record does not match the layout).
__device__ __forceinline__ bool grp_key(const GrpParams &p, uint32_t i, const uint8_t *&key, uint32_t &klen)
{
    const uint64_t o0 = grp_offset(p, i), o1 = grp_offset(p, (uint64_t) i + 1);
    key = p.blob + o0;
    if (o1 <= o0 || o1 - o0 > GEN_MAX_RECORD) return false;
    return gen_key_len(p.lay, key, (uint32_t) (o1 - o0), klen);
}

// Is the key at b (a record's first byte) the key a[0..alen)?  Only alen is
// needed: keys of one layout that agree over the length of one of them, its
// terminating NULs included, are equal; the first difference lies inside both.
// `b_room` = bytes that may be read at b.
__device__ __forceinline__ bool grp_same_key(const uint8_t *a, uint32_t alen, const uint8_t *b, uint64_t b_room)
{
    uint32_t j = 0;
    for (; j + 8 <= alen && j + 8 <= b_room; j += 8)
        if (reinterpret_cast<const GrpU64 *>(a + j)->v != reinterpret_cast<const GrpU64 *>(b + j)->v) return false;
    for (; j < alen; j++)
        if (j >= b_room || a[j] != b[j]) return false;
    return true;
}

// One thread per record.  The hash is the reference's (hashlittle over the key
// bytes); the slot is taken from its high bits after a multiplication, because
// all keys of a rank share hash % nranks (the reference divides by the number
// of ranks for the same reason, src/hashbucket.h:111).
__global__ __launch_bounds__(GRP_THREADS) void k_grp_insert(GrpParams p)
{
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    const uint64_t blob_bytes = grp_offset(p, p.n); // the records end here: nothing is read beyond
    for (uint64_t i64 = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i64 < p.n; i64 += stride) {
        const uint32_t i = (uint32_t) i64;
        const uint8_t *key;
        uint32_t klen = 0;
        if (!grp_key(p, i, key, klen)) {
            raise_error(p.ctrl, MGS_ERR_MALFORMED);
            p.slot_or_rank[i] = 0;
            continue;
        }
        const uint32_t h = hashlittle_bytes(key, klen);
        if (p.dest_out) p.dest_out[i] = (uint8_t) fastmod(h, p.fm);
        uint32_t s = (h * 0x9E3779B1u) >> p.slot_shift;
        const uint32_t place = grp_place(p, i);
        const unsigned long long mine = ((unsigned long long) h << 32) | place;
        const uint32_t ts = p.table_stride;
        unsigned long long k8 = 0; // the key's first 8 bytes, zero-padded
        if (ts == 2) {
            if ((uint64_t) (key - p.blob) + 8 <= blob_bytes) {
                k8 = reinterpret_cast<const GrpU64 *>(key)->v;
                if (klen < 8) k8 &= (1ull << (8 * klen)) - 1ull;
            }
            else {
                for (uint32_t b = 0; b < klen && b < 8; b++) k8 |= (unsigned long long) key[b] << (8 * b);
            }
        }
        bool cand = false;
        for (uint32_t probes = 0; probes <= p.slot_mask; probes++) { // the table is at most half full
            unsigned long long *e = p.table + (size_t) s * ts;
            unsigned long long cur = __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long cur8 = ts == 2 ? __hip_atomic_load(e + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : GRP_EMPTY64;
            if (cur == GRP_EMPTY64) {
                cur = atomicCAS(e, GRP_EMPTY64, mine);
                if (cur == GRP_EMPTY64) { // this record opened the slot
                    if (ts == 2) __hip_atomic_store(e + 1, k8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cand = true;
                    break;
                }
                cur8 = GRP_EMPTY64; // opened by another record just now: its key bytes may not be there yet
            }
            if ((uint32_t) (cur >> 32) == h) { // same hash: compare the keys
                bool same;
                if (cur == mine) same = true;
                else if (cur8 != GRP_EMPTY64 && cur8 != k8) same = false;
                else if (cur8 != GRP_EMPTY64 && klen <= 8) same = true; // all of the key is in the slot
                else { // in the stream itself
                    const uint8_t *okey = grp_key_at(p, (uint32_t) cur);
                    same = grp_same_key(key, klen, okey, blob_bytes - (uint64_t) (okey - p.blob));
                }
                if (same) {
                    if (mine < cur) atomicMin(e, mine); // the slot keeps the first occurrence
                    cand = mine <= cur;
                    break;
                }
            }
            s = (s + 1) & p.slot_mask;
        }
        p.slot_or_rank[i] = s | (cand ? p.cand_bit : 0u);
    }
}

// First occurrences per tile of GRP_TILE records: their number, and one bit per
// record (wave `w` of the tile takes records k * GRP_THREADS + 64 w + lane).
__global__ __launch_bounds__(GRP_THREADS) void k_grp_count(GrpParams p)
{
    __shared__ uint32_t wsum[GRP_THREADS / 64];
    const uint64_t t0 = (uint64_t) blockIdx.x * GRP_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        const uint64_t i = t0 + (uint64_t) k * GRP_THREADS + threadIdx.x;
        bool first = false;
        if (i < p.n) {
            const uint32_t sl = p.slot_or_rank[i];
            if (p.cand_bit == 0 || (sl & p.cand_bit) != 0)
                first = (uint32_t) p.table[(size_t) (sl & ~p.cand_bit) * p.table_stride] == grp_place(p, i);
        }
        const uint64_t bits = __ballot(first);
        if (lane == 0) p.first_bits[(t0 + (uint64_t) k * GRP_THREADS) / 64 + w] = bits;
        c += (uint32_t) __popcll(bits);
    }
    if (lane == 0) wsum[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int ww = 0; ww < GRP_THREADS / 64; ww++) tot += wsum[ww];
        p.tile_count[blockIdx.x] = tot;
    }
}

// One workgroup: exclusive sums of the tile counts in place, total at [ntiles].
__global__ __launch_bounds__(1024) void k_grp_scan(GrpParams p)
{
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t t0 = 0; t0 < p.ntiles; t0 += 1024) {
        const uint32_t t = t0 + threadIdx.x;
        const uint32_t v = t < p.ntiles ? p.tile_count[t] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v, lane);
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (int ww = 0; ww < w; ww++) before += wsum[ww];
        if (t < p.ntiles) p.tile_count[t] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        p.tile_count[p.ntiles] = carry;
        p.result[0] = carry;
        if ((uint64_t) carry > p.group_cap) raise_error(p.ctrl, MGS_ERR_ARG);
    }
}

// Rank of every first occurrence, left in rank_table[slot]: the tile's base plus
// the first occurrences before it in the tile, counted on the tile's bits.
__global__ __launch_bounds__(GRP_THREADS) void k_grp_rank(GrpParams p)
{
    constexpr int WORDS = GRP_TILE / 64;
    static_assert(WORDS <= 64, "one wave scans the tile's words");
    __shared__ uint64_t word[WORDS];
    __shared__ uint32_t before[WORDS];
    const uint64_t t0 = (uint64_t) blockIdx.x * GRP_TILE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (w == 0) {
        const uint64_t bits = lane < WORDS ? p.first_bits[t0 / 64 + lane] : 0ull;
        const uint32_t c = (uint32_t) __popcll(bits);
        const uint32_t incl = wave_incl_scan_u32(c, lane);
        if (lane < WORDS) {
            word[lane] = bits;
            before[lane] = incl - c;
        }
    }
    __syncthreads();
    const uint32_t base = p.tile_count[blockIdx.x];
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        const int wi = k * (GRP_THREADS / 64) + w;
        const uint64_t bits = word[wi];
        if ((bits >> lane) & 1ull) {
            const uint64_t i = t0 + (uint64_t) wi * 64 + lane;
            p.rank_table[p.slot_or_rank[i] & ~p.cand_bit] = base + before[wi] + (uint32_t) __popcll(bits & ((1ull << lane) - 1ull));
        }
    }
}

// Every record's sort record and the digit histograms of all passes (the number of groups
// is known on the device by now: k_grp_scan).
__global__ __launch_bounds__(GRP_THREADS) void k_grp_keys(GrpParams p)
{
    __shared__ uint32_t hist[GRP_SORT_MAX_PASSES * GRP_SORT_BINS];
    if (p.ctrl->head.error != 0) return; // a malformed record: its value must not be looked for
    for (int j = threadIdx.x; j < GRP_SORT_MAX_PASSES * GRP_SORT_BINS; j += GRP_THREADS) hist[j] = 0;
    __syncthreads();
    const GrpSortPlan plan = grp_sort_plan(p.result[0]);
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += stride) {
        const uint32_t rank = p.rank_table[p.slot_or_rank[i] & ~p.cand_bit];
        unsigned long long v = 0;
        if (p.vbytes) {
            const uint8_t *src = p.blob + grp_offset(p, i + 1) - p.vbytes; // the value ends the record
            if (p.vbytes >= 8) v = reinterpret_cast<const GrpU64 *>(src)->v;
            else
                for (uint32_t b = 0; b < p.vbytes; b++) v |= (unsigned long long) src[b] << (8 * b);
        }
        p.rec_a[i] = make_uint4(rank, (uint32_t) i, (uint32_t) v, (uint32_t) (v >> 32));
        // one LDS add per distinct digit value of the wave (skewed ranks put most lanes on one value)
        const uint64_t active = __ballot(true);
        for (int s = 0; s < plan.npass; s++) {
            const uint32_t d = (rank >> plan.shift[s]) & ((1u << plan.width[s]) - 1u);
            uint64_t m = active;
#pragma unroll
            for (uint32_t bit = 0; bit < 6; bit++) {
                const bool one = (d >> bit) & 1u;
                const uint64_t bk = __ballot(one);
                m &= one ? bk : ~bk;
            }
            if (lanes_below(m) == 0) atomicAdd(&hist[s * GRP_SORT_BINS + d], (uint32_t) __popcll(m));
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < GRP_SORT_MAX_PASSES * GRP_SORT_BINS; j += GRP_THREADS)
        if (hist[j]) atomicAdd(&p.hist[j], hist[j]);
}

// One wave: the partitions of every pass, end to end in the order of the digit's values.
__global__ __launch_bounds__(64) void k_grp_sort_setup(GrpParams p)
{
    const GrpSortPlan plan = grp_sort_plan(p.result[0]);
    const int lane = threadIdx.x;
    for (int s = 0; s < plan.npass; s++) {
        const uint32_t bins = 1u << plan.width[s];
        const uint32_t c = (uint32_t) lane < bins ? p.hist[s * GRP_SORT_BINS + lane] : 0u;
        const uint32_t incl = wave_incl_scan_u32(c, lane);
        uint4 *out = (s & 1) ? p.rec_a : p.rec_b;
        CtrlHead &H = p.sort_ctrl[s].head;
        if ((uint32_t) lane < bins) {
            H.part_base[lane] = (uint64_t) (uintptr_t) (out + (incl - c));
            H.part_cap[lane] = (int64_t) c * 16;
            H.part_init[lane] = 0;
        }
    }
}

// One thread per sorted position: the record's value, and the group's entry
// where a group begins.
__global__ __launch_bounds__(GRP_THREADS) void k_grp_emit(GrpParams p)
{
    if (p.ctrl->head.error != 0) return;
    for (int s = 0; s < GRP_SORT_MAX_PASSES; s++) { // a sort pass that gave up: nothing is sorted
        const int e = p.sort_ctrl[s].head.error;
        if (e != 0) {
            if (blockIdx.x == 0 && threadIdx.x == 0) raise_error(p.ctrl, e);
            return;
        }
    }
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    for (uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; j < p.n; j += stride) {
        const uint4 r = p.sorted[j];
        const uint32_t g = r.x, i = r.y;
        if (j == 0 || p.sorted[j - 1].x != g) {
            p.group_first[g] = i;
            p.group_start[g] = (uint32_t) j;
        }
        if (p.vbytes) {
            uint8_t *dst = p.values + j * p.vbytes;
            const unsigned long long v = ((unsigned long long) r.w << 32) | r.z;
            if (p.vbytes == 8) {
                reinterpret_cast<GrpU64 *>(dst)->v = v;
            }
            else if (p.vbytes < 8) {
                for (uint32_t b = 0; b < p.vbytes; b++) dst[b] = (uint8_t) (v >> (8 * b));
            }
            else { // wider values are fetched from the record
                const uint8_t *src = p.blob + grp_offset(p, (uint64_t) i + 1) - p.vbytes;
                reinterpret_cast<GrpU64 *>(dst)->v = v;
                uint32_t b = 8;
                for (; b + 8 <= p.vbytes; b += 8)
                    reinterpret_cast<GrpU64 *>(dst + b)->v = reinterpret_cast<const GrpU64 *>(src + b)->v;
                for (; b < p.vbytes; b++) dst[b] = src[b];
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) p.group_start[p.tile_count[p.ntiles]] = p.n;
}

// src/kmvcontainer.h:98-119: groups are added to a value buffer until its bytes
// reach page_bytes; the next group opens a new buffer.  One wave; the last
// group of a buffer is found by a search that probes 64 places at a time.
__global__ __launch_bounds__(64) void k_grp_pages(GrpParams p)
{
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    if (p.ctrl->head.error != 0 || p.vbytes == 0) { // groups without value bytes never fill a buffer
        if (lane == 0) p.result[1] = 0;
        return;
    }
    const uint64_t G = p.result[0];
    uint64_t g0 = 0, ncuts = 0;
    while (g0 < G) {
        // smallest e in (g0, G] with (group_start[e] - group_start[g0]) * vbytes >= page_bytes
        const uint64_t base = p.group_start[g0];
        if (((uint64_t) p.group_start[G] - base) * p.vbytes < (uint64_t) p.page_bytes) break;
        uint64_t lo = g0 + 1, hi = G; // the answer is in [lo, hi]
        while (lo < hi) {
            const uint64_t step = (hi - lo + 63) / 64;
            const uint64_t probe = min(hi, lo + (uint64_t) lane * step);
            const bool full = ((uint64_t) p.group_start[probe] - base) * p.vbytes >= (uint64_t) p.page_bytes;
            const uint64_t k = (uint64_t) __popcll(~__ballot(full)); // the sizes ascend: the first k probes are not full
            // probe k is the first full one (k >= 1 unless lo itself is full)
            const uint64_t nhi = min(hi, lo + k * step);
            lo = k ? min(hi, lo + (k - 1) * step + 1) : lo;
            hi = nhi;
        }
        if (lo >= G) break; // the buffer ends with the last group
        if (ncuts < p.max_cuts && lane == 0) p.cuts[ncuts] = lo; // group `lo` opens the next buffer
        ncuts++;
        g0 = lo;
    }
    if (lane == 0) p.result[1] = ncuts;
}

} // namespace mgs
#endif
