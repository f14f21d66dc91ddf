/*
 * CPU oracle for Mimir's map-side shuffle transform -- TEST INFRASTRUCTURE
 * ONLY (see mimir_oracle.h).  Plain C restatement of the reference
 * algorithm; every function cites the reference lines it follows.
 */
#include "mimir_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---- lookup3 hashlittle -------------------------------------------------
 * src/hash.cpp:14 (rot), :60-80 (mix), :107-123 (final), :152-163 (state
 * set-up), :344-400 (byte-at-a-time body, valid for every alignment; the
 * aligned variants at :166-343 give the same value on little-endian). */
static inline uint32_t rotl32(uint32_t x, int k)
{
    return (x << k) | (x >> (32 - k));
}

static inline uint32_t le32(const uint8_t *p)
{
    return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16)
           | ((uint32_t) p[3] << 24);
}

uint32_t mo_hashlittle(const void *key, size_t length, uint32_t initval)
{
    const uint8_t *k = (const uint8_t *) key;
    uint32_t a, b, c;
    a = b = c = 0xdeadbeefu + (uint32_t) length + initval;

    while (length > 12) {
        a += le32(k);
        b += le32(k + 4);
        c += le32(k + 8);
        /* mix */
        a -= c; a ^= rotl32(c, 4);  c += b;
        b -= a; b ^= rotl32(a, 6);  a += c;
        c -= b; c ^= rotl32(b, 8);  b += a;
        a -= c; a ^= rotl32(c, 16); c += b;
        b -= a; b ^= rotl32(a, 19); a += c;
        c -= b; c ^= rotl32(b, 4);  b += a;
        length -= 12;
        k += 12;
    }

    if (length == 0) return c; /* :394-395, no final() for an empty tail */

    /* tail: bytes 0-3 -> a, 4-7 -> b, 8-11 -> c, little-endian, zero padded */
    uint32_t t[3] = {0, 0, 0};
    for (size_t i = 0; i < length; i++)
        t[i >> 2] |= (uint32_t) k[i] << (8 * (i & 3));
    a += t[0];
    b += t[1];
    c += t[2];

    /* final */
    c ^= b; c -= rotl32(b, 14);
    a ^= c; a -= rotl32(c, 11);
    b ^= a; b -= rotl32(a, 25);
    c ^= b; c -= rotl32(b, 16);
    a ^= c; a -= rotl32(c, 4);
    b ^= a; b -= rotl32(a, 14);
    c ^= b; c -= rotl32(b, 24);
    return c;
}

/* ---- geometry: src/collectiveshuffler.h:33-53 (MEMPAGE_SIZE = 4096 and
 * MAX_COMM_SIZE = 2^30 from src/globals.h:33,37) ------------------------- */
int mo_geometry(int64_t comm_buf_size, int nranks, int64_t *buf_size,
                int *unit_size, int *type_log_bytes)
{
    int unit = 4096;
    while ((int64_t) unit * (int64_t) nranks > comm_buf_size) unit /= 2;
    if (unit < 64) return -1;
    int64_t bs = (comm_buf_size / unit / nranks) * unit;
    int tlb = 0;
    int type_bytes = 1;
    while ((int64_t) type_bytes * ((int64_t) 1 << 30) < bs * nranks) {
        type_bytes <<= 1;
        tlb++;
    }
    if (buf_size) *buf_size = bs;
    if (unit_size) *unit_size = unit;
    if (type_log_bytes) *type_log_bytes = tlb;
    return 0;
}

/* bytes taken by `count` NUL-terminated strings starting at p, scanning at
 * most `avail` bytes; -1 if they do not terminate inside avail
 * (bytestream<char*>::size, src/serializer.h:193-198). */
static int64_t cstr_span(const uint8_t *p, int64_t avail, int count)
{
    int64_t pos = 0;
    for (int s = 0; s < count; s++) {
        const void *z = (pos < avail) ? memchr(p + pos, 0, (size_t)(avail - pos))
                                      : NULL;
        if (!z) return -1;
        pos = (const uint8_t *) z - p + 1;
    }
    return pos;
}

int64_t mo_key_len(const mo_layout *lay, const uint8_t *rec, int64_t rec_len)
{
    int64_t klen;
    if (lay->key_kind == MO_KEY_FIXED) {
        klen = lay->key_bytes;
        if (klen > rec_len) return -1;
    }
    else {
        klen = cstr_span(rec, rec_len, lay->key_count > 0 ? lay->key_count : 1);
        if (klen < 0) return -1;
    }
    int64_t vlen;
    switch (lay->val_kind) {
    case MO_VAL_VOID: vlen = 0; break;
    case MO_VAL_FIXED: vlen = lay->val_bytes; break;
    default:
        vlen = cstr_span(rec + klen, rec_len - klen,
                         lay->val_count > 0 ? lay->val_count : 1);
        if (vlen < 0) return -1;
    }
    if (klen + vlen != rec_len) return -1;
    return klen;
}

static inline void rec_span(const uint64_t *rec_off, int64_t fixed_len,
                            uint64_t i, uint64_t *off, int64_t *len)
{
    if (rec_off) {
        *off = rec_off[i];
        *len = (int64_t)(rec_off[i + 1] - rec_off[i]);
    }
    else {
        *off = i * (uint64_t) fixed_len;
        *len = fixed_len;
    }
}

int mo_dest(const mo_layout *lay, const uint8_t *blob, const uint64_t *rec_off,
            int64_t fixed_len, uint64_t n, uint32_t nranks, uint32_t *hash_out,
            int32_t *dest_out)
{
    for (uint64_t i = 0; i < n; i++) {
        uint64_t off;
        int64_t len;
        rec_span(rec_off, fixed_len, i, &off, &len);
        int64_t klen = mo_key_len(lay, blob + off, len);
        if (klen < 0) return -1;
        uint32_t h = mo_hashlittle(blob + off, (size_t) klen, 0);
        if (hash_out) hash_out[i] = h;
        if (dest_out) dest_out[i] = (int32_t)(h % nranks);
    }
    return 0;
}

uint64_t mo_pack_round(const mo_layout *lay, const uint8_t *blob,
                       const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                       uint64_t start, uint32_t nranks, int32_t self,
                       int64_t buf_size, uint8_t *send_image,
                       int64_t *send_offset, uint8_t *local_out,
                       int64_t local_cap, int64_t *local_bytes,
                       uint64_t *local_records, uint64_t *packed_records)
{
    for (uint32_t d = 0; d < nranks; d++) send_offset[d] = 0;
    return mo_pack_more(lay, blob, rec_off, fixed_len, n, start, nranks, self,
                        buf_size, send_image, send_offset, local_out,
                        local_cap, local_bytes, local_records, packed_records);
}

uint64_t mo_pack_more(const mo_layout *lay, const uint8_t *blob,
                      const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                      uint64_t start, uint32_t nranks, int32_t self,
                      int64_t buf_size, uint8_t *send_image,
                      int64_t *send_offset, uint8_t *local_out,
                      int64_t local_cap, int64_t *local_bytes,
                      uint64_t *local_records, uint64_t *packed_records)
{
    int64_t lbytes = 0;
    uint64_t lrec = 0, prec = 0;
    uint64_t i = start;
    (void) nranks;

    for (; i < n; i++) {
        uint64_t off;
        int64_t len;
        rec_span(rec_off, fixed_len, i, &off, &len);
        const uint8_t *rec = blob + off;
        int64_t klen = mo_key_len(lay, rec, len);
        if (klen < 0) return UINT64_MAX;
        /* get_target_rank, src/baseshuffler.h:275-277 */
        uint32_t dest = mo_hashlittle(rec, (size_t) klen, 0) % nranks;

        if ((int32_t) dest == self) { /* src/collectiveshuffler.h:135-141 */
            if (local_out) {
                if (lbytes + len > local_cap) return UINT64_MAX;
                memcpy(local_out + lbytes, rec, (size_t) len);
            }
            lbytes += len;
            lrec++;
            continue;
        }
        /* kv_to_bytes == -1  <=>  bufsize < needed (src/serializer.h:47,
         * :90, :150): the record ends the round and opens the next one
         * (src/collectiveshuffler.h:148-160). */
        if (buf_size - send_offset[dest] < len) {
            if (len > buf_size) return UINT64_MAX; /* :156-159 LOG_ERROR */
            break;
        }
        if (send_image)
            memcpy(send_image + (int64_t) dest * buf_size + send_offset[dest],
                   rec, (size_t) len);
        send_offset[dest] += len; /* :162 */
        prec++;                   /* :163 kvcount */
    }
    if (local_bytes) *local_bytes = lbytes;
    if (local_records) *local_records = lrec;
    if (packed_records) *packed_records = prec;
    return i;
}

int64_t mo_recv_image(uint32_t nranks, int32_t me, int64_t buf_size,
                      const uint8_t *const *send_images,
                      const int64_t *const *send_offsets, uint8_t *recv_image,
                      int64_t *recv_count)
{
    int64_t pos = 0;
    for (uint32_t src = 0; src < nranks; src++) {
        int64_t cnt = send_offsets[src][me];
        if (recv_count) recv_count[src] = cnt;
        if (recv_image && cnt > 0)
            memcpy(recv_image + pos, send_images[src] + (int64_t) me * buf_size,
                   (size_t) cnt);
        pos += cnt;
    }
    return pos;
}

int64_t mo_parse_records(const mo_layout *lay, const uint8_t *buf,
                         int64_t nbytes, uint64_t *rec_off_out,
                         int64_t max_records)
{
    int64_t pos = 0, nrec = 0;
    while (pos < nbytes) {
        int64_t klen, vlen;
        /* kv_from_bytes, src/serializer.h:376-403 */
        if (lay->key_kind == MO_KEY_FIXED)
            klen = lay->key_bytes;
        else
            klen = cstr_span(buf + pos, nbytes - pos,
                             lay->key_count > 0 ? lay->key_count : 1);
        if (klen < 0 || pos + klen > nbytes) return -1;
        switch (lay->val_kind) {
        case MO_VAL_VOID: vlen = 0; break;
        case MO_VAL_FIXED: vlen = lay->val_bytes; break;
        default:
            vlen = cstr_span(buf + pos + klen, nbytes - pos - klen,
                             lay->val_count > 0 ? lay->val_count : 1);
        }
        if (vlen < 0 || pos + klen + vlen > nbytes) return -1;
        if (klen + vlen == 0) return -1;
        if (nrec >= max_records) return -1;
        rec_off_out[nrec++] = (uint64_t) pos;
        pos += klen + vlen;
    }
    rec_off_out[nrec] = (uint64_t) pos;
    return nrec;
}

int64_t mo_page_cuts(const uint64_t *rec_off, int64_t n, int64_t page_bytes,
                     int64_t first_fill, uint64_t *cut_record_out,
                     int64_t max_cuts)
{
    int64_t fill = first_fill, ncuts = 0;
    for (int64_t i = 0; i < n; i++) {
        const int64_t len = (int64_t) (rec_off[i + 1] - rec_off[i]);
        /* kv_to_bytes(ptr, pagesize - datasize) == -1, src/kvcontainer.h:140-142 */
        if (len > page_bytes - fill) {
            if (ncuts >= max_cuts) return -1;
            cut_record_out[ncuts++] = (uint64_t) i; /* add_page(), :143 */
            fill = 0;
            if (len > page_bytes) return -1; /* :147-149 */
        }
        fill += len;
    }
    return ncuts;
}

int64_t mo_tokenize(const uint8_t *text, int64_t nbytes, uint64_t value,
                    uint8_t *out, int64_t out_cap, uint64_t *rec_off_out,
                    int64_t max_records)
{
    int64_t start = 0, win = nbytes, nrec = 0, opos = 0;
    while (win > 0) {
        /* to_line: the line ends at the first '\n' of the window; in the last
         * block a line without one ends with the window (src/fileparser.h:25-38) */
        int64_t line = 0;
        while (line < win && text[start + line] != '\n') line++;
        /* key_from_bytes of a char* key = strlen + 1 (src/serializer.h:193-205):
         * the callback's line ends at the first NUL, and so does the reader's step */
        int64_t len = 0;
        while (len < line && text[start + len] != 0) len++;
        /* strtok_r(line, " "): maximal runs of bytes other than ' ' */
        int64_t i = 0;
        while (i < len) {
            while (i < len && text[start + i] == ' ') i++;
            int64_t w0 = i;
            while (i < len && text[start + i] != ' ') i++;
            const int64_t wl = i - w0;
            if (wl > 0 && wl < 1024) { /* examples/wordcount.cpp:72 */
                if (nrec >= max_records || opos + wl + 9 > out_cap) return -1;
                rec_off_out[nrec++] = (uint64_t) opos;
                memcpy(out + opos, text + start + w0, (size_t) wl);
                out[opos + wl] = 0;
                memcpy(out + opos + wl + 1, &value, 8);
                opos += wl + 9;
            }
        }
        /* src/filereader.h:163-170: past the line's terminator, or the window is used up */
        const int64_t move = len + 1;
        if (move >= win) {
            win = 0;
        }
        else {
            start += move;
            win -= move;
        }
    }
    rec_off_out[nrec] = (uint64_t) opos;
    return nrec;
}

int64_t mo_kmv_convert(const mo_layout *lay, const uint8_t *blob,
                       const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                       uint32_t hashscale, int64_t page_bytes,
                       uint32_t *group_first, uint64_t *group_start,
                       uint8_t *values_out, uint64_t *cut_group_out,
                       int64_t max_cuts, int64_t *n_cuts)
{
    if (lay->val_kind == MO_VAL_CSTR) return -1;
    const int64_t vbytes = lay->val_kind == MO_VAL_FIXED ? lay->val_bytes : 0;
    const uint32_t nbucket = 1024 * 1024; /* BUCKET_COUNT, src/config.cpp:15 */
    if (hashscale == 0) hashscale = 1;
    /* entries in insertion order; chains through `next` with insertion at the head
     * (src/hashbucket.h:279-287) */
    typedef struct { uint64_t off; int64_t klen; int64_t valbytes; int64_t fill; int64_t next; } entry;
    int64_t *bucket = (int64_t *) malloc(sizeof(int64_t) * nbucket);
    entry *ent = (entry *) malloc(sizeof(entry) * (n ? n : 1));
    int64_t *group_of = (int64_t *) malloc(sizeof(int64_t) * (n ? n : 1));
    int64_t G = 0, rc = -1;
    if (!bucket || !ent || !group_of) goto done;
    for (uint32_t b = 0; b < nbucket; b++) bucket[b] = -1;
    for (uint64_t i = 0; i < n; i++) { /* pass 1, :83-96 */
        uint64_t off;
        int64_t len;
        rec_span(rec_off, fixed_len, i, &off, &len);
        const int64_t klen = mo_key_len(lay, blob + off, len);
        if (klen < 0 || klen + vbytes != len) goto done;
        const uint32_t ib = (mo_hashlittle(blob + off, (size_t) klen, 0) / hashscale) % nbucket;
        int64_t e = bucket[ib];
        while (e >= 0 && !(ent[e].klen == klen && memcmp(blob + ent[e].off, blob + off, (size_t) klen) == 0))
            e = ent[e].next;
        if (e < 0) {
            e = G++;
            ent[e].off = off;
            ent[e].klen = klen;
            ent[e].valbytes = vbytes;
            ent[e].next = bucket[ib];
            bucket[ib] = e;
            group_first[e] = (uint32_t) i;
        }
        else {
            ent[e].valbytes += vbytes;
        }
        group_of[i] = e;
    }
    { /* slices in entry order, :98-130 */
        int64_t totalsize = 0, pos = 0, nc = 0;
        for (int64_t e = 0; e < G; e++) {
            group_start[e] = (uint64_t) (vbytes ? pos / vbytes : 0);
            ent[e].fill = pos;
            pos += ent[e].valbytes;
            totalsize += ent[e].valbytes;
            if (totalsize >= page_bytes && e + 1 < G) { /* this buffer is closed; the next entry opens one */
                if (nc >= max_cuts) goto done;
                cut_group_out[nc++] = (uint64_t) (e + 1);
                totalsize = 0;
            }
        }
        *n_cuts = nc;
    }
    if (vbytes == 0) { /* no value bytes: count the records per group instead */
        uint64_t *cnt = (uint64_t *) calloc((size_t) (G ? G : 1), sizeof(uint64_t));
        if (!cnt) goto done;
        for (uint64_t i = 0; i < n; i++) cnt[group_of[i]]++;
        uint64_t acc = 0;
        for (int64_t e = 0; e < G; e++) {
            group_start[e] = acc;
            acc += cnt[e];
        }
        free(cnt);
    }
    group_start[G] = n;
    for (uint64_t i = 0; i < n && vbytes; i++) { /* pass 2, :132-144 */
        uint64_t off;
        int64_t len;
        rec_span(rec_off, fixed_len, i, &off, &len);
        entry *e = &ent[group_of[i]];
        memcpy(values_out + e->fill, blob + off + len - vbytes, (size_t) vbytes);
        e->fill += vbytes;
    }
    rc = G;
done:
    free(bucket);
    free(ent);
    free(group_of);
    return rc;
}

uint64_t mo_combine_round(const mo_layout *lay, const uint8_t *blob,
                          const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                          uint64_t start, uint32_t nranks, int32_t self,
                          int64_t buf_size, int op, uint8_t *send_image,
                          int64_t *send_offset, uint8_t *local_out,
                          int64_t local_cap, int64_t *local_bytes,
                          uint64_t *local_records, uint64_t *packed_records)
{
    if (lay->val_kind != MO_VAL_FIXED || lay->val_bytes != 8 || !send_image) return UINT64_MAX;
    const uint32_t nbucket = 1024 * 1024; /* BUCKET_COUNT; HashBucket<CombinerVal>() has scale 1 (:47) */
    typedef struct { uint8_t *kv; int64_t klen; int64_t next; } entry;
    int64_t *bucket = (int64_t *) malloc(sizeof(int64_t) * nbucket);
    entry *ent = (entry *) malloc(sizeof(entry) * (n - start ? n - start : 1));
    int64_t nent = 0;
    uint64_t i = start, rc = UINT64_MAX;
    if (!bucket || !ent) goto done;
    for (uint32_t b = 0; b < nbucket; b++) bucket[b] = -1;
    for (uint32_t d = 0; d < nranks; d++) send_offset[d] = 0;
    *local_bytes = 0;
    *local_records = 0;
    *packed_records = 0;
    for (; i < n; i++) {
        uint64_t off;
        int64_t len;
        rec_span(rec_off, fixed_len, i, &off, &len);
        const uint8_t *rec = blob + off;
        const int64_t klen = mo_key_len(lay, rec, len);
        if (klen < 0 || klen + 8 != len) goto done;
        const uint32_t h = mo_hashlittle(rec, (size_t) klen, 0);
        const int32_t target = (int32_t) (h % nranks);
        if (target == self) { /* :67-83 */
            if (local_out) {
                if (*local_bytes + len > local_cap) goto done;
                memcpy(local_out + *local_bytes, rec, (size_t) len);
            }
            *local_bytes += len;
            *local_records += 1;
            continue;
        }
        if (len > buf_size) goto done; /* :86-88 */
        const uint32_t ib = h % nbucket;
        int64_t e = bucket[ib];
        while (e >= 0 && !(ent[e].klen == klen && memcmp(ent[e].kv, rec, (size_t) klen) == 0)) e = ent[e].next;
        if (e < 0) { /* a new key, :94-135 */
            if (send_offset[target] + len > buf_size) break; /* the round ends here, :122-127 */
            uint8_t *kv = send_image + (int64_t) target * buf_size + send_offset[target];
            memcpy(kv, rec, (size_t) len);
            send_offset[target] += len;
            ent[nent].kv = kv;
            ent[nent].klen = klen;
            ent[nent].next = bucket[ib];
            bucket[ib] = nent++;
            *packed_records += 1;
        }
        else { /* combine in place, :137-151 */
            uint64_t a, b, r;
            memcpy(&a, ent[e].kv + klen, 8);
            memcpy(&b, rec + klen, 8);
            r = op == MO_COMBINE_SUM_U64 ? a + b : op == MO_COMBINE_MIN_U64 ? (a < b ? a : b) : (a > b ? a : b);
            memcpy(ent[e].kv + klen, &r, 8);
        }
    }
    rc = i;
done:
    free(bucket);
    free(ent);
    return rc;
}
