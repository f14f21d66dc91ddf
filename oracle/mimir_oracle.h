/*
 * CPU oracle for Mimir's map-side shuffle transform.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mimir_amd/,
 * include/) may link, import or call this file; it is the checker the
 * tests, the smoke test and the bench's cpu_baseline leg compare the HIP
 * path against.
 *
 * Each function restates, in plain C, the behaviour of the reference code
 * cited next to it (paths relative to the reference checkout).  Records are
 * handed over in "wire form": the bytes `Serializer::kv_to_bytes` would
 * produce for one <key,value> pair, i.e. key bytes immediately followed by
 * value bytes, no header, no padding (src/serializer.h:348-364).
 */
#ifndef MIMIR_ORACLE_H
#define MIMIR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MO_KEY_FIXED = 0, MO_KEY_CSTR = 1 };
enum { MO_VAL_VOID = 0, MO_VAL_FIXED = 1, MO_VAL_CSTR = 2 };

/* How a wire-form record splits into key and value.
 *   key_kind FIXED : key is key_bytes = sizeof(K)*keycount raw bytes
 *                    (bytestream<T>, src/serializer.h:40-79)
 *   key_kind CSTR  : key is key_count NUL-terminated strings, NULs included
 *                    (bytestream<char*>, src/serializer.h:141-201)
 *   val_kind VOID  : no value bytes (bytestream<void>, :203-223)
 *   val_kind FIXED : val_bytes = sizeof(V)*valcount raw bytes
 *   val_kind CSTR  : val_count NUL-terminated strings
 */
typedef struct mo_layout {
    int32_t key_kind;
    int32_t key_bytes;
    int32_t key_count;
    int32_t val_kind;
    int32_t val_bytes;
    int32_t val_count;
} mo_layout;

/* lookup3 hashlittle, byte-at-a-time form (src/hash.cpp:152-163, :344-400). */
uint32_t mo_hashlittle(const void *key, size_t length, uint32_t initval);

/* Send-buffer geometry of CollectiveShuffler's constructor
 * (src/collectiveshuffler.h:33-53).  Returns 0, or -1 where the reference
 * aborts with "communication buffer is too small" (unit_size < 64). */
int mo_geometry(int64_t comm_buf_size, int nranks, int64_t *buf_size,
                int *unit_size, int *type_log_bytes);

/* Number of key bytes that get hashed for one wire-form record
 * (Serializer::get_key_bytes, src/serializer.h:418-421).  -1 if the record is
 * malformed for the layout. */
int64_t mo_key_len(const mo_layout *lay, const uint8_t *rec, int64_t rec_len);

/* hash and destination of every record:
 *   hash = hashlittle(keybytes, keysize, 0)   (src/serializer.h:464-480)
 *   dest = hash % (uint32_t) nranks           (src/baseshuffler.h:275-277)
 * rec_off: n+1 byte offsets into blob, or NULL for fixed-width records of
 * fixed_len bytes each.  hash_out / dest_out may be NULL.  Returns 0 or -1. */
int mo_dest(const mo_layout *lay, const uint8_t *blob, const uint64_t *rec_off,
            int64_t fixed_len, uint64_t n, uint32_t nranks, uint32_t *hash_out,
            int32_t *dest_out);

/* One round of CollectiveShuffler::write (src/collectiveshuffler.h:131-166)
 * over records [start, n): every record whose destination is `self` is
 * appended, in emit order, to local_out (the `out->write` bypass, :135-141);
 * every other record is appended to partition `dest` of send_image
 * (send_image + dest*buf_size + send_offset[dest], :143-146).  The round
 * ends before the first record that does not fit its partition
 * (kv_to_bytes returning -1, :148) or at n.  self = -1 disables the bypass.
 *
 * send_image : nranks*buf_size bytes (may be NULL: sizes only)
 * send_offset: nranks entries, bytes used per partition on return
 * local_out  : capacity local_cap bytes (may be NULL: sizes only)
 * Returns the index of the first record NOT consumed (== n when all fit),
 * or UINT64_MAX on a malformed record / a record larger than buf_size /
 * local_out overflow. */
uint64_t mo_pack_round(const mo_layout *lay, const uint8_t *blob,
                       const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                       uint64_t start, uint32_t nranks, int32_t self,
                       int64_t buf_size, uint8_t *send_image,
                       int64_t *send_offset, uint8_t *local_out,
                       int64_t local_cap, int64_t *local_bytes,
                       uint64_t *local_records, uint64_t *packed_records);

/* Same loop, continuing a round that is already partly filled: send_offset
 * holds the bytes in each partition on entry (write() called after earlier
 * write() calls of the same round) and is advanced. */
uint64_t mo_pack_more(const mo_layout *lay, const uint8_t *blob,
                      const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                      uint64_t start, uint32_t nranks, int32_t self,
                      int64_t buf_size, uint8_t *send_image,
                      int64_t *send_offset, uint8_t *local_out,
                      int64_t local_cap, int64_t *local_bytes,
                      uint64_t *local_records, uint64_t *packed_records);

/* Receive image of one exchange (src/collectiveshuffler.h:630-645): for
 * src = 0..nranks-1 the bytes partition `me` of that source's send image,
 * concatenated without gaps (type_log_bytes == 0).
 * send_images[src] points at that rank's nranks*buf_size image and
 * send_offsets[src] at its nranks offsets.  Returns bytes written. */
int64_t mo_recv_image(uint32_t nranks, int32_t me, int64_t buf_size,
                      const uint8_t *const *send_images,
                      const int64_t *const *send_offsets, uint8_t *recv_image,
                      int64_t *recv_count);

/* Walk a receive image the way save_data does (src/collectiveshuffler.h:
 * 563-585): split `nbytes` of back-to-back wire records into record offsets.
 * rec_off_out needs room for max_records+1 entries.  Returns the record
 * count, or -1 when a record is malformed / runs past nbytes. */
int64_t mo_parse_records(const mo_layout *lay, const uint8_t *buf,
                         int64_t nbytes, uint64_t *rec_off_out,
                         int64_t max_records);

/* KVContainer::write (src/kvcontainer.h:133-160) over a parsed stream: every
 * record is appended to the current page; when it does not fit in what is left
 * of the page (an exact fit does fit, src/serializer.h:348-364) a new page is
 * opened and the record goes there.  rec_off[0..n] are the record offsets,
 * first_fill the bytes already in the current page.  Writes the index of the
 * first record of every new page to cut_record_out (room for max_cuts) and
 * returns their number, or -1 when a record is larger than a page (:147-149)
 * or max_cuts is too small. */
int64_t mo_page_cuts(const uint64_t *rec_off, int64_t n, int64_t page_bytes,
                     int64_t first_fill, uint64_t *cut_record_out,
                     int64_t max_cuts);

/* Map side of wordcount: FileReader::read hands out the text line by line
 * (FileParser::to_line cuts at the first '\n', or at the end of the last
 * block: src/fileparser.h:21-41; the reader then advances by strlen + 1:
 * src/filereader.h:152-171, so a NUL ends a line as well) and the map callback
 * splits each line at spaces the way strtok_r does and writes <word, value>
 * for words shorter than 1024 bytes (examples/wordcount.cpp:66-82) in wire form
 * (word, NUL, 8 value bytes: src/serializer.h:63-83, :157-168).  The text is
 * one last block: whole lines, the last one need not end with '\n'.
 * Writes the records to out (capacity out_cap) and rec_off_out[0..n] (room for
 * max_records + 1); returns n, or -1 when something does not fit. */
int64_t mo_tokenize(const uint8_t *text, int64_t nbytes, uint64_t value,
                    uint8_t *out, int64_t out_cap, uint64_t *rec_off_out,
                    int64_t max_records);

/* KV -> KMV: KMVContainer::convert, src/kmvcontainer.h:74-150, over the
 * records blob/rec_off (or fixed_len) as the KV container would hand them out.
 * Pass 1 (:83-96): every key is looked up in a chained hash table, bucket
 * (hashlittle(key) / hashscale) % nbucket (src/hashbucket.h:111); a new key is
 * appended to the entry list (:263-287), which HashBucket::next walks in that
 * order (:301-342).  Then the unique keys get consecutive slices of value
 * buffers in that order; a buffer is closed with the entry that takes it to
 * page_bytes or beyond (:98-119).  Pass 2 (:132-144) appends every record's
 * value to its key's slice.  Values must be fixed-width or absent.
 * Out: group_first[g] = index of the first record with the g-th unique key,
 * group_start[0..G] = records in groups before g, values_out = the values
 * grouped, cut_group_out = every group that opens a new value buffer.
 * Returns G, or -1 (bad record, string values, too many cuts). */
int64_t mo_kmv_convert(const mo_layout *lay, const uint8_t *blob,
                       const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                       uint32_t hashscale, int64_t page_bytes,
                       uint32_t *group_first, uint64_t *group_start,
                       uint8_t *values_out, uint64_t *cut_group_out,
                       int64_t max_cuts, int64_t *n_cuts);

/* One round of CombineCollectiveShuffler::write
 * (src/combinecollectiveshuffler.h:63-180) over records [start, n), for an
 * 8-byte value and a combiner that is one of the declared operations below
 * (the reference calls a user callback, :139; wordcount's is a sum,
 * examples/wordcount.cpp:96-100).  A record whose destination is `self` goes
 * to local_out as it is (:67-83).  Any other record is looked up in the
 * round's hash bucket (:91-92): a new key is appended to its partition and
 * remembered (:94-135), a known key has its value combined in place (:137-151;
 * a fixed-width value never changes size, so no slices arise).  The round
 * ends before the first NEW key that does not fit its partition (:122-127);
 * garbage_collection() then clears the bucket (:237), so the next round
 * starts with no keys known.  Arguments and return as mo_pack_round;
 * packed_records counts the new keys (kvcount, :134). */
enum { MO_COMBINE_SUM_U64 = 0, MO_COMBINE_MIN_U64 = 1, MO_COMBINE_MAX_U64 = 2 };
uint64_t mo_combine_round(const mo_layout *lay, const uint8_t *blob,
                          const uint64_t *rec_off, int64_t fixed_len, uint64_t n,
                          uint64_t start, uint32_t nranks, int32_t self,
                          int64_t buf_size, int op, uint8_t *send_image,
                          int64_t *send_offset, uint8_t *local_out,
                          int64_t local_cap, int64_t *local_bytes,
                          uint64_t *local_records, uint64_t *packed_records);

#ifdef __cplusplus
}
#endif
#endif
