/*
 * mimir_gpu_shuffle.h -- C ABI of the MI355X (gfx950) map-side shuffle
 * transform: hash every <key,value> record with lookup3 hashlittle, take
 * `hash % nranks` as its destination, and pack the records, in stable emit
 * order, into per-destination partitions of a send image.
 *
 * The library stands in for the per-record work of Mimir's
 * CollectiveShuffler / NBCollectiveShuffler.  Every entry point names the
 * reference code (paths relative to a Mimir checkout) whose behaviour it
 * reproduces bit for bit.
 *
 * Conventions
 *   - plain C, no C++/torch types; pointers prefixed d_ are DEVICE pointers
 *     on the handle's HIP device, everything else is host memory;
 *   - every function returns 0 (MGS_OK) or a negative mgs_status, and leaves
 *     a human-readable message retrievable with mgs_last_error();
 *   - a handle is bound to one HIP device and must be driven by one host
 *     thread at a time (the reference is single-threaded per rank,
 *     examples/wordcount.cpp:33);
 *   - records travel in WIRE FORM: the bytes Serializer::kv_to_bytes writes
 *     for one pair -- key bytes immediately followed by value bytes, no
 *     header, no padding (src/serializer.h:348-364).  A char* key is its
 *     characters plus the terminating NUL (src/serializer.h:145-157).
 */
#ifndef MIMIR_GPU_SHUFFLE_H
#define MIMIR_GPU_SHUFFLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGS_MAX_RANKS 256 /* destinations per handle (dest ids are one byte) */

typedef enum mgs_status {
    MGS_OK = 0,
    MGS_ERR_ARG = -1,          /* bad argument / unsupported layout          */
    MGS_ERR_HIP = -2,          /* a HIP runtime call failed                  */
    MGS_ERR_RECORD_SIZE = -3,  /* a record is larger than buf_size
                                  (src/collectiveshuffler.h:156-159)          */
    MGS_ERR_LOCAL_OVERFLOW = -4, /* local bypass stream ran out of room      */
    MGS_ERR_MALFORMED = -5,    /* record does not match the layout           */
    MGS_ERR_TIMEOUT = -6,      /* in-kernel wait gave up (device not ours?)  */
    MGS_ERR_STATE = -7         /* call sequence error                        */
} mgs_status;

/* ---- record layout: which bytes of a wire-form record are the key --------
 * Mirrors the template dispatch of bytestream<T> (src/serializer.h:40-223):
 *   MGS_KEY_FIXED  K is a value type; key = key_bytes = sizeof(K)*keycount
 *   MGS_KEY_CSTR   K is char* / const char*; key = key_count C strings, NULs
 *                  included
 *   MGS_VAL_VOID / MGS_VAL_FIXED / MGS_VAL_CSTR likewise for V.            */
enum { MGS_KEY_FIXED = 0, MGS_KEY_CSTR = 1 };
enum { MGS_VAL_VOID = 0, MGS_VAL_FIXED = 1, MGS_VAL_CSTR = 2 };

typedef struct mgs_layout {
    int32_t key_kind;
    int32_t key_bytes; /* FIXED only */
    int32_t key_count; /* CSTR only, >= 1 */
    int32_t val_kind;
    int32_t val_bytes; /* FIXED only */
    int32_t val_count; /* CSTR only, >= 1 */
} mgs_layout;

/* ---- which kernels to run ------------------------------------------------ */
enum {
    MGS_ALGO_AUTO = 0,     /* fastest kernel that supports the layout        */
    MGS_ALGO_GENERIC = 1,  /* three-pass histogram / scan / scatter, any
                              layout, any nranks <= MGS_MAX_RANKS             */
    MGS_ALGO_ONESWEEP16 = 2, /* single-pass chained-scan kernel for 16-byte
                              fixed-width records, nranks <= 64               */
    MGS_ALGO_ONESWEEPV = 3  /* single-pass chained-scan kernel for
                              byte-granular records: variable width, or fixed
                              widths up to 79 bytes; nranks <= 64 (63 with a
                              local rank).  A variable-width batch whose
                              records are too long for the LDS staging of the
                              tile in use falls back to GENERIC inside the
                              call; the tile follows the record length seen in
                              the handle's earlier calls                      */
};

typedef struct mgs_desc {
    int32_t nranks;       /* P: number of destinations (shuffle_size)        */
    int32_t rank;         /* this rank; its records bypass the send image
                             (src/collectiveshuffler.h:135-141).  -1: no
                             bypass, all P partitions are packed             */
    mgs_layout layout;
    int64_t buf_size;     /* bytes per destination partition; send image is
                             nranks*buf_size (src/collectiveshuffler.h:46)   */
    int32_t device;       /* HIP device ordinal                              */
    int32_t flags;        /* 0, or MGS_FLAG_* below                          */
    uint64_t max_records; /* most records one mgs_pack_device call covers    */
} mgs_desc;

/* mgs_desc.flags: which of the two byte-granular one-sweep kernels
 * MGS_ALGO_ONESWEEPV runs.  Default (0): one-round calls take the kernel that
 * regroups a tile in LDS and stores whole aligned lines (faster for a single
 * round), calls that cut several rounds take the kernel that cuts them itself.
 * The flags pin one of them for every call (measurement, tests); the
 * LDS-image kernel packs one round per call. */
#define MGS_FLAG_OSV_ROUNDS_KERNEL 1
#define MGS_FLAG_OSV_IMAGE_KERNEL 2
/* The LDS-image kernel comes in two forms.  Records of ONE string key with a
 * fixed-width or absent value, and fixed widths up to 32 bytes, take the form
 * that loads every record straight into registers (k_onesweepr); every other
 * layout takes the form that stages a tile's bytes in LDS first (k_onesweepv).
 * This flag keeps the staged form for every call (measurement, tests). */
#define MGS_FLAG_OSV_STAGED_KERNEL 4

typedef struct mgs_handle mgs_handle;

/* Result of one pack call = one (possibly partial) shuffle round. */
typedef struct mgs_round {
    uint64_t next;           /* index of the first record NOT consumed: the
                                record that did not fit its partition
                                (kv_to_bytes == -1, src/collectiveshuffler.h:
                                148) and must open the next round; == n_end
                                when everything fitted                       */
    uint64_t packed_records; /* records written to the send image (the
                                reference's kvcount, :163)                   */
    uint64_t local_records;  /* records that took the local bypass           */
    int64_t local_bytes;     /* bytes appended to d_local_out                */
    int64_t send_offset[MGS_MAX_RANKS]; /* bytes used in each partition after
                                this call (the reference's send_offset[],
                                :162); entry `rank` stays 0                  */
} mgs_round;

/* Send-buffer geometry of the CollectiveShuffler / NBCollectiveShuffler
 * constructors (src/collectiveshuffler.h:33-53, src/nbcollectiveshuffler.h:
 * 48-67): unit_size = 4096 halved until unit_size*nranks <= comm_buf_size,
 * buf_size = (comm_buf_size/unit_size/nranks)*unit_size, type_log_bytes = the
 * log2 of the MPI count unit.  Returns MGS_ERR_ARG where the reference aborts
 * ("communication buffer is too small", unit_size < 64).  Host only. */
int mgs_geometry(int64_t comm_buf_size, int nranks, int64_t *buf_size,
                 int *unit_size, int *type_log_bytes);

/* lookup3 hashlittle on the host (src/hash.cpp:152-400) -- the same routine
 * the kernels run, exported so callers and tests can check it without a GPU. */
uint32_t mgs_hashlittle_host(const void *key, size_t length, uint32_t initval);

/* Destination rule of BaseShuffler::get_target_rank with load balancing off
 * (src/baseshuffler.h:275-277), evaluated with the kernels' division-free
 * modulo.  Host only. */
uint32_t mgs_dest_of_hash_host(uint32_t hash, uint32_t nranks);

/* Create / destroy the per-rank state.  Plays the role of
 * CollectiveShuffler's constructor + open() (src/collectiveshuffler.h:25-109):
 * validates the geometry and allocates the device scratch the kernels need
 * (never the send image -- the caller owns that). */
int mgs_create(const mgs_desc *desc, mgs_handle **out);
void mgs_destroy(mgs_handle *h);

/* Message for the last failure on this handle (h == NULL: last failure of
 * mgs_create on this thread).  Never NULL. */
const char *mgs_last_error(const mgs_handle *h);

/* The hot path: CollectiveShuffler::write (src/collectiveshuffler.h:131-166)
 * applied to records [start, n_end) of a device-resident stream, as one
 * kernel sequence on `stream`.
 *
 *   d_records   wire-form records, back to back
 *   d_rec_off   n_end+1 byte offsets of the records inside d_records, or NULL
 *               when the layout is fixed-width (record i at i*record_bytes)
 *   init_offset nranks host values: bytes already in each partition (NULL =
 *               all zero).  Lets a round be filled by several calls.
 *   d_send_image nranks*buf_size bytes; partition d starts at d*buf_size
 *               (src/collectiveshuffler.h:143-144)
 *   d_local_out records whose destination is desc.rank, appended in emit
 *               order (the out->write bypass, :135-141); capacity local_cap
 *               bytes.  May be NULL when desc.rank == -1.  Size it for the
 *               local records of the whole examined range [start, min(n_end,
 *               start + max_records)), not only of the part before the round
 *               cut: the kernels place a tile's records before they know
 *               where the round ends, and a local record that has no room
 *               is MGS_ERR_LOCAL_OVERFLOW even when it lies after the cut.
 *   d_dest_out  optional, one byte per record of [start, n_end): its
 *               destination rank (hashlittle(key) % nranks)
 *   stream      hipStream_t the kernels are enqueued on; the inputs must be
 *               complete in that stream's order.  NULL: a stream the handle
 *               owns, created as a blocking stream -- its work starts after
 *               everything enqueued on the legacy default stream before the
 *               call (a hipMemcpy of the records, a kernel launched on stream
 *               0) and later default-stream work waits for it; work on other
 *               non-blocking streams is NOT waited for.  The same holds for
 *               every entry point below that takes a stream.
 *
 * At most desc.max_records records are examined; the call stops early at the
 * first record that does not fit its partition.  Blocks until the result is
 * on the host.  Bytes of a partition beyond send_offset[] are unspecified
 * (as in the reference, src/serializer.h:348-364). */
int mgs_pack_device(mgs_handle *h, const void *d_records,
                    const uint32_t *d_rec_off, uint64_t n_end, uint64_t start,
                    const int64_t *init_offset, void *d_send_image,
                    void *d_local_out, int64_t local_cap, uint8_t *d_dest_out,
                    int algo, void *stream, mgs_round *out);

/* Same kernels, enqueue only: returns as soon as the work is on `stream`.
 * mgs_pack_result() then waits for that work and fetches the result.  One
 * pack call may be in flight per handle; until its result has been fetched
 * the handle's other device calls (unpack, tokenize, group, combine) return
 * MGS_ERR_STATE, and another mgs_pack_device[_async] first waits for it.
 * (Callers that used to rely on unpack/tokenize/group/combine waiting for a
 * pack in flight call mgs_pack_result first.)  A call that packs several
 * rounds answers only to its own result call: mgs_pack_result on it returns
 * MGS_ERR_STATE and leaves it in flight, so mgs_pack_rounds_result /
 * mgs_pack_streams_result still fetch it. */
int mgs_pack_device_async(mgs_handle *h, const void *d_records,
                          const uint32_t *d_rec_off, uint64_t n_end,
                          uint64_t start, const int64_t *init_offset,
                          void *d_send_image, void *d_local_out,
                          int64_t local_cap, uint8_t *d_dest_out, int algo,
                          void *stream);
int mgs_pack_result(mgs_handle *h, mgs_round *out);

/* Many rounds per call.  CollectiveShuffler::write ends a round -- exchange_kv,
 * every partition empty again -- whenever one record does not fit its partition
 * (src/collectiveshuffler.h:148-160), so a large batch is a sequence of rounds
 * whose boundaries depend on the data.  This call packs records [start, n_end)
 * into up to max_rounds send images in one go: image k, at d_send_images +
 * k * image_stride (image_stride >= nranks * buf_size), is byte for byte the
 * send buffer of the k-th exchange, and the per-round arrays say what
 * mgs_pack_device would have returned round after round:
 *
 *   round_next[k]            index of the first record after round k
 *   round_send_offset[k*nranks + b]  bytes of partition b in image k
 *   round_local_records[k] / round_local_bytes[k]
 *                            local-bypass records emitted during round k: the
 *                            local stream in d_local_out is continuous, round k's
 *                            records follow those of round k-1
 *   out->n_rounds            rounds begun (>= 1); rounds 0 .. n_rounds-2 ended
 *                            with a record that did not fit.  out->next == n_end:
 *                            the last round is still open (its send_offset may be
 *                            handed to the next call as init_offset); out->next <
 *                            n_end: max_rounds ran out, the last round ended at
 *                            record out->next, which opens the round after it
 *   init_offset              bytes already in the partitions of image 0
 *
 * For 16-byte records (MGS_ALGO_ONESWEEP16) the rounds are cut inside the
 * one-sweep kernel: one launch for the whole batch, each record still read once
 * and written once; the other layouts run one kernel sequence per round.  Any
 * array pointer may be NULL.  The _async / _result pair exists for the
 * in-kernel case only (MGS_ERR_ARG otherwise). */
typedef struct mgs_rounds {
    uint32_t n_rounds;
    uint32_t reserved;
    uint64_t next;           /* first record not consumed */
    uint64_t packed_records; /* over all rounds */
    uint64_t local_records;
    int64_t local_bytes;
} mgs_rounds;

int mgs_pack_rounds_device(mgs_handle *h, const void *d_records,
                           const uint32_t *d_rec_off, uint64_t n_end,
                           uint64_t start, const int64_t *init_offset,
                           void *d_send_images, int64_t image_stride,
                           uint32_t max_rounds, void *d_local_out,
                           int64_t local_cap, uint8_t *d_dest_out, int algo,
                           void *stream, mgs_rounds *out, uint64_t *round_next,
                           int64_t *round_send_offset,
                           uint64_t *round_local_records,
                           int64_t *round_local_bytes);
int mgs_pack_rounds_device_async(mgs_handle *h, const void *d_records,
                                 const uint32_t *d_rec_off, uint64_t n_end,
                                 uint64_t start, const int64_t *init_offset,
                                 void *d_send_images, int64_t image_stride,
                                 uint32_t max_rounds, void *d_local_out,
                                 int64_t local_cap, uint8_t *d_dest_out,
                                 int algo, void *stream);
int mgs_pack_rounds_result(mgs_handle *h, mgs_rounds *out, uint64_t *round_next,
                           int64_t *round_send_offset,
                           uint64_t *round_local_records,
                           int64_t *round_local_bytes);

/* Many rounds per call, streams layout.  The same rounds as mgs_pack_rounds_device
 * (src/collectiveshuffler.h:131-166: a round ends before the first record that
 * does not fit its partition of buf_size bytes), but the partitions of a round
 * are not copied to an image of their own: the call appends the records of
 * [start, n_end) to nranks destination streams -- stream b at d_streams +
 * b * stream_cap, stream_cap bytes long, stream_fill[b] bytes already there
 * (NULL = empty) -- and reports every round as a view into the streams:
 *
 *   round_base[k*nranks + b]         byte of stream b at which partition b of
 *                                    round k begins
 *   round_send_offset[k*nranks + b]  its length: the send_offset of the k-th
 *                                    exchange (src/collectiveshuffler.h:162)
 *
 * Bytes [round_base, round_base + round_send_offset) of stream b are byte for
 * byte what the reference's send buffer holds at b*buf_size in that exchange;
 * an all-to-all-v takes them with displacement b*stream_cap + round_base in
 * place of b*buf_size (src/collectiveshuffler.h:630-645).  The other per-round
 * arrays, out->n_rounds and out->next mean what they mean for
 * mgs_pack_rounds_device.  open_round_base[b] (NULL = stream_fill) is the byte
 * of stream b at which the round that is still open began: pass the last
 * round's round_base of the previous call to go on in the same streams.
 * out->next < n_end also when a stream is full: the last reported round is then
 * open and ends with record out->next - 1.
 *
 * The sweep over the records knows nothing of rounds, so no tile ever waits
 * for a round cut: one workgroup of the same launch trails the chained scan's
 * prefixes and finds the cuts (mimir_amd/csrc/mgs_round_finder.h).  Layouts:
 * what the one-sweep kernels take (at most 64 ranks, 63 with a local rank
 * unless records are 16 bytes; fixed widths up to 79 bytes or variable width;
 * 16-byte records need 16-byte aligned buffers, fills and stream_cap);
 * MGS_ERR_ARG otherwise.  Variable-width records too long for the kernel's LDS
 * staging: the blocking call packs the batch round by round with the generic
 * kernels and appends each round's partitions to the streams (it then begins a
 * round only while every stream has room for a whole partition); the _async /
 * _result pair returns MGS_ERR_RECORD_SIZE. */
int mgs_pack_streams_device(mgs_handle *h, const void *d_records,
                            const uint32_t *d_rec_off, uint64_t n_end,
                            uint64_t start, const int64_t *stream_fill,
                            const int64_t *open_round_base, void *d_streams,
                            int64_t stream_cap, uint32_t max_rounds,
                            void *d_local_out, int64_t local_cap,
                            uint8_t *d_dest_out, void *stream, mgs_rounds *out,
                            uint64_t *round_next, int64_t *round_send_offset,
                            int64_t *round_base, uint64_t *round_local_records,
                            int64_t *round_local_bytes);
int mgs_pack_streams_device_async(mgs_handle *h, const void *d_records,
                                  const uint32_t *d_rec_off, uint64_t n_end,
                                  uint64_t start, const int64_t *stream_fill,
                                  const int64_t *open_round_base,
                                  void *d_streams, int64_t stream_cap,
                                  uint32_t max_rounds, void *d_local_out,
                                  int64_t local_cap, uint8_t *d_dest_out,
                                  void *stream);
int mgs_pack_streams_result(mgs_handle *h, mgs_rounds *out, uint64_t *round_next,
                            int64_t *round_send_offset, int64_t *round_base,
                            uint64_t *round_local_records,
                            int64_t *round_local_bytes);

/* mgs_pack_device for callers that cannot allocate: tight serving loops,
 * harnesses that forbid hipMalloc, code that must not grow the process's device
 * memory.  (It is a blocking call -- it ends by waiting for the stream and
 * reading the result back -- so it is not for use under HIP graph capture.)  It
 * has no counterpart in the reference.  Same arguments, same kernels and same result
 * as mgs_pack_device with MGS_ALGO_AUTO, but the call allocates nothing: the
 * device-side control block and the chained-scan arrays of the one-sweep
 * kernels live in a static device array of the library (MGS_NOALLOC_SCAN_BYTES
 * for the scan arrays; every GPU the library is used on holds its copy until the
 * process ends), and the host side uses what mgs_create made for the handle.
 *
 * Limits (MGS_ERR_ARG with a message otherwise): layouts and rank counts the
 * one-sweep kernels take (at most 64 ranks, 63 with a local rank for other than
 * 16-byte records; fixed widths up to 75 bytes or variable width), and
 * ceil(n / tile) * (8*columns + 16*ceil(columns/4)) <= MGS_NOALLOC_SCAN_BYTES
 * with columns = nranks (+1 with a local rank, except 16-byte records) and
 * tile = 3584 records of 16 bytes, 1792 / 896 / 448 otherwise (the largest
 * whose bytes fit 35 KB): e.g. 1.5e8 <u64,u64> records or 6e7 wordcount
 * records per call at 8 ranks.  Variable-width records too long for the
 * kernel's LDS staging give MGS_ERR_RECORD_SIZE (mgs_pack_device falls back to
 * its generic kernels there).
 *
 * All calls on one GPU share that GPU's array, whatever the handle: the array
 * keeps the previous call's contents (each call clears what it needs zeroed),
 * and two calls that overlap in time on one GPU race -- serialise them. */
#define MGS_NOALLOC_SCAN_BYTES (4u << 20)
int mgs_pack_device_noalloc(mgs_handle *h, const void *d_records,
                            const uint32_t *d_rec_off, uint64_t n_end,
                            uint64_t start, const int64_t *init_offset,
                            void *d_send_image, void *d_local_out,
                            int64_t local_cap, uint8_t *d_dest_out,
                            void *stream, mgs_round *out);

/* Name of the kernel path the last pack call on this handle ran
 * ("generic" / "onesweep16" / "onesweepv" / "onesweepd"). */
const char *mgs_last_algo(const mgs_handle *h);

/* Name of the dominant kernel of that path ("k_gen_scatter", "k_onesweep16",
 * "k_onesweepv", "k_onesweepr", "k_onesweepd"): what mgs_last_kernel_ms times
 * and what a profiler's kernel trace shows. */
const char *mgs_last_kernel(const mgs_handle *h);

/* The descriptor the handle was created with. */
int mgs_get_desc(const mgs_handle *h, mgs_desc *out);

/* ---- receive side: record boundaries and container page cuts ---------------
 * What follows the exchange in the reference: save_data parses the receive
 * buffer record by record (kv_from_bytes, src/collectiveshuffler.h:563-585,
 * src/serializer.h:376-403) and KVContainer::write appends each record to the
 * current page, opening a new page when a record does not fit
 * (src/kvcontainer.h:133-160).  Pages hold whole records back to back, so a
 * page is a contiguous slice of the stream; this call finds, on the device,
 * where the records start and at which records new pages begin, so that the
 * payload never has to visit the host:
 *
 *   d_stream / nbytes   wire-form records back to back (a receive image, or
 *                       several of them concatenated); nbytes < 4 GiB
 *   d_rec_off           out, device: d_rec_off[i] = first byte of record i,
 *                       d_rec_off[n] = nbytes (the same form mgs_pack_device
 *                       takes as input); rec_off_capacity entries available
 *   page_bytes          the container's page size (64 MiB in the reference)
 *   first_page_fill     bytes already in the container's current page
 *   cuts / n_cuts       out, host: every record at which a NEW page begins;
 *                       records before cuts[0] continue the current page
 *
 * Layouts: fixed-width records; one NUL-terminated key followed by a
 * fixed-width value of at most 62 bytes, or by no value.  Other layouts
 * return MGS_ERR_ARG (parse those on the host).  MGS_ERR_MALFORMED: the stream
 * does not end at a record boundary.  MGS_ERR_RECORD_SIZE: a record is larger
 * than a page.  Blocking. */
typedef struct mgs_page_cut {
    uint64_t first_record; /* index of the first record of the new page */
    uint64_t first_byte;   /* its offset in the stream */
} mgs_page_cut;

int mgs_unpack_device(mgs_handle *h, const void *d_stream, uint64_t nbytes,
                      uint32_t *d_rec_off, uint64_t rec_off_capacity,
                      int64_t page_bytes, int64_t first_page_fill,
                      uint64_t *n_records, mgs_page_cut *cuts, int max_cuts,
                      int *n_cuts, void *stream);

/* ---- map side of wordcount: text -> <word, count> records ------------------
 * What precedes the path in the reference for text inputs: FileReader::read
 * frames the input into lines (FileParser::to_line cuts at '\n',
 * src/fileparser.h:21-41; the reader advances by strlen + 1,
 * src/filereader.h:138-183, so a NUL ends a line too) and the map callback
 * splits each line at spaces with strtok_r and writes <word, 1> for every word
 * shorter than 1024 bytes (examples/wordcount.cpp:66-82).  This call does the
 * same on a device-resident text and leaves the records in wire form (word,
 * NUL, 8 value bytes: src/serializer.h:63-83, :157-168), in text order, with
 * their offsets -- the input form of mgs_pack_device:
 *
 *   d_text / nbytes     whole lines (the last one need not end with '\n');
 *                       nbytes < 4 GiB
 *   value               the value written with every word (wordcount: 1)
 *   d_records           out, device: the records back to back;
 *                       records_capacity bytes available (text bytes + 9 per
 *                       word at most: 5 * nbytes + 9 always suffices)
 *   d_rec_off           out, device: d_rec_off[i] = first byte of record i,
 *                       d_rec_off[n] = total bytes; rec_off_capacity entries
 *                       (nbytes / 2 + 2 always suffice)
 *
 * The handle's layout must be one NUL-terminated key and an 8-byte value.
 * MGS_ERR_ARG when an output does not fit or would pass 4 GiB.  Blocking. */
int mgs_tokenize_device(mgs_handle *h, const void *d_text, uint64_t nbytes,
                        uint64_t value, void *d_records, uint64_t records_capacity,
                        uint32_t *d_rec_off, uint64_t rec_off_capacity,
                        uint64_t *n_records, uint64_t *n_bytes, void *stream);

/* ---- reduce side: KV -> KMV (group by key) ----------------------------------
 * What the reduce phase starts with in the reference: KMVContainer::convert
 * (src/kmvcontainer.h:74-150) looks every record's key up in a hash table whose
 * entries keep the order of first insertion (src/hashbucket.h:185-290,
 * :301-342), gives the unique keys consecutive slices of value buffers in that
 * order and appends every record's value to its key's slice in stream order.
 * The reduce callback then sees each unique key once, in the order of first
 * occurrence, with its values in stream order (src/mimircontext.h:397-417).
 * This call computes that grouping for a device-resident KV stream:
 *
 *   d_records / d_rec_off / n_records   as for mgs_pack_device (d_rec_off NULL
 *                       for fixed-width records); n_records <= 2^30
 *   d_group_first       out, device [group_capacity]: the index of the first
 *                       record with the g-th unique key (the key is that
 *                       record's key)
 *   d_group_start       out, device [group_capacity + 1]: the number of records
 *                       in groups before g; d_group_start[n_groups] = n_records
 *   d_values            out, device [n_records * value bytes]: the values,
 *                       grouped: group g's are entries d_group_start[g] ..
 *                       d_group_start[g+1]-1, in stream order
 *   page_bytes / cuts   the groups at which a new value buffer begins: a
 *                       buffer is closed with the group that takes it to
 *                       page_bytes or beyond (DATA_PAGE_SIZE, 64 MiB, in the
 *                       reference, :105)
 *
 * Layouts: any key; the value fixed-width or absent (with string values the
 * reference sizes the slices with a stale length, :84-94, so there is no
 * defined result to match).  MGS_ERR_ARG when there are more groups than
 * group_capacity or more cuts than max_cuts.  Blocking.  The stable sort of
 * the records by group inside is a radix sort whose passes are launches of the
 * library's own 16-byte partition kernel (no external sort library). */
int mgs_group_device(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off,
                     uint64_t n_records, uint32_t *d_group_first, uint32_t *d_group_start,
                     uint64_t group_capacity, void *d_values, int64_t page_bytes,
                     uint64_t *n_groups, uint64_t *cuts, int max_cuts, int *n_cuts,
                     void *stream);

/* ---- send side with a combiner ----------------------------------------------
 * One round of CombineCollectiveShuffler::write
 * (src/combinecollectiveshuffler.h:63-180) for an 8-byte value and a combiner
 * that is one of the operations below (the reference calls a user callback,
 * :139; wordcount's, examples/wordcount.cpp:96-100, is MGS_COMBINE_SUM_U64;
 * a job declares the operation instead of passing a callback).  Records
 * [start, n_end): a record whose destination is this rank goes to d_local_out
 * unchanged (:67-83); of the others only the first with each key is written
 * to its partition, in emit order (:94-135), every later one is combined into
 * it (:137-151); the round ends before the first NEW key that does not fit its
 * partition (:122-127) -- out->next -- and the next call starts with no key
 * known, as after garbage_collection() (:237).  Partitions are written from
 * their beginning (one round per call); out->packed_records counts the new
 * keys (kvcount, :134).  Arguments as mgs_pack_device.  At most 64 ranks; the
 * records of one call must be under 4 GiB.  Blocking.  Feed about a round's
 * worth of records per call: keys that first occur after the round's end cost
 * table work for nothing. */
typedef enum mgs_combine_op {
    MGS_COMBINE_SUM_U64 = 0,
    MGS_COMBINE_MIN_U64 = 1,
    MGS_COMBINE_MAX_U64 = 2
} mgs_combine_op;

int mgs_combine_device(mgs_handle *h, const void *d_records, const uint32_t *d_rec_off,
                       uint64_t n_end, uint64_t start, int op, void *d_send_image,
                       void *d_local_out, int64_t local_cap, void *stream, mgs_round *out);

/* ---- host-staged front end: what a Writable::write() shim calls -----------
 * The reference serialises every pair into the send buffer inside write()
 * (src/collectiveshuffler.h:143-146) because the caller's pointers die when
 * the call returns.  A stager makes the same copy into a pinned host buffer,
 * in emit order and wire form, and owns the device send image of the current
 * round.
 *
 *   mgs_stager_put    append one pair (key bytes, value bytes).  0 = staged,
 *                     1 = no room: call mgs_stager_pack, then put again.
 *   mgs_stager_pack   move what is staged to the GPU and run the pack kernels
 *                     on it, continuing the current round.  *drained = 1: all
 *                     staged records were consumed.  *drained = 0: a partition
 *                     is full (kv_to_bytes == -1, :148): exchange the round,
 *                     call mgs_stager_round_done, call mgs_stager_pack again.
 *                     *local_records: out->local_bytes bytes of wire-form
 *                     records whose destination is this rank, emit order (the
 *                     out->write bypass, :135-141); valid until the next call.
 *   mgs_stager_send_image_host / _device
 *                     the round's send image -- partition d at d*buf_size,
 *                     send_offset[d] valid bytes (:143-144, :635) -- as a
 *                     pinned host copy (for MPI_Alltoallv on host memory) or
 *                     in place on the GPU (for RCCL / GPU-aware MPI).
 *   mgs_stager_round_done
 *                     the exchange is over: send_offset[*] = 0 (:657).
 *   mgs_stager_pending records staged but not packed yet.
 *   mgs_stager_set_combine
 *                     the combiner shuffler (src/combinecollectiveshuffler.h:
 *                     63-180) with a declared operation (mgs_combine_op; -1 =
 *                     off): mgs_stager_pack then runs mgs_combine_device over
 *                     the current round, whose records stay staged until it
 *                     ends (a partition cut, or an exchange asked for by the
 *                     caller); when the staging buffer fills first they are
 *                     replaced by the round's combined partitions.  The
 *                     staging buffer must hold two send images.  out->
 *                     local_bytes / packed_records of a pack are what that
 *                     call added. */
typedef struct mgs_stager mgs_stager;
int mgs_stager_create(mgs_handle *h, int64_t stage_bytes, mgs_stager **out);
void mgs_stager_destroy(mgs_stager *s);
const char *mgs_stager_last_error(const mgs_stager *s);
int mgs_stager_put(mgs_stager *s, const void *key, uint32_t klen,
                   const void *val, uint32_t vlen);
uint64_t mgs_stager_pending(const mgs_stager *s);
int mgs_stager_pack(mgs_stager *s, mgs_round *out, const void **local_records,
                    int *drained);
/* mgs_stager_pack in two halves, for a caller with work of its own while the GPU
 * packs (the non-blocking shuffler tests its exchanges in flight,
 * src/nbcollectiveshuffler.h:279-387): _begin enqueues the upload and the pack
 * kernels (mgs_pack_device_async), _end waits and reports as mgs_stager_pack does.
 * No other stager call between the two: put, the send image views and
 * mgs_stager_round_done return MGS_ERR_STATE until _end has run. */
int mgs_stager_pack_begin(mgs_stager *s);
int mgs_stager_pack_end(mgs_stager *s, mgs_round *out,
                        const void **local_records, int *drained);
int mgs_stager_send_image_host(mgs_stager *s, const void **host_image,
                               const int64_t **send_offset);
/* The host view copied straight into a buffer of the caller's (nranks*buf_size
 * bytes, partition d at d*buf_size): the message buffers of the non-blocking
 * shuffler (src/nbcollectiveshuffler.h:426-449, :222-277).  mgs_host_alloc /
 * mgs_host_free hand out page-locked memory, which the copy reaches directly (the
 * reference's buffers come from mem_aligned_malloc, :434-437). */
int mgs_stager_send_image_to(mgs_stager *s, void *host_dst,
                             const int64_t **send_offset);
void *mgs_host_alloc(size_t bytes);
void mgs_host_free(void *p);
/* Device memory on GPU `device`, and a blocking copy from it into host memory
 * (page-locked or not), for host code that keeps device buffers of its own
 * without being built against HIP: the destination streams of a whole map in
 * the shuffler for device-resident batches (mimir_gpu::StreamsGpuShuffler; the
 * reference's buffers come from mem_aligned_malloc, src/collectiveshuffler.h:
 * 95-98).  NULL / MGS_ERR_HIP on failure. */
void *mgs_device_alloc(int device, size_t bytes);
void mgs_device_free(int device, void *p);
int mgs_copy_to_host(int device, void *host_dst, const void *d_src, size_t bytes);
int mgs_stager_send_image_device(mgs_stager *s, void **d_image,
                                 const int64_t **send_offset);
int mgs_stager_round_done(mgs_stager *s);
/* mgs_stager_put inlined by the caller (the per-record write() of a shim costs a
 * call into the library otherwise): the staging buffer itself.  The caller
 * appends wire-form records at bytes + used, stores each record's end in
 * off[nrec + 1] (a fixed-width layout too), stops at cap bytes / max_rec
 * records, and hands the new fill back with mgs_stager_commit before any other
 * stager call.  writable == 0: use mgs_stager_put (the buffer's contents are in
 * the pack kernels' hands, or the stager is combining). */
typedef struct mgs_stage_window {
    uint8_t *bytes;
    uint32_t *off;
    int64_t cap;
    uint64_t max_rec;
    int64_t used;
    uint64_t nrec;
    int32_t writable;
    int32_t reserved;
} mgs_stage_window;
int mgs_stager_window(mgs_stager *s, mgs_stage_window *w);
int mgs_stager_commit(mgs_stager *s, int64_t used, uint64_t nrec);
/* For a transport that moves device memory (RCCL, GPU-aware MPI, peer copies):
 * the device receive image -- nranks*buf_size bytes, the reference's recv_buffer
 * (src/collectiveshuffler.h:97-98), to be filled source by source without gaps
 * (:637-639) -- and save_data (:563-585) on its first nbytes bytes: the record
 * boundaries are found on the GPU (mgs_unpack_device), bytes and offsets are
 * copied to pinned host memory for out->write.  *host_rec_off is NULL for
 * layouts mgs_unpack_device does not take (walk the bytes then). */
int mgs_stager_recv_image_device(mgs_stager *s, void **d_recv);
int mgs_stager_recv_to_host(mgs_stager *s, int64_t nbytes,
                            const void **host_records,
                            const uint32_t **host_rec_off,
                            uint64_t *n_records);
int mgs_stager_set_combine(mgs_stager *s, int op);

/* Device time of the dominant kernel of the last completed pack call (the
 * one-sweep kernel, or the generic path's scatter kernel), from HIP events
 * recorded around that launch on the stream it ran on.  Measurement aid; has
 * no counterpart in the reference. */
int mgs_last_kernel_ms(mgs_handle *h, float *ms);

/* Tuning / calibration aids (no counterpart in the reference).
 * mgs_set_tuning selects which compiled instantiation of the one-sweep kernel
 * later pack calls run (0 = the default); mgs_tuning_name names it (NULL past
 * the end of the list).  mgs_copy_calibration streams n_records 16-byte
 * records from d_in to d_out with the same tile loop and nothing else, and
 * reports the device time: what a perfect partitioner could reach here. */
int mgs_set_tuning(mgs_handle *h, int variant);
int mgs_get_tuning(const mgs_handle *h);
int mgs_debug_stats(mgs_handle *h, uint64_t out[8]);
/* Diagnostic: the round finder of the last completed streams-layout call: [0]
 * its wall time in 100 MHz ticks, [1..3] shader cycles spent searching for the
 * tiles that hold cuts, hashing those tiles, walking them. */
int mgs_finder_stats(mgs_handle *h, uint64_t out[4]); /* counters of a "-stats" variant */
const char *mgs_tuning_name(int variant);
int mgs_copy_calibration(mgs_handle *h, const void *d_in, void *d_out,
                         uint64_t n_records, void *stream, float *ms);

/* ---- synthetic inputs for measurement (seeded; the reference generators
 * seed from std::random_device, generator/gen_words.cpp:85-86) -------------
 * n records {u64 key, u64 val}: key = splitmix64 of (seed, first+i),
 * val = first+i. */
int mgs_synth_u64_records(void *d_out, uint64_t n, uint64_t first,
                          uint64_t seed, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MIMIR_GPU_SHUFFLE_H */
