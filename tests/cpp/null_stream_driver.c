/* Test driver for the stream == NULL contract of include/mimir_gpu_shuffle.h, in
 * plain C like a caller outside PyTorch would be written.
 *
 *   null_stream_driver <records.bin> <n> <P> <rank> <buf_size> <seed> <out.bin>
 *
 * Case 1: the n 16-byte records of the file are read into pageable host memory,
 * copied to the GPU with a plain hipMemcpy, and packed right away with
 * mgs_pack_device(..., stream = NULL, ...).
 * Case 2: n records are produced on the GPU by a kernel launched on the
 * default stream (mgs_synth_u64_records with stream = NULL) and packed right
 * away, again with stream = NULL and no synchronisation in between.
 * For each case the round result, the used part of every partition and the
 * local-bypass bytes are appended to <out.bin>; the Python test compares them
 * with the oracle. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mimir_gpu_shuffle.h"

#define CHECK_HIP(x)                                                                  \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) {                                                       \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

static int dump(FILE *f, const mgs_round *r, int P, int64_t buf_size, const void *d_image,
                const void *d_local)
{
    uint8_t *host = (uint8_t *) malloc((size_t) buf_size > 16 ? (size_t) buf_size : 16);
    int64_t head[4 + MGS_MAX_RANKS];
    head[0] = (int64_t) r->next;
    head[1] = (int64_t) r->packed_records;
    head[2] = (int64_t) r->local_records;
    head[3] = r->local_bytes;
    for (int b = 0; b < P; b++) head[4 + b] = r->send_offset[b];
    fwrite(head, sizeof(int64_t), (size_t) (4 + P), f);
    for (int b = 0; b < P; b++) {
        if (r->send_offset[b] == 0) continue;
        if (hipMemcpy(host, (const uint8_t *) d_image + (size_t) b * (size_t) buf_size,
                      (size_t) r->send_offset[b], hipMemcpyDeviceToHost) != hipSuccess)
            return 2;
        fwrite(host, 1, (size_t) r->send_offset[b], f);
    }
    free(host);
    if (r->local_bytes > 0) {
        uint8_t *loc = (uint8_t *) malloc((size_t) r->local_bytes);
        if (hipMemcpy(loc, d_local, (size_t) r->local_bytes, hipMemcpyDeviceToHost) != hipSuccess) return 2;
        fwrite(loc, 1, (size_t) r->local_bytes, f);
        free(loc);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 8) {
        fprintf(stderr, "usage: %s records.bin n P rank buf_size seed out.bin\n", argv[0]);
        return 1;
    }
    const uint64_t n = strtoull(argv[2], NULL, 10);
    const int P = atoi(argv[3]), rank = atoi(argv[4]);
    const int64_t buf_size = strtoll(argv[5], NULL, 10);
    const uint64_t seed = strtoull(argv[6], NULL, 10);

    uint8_t *host = (uint8_t *) malloc(16 * n); /* pageable on purpose */
    FILE *in = fopen(argv[1], "rb");
    if (!in || fread(host, 16, n, in) != n) {
        fprintf(stderr, "cannot read %llu records from %s\n", (unsigned long long) n, argv[1]);
        return 1;
    }
    fclose(in);

    mgs_desc desc;
    memset(&desc, 0, sizeof(desc));
    desc.nranks = P;
    desc.rank = rank;
    desc.layout.key_kind = MGS_KEY_FIXED;
    desc.layout.key_bytes = 8;
    desc.layout.key_count = 1;
    desc.layout.val_kind = MGS_VAL_FIXED;
    desc.layout.val_bytes = 8;
    desc.layout.val_count = 1;
    desc.buf_size = buf_size;
    desc.device = 0;
    desc.max_records = n;
    mgs_handle *h = NULL;
    if (mgs_create(&desc, &h) != MGS_OK) {
        fprintf(stderr, "mgs_create: %s\n", mgs_last_error(NULL));
        return 2;
    }
    void *d_rec = NULL, *d_image = NULL, *d_local = NULL;
    CHECK_HIP(hipMalloc(&d_rec, 16 * n));
    CHECK_HIP(hipMalloc(&d_image, (size_t) P * (size_t) buf_size));
    CHECK_HIP(hipMalloc(&d_local, 16 * n));
    FILE *out = fopen(argv[7], "wb");
    if (!out) return 1;
    mgs_round r;

    /* case 1: blocking copy from pageable memory, then pack on stream NULL */
    CHECK_HIP(hipMemcpy(d_rec, host, 16 * n, hipMemcpyHostToDevice));
    if (mgs_pack_device(h, d_rec, NULL, n, 0, NULL, d_image, rank >= 0 ? d_local : NULL, (int64_t) (16 * n), NULL,
                        MGS_ALGO_AUTO, NULL, &r) != MGS_OK) {
        fprintf(stderr, "pack 1: %s\n", mgs_last_error(h));
        return 3;
    }
    if (dump(out, &r, P, buf_size, d_image, d_local)) return 2;

    /* case 2: records written by a kernel on the default stream, then pack on stream NULL */
    CHECK_HIP(hipMemset(d_rec, 0, 16 * n));
    CHECK_HIP(hipDeviceSynchronize());
    if (mgs_synth_u64_records(d_rec, n, 0, seed, 0, NULL) != MGS_OK) return 3;
    if (mgs_pack_device(h, d_rec, NULL, n, 0, NULL, d_image, rank >= 0 ? d_local : NULL, (int64_t) (16 * n), NULL,
                        MGS_ALGO_AUTO, NULL, &r) != MGS_OK) {
        fprintf(stderr, "pack 2: %s\n", mgs_last_error(h));
        return 3;
    }
    if (dump(out, &r, P, buf_size, d_image, d_local)) return 2;
    fclose(out);
    mgs_destroy(h);
    (void) hipFree(d_rec);
    (void) hipFree(d_image);
    (void) hipFree(d_local);
    free(host);
    return 0;
}
