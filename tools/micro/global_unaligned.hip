// Microbenchmark: how fast can 64-lane waves scatter short records to 8
// partition streams when the records are not 16 bytes long?  Each row of 64
// records holds 8 records for each of 8 destinations (lanes 8 apart share a
// destination), as a hash-partitioned tile would; the store pattern is what a
// kernel that writes records straight to their partitions would issue.
//   0  16-byte records, one aligned 16-byte store each (the yardstick)
//   1  14-byte records, two overlapping 8-byte stores at unaligned addresses
//   2  14-byte records, naturally aligned pieces (2/4/8 bytes) chosen per record
//   3  22-byte records, two overlapping 16-byte stores at unaligned addresses
//   4  14-byte records, 8 + 4 + 2 bytes at unaligned addresses
//   5  20-byte records, 16 + 4 bytes (4-byte aligned addresses)
// Usage: global_unaligned [log2 records] [reps]; prints GB/s written per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

struct __attribute__((packed)) U16 { unsigned short v; };
struct __attribute__((packed)) U32 { unsigned int v; };
struct __attribute__((packed)) U64 { unsigned long long v; };
struct __attribute__((packed)) U128 { unsigned int x, y, z, w; };

template <int VARIANT, int L>
__global__ __launch_bounds__(256) void k_scatter(unsigned char *out, unsigned long long part_stride,
                                                 unsigned long long rows)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long wave = (unsigned long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    const unsigned long long nwaves = (unsigned long long) gridDim.x * 4;
    for (unsigned long long row = wave; row < rows; row += nwaves) {
        const unsigned d = (unsigned) (lane * 5 + row) & 7u;
        const unsigned long long pos = row * 8 + (unsigned) (lane >> 3);
        unsigned char *dst = out + d * part_stride + pos * L;
        const unsigned long long a = row * 64 + lane, b = a * 0x9E3779B97F4A7C15ull;
        if (VARIANT == 0) {
            *reinterpret_cast<uint4 *>(dst) = make_uint4((unsigned) a, (unsigned) (a >> 32), (unsigned) b, (unsigned) (b >> 32));
        }
        else if (VARIANT == 1) {
            reinterpret_cast<U64 *>(dst)->v = a;
            reinterpret_cast<U64 *>(dst + L - 8)->v = b;
        }
        else if (VARIANT == 2) {
            unsigned char *p = dst, *e = dst + L;
            unsigned long long v = a;
            if (((uintptr_t) p & 2) && p + 2 <= e) { *reinterpret_cast<unsigned short *>(p) = (unsigned short) v; p += 2; v >>= 16; }
            if (((uintptr_t) p & 4) && p + 4 <= e) { *reinterpret_cast<unsigned int *>(p) = (unsigned int) v; p += 4; v = b; }
            if (p + 8 <= e) { *reinterpret_cast<unsigned long long *>(p) = v; p += 8; v = b >> 7; }
            if (p + 4 <= e) { *reinterpret_cast<unsigned int *>(p) = (unsigned int) v; p += 4; v >>= 32; }
            if (p + 2 <= e) { *reinterpret_cast<unsigned short *>(p) = (unsigned short) v; p += 2; }
        }
        else if (VARIANT == 3) {
            *reinterpret_cast<U128 *>(dst) = U128{(unsigned) a, (unsigned) (a >> 32), (unsigned) b, (unsigned) (b >> 32)};
            *reinterpret_cast<U128 *>(dst + L - 16) = U128{(unsigned) b, (unsigned) (a >> 32), (unsigned) a, (unsigned) (b >> 32)};
        }
        else if (VARIANT == 4) {
            reinterpret_cast<U64 *>(dst)->v = a;
            reinterpret_cast<U32 *>(dst + 8)->v = (unsigned) b;
            reinterpret_cast<U16 *>(dst + 12)->v = (unsigned short) (b >> 32);
        }
        else if (VARIANT == 5) {
            *reinterpret_cast<U128 *>(dst) = U128{(unsigned) a, (unsigned) (a >> 32), (unsigned) b, (unsigned) (b >> 32)};
            reinterpret_cast<U32 *>(dst + 16)->v = (unsigned) (b >> 13);
        }
    }
}

template <int VARIANT, int L>
static void run(const char *what, unsigned char *out, unsigned long long n, int reps)
{
    const unsigned long long rows = n / 64, part_stride = (rows * 8 * L + 4095) / 4096 * 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k_scatter<VARIANT, L>), dim3(256 * 8), dim3(256), 0, 0, out, part_stride, rows);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL((k_scatter<VARIANT, L>), dim3(256 * 8), dim3(256), 0, 0, out, part_stride, rows);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    printf("{\"variant\": %d, \"what\": \"%s\", \"record_bytes\": %d, \"records\": %llu, \"ms\": %.4f, "
           "\"G_records_per_s\": %.2f, \"GB_per_s_written\": %.1f}\n",
           VARIANT, what, L, n, ms, n / ms / 1e6, (double) n * L / ms / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 28, reps = argc > 2 ? atoi(argv[2]) : 5;
    const unsigned long long n = 1ull << lg;
    unsigned char *out;
    if (hipMalloc(&out, n * 24 + (64u << 20)) != hipSuccess) return 1;
    run<0, 16>("16 B aligned, one 16-byte store", out, n, reps);
    run<1, 14>("14 B, two overlapping unaligned 8-byte stores", out, n, reps);
    run<2, 14>("14 B, naturally aligned pieces", out, n, reps);
    run<4, 14>("14 B, 8+4+2 at unaligned addresses", out, n, reps);
    run<3, 22>("22 B, two overlapping unaligned 16-byte stores", out, n, reps);
    run<5, 20>("20 B, 16+4 at 4-byte aligned addresses", out, n, reps);
    hipFree(out);
    return 0;
}
