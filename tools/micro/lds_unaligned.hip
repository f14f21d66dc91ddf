// Micro-benchmark: cost of unaligned LDS accesses on gfx950.
// hipcc --offload-arch=gfx950 -O3 -o lds_unaligned lds_unaligned.hip && ./lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct __attribute__((packed)) U128 { uint32_t x, y, z, w; };
struct __attribute__((packed)) U32 { uint32_t v; };
struct __attribute__((packed)) U64 { unsigned long long v; };

template <int MODE>
__global__ __launch_bounds__(512) void k(uint32_t *out, unsigned long long *cycles, int shift, int stride)
{
    __shared__ __attribute__((aligned(16))) uint8_t s[36864 + 64];
    const int t = threadIdx.x;
    for (int i = t; i < (36864 + 64) / 4; i += blockDim.x) ((uint32_t *) s)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; it++) {
        const uint32_t off = ((uint32_t) (t * stride + it * 16 * 37) % 36000u & ~15u) + shift;
        if (MODE == 0) { // one 16-byte read at `off`
            const U128 v = *reinterpret_cast<const U128 *>(s + off);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
        else if (MODE == 1) { // five aligned dwords + funnel shift
            const uint32_t *w = reinterpret_cast<const uint32_t *>(s + (off & ~3u));
            const uint32_t a = w[0], b = w[1], c = w[2], d = w[3], e = w[4];
            const uint32_t sh = off & 3u;
            acc += __builtin_amdgcn_alignbyte(b, a, sh) ^ __builtin_amdgcn_alignbyte(c, b, sh)
                   ^ __builtin_amdgcn_alignbyte(d, c, sh) ^ __builtin_amdgcn_alignbyte(e, d, sh);
        }
        else if (MODE == 2) { // four unaligned dwords
            acc += reinterpret_cast<const U32 *>(s + off)->v ^ reinterpret_cast<const U32 *>(s + off + 4)->v
                   ^ reinterpret_cast<const U32 *>(s + off + 8)->v ^ reinterpret_cast<const U32 *>(s + off + 12)->v;
        }
        else if (MODE == 3) { // two unaligned 8-byte reads
            acc += (uint32_t) (reinterpret_cast<const U64 *>(s + off)->v ^ reinterpret_cast<const U64 *>(s + off + 8)->v);
        }
        else if (MODE == 5) {
            reinterpret_cast<U64 *>(s + off)->v = ((unsigned long long) acc << 32) | it;
            acc += it;
        }
        else if (MODE == 6) {
            reinterpret_cast<U32 *>(s + off)->v = acc;
            acc += it;
        }
        else if (MODE == 7) {
            struct __attribute__((packed)) U16 { uint16_t v; };
            reinterpret_cast<U16 *>(s + off)->v = (uint16_t) acc;
            acc += it;
        }
        else if (MODE == 8) {
            s[off] = (uint8_t) acc;
            acc += it;
        }
        else if (MODE == 4) { // unaligned 16-byte write
            U128 v = {acc, (uint32_t) it, (uint32_t) t, 7u};
            *reinterpret_cast<U128 *>(s + off) = v;
            acc += it;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + t] = acc;
    if (t == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int shift, int stride)
{
    uint32_t *out;
    unsigned long long *cyc, h[256];
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 256 * 8);
    for (int r = 0; r < 2; r++) k<MODE><<<256, 512>>>(out, cyc, shift, stride);
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long sum = 0;
    for (int i = 0; i < 256; i++) sum += h[i];
    printf("%-34s shift %2d stride %3d: %7.1f cycles per wave-wide access (8 waves/CU)\n", name, shift, stride,
           (double) sum / 256 / 64);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int stride : {16, 56}) {
        for (int shift : {0, 1, 2, 4, 8}) {
            run<0>("16-byte read", shift, stride);
            run<4>("16-byte write", shift, stride);
            run<5>("8-byte write", shift, stride);
            run<6>("4-byte write", shift, stride);
            run<7>("2-byte write", shift, stride);
            run<8>("1-byte write", shift, stride);
        }
    }
    return 0;
}
