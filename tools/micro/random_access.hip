// Microbenchmark: the rate of independent random 8-byte reads (and of 8-byte atomic-min
// updates) on a table in HBM, as a function of the table's size -- the roofline that bounds
// the open-addressing insert of mgs_group_device / mgs_combine_device, which makes two
// dependent random accesses per record (its slot, then the key the slot points at).
// Usage: random_access [accesses]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <int MODE> // 0: read, 1: atomicMin, 2: two dependent reads
__global__ void k_random(unsigned long long *table, uint64_t slots, uint64_t n, unsigned long long *sink)
{
    const uint64_t stride = (uint64_t) gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t s = mix64(i) % slots;
        if (MODE == 0) acc += table[s];
        else if (MODE == 1) atomicMin(&table[s], (unsigned long long) i);
        else acc += table[(table[s] + s) % slots];
    }
    if (acc == 0x1234567887654321ull) *sink = acc;
}

int main(int argc, char **argv)
{
    const uint64_t n = argc > 1 ? (uint64_t) atof(argv[1]) : 400000000ull;
    unsigned long long *sink;
    hipMalloc(&sink, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (uint64_t mb : {16ull, 256ull, 1024ull, 8192ull}) {
        const uint64_t slots = mb * 1024 * 1024 / 8;
        unsigned long long *table;
        if (hipMalloc(&table, slots * 8) != hipSuccess) break;
        hipMemset(table, 0x11, slots * 8);
        for (int mode = 0; mode < 3; mode++) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k_random<0>, dim3(256 * 8), dim3(256), 0, 0, table, slots, n, sink);
                else if (mode == 1) hipLaunchKernelGGL(k_random<1>, dim3(256 * 8), dim3(256), 0, 0, table, slots, n, sink);
                else hipLaunchKernelGGL(k_random<2>, dim3(256 * 8), dim3(256), 0, 0, table, slots, n, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            printf("{\"table_MiB\": %llu, \"mode\": \"%s\", \"accesses\": %llu, \"ms\": %.3f, \"G_per_s\": %.2f}\n",
                   (unsigned long long) mb, mode == 0 ? "read" : (mode == 1 ? "atomic_min" : "two_dependent_reads"),
                   (unsigned long long) n, best, (double) n / best / 1e6);
        }
        hipFree(table);
    }
    return 0;
}
