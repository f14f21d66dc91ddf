// The instantiations of k_onesweepd, a translation unit of their own.
#define MGS_TEMPLATES_ONLY
#define MGS_OSD_UNIT
#include "mgs_onesweepd.h"
#include "mgs_kernel_tables.h"

namespace mgs {

void osd_launch(const OvParams &q, int ki, int lm, uint32_t grid, hipStream_t stream)
{
    typedef void (*OvKernel)(OvParams);
    constexpr int kOvThreads = 512;
#define ODKL(K, V, L) {k_onesweepd<kOvThreads, K, V, 4, L, 0>, k_onesweepd<kOvThreads, K, V, 4, L, 1>, \
           k_onesweepd<kOvThreads, K, V, 4, L, 2>}
#define ODK(K, V) ODKL(K, V, 8)
    static const OvKernel kernels[25][3] = {ODK(1, 0), ODK(2, 0), ODK(3, 0), ODK(4, 0), ODK(4, OSV_FAKE_CHAIN),
                    ODK(1, OSV_NT_LOAD), ODK(2, OSV_NT_LOAD), ODK(3, OSV_NT_LOAD),
                    ODK(4, OSV_NT_LOAD), ODK(1, OSV_STATS), ODK(2, OSV_STATS),
                    ODK(3, OSV_STATS), ODK(4, OSV_STATS),
                    ODK(1, OSV_STATS | OSV_NO_PREFETCH), ODK(2, OSV_STATS | OSV_NO_PREFETCH),
                    ODK(3, OSV_STATS | OSV_NO_PREFETCH), ODK(4, OSV_STATS | OSV_NO_PREFETCH),
                    // experiments: a wider window of the chained scan
                    ODKL(1, 0, 16), ODKL(2, 0, 16), ODKL(3, 0, 16), ODKL(4, 0, 16),
                    ODKL(1, OSV_STATS | OSV_NO_PREFETCH, 16), ODKL(2, OSV_STATS | OSV_NO_PREFETCH, 16),
                    ODKL(3, OSV_STATS | OSV_NO_PREFETCH, 16), ODKL(4, OSV_STATS | OSV_NO_PREFETCH, 16)};
#undef ODK
#undef ODKL
    hipLaunchKernelGGL(kernels[ki][lm], dim3(grid), dim3(kOvThreads), 0, stream, q);
}

void osd_check_launch(const OvParams &q, uint32_t records_per_wave, uint32_t waves, uint32_t ntiles, hipStream_t stream)
{
    const uint64_t items = (uint64_t) ntiles * waves;
    hipLaunchKernelGGL(k_osd_check, dim3((unsigned) ((items + 255) / 256)), dim3(256), 0, stream, q, records_per_wave,
                       waves);
}

} // namespace mgs
