// The instantiations of k_onesweepv, a translation unit of their own.
#define MGS_TEMPLATES_ONLY
#include "mgs_onesweepv.h"
#include "mgs_kernel_tables.h"

namespace mgs {

void osv_launch(const OvParams &q, int k, int lm, uint32_t grid, hipStream_t stream)
{
    typedef void (*OvKernel)(OvParams);
    constexpr int kOvThreads = 512;
#define OVK(K, V) {k_onesweepv<kOvThreads, K, V, 4, 8, 0>, k_onesweepv<kOvThreads, K, V, 4, 8, 1>, \
                   k_onesweepv<kOvThreads, K, V, 4, 8, 2>}
    static const OvKernel kernels[3][3] = {OVK(1, 0), OVK(2, 0), OVK(4, 0)};
#undef OVK
    const int ki = k == 1 ? 0 : (k == 2 ? 1 : 2);
    hipLaunchKernelGGL(kernels[ki][lm], dim3(grid), dim3(kOvThreads), 0, stream, q);
}

} // namespace mgs
