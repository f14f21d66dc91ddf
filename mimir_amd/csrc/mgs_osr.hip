// The instantiations of k_onesweepr, a translation unit of their own.
#define MGS_TEMPLATES_ONLY
#include "mgs_onesweepr.h"
#include "mgs_onesweepr_api.h"

namespace mgs {

namespace {
typedef void (*OsrKernel)(OvParams);
struct OsrEntry {
    int k, nq, lm;
    bool a4, stats;
    OsrKernel fn;
};
#define OSR(K, NQ, LM, A4, V) \
    {K, NQ, LM, A4, (V) != 0, k_onesweepr<OSR_THREADS, K, NQ, V, 4, 8, LM, A4, (NQ == 1 ? OSR_IMG1 : OSR_IMG2)>}
const OsrEntry kOsr[] = {
    // one string key + fixed-width or absent value
    OSR(4, 1, 1, false, 0), OSR(4, 2, 1, false, 0), OSR(2, 2, 1, false, 0), OSR(1, 2, 1, false, 0),
    // fixed widths up to 16 / 32 bytes; multiples of four bytes move as dwords
    OSR(4, 1, 2, false, 0), OSR(4, 2, 2, false, 0), OSR(4, 1, 2, true, 0), OSR(4, 2, 2, true, 0),
};
#undef OSR
const OsrEntry *find(int k, int nq, int lm, bool a4, bool stats)
{
    for (const OsrEntry &e : kOsr)
        if (e.k == k && e.nq == nq && e.lm == lm && e.a4 == a4 && e.stats == stats) return &e;
    return nullptr;
}
} // namespace

int osr_image_bytes(int nq) { return nq == 1 ? OSR_IMG1 : OSR_IMG2; }

bool osr_has(int k, int nq, int lm, bool a4) { return find(k, nq, lm, a4, false) != nullptr; }

bool osr_launch(const OvParams &q, int k, int nq, int lm, bool a4, bool stats, uint32_t grid, hipStream_t stream)
{
    const OsrEntry *e = find(k, nq, lm, a4, stats);
    if (!e) e = find(k, nq, lm, a4, false);
    if (!e) return false;
    hipLaunchKernelGGL(e->fn, dim3(grid), dim3(OSR_THREADS), 0, stream, q);
    return true;
}

} // namespace mgs
