// The instantiations of k_onesweepr, a translation unit of their own.
#define MGS_TEMPLATES_ONLY
#include "mgs_onesweepr.h"
#include "mgs_onesweepr_api.h"

namespace mgs {

namespace {
typedef void (*OsrKernel)(OvParams);
struct OsrEntry {
    int k, nq, lm;
    bool a4, pre;
    int var; // OSV_* bits of the instantiation (0: production)
    OsrKernel fn;
};
// NQ: 16-byte register quads per record; 3 stands for a quad and a dword (records up to 20 bytes).
// The scan wave looks back four rows of tiles per poll for string keys, eight for fixed widths (measured:
// tools/ab_look.sh; sixteen cost 7 % on every shape).
#define OSR(K, NQ, LM, A4, PRE, V)                                                                                   \
    {K, NQ, LM, A4, PRE, V,                                                                                          \
     k_onesweepr<OSR_THREADS, K, (NQ == 3 ? 2 : NQ), V, 4, (LM == 1 ? 4 : 8), LM, A4, (NQ == 1 ? OSR_IMG1 : OSR_IMG2), PRE, (NQ == 3 ? 1 : 4)>}
const OsrEntry kOsr[] = {
    // one string key + fixed-width or absent value
    OSR(4, 2, 1, false, false, 0), OSR(3, 2, 1, false, true, 0), OSR(3, 2, 1, false, false, 0), OSR(2, 2, 1, false, true, 0),
    OSR(1, 2, 1, false, true, 0), OSR(4, 3, 1, false, false, 0), OSR(4, 3, 1, false, true, 0),
    // fixed widths up to 16 / 32 bytes; multiples of four bytes move as dwords.  (The early-request form of
    // a (k, nq) comes first: it is the one a call gets that does not pin PRE.)
    OSR(5, 1, 2, false, true, 0), OSR(4, 1, 2, false, true, 0), OSR(4, 2, 2, false, true, 0), OSR(4, 2, 2, false, false, 0),
    OSR(3, 2, 2, false, true, 0),
    OSR(5, 1, 2, true, true, 0), OSR(4, 1, 2, true, true, 0), OSR(4, 2, 2, true, true, 0), OSR(4, 2, 2, true, false, 0),
    OSR(3, 2, 2, true, true, 0),
    // diagnostic builds: cycles per phase
    OSR(4, 2, 1, false, false, OSV_STATS), OSR(3, 2, 1, false, true, OSV_STATS), OSR(4, 2, 2, true, false, OSV_STATS),
    OSR(4, 1, 2, true, true, OSV_STATS), OSR(4, 3, 1, false, true, OSV_STATS),
    OSR(4, 2, 1, false, false, OSV_STATS | OSV_NO_PREFETCH), OSR(4, 2, 1, false, false, OSV_STATS | OSV_FAKE_CHAIN),
    // timing only (wrong output): no chained scan, positions are made up -- prices the chain
    OSR(4, 2, 1, false, false, OSV_FAKE_CHAIN), OSR(3, 2, 1, false, true, OSV_FAKE_CHAIN), OSR(4, 2, 2, true, false, OSV_FAKE_CHAIN),
};
#undef OSR
const OsrEntry *find(int k, int nq, int lm, bool a4, int pre, int var)
{
    for (const OsrEntry &e : kOsr)
        if (e.k == k && e.nq == nq && e.lm == lm && e.a4 == a4 && (pre < 0 || e.pre == (pre != 0)) && e.var == var)
            return &e;
    return nullptr;
}
} // namespace

int osr_image_bytes(int nq) { return nq == 1 ? OSR_IMG1 : OSR_IMG2; }

bool osr_has(int k, int nq, int lm, bool a4) { return find(k, nq, lm, a4, -1, 0) != nullptr; }

bool osr_launch(const OvParams &q, int k, int nq, int lm, bool a4, int pre, int var, uint32_t grid, hipStream_t stream)
{
    const OsrEntry *e = find(k, nq, lm, a4, pre, var);
    if (!e) e = find(k, nq, lm, a4, -1, var);
    if (!e) e = find(k, nq, lm, a4, pre, 0);
    if (!e) e = find(k, nq, lm, a4, -1, 0);
    if (!e) return false;
    hipLaunchKernelGGL(e->fn, dim3(grid), dim3(OSR_THREADS), 0, stream, q);
    return true;
}

} // namespace mgs
