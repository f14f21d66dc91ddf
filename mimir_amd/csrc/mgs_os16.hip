// The instantiations of the 16-byte one-sweep kernels, a translation unit of their own.
#define MGS_TEMPLATES_ONLY
#include "mgs_onesweep16.h"
#include "mgs_onesweep16w.h"
#include "mgs_kernel_tables.h"

namespace mgs {

namespace {
#define MINW_(T, B) ((B * (T / 64) + 3) / 4)
#define SAME7(k) {k, k, k, k, k, k, k}
#define OSV(name, T, I, V, B) {name, T, I, B, T * I, false, SAME7((k_onesweep16<T, I, V, MINW_(T, B)>))}
#define OSQ(name, T, I, V, B, L)                                                                    \
    {name, T, I, B, (T / 64 - 1) * 64 * I, true,                                                    \
     {nullptr, k_onesweep16q<T, I, V, MINW_(T, B), L, 1>, k_onesweep16q<T, I, V, MINW_(T, B), L, 2>, \
      k_onesweep16q<T, I, V, MINW_(T, B), L, 3>, k_onesweep16q<T, I, V, MINW_(T, B), L, 4>,          \
      k_onesweep16q<T, I, V, MINW_(T, B), L, 5>, k_onesweep16q<T, I, V, MINW_(T, B), L, 6>}}
static const OsVariant kOsVariants[] = {
    OSQ("q512x8-w8", 512, 8, OSV_EARLY_AGG, 2, 8),                      // 0: production
    OSQ("q512x8-w8-stats", 512, 8, OSV_EARLY_AGG | OSV_STATS, 2, 8),
    OSQ("q512x8-fakechain", 512, 8, OSV_FAKE_CHAIN, 2, 8),              // 2
    OSQ("q512x8-w4", 512, 8, OSV_EARLY_AGG, 2, 4),
    OSQ("q512x8-w12", 512, 8, OSV_EARLY_AGG, 2, 12),                    // 4
    OSQ("q512x8-w12-stats", 512, 8, OSV_EARLY_AGG | OSV_STATS, 2, 12),
    OSQ("q512x8-w8-late", 512, 8, 0, 2, 8),                             // 6: counts published by the scan wave
    OSQ("q512x8-w8-late-stats", 512, 8, OSV_STATS, 2, 8),
    OSV("t512x8", 512, 8, 0, 2),                            // 8: LDS image, every wave works, wave 0 walks
    OSV("t512x8-fakechain", 512, 8, OSV_FAKE_CHAIN, 2),
    OSV("t512x8-stats", 512, 8, OSV_STATS, 2),              // 10
};
// one pass of a radix sort of 16-byte records (OSV_DIGIT): not a pack kernel, so not in the list above
static const OsVariant kOsDigit = OSQ("q512x8-w8-digit", 512, 8, OSV_EARLY_AGG | OSV_DIGIT, 2, 8);
} // namespace

const OsVariant *os16_digit_variant() { return &kOsDigit; }

const OsVariant *os16_variants(int *count)
{
    if (count) *count = (int) (sizeof(kOsVariants) / sizeof(kOsVariants[0]));
    return kOsVariants;
}

} // namespace mgs
