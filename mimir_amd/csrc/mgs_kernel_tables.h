// The instantiations of the kernel templates live in translation units of their own
// (mgs_os16.hip, mgs_osv.hip, mgs_osd.hip, mgs_osr.hip), so that they compile side by side
// and a change to one family rebuilds only that one.  This header is what the C ABI's
// translation unit sees of them.
#ifndef MGS_KERNEL_TABLES_H
#define MGS_KERNEL_TABLES_H

#include <hip/hip_runtime.h>

#include "mgs_onesweepv.h"

namespace mgs {

// ---- 16-byte records (mgs_os16.hip) ---------------------------------------------------
typedef void (*OsKernel)(OsParams);
struct OsVariant {
    const char *name;
    int threads, items, blocks_per_cu, tile;
    bool agg16;         // uses the 16-bit count array of the windowed scan
    OsKernel kernel[7]; // by number of destination bits (1..6); [0] unused
};
// Variant 0 is what production calls run; the rest are there to be timed against it
// (mgs_set_tuning, tools/tune_onesweep.py): each prices one design decision on the hardware.
const OsVariant *os16_variants(int *count);
// The production kernel with the destination taken from a bit field of a record's first word
// (OSV_DIGIT): one pass of a stable radix sort of 16-byte records.
const OsVariant *os16_digit_variant();

// ---- byte-granular records, LDS staging (mgs_osv.hip) ------------------------------------
// k: records per worker lane (1, 2, 4); lm: layout mode 0..2
void osv_launch(const OvParams &q, int k, int lm, uint32_t grid, hipStream_t stream);

// ---- byte-granular records, direct scatter with round cuts (mgs_osd.hip) ------------------
constexpr int OSD_KINDS = 25;
// ki: row of the instantiation table (mgs_osd.hip), lm: layout mode 0..2
void osd_launch(const OvParams &q, int ki, int lm, uint32_t grid, hipStream_t stream);
void osd_check_launch(const OvParams &q, uint32_t records_per_wave, uint32_t waves, uint32_t ntiles, hipStream_t stream);

} // namespace mgs
#endif
