// Host-side entry to the k_onesweepr instantiations (compiled in mgs_osr.hip).
#ifndef MGS_ONESWEEPR_API_H
#define MGS_ONESWEEPR_API_H

#include <hip/hip_runtime.h>

#include "mgs_onesweepv.h"

namespace mgs {

// bytes of the LDS image of the instantiations with nq 16-byte register quads per record
int osr_image_bytes(int nq);
// Is there an instantiation for k records per lane, nq quads, layout mode lm, aligned-dword moves a4?
bool osr_has(int k, int nq, int lm, bool a4);
// Launch it on `stream` (false: no such instantiation; nothing was launched).  pre: 1 / 0 = the
// instantiation that requests a tile's first quads a phase early / at the end of the trip
// before, -1 = whichever exists (the first in the table).  var: OSV_* bits of a diagnostic
// instantiation (OSV_STATS, OSV_FAKE_CHAIN), 0 = the production one, which also stands in when
// the diagnostic one does not exist.
bool osr_launch(const OvParams &q, int k, int nq, int lm, bool a4, int pre, int var, uint32_t grid,
                hipStream_t stream);

} // namespace mgs
#endif
