// mdrp_tu.hip - one of the library's secondary translation units: compiled once per group with -DMDRP_TU=<group> (mdrp_amd/build.py),
// it holds the explicit instantiations of that group's kernels (mdrp_instances.h) and nothing else.  No host code, no state.
#define MDRP_SECONDARY_TU 1
#include "mdrp_kernels.h"
#include "mdrp_classic.h"
#define MDRP_INST
#include "mdrp_instances.h"
namespace mdrp {
#if MDRP_TU == 1
MDRP_INSTANCES_FINAL_64
#elif MDRP_TU == 2
MDRP_INSTANCES_FINAL_256
#elif MDRP_TU == 3
MDRP_INSTANCES_CLASSIC
#else
#error "MDRP_TU must be 1 (k_final, 64 lanes), 2 (k_final, 256 lanes) or 3 (5- / 6- / 7-point baselines)"
#endif
} // namespace mdrp
