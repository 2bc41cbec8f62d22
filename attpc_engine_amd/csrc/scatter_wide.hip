// scatter_wide.hip -- the scatter kernel with u64 electrons per table slot (one 1024-thread workgroup, 8192 slots of
// 12 bytes per CU: the table of rounds 1 and 2).  Same source as scatter.hip, other build-time constants.  The default
// builds keep u32 sums (a third more slots in the same LDS) and hand a window in which a sum could wrap to
// lone_bucket_kernel; a detector where that happens all the time (electrons x gain per pad beyond 2^31: a gain far
// beyond the AT-TPC's) is switched to this build by the host (abi.hip: prefer_wide).
#define ATTPC_SC_VARIANT wide
#define ATTPC_SC_WIDE_CHARGE 1
#include "scatter.hip"
