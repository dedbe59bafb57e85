// nhip_bnb_instr.hip -- the branch-and-bound matcher's kernels with their instrumentation compiled in (per-phase
// shader clocks, work counters, per-workgroup timestamps, the NHIP_BNB_DEBUG timing switches).  Launched instead of the
// product kernels only when NHIP_BNB_INSTRUMENT=1 is set (nhip_bnb.hip, launch_csm_bnb); same source, same records.
#define NHIP_BNB_INSTR 1
#include "nhip_bnb.hip"
