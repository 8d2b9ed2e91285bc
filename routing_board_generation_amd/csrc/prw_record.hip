// prw_record.hip -- the recording rollout (connector_rollout_kernel<LPB, record>) as a translation
// unit of its own, so that it can be compiled with the compiler's default instruction scheduler:
// the rest of the library (csrc/prw_kernels.hip) is built with -amdgpu-sched-strategy=max-ilp,
// which speeds the walk up by a tenth and slows this one kernel down by as much
// (routing_board_generation_amd/_native.py::build_native).  Same source, other half of it.
#define RBG_TU_RECORD 1
#include "prw_kernels.hip"
