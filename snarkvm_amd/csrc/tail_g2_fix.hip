// tail_g2_fix.hip - the one-wave kernels that recompute, with the plain addition law (doubling inlined: the only full Fq2 additions of the tail), the fold outputs and
// bit planes the G2 tail kernels flagged because an addition met equal x coordinates (msm.hip.h::msm_fold_fix_kernel, msm_bitplane_fix_kernel).
#define SV_TU_TAIL
#include "msm.hip.h"

namespace sv {
#ifndef SV_NO_G2
SV_TAIL_FIX_KERNELS(, fq2_t)
#endif
}  // namespace sv
