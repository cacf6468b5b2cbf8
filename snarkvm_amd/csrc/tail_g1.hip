// tail_g1.hip - the G1 fold / bit-plane kernels (msm.hip.h: msm_fold_kernel, msm_bitplane_kernel over the exact and the lazy tail arithmetic), instantiated here and
// nowhere else: a translation unit of their own so that the build compiles them beside the units that launch them (snarkvm_amd/build.py).
#define SV_TU_TAIL
#include "msm.hip.h"

namespace sv {
SV_TAIL_KERNELS(, fq_t)
SV_TAIL_KERNELS(, fqz_t)
}  // namespace sv
