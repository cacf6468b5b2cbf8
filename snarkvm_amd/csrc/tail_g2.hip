// tail_g2.hip - the G2 fold kernels (msm.hip.h::msm_fold_kernel over fq2_t; block_sum with the sixteen-lane cooperative additions of hex2.hip.h), instantiated here
// and nowhere else: with the bit-plane kernels (tail_g2_planes.hip) the largest functions of the library, compiled beside the units that launch them (snarkvm_amd/build.py).
#define SV_TU_TAIL
#include "msm.hip.h"

namespace sv {
#ifndef SV_NO_G2
SV_TAIL_FOLD_KERNELS(, fq2_t)
#endif
}  // namespace sv
