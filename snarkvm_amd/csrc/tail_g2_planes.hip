// tail_g2_planes.hip - the G2 bit-plane kernels (msm.hip.h::msm_bitplane_kernel over fq2_t), see tail_g2.hip.
#define SV_TU_TAIL
#include "msm.hip.h"

namespace sv {
#ifndef SV_NO_G2
SV_TAIL_PLANE_KERNELS(, fq2_t)
#endif
}  // namespace sv
