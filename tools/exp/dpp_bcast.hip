// Hardware check of the DPP controls hex2.hip.h relies on (gfx950): row_newbcast:N must give every lane of a 16-lane row the value of lane N of ITS row.
//   hipcc --offload-arch=gfx950 -O2 tools/exp/dpp_bcast.hip -o tools/exp/dpp_bcast && tools/exp/dpp_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N>
__device__ int bc(int v) { return __builtin_amdgcn_mov_dpp(v, 0x150 + N, 0xf, 0xf, true); }
__global__ void k(int* out) {
    const int v = 1000 + (int)threadIdx.x;
    out[threadIdx.x * 4 + 0] = bc<0>(v);
    out[threadIdx.x * 4 + 1] = bc<6>(v);
    out[threadIdx.x * 4 + 2] = bc<15>(v);
    out[threadIdx.x * 4 + 3] = __builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, true);  // quad_perm:[2,3,0,1]
}
int main() {
    int* d;
    hipMalloc(&d, 64 * 4 * sizeof(int));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[256];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 64; t++) {
        const int row = t & ~15;
        if (h[t * 4 + 0] != 1000 + row + 0 || h[t * 4 + 1] != 1000 + row + 6 || h[t * 4 + 2] != 1000 + row + 15 || h[t * 4 + 3] != 1000 + (t ^ 2)) bad++;
    }
    printf("row_newbcast / quad_perm[2,3,0,1] check: %s (%d lanes differ); lane 21 sees %d %d %d %d\n", bad ? "FAIL" : "OK", bad, h[84], h[85], h[86], h[87]);
    return bad != 0;
}
