// g2lazy_dev.hip - round-4 diagnostic: the lazy G2 mixed addition (ffl2.hip.h) as a device kernel against the same chain on the host.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/g2lazy_dev.hip -o tools/exp/g2lazy_dev ; run: tools/exp/g2lazy_dev /tmp/g2pts.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#include "../../snarkvm_amd/csrc/ffl2.hip.h"
using namespace sv;

struct slot_t {
    fq2l_t x, y;
};
static __device__ __forceinline__ void exceptional_add(xyzz_lazy2_t* acc, const fq2l_t* px, const fq2l_t* py, bool neg) {
    xyzz_t<fq2_t> ex = acc->to_exact();
    const fq_t c348 = fq_t::from_table(FqLConv::C348);
    fq_t t[4];
#pragma unroll
    for (int i = 0; i < 13; i++) {
        t[0].v[i] = (uint32_t)px->c0.v[i], t[1].v[i] = (uint32_t)px->c1.v[i];
        t[2].v[i] = (uint32_t)py->c0.v[i], t[3].v[i] = (uint32_t)py->c1.v[i];
    }
    ex.add_affine({{t[0] * c348, t[1] * c348}, {t[2] * c348, t[3] * c348}}, neg);
    *acc = xyzz_lazy2_t::from_exact(ex);
}
__global__ void __launch_bounds__(256, 1) k_chain(const slot_t* pts, int npts, int iters, g2_lazy_partial_t* out, int* flags) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    xyzz_lazy2_t acc = xyzz_lazy2_t::infinity();
    int bad = 0;
    for (int i = 0; i < iters; i++) {
        const slot_t s = pts[(t * 7 + i / 2) % npts];  // every point twice in a row: the second addition is a doubling
        if (!acc.madd(s.x, s.y, false)) {
            bad++;
            xyzz_lazy2_t tmp = acc;
            const fq2l_t tx = s.x, ty = s.y;
            exceptional_add(&tmp, &tx, &ty, false);
            acc = tmp;
        }
    }
    acc.store_raw(&out[t]);
    flags[t] = bad;
}
int main(int argc, char** argv) {
    FILE* f = fopen(argc > 1 ? argv[1] : "/tmp/g2pts.bin", "rb");
    if (!f) { printf("no points file\n"); return 1; }
    std::vector<unsigned char> buf(48 * 200);
    if (fread(buf.data(), 1, buf.size(), f) != buf.size()) { printf("short read\n"); return 1; }
    fclose(f);
    std::vector<slot_t> pts;
    for (int i = 0; i < 48; i++) {
        const uint32_t* src = (const uint32_t*)(buf.data() + 200 * i);
        const fq2_t x = fq2_t::from_raw_words(src), y = fq2_t::from_raw_words(src + 24);
        pts.push_back({{fql_canonical_from_exact(x.c0), fql_canonical_from_exact(x.c1)}, {fql_canonical_from_exact(y.c0), fql_canonical_from_exact(y.c1)}});
    }
    const int T = 64, iters = 5;
    slot_t* d_pts;
    g2_lazy_partial_t* d_out;
    int* d_flags;
    hipMalloc(&d_pts, pts.size() * sizeof(slot_t));
    hipMalloc(&d_out, T * sizeof(g2_lazy_partial_t));
    hipMalloc(&d_flags, T * sizeof(int));
    hipMemcpy(d_pts, pts.data(), pts.size() * sizeof(slot_t), hipMemcpyHostToDevice);
    for (int it = 1; it <= iters; it++) {
        printf("launch iters=%d ...\n", it);
        fflush(stdout);
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(T), 0, 0, d_pts, (int)pts.size(), it, d_out, d_flags);
        hipError_t e = hipDeviceSynchronize();
        printf("   sync: %s\n", hipGetErrorString(e));
        std::vector<g2_lazy_partial_t> out(T);
        std::vector<int> flags(T);
        hipMemcpy(out.data(), d_out, T * sizeof(g2_lazy_partial_t), hipMemcpyDeviceToHost);
        hipMemcpy(flags.data(), d_flags, T * sizeof(int), hipMemcpyDeviceToHost);
        int mism = 0, bad = 0;
        for (int t = 0; t < T; t++) {
            xyzz_lazy2_t acc = xyzz_lazy2_t::infinity();
            for (int i = 0; i < it; i++) {
                const slot_t s = pts[(t * 7 + i / 2) % pts.size()];
                if (!acc.madd(s.x, s.y, false)) {
                    xyzz_t<fq2_t> ex = acc.to_exact();
                    const fq_t c348 = fq_t::from_table(FqLConv::C348);
                    fq_t q[4];
                    for (int l = 0; l < 13; l++) q[0].v[l] = (uint32_t)s.x.c0.v[l], q[1].v[l] = (uint32_t)s.x.c1.v[l], q[2].v[l] = (uint32_t)s.y.c0.v[l], q[3].v[l] = (uint32_t)s.y.c1.v[l];
                    ex.add_affine({{q[0] * c348, q[1] * c348}, {q[2] * c348, q[3] * c348}}, false);
                    acc = xyzz_lazy2_t::from_exact(ex);
                }
            }
            g2_lazy_partial_t want;
            acc.store_raw(&want);
            const xyzz_t<fq2_t> a = xyzz_lazy2_t::exact_from_raw(&out[t]), b = xyzz_lazy2_t::exact_from_raw(&want);
            if (!(a.x == b.x && a.y == b.y && a.zz == b.zz && a.zzz == b.zzz)) mism++;
            bad += flags[t];
        }
        printf("   iters=%d: %d of %d threads differ from the host chain, %d exceptional returns\n", it, mism, T, bad);
    }
    return 0;
}
