// tools/exp/g2call.hip - is the G2 mixed addition bound by instruction fetch?  The inlined Fq2 XYZZ + affine addition is ~146 KB of
// straight-line code (the instruction cache of a CU pair holds 64 KB).  Variants of the same register-resident chain of mixed
// additions (ec.hip.h xyzz_t<F>::add_affine), one wave per SIMD like msm_accumulate_seg_kernel<fq2_t, 1, *>:
//   -DVARIANT=0  everything inlined (the product's code)
//   -DVARIANT=1  the Fq two-product reduction, product and square are out-of-line functions (operands by value / pointer: the
//                compiler's calling convention moves what does not fit 16 argument registers through scratch)
//   -DVARIANT=2  the Fq2 product and square as out-of-line functions
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVARIANT=k tools/exp/g2call.hip -o tools/exp/g2call_k
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../snarkvm_amd/csrc/ec.hip.h"
using namespace sv;
#ifndef VARIANT
#define VARIANT 0
#endif

#if VARIANT == 1
static __device__ __noinline__ fq_t fq_dop_ool(const fq_t a, const fq_t* b, const fq_t* c, const fq_t* d) { return fq_t::diff_of_products(a, *b, *c, *d); }
static __device__ __noinline__ fq_t fq_mul_ool(const fq_t a, const fq_t* b) { return a * *b; }
static __device__ __noinline__ fq_t fq_sqr_ool(const fq_t a) { return a.sqr(); }
#endif

struct fq2x_t : fq2_t {  // fq2_t with out-of-line products
    __device__ fq2x_t() {}
    __device__ fq2x_t(const fq2_t& a) : fq2_t(a) {}
    __device__ static fq2x_t zero() { return fq2_t::zero(); }
    __device__ static fq2x_t one() { return fq2_t::one(); }
    __device__ fq2x_t operator+(const fq2x_t& b) const { return (const fq2_t&)*this + (const fq2_t&)b; }
    __device__ fq2x_t operator-(const fq2x_t& b) const { return (const fq2_t&)*this - (const fq2_t&)b; }
    __device__ fq2x_t neg() const { return fq2_t::neg(); }
    __device__ fq2x_t dbl() const { return fq2_t::dbl(); }
#if VARIANT == 0
    __device__ fq2x_t operator*(const fq2x_t& b) const { return (const fq2_t&)*this * (const fq2_t&)b; }
    __device__ fq2x_t sqr() const { return fq2_t::sqr(); }
#elif VARIANT == 1
    __device__ fq2x_t operator*(const fq2x_t& b) const {
        const fq_t m5 = mul5(c1), n1 = c1.neg();
        fq2_t r;
        r.c0 = fq_dop_ool(c0, &b.c0, &m5, &b.c1);
        r.c1 = fq_dop_ool(c0, &b.c1, &n1, &b.c0);
        return r;
    }
    __device__ fq2x_t sqr() const {
        fq_t a = fq_sqr_ool(c0), bb = fq_sqr_ool(c1), m = fq_mul_ool(c0, &c1);
        fq2_t r = {a - mul5(bb), m.dbl()};
        return r;
    }
#else
    static __device__ __noinline__ void mul_ool(fq2_t* r, const fq2_t* a, const fq2_t* b) { *r = *a * *b; }
    static __device__ __noinline__ void sqr_ool(fq2_t* r, const fq2_t* a) { *r = a->sqr(); }
    __device__ fq2x_t operator*(const fq2x_t& b) const {
        fq2_t r, x = *this, y = b;
        mul_ool(&r, &x, &y);
        return r;
    }
    __device__ fq2x_t sqr() const {
        fq2_t r, x = *this;
        sqr_ool(&r, &x);
        return r;
    }
#endif
    __device__ static fq2x_t diff_of_products(const fq2x_t& a, const fq2x_t& b, const fq2x_t& c, const fq2x_t& d) { return a * b - c * d; }
};

__device__ __forceinline__ fq_t seed_fq(uint32_t s) {
    fq_t a;
    for (int i = 0; i < 13; i++) {
        s = s * 1664525u + 1013904223u;
        a.v[i] = s & LIMB_MASK;
    }
    a.v[12] &= 0x00ffffffu;
    return a;
}

__global__ void __launch_bounds__(256, 1) k_madd(uint32_t* out, int iters) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    fq2_t sx = {seed_fq(t * 4 + 1), seed_fq(t * 4 + 2)}, sy = {seed_fq(t * 4 + 3), seed_fq(t * 4 + 4)};
    aff_t<fq2x_t> p = {fq2x_t(sx), fq2x_t(sy)};
    xyzz_t<fq2x_t> acc = {fq2x_t(sy), fq2x_t(sx), fq2x_t::one(), fq2x_t::one()};
    for (int it = 0; it < iters; it++) {
        acc.add_affine(p, (it & 1) != 0);
        p.x = p.x + acc.zz;  // a different point every time
    }
    uint32_t s = 0;
    for (int i = 0; i < 13; i++) s ^= acc.x.c0.v[i] + 3 * acc.y.c1.v[i] + 5 * acc.zz.c0.v[i] + 7 * acc.zzz.c1.v[i];
    out[t] = s;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    uint32_t* d;
    const int blocks = prop.multiProcessorCount * 4;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms;
    // one wave alone: latency of one addition
    hipLaunchKernelGGL(k_madd, dim3(1), dim3(64), 0, 0, d, 4);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_madd, dim3(1), dim3(64), 0, 0, d, 100);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double lat = ms * 1e3 / 100;
    // the whole chip, one wave per SIMD (one block of 256 per CU), 4 rounds
    const int iters = 64;
    hipLaunchKernelGGL(k_madd, dim3(blocks), dim3(256), 0, 0, d, 2);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_madd, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    uint32_t h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("VARIANT %d: single wave %.2f us per G2 mixed addition; whole chip, 1 wave / SIMD: %.3f G additions/s = %.2f us per wave-addition (checksum %08x %08x)\n",
           VARIANT, lat, (double)blocks * 256 * iters / (ms * 1e-3) * 1e-9, ms * 1e3 / (iters * 4), h[0], h[1]);
    return 0;
}
