// tools/ecbench.hip - wall-clock ALU ceilings of the product's own field / curve arithmetic (ff.hip.h, ec.hip.h) on the whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ecbench.hip -o tools/ecbench
// Every kernel is a register-resident dependent chain per lane (no memory traffic in the loop), launched with enough blocks
// for k waves per SIMD (k = the __launch_bounds__ request), timed with HIP events.  Prints G operations/s chip-wide; the
// accumulate kernel's `alu_roofline` in bench.py is (mixed additions per launch) / (time) / (the madd ceiling printed here).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../snarkvm_amd/csrc/msm.hip.h"
#include "../snarkvm_amd/csrc/ffl.hip.h"

using namespace sv;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ fq_t seed_fq(uint32_t s) {
    fq_t a;
    for (int i = 0; i < 13; i++) {
        s = s * 1664525u + 1013904223u;
        a.v[i] = s & LIMB_MASK;
    }
    a.v[12] &= 0x00ffffffu;  // < q
    return a;
}

// op: 0 mul, 1 sqr, 2 add+sub pair, 3 diff_of_products
template <int MINW, int op>
__global__ void __launch_bounds__(256, MINW) k_field(uint32_t* out, int iters, int) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    fq_t a = seed_fq(tid * 3 + 1), b = seed_fq(tid * 7 + 5);
    for (int it = 0; it < iters; it++) {
        if (op == 0) a = a * b;
        else if (op == 1) a = a.sqr();
        else if (op == 2) { a = a + b; b = b - a; }
        else a = fq_t::diff_of_products(a, b, b, a);
    }
    uint32_t x = 0;
    for (int i = 0; i < 13; i++) x ^= a.v[i] ^ b.v[i];
    out[tid] = x;
}
// op: 0 add_affine (madd), 1 add (xyzz + xyzz), 2 dbl
template <int MINW, int op>
__global__ void __launch_bounds__(256, MINW) k_point(uint32_t* out, int iters, int) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    g1_aff_t p = {seed_fq(tid * 3 + 1), seed_fq(tid * 7 + 5)};  // not on the curve: the formulas do not care
    g1_xyzz_t acc = {seed_fq(tid + 11), seed_fq(tid + 13), seed_fq(tid + 17), seed_fq(tid + 19)};
    g1_xyzz_t q = acc;
    q.x = q.x + p.x;
    for (int it = 0; it < iters; it++) {
        if (op == 0) acc.add_affine(p, (it & 1) != 0);
        else if (op == 1) acc.add(q);
        else acc = acc.dbl();
    }
    uint32_t x = 0;
    for (int i = 0; i < 13; i++) x ^= acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
    out[tid] = x;
}

// the lazily reduced arithmetic of ffl.hip.h: op 0 madd (xyzz_lazy_t::madd), 1 mul, 2 sqr, 3 diff_of_products
template <int MINW, int op>
__global__ void __launch_bounds__(256, MINW) k_lazy(uint32_t* out, int iters, int) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const fql_t px = fql_t::from_limbs(seed_fq(tid * 3 + 1)), py = fql_t::from_limbs(seed_fq(tid * 7 + 5));
    xyzz_lazy_t acc;
    acc.x = fql_t::from_limbs(seed_fq(tid + 11));
    acc.y = fql_t::from_limbs(seed_fq(tid + 13));
    acc.zz = fql_t::from_limbs(seed_fq(tid + 17));
    acc.zzz = fql_t::from_limbs(seed_fq(tid + 19));
    acc.inf = false;
    uint32_t bad = 0;
    for (int it = 0; it < iters; it++) {
        if (op == 0) bad += acc.madd(px, py, (it & 1) != 0) ? 0u : 1u;
        else if (op == 1) acc.x = fql_t::mul(acc.x, py);
        else if (op == 2) acc.x = fql_t::sqr(acc.x - px).normalized();
        else acc.y = fql_t::diff_of_products(acc.zz - px, acc.zzz - py, acc.y, px);
    }
    uint32_t x = bad;
    for (int i = 0; i < 13; i++) x ^= (uint32_t)(acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i]);
    out[tid] = x;
}

// i-cache experiment for the latency-bound tails: the same dependent chain of XYZZ additions executed by ONE wave, either
// through one inlined call site inside a loop (44 KB of straight-line code, warm after the first trip) or through SITES distinct
// inlined sites per trip (SITES x 44 KB: every addition streams cold code), or through one out-of-line function.
__device__ __noinline__ g1_xyzz_t add_outlined(g1_xyzz_t a, const g1_xyzz_t b) {
    a.add(b);
    return a;
}
template <int SITES>
__global__ void __launch_bounds__(64) k_sites(uint32_t* out, int iters, int mode) {
    const uint32_t tid = threadIdx.x;
    g1_xyzz_t acc = {seed_fq(tid + 11), seed_fq(tid + 13), seed_fq(tid + 17), seed_fq(tid + 19)};
    g1_xyzz_t q = acc;
    q.x = q.x + seed_fq(tid * 3 + 1);
    for (int it = 0; it < iters; it++) {
        if (mode == 1) {
            acc = add_outlined(acc, q);
        } else {
#define ADD_SITE acc.add(q); asm volatile("" ::: "memory");
            ADD_SITE
            if (SITES >= 2) { ADD_SITE }
            if (SITES >= 4) { ADD_SITE ADD_SITE }
            if (SITES >= 8) { ADD_SITE ADD_SITE ADD_SITE ADD_SITE }
        }
    }
    uint32_t x = 0;
    for (int i = 0; i < 13; i++) x ^= acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
    out[tid] = x;
}
template <class K>
static double time_single(K kern, int iters, int mode, uint32_t* d_out, hipEvent_t e0, hipEvent_t e1) {
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d_out, 2, mode);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d_out, iters, mode);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3;
}

// the tail's building blocks: a dependent chain of quad-cooperative additions (msm.hip.h::quad_add), and whole block sums
// (mode 0: 64 threads, 1: 256 threads) of one point per lane
__global__ void __launch_bounds__(256, 2) k_quad(uint32_t* out, int iters, int) {
    const uint32_t tid = threadIdx.x >> 2;  // the four lanes of a quad hold identical operands
    g1_xyzz_t acc = {seed_fq(tid + 11), seed_fq(tid + 13), seed_fq(tid + 17), seed_fq(tid + 19)};
    g1_xyzz_t q = acc;
    q.x = q.x + seed_fq(tid * 3 + 1);
    for (int it = 0; it < iters; it++) quad_add(acc, q);
    uint32_t x = 0;
    for (int i = 0; i < 13; i++) x ^= acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
    out[threadIdx.x] = x;
}
__global__ void __launch_bounds__(256, 2) k_block_sum(uint32_t* out, int iters, int) {
    __shared__ xyzz_mem_t<fq_t> sh[16];
    const uint32_t tid = threadIdx.x;
    uint32_t x = 0;
    for (int it = 0; it < iters; it++) {
        g1_xyzz_t acc = {seed_fq(tid + 11 + it), seed_fq(tid + 13), seed_fq(tid + 17), seed_fq(tid + 19)};
        block_sum<fq_t>(acc, sh);
        for (int i = 0; i < 13; i++) x ^= acc.x.v[i] ^ acc.y.v[i] ^ acc.zz.v[i] ^ acc.zzz.v[i];
        __syncthreads();
    }
    out[blockIdx.x * blockDim.x + tid] = x;
}
template <class K>
static double time_block(K kern, int blocks, int threads, int iters, uint32_t* d_out, hipEvent_t e0, hipEvent_t e1) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 2, 0);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 0);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / iters;
}

// The arithmetic of one radix-2^8 NTT pass on a pair of elements, register resident (no LDS, no memory): 8 lazy DIF butterflies
// (ntt.hip.h::lazy_butterfly: add_lazy, sub_lazy + 2^s r, mul_lazy by a twiddle) followed by the closing product and the
// conditional subtraction of both outputs.  Chip-wide rate in G butterflies/s = the arithmetic ceiling of ntt_pass_kernel_v2.
__device__ __forceinline__ fr_t seed_fr(uint32_t s) {
    fr_t a;
    for (int i = 0; i < 9; i++) {
        s = s * 1664525u + 1013904223u;
        a.v[i] = s & LIMB_MASK;
    }
    a.v[8] &= 0x000fffffu;  // < r
    return a;
}
template <int MINW>
__global__ void __launch_bounds__(256, MINW) k_ntt_pass(uint32_t* out, int iters, int) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    fr_t u = seed_fr(tid * 3 + 1), v = seed_fr(tid * 7 + 5);
    const fr_t w = seed_fr(tid * 11 + 3), wc = seed_fr(tid * 13 + 7);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int st = 0; st < 8; st++) {
            uint32_t kp[9];
            fr_t::mod_shl(kp, st);
            const fr_t sum = fr_t::add_lazy(u, v);
            v = fr_t::sub_lazy(u, v, kp).mul_lazy(w);
            u = sum;
        }
        u = u.mul_lazy(wc).reduce_lazy();
        v = v.mul_lazy(wc).reduce_lazy();
    }
    uint32_t x = 0;
    for (int i = 0; i < 9; i++) x ^= u.v[i] ^ v.v[i];
    out[tid] = x;
}
template <class K>
static double run(K kern, int blocks, int iters, int op, uint32_t* d_out, hipEvent_t e0, hipEvent_t e1) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 4, op);  // warm-up
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, iters, op);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)blocks * 256 * iters / (ms * 1e-3) * 1e-9;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("ecbench on %s, %d CUs: G operations/s chip-wide at k resident waves per SIMD (grid = CUs * k blocks of 256, x4 rounds)\n", prop.gcnArchName, cus);
    const char* fnames[4] = {"Fq mul", "Fq sqr", "Fq add+sub pair", "Fq diff_of_products"};
    printf("%-24s %10s %10s %10s %10s\n", "op", "k=1", "k=2", "k=3", "k=4");
#define ROW_F(OP, IT) printf("%-24s %10.2f %10.2f %10.2f %10.2f\n", fnames[OP], run(k_field<1, OP>, cus * 1 * 4, IT, OP, d_out, e0, e1), \
                             run(k_field<2, OP>, cus * 2 * 4, IT, OP, d_out, e0, e1), run(k_field<3, OP>, cus * 3 * 4, IT, OP, d_out, e0, e1), \
                             run(k_field<4, OP>, cus * 4 * 4, IT, OP, d_out, e0, e1));
    ROW_F(0, 2000) ROW_F(1, 2000) ROW_F(2, 2000) ROW_F(3, 2000)
    const char* pnames[3] = {"G1 madd (xyzz+affine)", "G1 add (xyzz+xyzz)", "G1 dbl (xyzz)"};
#define ROW_P(OP, IT) printf("%-24s %10.3f %10.3f %10.3f %10.3f\n", pnames[OP], run(k_point<1, OP>, cus * 1 * 4, IT, OP, d_out, e0, e1), \
                             run(k_point<2, OP>, cus * 2 * 4, IT, OP, d_out, e0, e1), run(k_point<3, OP>, cus * 3 * 4, IT, OP, d_out, e0, e1), \
                             run(k_point<4, OP>, cus * 4 * 4, IT, OP, d_out, e0, e1));
    ROW_P(0, 400) ROW_P(1, 400) ROW_P(2, 400)
    const char* lnames[4] = {"lazy G1 madd (ffl.hip.h)", "lazy Fq mul", "lazy Fq sqr (+sub, norm)", "lazy Fq diff_of_products"};
#define ROW_L(OP, IT) printf("%-24s %10.3f %10.3f %10.3f %10.3f\n", lnames[OP], run(k_lazy<1, OP>, cus * 1 * 4, IT, OP, d_out, e0, e1), \
                             run(k_lazy<2, OP>, cus * 2 * 4, IT, OP, d_out, e0, e1), run(k_lazy<3, OP>, cus * 3 * 4, IT, OP, d_out, e0, e1), \
                             run(k_lazy<4, OP>, cus * 4 * 4, IT, OP, d_out, e0, e1));
    ROW_L(0, 400) ROW_L(1, 2000) ROW_L(2, 2000) ROW_L(3, 2000)
    // single-wave latency of one dependent group operation (the tails of a small MSM): one block of 64 threads
#define LAT(OP) { const int it = 200; hipLaunchKernelGGL((k_point<1, OP>), dim3(1), dim3(64), 0, 0, d_out, 4, OP); CHECK(hipEventRecord(e0)); \
        hipLaunchKernelGGL((k_point<1, OP>), dim3(1), dim3(64), 0, 0, d_out, it, OP); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); \
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); printf("single wave, dependent chain: %-24s %.2f us per operation\n", pnames[OP], ms * 1e3 / it); }
    LAT(0) LAT(1) LAT(2)
    printf("single wave, XYZZ additions, code locality: 1 inlined site in a loop %.2f us/add | 2 sites %.2f | 4 sites %.2f | 8 sites %.2f | one out-of-line function %.2f\n",
           time_single(k_sites<1>, 64, 0, d_out, e0, e1) / 64, time_single(k_sites<2>, 32, 0, d_out, e0, e1) / 64, time_single(k_sites<4>, 16, 0, d_out, e0, e1) / 64,
           time_single(k_sites<8>, 8, 0, d_out, e0, e1) / 64, time_single(k_sites<1>, 64, 1, d_out, e0, e1) / 64);
    {
        const int it = 200;
        double r[4] = {run(k_ntt_pass<1>, cus * 1 * 4, it, 0, d_out, e0, e1), run(k_ntt_pass<2>, cus * 2 * 4, it, 0, d_out, e0, e1),
                       run(k_ntt_pass<3>, cus * 3 * 4, it, 0, d_out, e0, e1), run(k_ntt_pass<4>, cus * 4 * 4, it, 0, d_out, e0, e1)};
        // one iteration = 8 butterflies + 2 closing products on a pair = one pass over two elements
        printf("%-24s %10.2f %10.2f %10.2f %10.2f   (G element-passes/s: 8 lazy butterflies + 2 closing products per pair of Fr elements)\n",
               "Fr NTT pass arithmetic", 2 * r[0], 2 * r[1], 2 * r[2], 2 * r[3]);
    }
    printf("single wave, quad-cooperative XYZZ addition (4 lanes share the 14 products): %.2f us per addition\n", time_block(k_quad, 1, 64, 200, d_out, e0, e1));
    printf("block_sum (1 plain level + quad levels): 64 threads %.1f us | 256 threads %.1f us | 384 blocks x 256 threads %.1f us | 512 blocks x 256 threads %.1f us\n",
           time_block(k_block_sum, 1, 64, 20, d_out, e0, e1), time_block(k_block_sum, 1, 256, 20, d_out, e0, e1), time_block(k_block_sum, 384, 256, 20, d_out, e0, e1),
           time_block(k_block_sum, 512, 256, 20, d_out, e0, e1));
    return 0;
}
