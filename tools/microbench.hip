// tools/microbench.hip - gfx950 integer/FP64 instruction-rate probes and Montgomery-product variants.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o tools/microbench
// Prints cycles per wave-instruction (s_memtime based) for 1/2/4 waves per SIMD, and ns/mul for the
// field-multiplication variants.  Used to choose the limb representation (DESIGN.md section "arithmetic").
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

#define REP8(X) X X X X X X X X
#define ITERS 2000

// Each probe: 8 independent register chains, 8x unrolled -> 64 instr per loop iteration.
#define PROBE_KERNEL(NAME, DECL, BODY, SINK)                                                   \
    __global__ void NAME(uint64_t* out, uint32_t seed) {                                       \
        DECL;                                                                                  \
        uint64_t t0 = __builtin_readcyclecounter();                                            \
        for (int it = 0; it < ITERS; it++) { REP8(BODY) }                                      \
        uint64_t t1 = __builtin_readcyclecounter();                                            \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                             \
        if (seed == 0xdeadbeef) out[1 + threadIdx.x] = SINK;                                   \
    }

#define DECL_U32 uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19, b = seed | 1
#define DECL_U64 uint64_t c0 = seed + threadIdx.x, c1 = c0 * 3, c2 = c0 * 5, c3 = c0 * 7, c4 = c0 * 11, c5 = c0 * 13, c6 = c0 * 17, c7 = c0 * 19; uint32_t a = seed * 77 + threadIdx.x, b = seed | 1
#define DECL_F64 double d0 = seed + threadIdx.x, d1 = d0 * 3, d2 = d0 * 5, d3 = d0 * 7, d4 = d0 * 11, d5 = d0 * 13, d6 = d0 * 17, d7 = d0 * 19, e = 1.0000001, f = 0.5

#define B8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define MAD64(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c##i) : "v"(a), "v"(b) : "vcc");
PROBE_KERNEL(k_mad_u64_u32, DECL_U64, B8(MAD64), c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
#define MAD64DEP(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b) : "vcc");
PROBE_KERNEL(k_mad_u64_u32_dep, DECL_U64, B8(MAD64DEP), c0)
#define MADC(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(c##i), "+v"(a##i) : "v"(a), "v"(b) : "vcc");
#define DECL_U64B uint64_t c0 = seed + threadIdx.x, c1 = c0 * 3, c2 = c0 * 5, c3 = c0 * 7, c4 = c0 * 11, c5 = c0 * 13, c6 = c0 * 17, c7 = c0 * 19; uint32_t a = seed * 77 + threadIdx.x, b = seed | 1, a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0
PROBE_KERNEL(k_madc_nop_pair, DECL_U64B, B8(MADC), c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define MADC2(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(c##i), "+v"(a##i) : "v"(a), "v"(b) : "vcc");
PROBE_KERNEL(k_madc_nonop_pair, DECL_U64B, B8(MADC2), c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_mul_lo_u32, DECL_U32, B8(MULLO), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_mul_hi_u32, DECL_U32, B8(MULHI), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_mad_u32_u24, DECL_U32, B8(MAD24), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_mul_hi_u32_u24, DECL_U32, B8(MULHI24), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define ADD32(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_add_u32, DECL_U32, B8(ADD32), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n s_nop 1\n v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
PROBE_KERNEL(k_addco_nop_addc, DECL_U32, B8(ADDCO), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define ADD64(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(c##i) : "v"(c7));
PROBE_KERNEL(k_lshl_add_u64, DECL_U64, ADD64(0) ADD64(1) ADD64(2) ADD64(3) ADD64(4) ADD64(5) ADD64(6) ADD64(0), c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ a ^ b)
#define SHR64(i) asm volatile("v_lshrrev_b64 %0, 3, %0" : "+v"(c##i));
PROBE_KERNEL(k_lshrrev_b64, DECL_U64, B8(SHR64), c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ a ^ b)
#define ALIGNB(i) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_alignbit_b32, DECL_U32, B8(ALIGNB), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##i) : "v"(e), "v"(f));
PROBE_KERNEL(k_fma_f64, DECL_F64, B8(FMA64), (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define FMA64DEP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d0) : "v"(e), "v"(f));
PROBE_KERNEL(k_fma_f64_dep, DECL_F64, B8(FMA64DEP), (uint64_t)(d0))
#define MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d##i) : "v"(e));
PROBE_KERNEL(k_mul_f64, DECL_F64, B8(MUL64), (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7))
#define FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_fma_f32, DECL_U32, B8(FMA32), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define AND32(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_and_b32, DECL_U32, B8(AND32), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define MADI64(i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(c##i) : "v"(a), "v"(b) : "vcc");
PROBE_KERNEL(k_mad_i64_i32, DECL_U64, B8(MADI64), c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7)
#define ASHR(i) asm volatile("v_ashrrev_i32 %0, 29, %0" : "+v"(a##i));
PROBE_KERNEL(k_ashrrev_i32, DECL_U32, B8(ASHR), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b)
#define BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(a##i));
PROBE_KERNEL(k_bfe_u32, DECL_U32, B8(BFE), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ b)
#define ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a##i) : "v"(b));
PROBE_KERNEL(k_add3_u32, DECL_U32, B8(ADD3), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
PROBE_KERNEL(k_cndmask_b32, DECL_U32, B8(CNDM), a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(c##i) : "v"(c7));
PROBE_KERNEL(k_pk_fma_f32, DECL_U64, PKFMA(0) PKFMA(1) PKFMA(2) PKFMA(3) PKFMA(4) PKFMA(5) PKFMA(6) PKFMA(0), c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ a ^ b)

typedef void (*probe_fn)(uint64_t*, uint32_t);
struct Probe { const char* name; probe_fn fn; };

// ---------------------------------------------------------------------------------- field-mul variants
static constexpr uint32_t QMOD[12] = {0x00000001u, 0x8508c000u, 0x30000000u, 0x170b5d44u, 0xba094800u, 0x1ef3622fu,
                                      0x00f5138fu, 0x1a22d9f3u, 0x6ca1493bu, 0xc63b05c0u, 0x17c510eau, 0x01ae3a46u};
struct F12 { uint32_t v[12]; };

// (C) plain C++ CIOS, 32-bit limbs
__host__ __device__ __forceinline__ F12 mul_cios(const F12& a, const F12& b) {
    constexpr int N = 12;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < N; j++) { c = (uint64_t)a.v[j] * b.v[i] + t[j] + c; t[j] = (uint32_t)c; c >>= 32; }
        c += t[N]; t[N] = (uint32_t)c; t[N + 1] = (uint32_t)(c >> 32);
        uint32_t m = 0u - t[0];
        c = (uint64_t)m * QMOD[0] + t[0]; c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) { c = (uint64_t)m * QMOD[j] + t[j] + c; t[j - 1] = (uint32_t)c; c >>= 32; }
        c += t[N]; t[N - 1] = (uint32_t)c; t[N] = t[N + 1] + (uint32_t)(c >> 32);
    }
    F12 d; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)t[i] - QMOD[i] - br; d.v[i] = (uint32_t)x; br = (uint32_t)(x >> 63); }
    F12 r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = br ? t[i] : d.v[i];
    return r;
}
// (A) product scanning with asm mad + addc (NOPS = wait states between the VCC write and read)
template <int NOPS>
__device__ __forceinline__ void madc(uint64_t& acc01, uint32_t& acc2, uint32_t a, uint32_t b) {
    if (NOPS == 2)
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc01), "+v"(acc2) : "v"(a), "v"(b) : "vcc");
    else
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc01), "+v"(acc2) : "v"(a), "v"(b) : "vcc");
}
template <int NOPS>
__device__ __forceinline__ void madck(uint64_t& acc01, uint32_t& acc2, uint32_t a, uint32_t k) {
    if (NOPS == 2)
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc01), "+v"(acc2) : "v"(a), "s"(k) : "vcc");
    else
        asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc01), "+v"(acc2) : "v"(a), "s"(k) : "vcc");
}
template <int NOPS>
__device__ __forceinline__ F12 mul_fips(const F12& a, const F12& b) {
    constexpr int N = 12;
    uint32_t m[N], t[N];
    uint64_t acc01 = 0; uint32_t acc2 = 0;
#pragma unroll
    for (int k = 0; k < 2 * N; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 0 && j < N) madc<NOPS>(acc01, acc2, a.v[i], b.v[j]); }
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 1 && j < N && i < k) madck<NOPS>(acc01, acc2, m[i], QMOD[j]); }
        if (k < N) {
            m[k] = 0u - (uint32_t)acc01;
            uint64_t s = acc01 + m[k]; acc2 += (s < acc01) ? 1u : 0u; acc01 = s;
        } else t[k - N] = (uint32_t)acc01;
        acc01 = (acc01 >> 32) | ((uint64_t)acc2 << 32); acc2 = 0;
    }
    F12 d; uint32_t br = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { uint64_t x = (uint64_t)t[i] - QMOD[i] - br; d.v[i] = (uint32_t)x; br = (uint32_t)(x >> 63); }
    F12 r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = br ? t[i] : d.v[i];
    return r;
}
// (B) 29-bit limbs (13), internal Montgomery radix 2^377, carry-free 64-bit column accumulation.
struct F13 { uint32_t v[13]; };
static constexpr uint32_t M29 = (1u << 29) - 1;
__host__ __device__ constexpr uint32_t q29(int i) {
    // limb i of q in radix 2^29, derived from the 32-bit words
    int bit = 29 * i, w = bit / 32, s = bit % 32;
    uint64_t lo = QMOD[w], hi = (w + 1 < 12) ? QMOD[w + 1] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> s) & M29);
}
__host__ __device__ __forceinline__ F13 mul29(const F13& a, const F13& b) {
    constexpr int N = 13;
    uint32_t m[N], t[N];
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 0 && j < N) acc += (uint64_t)a.v[i] * b.v[j]; }
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 1 && j < N && i < k) acc += (uint64_t)m[i] * q29(j); }
        if (k < N) { m[k] = (0u - (uint32_t)acc) & M29; acc += m[k]; }
        else t[k - N] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    // conditional subtract (signed limb arithmetic, no VCC chains)
    uint32_t d[N]; int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { int32_t x = (int32_t)t[i] - (int32_t)q29(i) + c; d[i] = (uint32_t)x & M29; c = x >> 29; }
    F13 r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = (c < 0) ? t[i] : d[i];
    return r;
}

// (C) 28-bit limbs (14), Montgomery radix 2^392: q / R = 2^-15, so the product of operands < 128 q comes out < 1.001 q
// WITHOUT a conditional subtraction ("almost reduced" arithmetic; DESIGN.md 8, next levers).  +16 % multiply-adds.
struct F14 { uint32_t v[14]; };
static constexpr uint32_t M28 = (1u << 28) - 1;
__host__ __device__ constexpr uint32_t q28(int i) {
    int bit = 28 * i, w = bit / 32, s = bit % 32;
    uint64_t lo = (w < 12) ? QMOD[w] : 0, hi = (w + 1 < 12) ? QMOD[w + 1] : 0;
    return (uint32_t)(((lo | (hi << 32)) >> s) & M28);
}
__host__ __device__ __forceinline__ F14 mul28_lazy(const F14& a, const F14& b) {
    constexpr int N = 14;
    uint32_t m[N];
    F14 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 0 && j < N) acc += (uint64_t)a.v[i] * b.v[j]; }
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 1 && j < N && i < k) acc += (uint64_t)m[i] * q28(j); }
        if (k < N) { m[k] = (0u - (uint32_t)acc) & M28; acc += m[k]; }
        else r.v[k - N] = (k == 2 * N - 1) ? (uint32_t)acc : ((uint32_t)acc & M28);
        acc >>= 28;
    }
    return r;
}
__global__ void k_mulbench28(F14* out, const F14* in, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    F14 a = in[tid & 1023], b = in[(tid + 1) & 1023];
    for (int it = 0; it < iters; it++) a = mul28_lazy(a, b);
    out[tid] = a;
}
// (B') the 29-bit product without its conditional subtraction (timing only: isolates what that step costs)
__host__ __device__ __forceinline__ F13 mul29_nosub(const F13& a, const F13& b) {
    constexpr int N = 13;
    uint32_t m[N];
    F13 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 2 * N; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 0 && j < N) acc += (uint64_t)a.v[i] * b.v[j]; }
#pragma unroll
        for (int i = 0; i < N; i++) { int j = k - i; if (j >= 1 && j < N && i < k) acc += (uint64_t)m[i] * q29(j); }
        if (k < N) { m[k] = (0u - (uint32_t)acc) & M29; acc += m[k]; }
        else r.v[k - N] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    return r;
}
__global__ void k_mulbench29_nosub(F13* out, const F13* in, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    F13 a = in[tid & 1023], b = in[(tid + 1) & 1023];
    for (int it = 0; it < iters; it++) a = mul29_nosub(a, b);
    out[tid] = a;
}

template <int V>
__global__ void k_mulbench(F12* out, const F12* in, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    F12 a = in[tid & 1023], b = in[(tid + 1) & 1023];
    for (int it = 0; it < iters; it++) {
        if (V == 0) a = mul_cios(a, b);
        if (V == 1) a = mul_fips<2>(a, b);
        if (V == 2) a = mul_fips<0>(a, b);
    }
    out[tid] = a;
}
__global__ void k_mulbench29(F13* out, const F13* in, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    F13 a = in[tid & 1023], b = in[(tid + 1) & 1023];
    for (int it = 0; it < iters; it++) a = mul29(a, b);
    out[tid] = a;
}

// --------------------------------------------------------------------------------- memory probes
__global__ void k_copy16(const uint4* __restrict__ in, uint4* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) out[i] = in[i];
}
__global__ void k_atomic_hist(const uint32_t* __restrict__ keys, uint32_t* hist, size_t n, uint32_t mask) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) atomicAdd(&hist[keys[i] & mask], 1u);
}
__global__ void k_fill_keys(uint32_t* keys, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) { uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 32; keys[i] = (uint32_t)z; }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    uint64_t* d_out;
    CHECK(hipMalloc(&d_out, 8 * 2048));
    Probe probes[] = {{"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_u64_u32(dep chain)", k_mad_u64_u32_dep},
                      {"mad+nop1+addc pair", k_madc_nop_pair}, {"mad+addc pair (no nop)", k_madc_nonop_pair},
                      {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u32_u24", k_mad_u32_u24},
                      {"v_mul_hi_u32_u24", k_mul_hi_u32_u24}, {"v_add_u32", k_add_u32}, {"add_co+nop1+addc pair", k_addco_nop_addc},
                      {"v_lshl_add_u64", k_lshl_add_u64}, {"v_lshrrev_b64", k_lshrrev_b64}, {"v_alignbit_b32", k_alignbit_b32},
                      {"v_fma_f64", k_fma_f64}, {"v_fma_f64(dep chain)", k_fma_f64_dep}, {"v_mul_f64", k_mul_f64}, {"v_fma_f32", k_fma_f32}};
    printf("%-28s %10s %10s %10s   (cycles per wave-instruction on one SIMD; block = 256/512/1024 threads on one CU)\n", "instr", "1w/SIMD", "2w/SIMD", "4w/SIMD");
    for (auto& p : probes) {
        double res[3];
        int bs[3] = {256, 512, 1024};
        for (int c = 0; c < 3; c++) {
            hipLaunchKernelGGL(p.fn, dim3(1), dim3(bs[c]), 0, 0, d_out, 12345u);
            CHECK(hipDeviceSynchronize());
            uint64_t cyc;
            CHECK(hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost));
            int waves_per_simd = bs[c] / 256;
            res[c] = (double)cyc / (ITERS * 64.0) / waves_per_simd;  // per wave-instruction per SIMD
        }
        printf("%-28s %10.2f %10.2f %10.2f\n", p.name, res[0], res[1], res[2]);
    }
    // ---- wall-clock, whole-chip calibration: wave-instructions per second with every CU loaded at 1/2/4/8 waves per SIMD.
    // No cycle counter involved: HIP events around a grid of (CUs * k) blocks of 256 threads, each lane issuing ITERS * 64
    // instructions.  "cyc/SIMD" = 1024 SIMDs * 2.4 GHz * time / wave-instructions (nominal clock; DVFS may run lower).
    {
        hipEvent_t c0, c1; CHECK(hipEventCreate(&c0)); CHECK(hipEventCreate(&c1));
        Probe cal[] = {{"v_fma_f32", k_fma_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_fma_f64", k_fma_f64}, {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32},
                       {"v_add3_u32", k_add3_u32}, {"v_ashrrev_i32", k_ashrrev_i32}, {"v_bfe_u32", k_bfe_u32},
                       {"v_alignbit_b32", k_alignbit_b32}, {"v_lshrrev_b64", k_lshrrev_b64}, {"v_lshl_add_u64", k_lshl_add_u64},
                       {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u32_u24", k_mad_u32_u24},
                       {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mad_u64_u32(dep chain)", k_mad_u64_u32_dep}, {"v_mad_i64_i32", k_mad_i64_i32}};
        printf("\nwall-clock whole-chip calibration (%d CUs): G wave-instr/s [cyc per wave-instr per SIMD at 2.4 GHz]\n", prop.multiProcessorCount);
        printf("%-28s %22s %22s %22s %22s\n", "instr", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD");
        for (auto& p : cal) {
            printf("%-28s", p.name);
            for (int k = 1; k <= 8; k *= 2) {
                const int grid = prop.multiProcessorCount * k;
                hipLaunchKernelGGL(p.fn, dim3(grid), dim3(256), 0, 0, d_out, 12345u);  // warm-up
                CHECK(hipEventRecord(c0));
                hipLaunchKernelGGL(p.fn, dim3(grid), dim3(256), 0, 0, d_out, 12345u);
                CHECK(hipEventRecord(c1)); CHECK(hipEventSynchronize(c1));
                float ms; CHECK(hipEventElapsedTime(&ms, c0, c1));
                const double winst = (double)grid * 4 * ITERS * 64.0;
                const double cyc = 4.0 * prop.multiProcessorCount * 2.4e9 * (ms * 1e-3) / winst;
                printf("   %9.1f G [%5.2f]", winst / (ms * 1e-3) * 1e-9, cyc);
            }
            printf("\n");
        }
        printf("sanity: v_fma_f64 at saturation x 64 lanes x 2 flop = vector FP64 TF/s (spec 78.6); v_fma_f32 -> FP32 (spec 157.3 incl. packed)\n\n");
    }
    // ---- field multiplication variants
    std::vector<F12> h(1024);
    uint64_t s = 88172645463325252ull;
    for (auto& e : h) { for (int i = 0; i < 12; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; e.v[i] = (uint32_t)s; } e.v[11] &= 0x00ffffff; }
    F12 *d_in, *d_o; F13 *d_in13, *d_o13;
    const int grid = 256 * 8, block = 256, iters = 200;
    CHECK(hipMalloc(&d_in, sizeof(F12) * 1024)); CHECK(hipMalloc(&d_o, sizeof(F12) * grid * block));
    CHECK(hipMalloc(&d_in13, sizeof(F13) * 1024)); CHECK(hipMalloc(&d_o13, sizeof(F13) * grid * block));
    CHECK(hipMemcpy(d_in, h.data(), sizeof(F12) * 1024, hipMemcpyHostToDevice));
    std::vector<F13> h13(1024);
    for (int e = 0; e < 1024; e++) for (int i = 0; i < 13; i++) { int bit = 29 * i, w = bit / 32, sh = bit % 32; uint64_t lo = h[e].v[w], hi = (w + 1 < 12) ? h[e].v[w + 1] : 0; h13[e].v[i] = (uint32_t)(((lo | (hi << 32)) >> sh) & M29); }
    CHECK(hipMemcpy(d_in13, h13.data(), sizeof(F13) * 1024, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<F12> ref(grid * block), got(grid * block);
    const char* names[3] = {"cios32 plain C++", "fips32 asm mad+nop1+addc", "fips32 asm mad+addc (no nop)"};
    for (int v = 0; v < 3; v++) {
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            if (v == 0) hipLaunchKernelGGL(k_mulbench<0>, dim3(grid), dim3(block), 0, 0, d_o, d_in, iters);
            if (v == 1) hipLaunchKernelGGL(k_mulbench<1>, dim3(grid), dim3(block), 0, 0, d_o, d_in, iters);
            if (v == 2) hipLaunchKernelGGL(k_mulbench<2>, dim3(grid), dim3(block), 0, 0, d_o, d_in, iters);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(got.data(), d_o, sizeof(F12) * grid * block, hipMemcpyDeviceToHost));
        if (v == 0) {
            ref = got;
            // host check of the plain variant on a few lanes
            int bad = 0;
            for (int t = 0; t < 64; t++) { F12 a = h[t & 1023], b = h[(t + 1) & 1023]; for (int it = 0; it < iters; it++) a = mul_cios(a, b); if (memcmp(&a, &got[t], sizeof a)) bad++; }
            printf("cios32 device-vs-host mismatches: %d / 64\n", bad);
        }
        size_t mism = 0; for (size_t i = 0; i < got.size(); i++) if (memcmp(&got[i], &ref[i], sizeof(F12))) mism++;
        double muls = (double)grid * block * iters;
        printf("%-34s %8.3f ms  %7.2f Gmul/s  mismatches vs cios: %zu\n", names[v], ms, muls / ms * 1e-6, mism);
    }
    {
        for (int rep = 0; rep < 2; rep++) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mulbench29, dim3(grid), dim3(block), 0, 0, d_o13, d_in13, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<F13> g13(64);
        CHECK(hipMemcpy(g13.data(), d_o13, sizeof(F13) * 64, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int t = 0; t < 64; t++) { F13 a = h13[t & 1023], b = h13[(t + 1) & 1023]; for (int it = 0; it < iters; it++) a = mul29(a, b); if (memcmp(&a, &g13[t], sizeof a)) bad++; }
        printf("%-34s %8.3f ms  %7.2f Gmul/s  device-vs-host mismatches: %d / 64\n", "mont29 (13x29-bit) plain C++", ms, (double)grid * block * iters / ms * 1e-6, bad);
    }
    {
        for (int rep = 0; rep < 2; rep++) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mulbench29_nosub, dim3(grid), dim3(block), 0, 0, d_o13, d_in13, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-34s %8.3f ms  %7.2f Gmul/s  (timing only)\n", "mont29 without the cond. subtract", ms, (double)grid * block * iters / ms * 1e-6);
        // 14 x 28-bit limbs, lazy: operands = the 13 x 29 inputs re-sliced (any value < 2^392 is a valid operand)
        std::vector<F14> h14(1024);
        for (int t = 0; t < 1024; t++) {
            for (int i = 0; i < 14; i++) h14[t].v[i] = 0;
            for (int bit = 0; bit < 377; bit++) if ((h13[t].v[bit / 29] >> (bit % 29)) & 1) h14[t].v[bit / 28] |= 1u << (bit % 28);
        }
        F14 *d_in14, *d_o14;
        CHECK(hipMalloc(&d_in14, sizeof(F14) * 1024)); CHECK(hipMalloc(&d_o14, sizeof(F14) * (size_t)grid * block));
        CHECK(hipMemcpy(d_in14, h14.data(), sizeof(F14) * 1024, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; rep++) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mulbench28, dim3(grid), dim3(block), 0, 0, d_o14, d_in14, iters); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); }
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<F14> g14(64);
        CHECK(hipMemcpy(g14.data(), d_o14, sizeof(F14) * 64, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int t = 0; t < 64; t++) { F14 a = h14[t & 1023], b = h14[(t + 1) & 1023]; for (int it = 0; it < iters; it++) a = mul28_lazy(a, b); if (memcmp(&a, &g14[t], sizeof a)) bad++; }
        printf("%-34s %8.3f ms  %7.2f Gmul/s  device-vs-host mismatches: %d / 64\n", "mont28 (14x28-bit) lazy, no sub", ms, (double)grid * block * iters / ms * 1e-6, bad);
    }
    // ---- memory probes
    {
        size_t n = (size_t)1 << 26;  // 1 GiB of uint4
        uint4 *a, *b; CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16)); CHECK(hipMemset(a, 1, n * 16));
        for (int rep = 0; rep < 3; rep++) { CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_copy16, dim3(256 * 16), dim3(256), 0, 0, a, b, n); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("copy 1 GiB (uint4): %.3f ms  -> %.2f TB/s (read+write)\n", ms, 2.0 * n * 16 / ms * 1e-9);
        size_t nk = (size_t)1 << 28; uint32_t* keys = (uint32_t*)a; uint32_t* hist = (uint32_t*)b;
        hipLaunchKernelGGL(k_fill_keys, dim3(256 * 16), dim3(256), 0, 0, keys, nk);
        for (uint32_t bits : {15u, 19u, 23u}) {
            CHECK(hipMemset(hist, 0, (size_t)4 << bits));
            CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_atomic_hist, dim3(256 * 16), dim3(256), 0, 0, keys, hist, nk, (1u << bits) - 1); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("global atomicAdd histogram, 2^28 keys into 2^%u bins: %.3f ms -> %.2f Gatomics/s\n", bits, ms, nk / ms * 1e-6);
        }
    }
    return 0;
}
