// tools/exp/madchain.hip - does pinning the column carry to the multiply-add chain pay?  Three builds of the lazy Fq product
// (ffl.hip.h fql_t::mul) in a register-resident dependent chain, whole chip, 2 waves per SIMD:
//   -DVARIANT=0  the compiler's own schedule (operand scanning over several column accumulators + one 64-bit add per column)
//   -DVARIANT=1  the first product of every column is an asm multiply-add whose addend is the carry
//   -DVARIANT=2  every multiply-add of a column is asm, one chain per column (the compiler puts s_nop 0 after each statement)
//   -DVARIANT=3  first product asm + the column sum made opaque around the quotient terms
// Measured on MI355X (round 3): 85.9 / 84.1 / 85.0 / 80.5 G products/s - the compiler's schedule stands; the extra 64-bit add per
// column costs no more than the s_nop 0 the compiler puts after every asm statement, and the multiply-adds run at 0.82 of the
// chip's issue rate either way.  Kept as the record of a negative result.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVARIANT=k tools/exp/madchain.hip -o tools/exp/madchain_k
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <utility>
#include "../../snarkvm_amd/csrc/ffl.hip.h"
namespace sv {
// Compile-time loops: the column routines below index limbs with constants the FRONT END knows (integral_constant), not with
// loop variables the optimiser has to unroll and fold - which it stops doing early enough once a loop body holds an asm statement.
template <int... Is, class Fn>
SV_HD void static_for_impl(std::integer_sequence<int, Is...>, Fn&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class Fn>
SV_HD void static_for(Fn&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}
// c + a * b as ONE multiply-add whose addend is c.  Product-scanning columns start with the carry of the previous column; left to
// itself the compiler starts every column from zero (to overlap it with the end of the previous one) and adds the carry with an
// extra 64-bit addition per column - 27 v_lshl_add_u64 per lazy Fq product, 18 per Fr product - although a wave issues its VALU
// instructions in order, one at a time, so the overlap buys nothing.  The asm statement pins the carry to the multiply-add.
SV_HD uint64_t mad_u64_carry(uint32_t a, uint32_t b, uint64_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SV_NO_MAD_CARRY)
    uint64_t co;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(co) : "v"(a), "v"(b));
    return c;
#else
    return c + (uint64_t)a * b;
#endif
}
// the same with a wave-uniform constant as the second factor (a limb of the modulus: scalar register or literal)
SV_HD int64_t mad_i64_carry_s(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SV_NO_MAD_CARRY)
    uint64_t co;  // the instruction's carry-out operand: never read
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(co) : "v"(a), "s"(b));
    return c;
#else
    return c + (int64_t)a * b;
#endif
}
SV_HD int64_t mad_i64_carry(int32_t a, int32_t b, int64_t c) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SV_NO_MAD_CARRY)
    uint64_t co;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(c), "=s"(co) : "v"(a), "v"(b));
    return c;
#else
    return c + (int64_t)a * b;
#endif
}
}  // namespace sv
using namespace sv;
#ifndef VARIANT
#define VARIANT 0
#endif
static constexpr int N = 13, STEPS = 14;
static constexpr uint32_t MASK = (1u << 29) - 1;

template <int K, int I>
struct col_ab {  // the a*b products of column K from limb I on, as one asm chain
    static __device__ __forceinline__ void run(int64_t& acc, const fql_t& a, const fql_t& b) {
        if constexpr (I < N) {
            constexpr int j = K - I;
            if constexpr (j >= 0 && j < N) {
                uint64_t co;
                asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(co) : "v"(a.v[I]), "v"(b.v[j]));
            }
            col_ab<K, I + 1>::run(acc, a, b);
        }
    }
};

__device__ __forceinline__ fql_t mul_v(const fql_t& a, const fql_t& b) {
#if VARIANT == 0
    return fql_t::mul(a, b);
#else
    uint32_t m[STEPS];
    fql_t r;
    int64_t acc = 0;
    static_for<N + STEPS>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#if VARIANT == 1
        constexpr int i0 = k < N ? 0 : k - N + 1;
        if constexpr (k > 0 && i0 < N) acc = mad_i64_carry(a.v[i0], b.v[k - i0], acc);
        static_for<N>([&](auto ic) {
            constexpr int i = decltype(ic)::value, j = k - i;
            if constexpr (j >= 0 && j < N && !(k > 0 && i == i0)) acc += (int64_t)a.v[i] * b.v[j];
        });
        static_for<STEPS>([&](auto ic) {
            constexpr int i = decltype(ic)::value, j = k - i;
            if constexpr (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FqL::MOD[j];
        });
#elif VARIANT == 2
        static_for<N>([&](auto ic) {
            constexpr int i = decltype(ic)::value, j = k - i;
            if constexpr (j >= 0 && j < N) acc = mad_i64_carry(a.v[i], b.v[j], acc);
        });
        static_for<STEPS>([&](auto ic) {
            constexpr int i = decltype(ic)::value, j = k - i;
            if constexpr (j >= 1 && j < N && i < k) acc = mad_i64_carry_s((int32_t)m[i], -FqL::MOD[j], acc);
        });
#else
        // VARIANT 3: the a*b products as compiler code started from the carry through ONE asm (the first product), the
        // quotient products as compiler code, but the column sum made opaque BEFORE the quotient terms so that the compiler
        // cannot pre-accumulate them in a second register pair
        constexpr int i0 = k < N ? 0 : k - N + 1;
        if constexpr (k > 0 && i0 < N) acc = mad_i64_carry(a.v[i0], b.v[k - i0], acc);
        static_for<N>([&](auto ic) {
            constexpr int i = decltype(ic)::value, j = k - i;
            if constexpr (j >= 0 && j < N && !(k > 0 && i == i0)) acc += (int64_t)a.v[i] * b.v[j];
        });
        asm("" : "+v"(acc));
        static_for<STEPS>([&](auto ic) {
            constexpr int i = decltype(ic)::value, j = k - i;
            if constexpr (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FqL::MOD[j];
        });
        asm("" : "+v"(acc));
#endif
        if constexpr (k < STEPS) m[k] = (uint32_t)acc & MASK;
        else r.v[k - STEPS] = (k == N + STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
        acc >>= 29;
    });
    SV_OPAQUE_13(r.v);
    return r;
#endif
}

__global__ void __launch_bounds__(256, 2) k_mul(int32_t* out, int iters) {
    fql_t a, b;
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int i = 0; i < 13; i++) a.v[i] = (t * 2654435761u + i * 40503u) & MASK, b.v[i] = (t * 40503u + i * 2654435761u) & MASK;
    for (int it = 0; it < iters; it++) {
        a = mul_v(a, b);
        b = mul_v(b, a);
    }
    int32_t s = 0;
    for (int i = 0; i < 13; i++) s ^= a.v[i] + 3 * b.v[i] + (i << 20);
    out[t] = s;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int blocks = prop.multiProcessorCount * 2 * 4, iters = 4000;
    int32_t* d;
    hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k_mul, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mul, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    int32_t h[4];
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("VARIANT %d: %.2f G lazy Fq products/s (checksum %08x %08x)\n", VARIANT, (double)blocks * 256 * iters * 2 / (ms * 1e-3) * 1e-9, h[0], h[1]);
    return 0;
}
