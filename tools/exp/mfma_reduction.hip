// mfma_reduction.hip - can the constant-operand half of a Montgomery product use the matrix unit?  (round-4 review, item 9; timeboxed)
//
// The accumulate kernel is bound by the issue rate of the 64-bit multiply-add (v_mad_i64_i32, quarter rate), and ~45 % of its multiply-adds
// are the  m * q  half of the reductions: 14 quotient digits m_k times the CONSTANT modulus q (ffl.hip.h: 168 + 14 multiply-adds per
// product).  A product by a constant is a Toeplitz(q) x [digit vectors] GEMM, which v_mfma_i32_16x16x64_i8 could run.  This file measures
// what that costs on gfx950 INCLUDING everything the kernel would have to do around the MFMAs, for the 64 values of one wave:
//
//   (A) the multiply-add form the kernel uses today:  P = m * q as 14 x 13 column multiply-adds  (182 v_mad_i64_i32) + carry normalisation
//   (B) the matrix form:  m (14 limbs of LB bits) -> 7-bit digits (signed i8 holds 0 .. 127; 8-bit digits do not fit the signed operand)
//       -> LDS transposition into the MFMA's B-operand layout -> 7 x 4 MFMAs against the Toeplitz matrix of q's digits (112 digit positions
//       x 64 values) -> LDS transposition of the 112 column sums back to "one value per lane" -> recombination into LB-bit limb columns
//       + the same carry normalisation.  LB = 29 is the kernel's limb width; LB = 28 is the BEST case for the matrix form (a limb is exactly
//       four 7-bit digits: extraction and recombination are aligned).
//   (C) what (B) needs in addition and (A) gets for free: the quotient digits UP FRONT.  The interleaved reduction reads m_k off the running
//       column (one AND); a GEMM needs all of m before it starts, i.e. the separate product  m = T_low * (-1/q) mod 2^(14 LB):
//       a 14 x 14 low-half product = 105 multiply-adds.
//
// (B) is checked against (A) on random inputs before anything is timed.  Output: cycles per wave-iteration (s_memtime) for A, B(29), B(28), C
// and the verdict  B + C  vs  A.   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/mfma_reduction.hip -o tools/exp/mfma_reduction
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                     \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(2);                                                              \
        }                                                                         \
    } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

// q = the BLS12-377 base field modulus (fq.rs:111-150) as 48 little-endian bytes
__constant__ uint8_t QB[48];
static const uint64_t Q64[6] = {0x8508c00000000001ull, 0x170b5d4430000000ull, 0x1ef3622fba094800ull, 0x1a22d9f300f5138full, 0xc63b05c06ca1493bull, 0x01ae3a4617c510eaull};

template <int LB>
struct cfg {
    static constexpr int NM = 14;                          // limbs of m (the quotient digits of a 14-step reduction)
    static constexpr int NQ = (377 + LB - 1) / LB;         // limbs of q: 13 (LB = 29) or 14 (LB = 28)
    static constexpr int NP = NM + NQ;                     // limb columns of the product
    static constexpr int DM = (NM * LB + 6) / 7;           // 7-bit digits of m: 58 / 56
    static constexpr int DQ = 54;                          // 7-bit digits of q (377 bits)
    static constexpr uint32_t MASK = (1u << LB) - 1;
};

template <int LB>
__device__ __forceinline__ void q_limbs(int32_t* q) {  // q as NQ limbs of LB bits
    for (int i = 0; i < cfg<LB>::NQ; i++) {
        uint64_t v = 0;
        for (int b = 0; b < LB; b++) {
            const int bit = i * LB + b;
            if (bit < 384 && ((QB[bit >> 3] >> (bit & 7)) & 1)) v |= 1ull << b;
        }
        q[i] = (int32_t)v;
    }
}
// carry normalisation of NP signed 64-bit limb columns -> a checksum of the normalised limbs (keeps the compiler honest, compares A with B)
template <int LB>
__device__ __forceinline__ uint32_t normalise_sum(const int64_t* col) {
    int64_t c = 0;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < cfg<LB>::NP; k++) {
        const int64_t x = col[k] + c;
        s = s * 31u + ((uint32_t)x & cfg<LB>::MASK);
        c = x >> LB;
    }
    return s + (uint32_t)c;
}

// (A): 14 x NQ multiply-adds.  `iters` dependent rounds (the checksum perturbs m) so that nothing is hoisted.
template <int LB>
__global__ void __launch_bounds__(64) k_mad(const uint32_t* __restrict__ m_in, uint32_t* __restrict__ out, int iters, long long* cycles) {
    typedef cfg<LB> C;
    int32_t q[C::NQ], m[C::NM];
    q_limbs<LB>(q);
    const size_t g = blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < C::NM; i++) m[i] = (int32_t)(m_in[g * C::NM + i] & C::MASK);
    uint32_t s = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        int64_t col[C::NP];
#pragma unroll
        for (int k = 0; k < C::NP; k++) {
            int64_t a = 0;
#pragma unroll
            for (int i = 0; i < C::NM; i++) {
                const int j = k - i;
                if (j >= 0 && j < C::NQ) a += (int64_t)m[i] * q[j];
            }
            col[k] = a;
        }
        s = normalise_sum<LB>(col);
        m[it % C::NM] = (m[it % C::NM] + (int32_t)(s & 7)) & (int32_t)C::MASK;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[g] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
// (C): the quotient digits up front: the low NM limbs of T_low * qinv (a 14 x 14 low-half product: 105 multiply-adds) + carries
template <int LB>
__global__ void __launch_bounds__(64) k_quot(const uint32_t* __restrict__ m_in, uint32_t* __restrict__ out, int iters, long long* cycles) {
    typedef cfg<LB> C;
    int32_t qi[C::NM], t[C::NM];
    const size_t g = blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < C::NM; i++) t[i] = (int32_t)(m_in[g * C::NM + i] & C::MASK), qi[i] = (int32_t)((0x9E3779B9u * (i + 1)) & C::MASK);  // any constant: the cost is what counts
    uint32_t s = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        int64_t c = 0;
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < C::NM; k++) {
            int64_t a = c;
#pragma unroll
            for (int i = 0; i <= k; i++) a += (int64_t)t[i] * qi[k - i];
            acc = acc * 31u + ((uint32_t)a & C::MASK);
            c = a >> LB;
        }
        s = acc;
        t[it % C::NM] = (t[it % C::NM] + (int32_t)(s & 7)) & (int32_t)C::MASK;
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[g] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

// (B): the matrix form.  One wave per workgroup; LDS: 64 x 64 digit bytes in, 64 x 112 column sums out.
template <int LB>
__global__ void __launch_bounds__(64) k_mfma(const uint32_t* __restrict__ m_in, uint32_t* __restrict__ out, int iters, long long* cycles) {
    typedef cfg<LB> C;
    __shared__ __attribute__((aligned(16))) uint8_t lds_in[64 * 64];
    __shared__ __attribute__((aligned(16))) int32_t lds_out[64 * 112];
    const int l = threadIdx.x, lo = l & 15, g4 = l >> 4;
    const size_t g = blockIdx.x * 64 + threadIdx.x;
    int32_t m[C::NM];
    for (int i = 0; i < C::NM; i++) m[i] = (int32_t)(m_in[g * C::NM + i] & C::MASK);
    // the Toeplitz operand: A_r holds rows 16 r + lo, k = 16 g4 + byte: q's 7-bit digit (row - k), built once (it is a constant of the kernel)
    v4i A[7];
    for (int r = 0; r < 7; r++) {
        uint32_t w[4] = {0, 0, 0, 0};
        for (int b = 0; b < 16; b++) {
            const int row = 16 * r + lo, k = 16 * g4 + b, d = row - k;
            uint32_t dig = 0;
            if (d >= 0 && d < C::DQ)
                for (int bit = 0; bit < 7; bit++) {
                    const int pos = 7 * d + bit;
                    if (pos < 384 && ((QB[pos >> 3] >> (pos & 7)) & 1)) dig |= 1u << bit;
                }
            w[b >> 2] |= dig << (8 * (b & 3));
        }
        A[r] = (v4i){(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    }
    uint32_t s = 0;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        // 1. m -> 7-bit digits, four per word (64 bytes per value; digits >= DM are zero)
        uint32_t dw[16];
#pragma unroll
        for (int wd = 0; wd < 16; wd++) {
            uint32_t x = 0;
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int d = 4 * wd + b;
                if (d < C::DM) {
                    const int bit = 7 * d, i = bit / LB, sh = bit % LB;
                    uint32_t v = (uint32_t)m[i] >> sh;
                    if (sh + 7 > LB && i + 1 < C::NM) v |= (uint32_t)m[i + 1] << (LB - sh);
                    x |= (v & 127u) << (8 * b);
                }
            }
            dw[wd] = x;
        }
        // 2. transposition: value l's 64 digit bytes -> LDS; B_t = bytes [16 g4, 16 g4 + 16) of value 16 t + lo
        uint4* in4 = (uint4*)lds_in;
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) in4[l * 4 + k4] = make_uint4(dw[4 * k4], dw[4 * k4 + 1], dw[4 * k4 + 2], dw[4 * k4 + 3]);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        v4i B[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint4 u = in4[(16 * t + lo) * 4 + g4];
            B[t] = (v4i){(int)u.x, (int)u.y, (int)u.z, (int)u.w};
        }
        // 3. 7 x 4 MFMAs: D[r][t] = rows 16 r + 4 g4 + reg, column (value) 16 t + lo
        v4i D[7][4];
#pragma unroll
        for (int r = 0; r < 7; r++)
#pragma unroll
            for (int t = 0; t < 4; t++) D[r][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[r], B[t], (v4i){0, 0, 0, 0}, 0, 0, 0);
        // 4. back: value (16 t + lo) gets its rows 16 r + 4 g4 .. + 3
        v4i* out4 = (v4i*)lds_out;
#pragma unroll
        for (int r = 0; r < 7; r++)
#pragma unroll
            for (int t = 0; t < 4; t++) out4[(16 * t + lo) * 28 + 4 * r + g4] = D[r][t];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        // 5. the 112 column sums (radix 2^7, each < 2^21) of value l -> LB-bit limb columns
        int64_t col[C::NP];
#pragma unroll
        for (int k = 0; k < C::NP; k++) col[k] = 0;
#pragma unroll
        for (int q4 = 0; q4 < 28; q4++) {
            const v4i c = out4[l * 28 + q4];
            const int cs[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int d = 4 * q4 + e;
                if (d < C::DM + C::DQ) {
                    const int bit = 7 * d, j = bit / LB, sh = bit % LB;
                    if (j < C::NP) col[j] += (int64_t)cs[e] << sh;  // < 2^21 << 28: 64 bits; the carry normalisation takes any column
                }
            }
        }
        s = normalise_sum<LB>(col);
        m[it % C::NM] = (m[it % C::NM] + (int32_t)(s & 7)) & (int32_t)C::MASK;
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    out[g] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int LB>
static void run(const char* label) {
    typedef cfg<LB> C;
    const int blocks = 2048, n = blocks * 64;
    std::vector<uint32_t> h_m((size_t)n * C::NM);
    uint64_t st = 0x5EED0000 + LB;
    for (auto& w : h_m) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        w = (uint32_t)(st >> 33);
    }
    uint32_t *d_m, *d_a, *d_b, *d_c;
    long long* d_cyc;
    CK(hipMalloc(&d_m, h_m.size() * 4));
    CK(hipMalloc(&d_a, n * 4));
    CK(hipMalloc(&d_b, n * 4));
    CK(hipMalloc(&d_c, n * 4));
    CK(hipMalloc(&d_cyc, 8));
    CK(hipMemcpy(d_m, h_m.data(), h_m.size() * 4, hipMemcpyHostToDevice));
    // correctness: one round of each form on the same inputs
    hipLaunchKernelGGL(k_mad<LB>, dim3(blocks), dim3(64), 0, 0, d_m, d_a, 1, d_cyc);
    hipLaunchKernelGGL(k_mfma<LB>, dim3(blocks), dim3(64), 0, 0, d_m, d_b, 1, d_cyc);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> a(n), b(n);
    CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int i = 0; i < n; i++) bad += a[i] != b[i];
    printf("%s: matrix form vs multiply-add form on %d random values: %s (%zu differ)\n", label, n, bad ? "MISMATCH" : "identical", bad);
    const int iters = 200;
    auto timed = [&](auto kern, uint32_t* d_o, const char* what) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_m, d_o, iters, d_cyc);  // warm-up
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(64), 0, 0, d_m, d_o, iters, d_cyc);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long cyc = 0;
        CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        const double per_value_ns = ms * 1e6 / ((double)n * iters);
        printf("  %-58s %8.3f ms  = %7.3f ns per value (whole chip, %d waves)   wave 0: %lld memtime ticks per round\n", what, ms, per_value_ns, blocks, cyc / iters);
        return per_value_ns;
    };
    const double ta = timed(k_mad<LB>, d_a, "(A) 14 x 13/14 multiply-adds + carries");
    const double tb = timed(k_mfma<LB>, d_b, "(B) digits -> LDS -> 28 MFMA i8 -> LDS -> limb columns + carries");
    const double tc = timed(k_quot<LB>, d_c, "(C) quotient digits up front (105 multiply-adds)");
    printf("  => matrix form (B + C) / multiply-add form (A) = %.2f   (B alone / A = %.2f)\n\n", (tb + tc) / ta, tb / ta);
    (void)hipFree(d_m), (void)hipFree(d_a), (void)hipFree(d_b), (void)hipFree(d_c), (void)hipFree(d_cyc);
}

int main() {
    uint8_t qb[48];
    for (int i = 0; i < 48; i++) qb[i] = (uint8_t)(Q64[i / 8] >> (8 * (i % 8)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(QB), qb, 48));
    run<29>("LB = 29 (the kernel's limbs)");
    run<28>("LB = 28 (a limb = four 7-bit digits: the matrix form's best case)");
    return 0;
}
