// group.hip.h - the two setup-time group operations of SURVEY.md §8f N4, built from the MSM engine's point arithmetic:
//
//   FixedBase::msm (algorithms/src/msm/fixed_base.rs:33-97): out_i = v_i * g for one base g and many scalars, by a window
//       table of multiples of g (`get_window_table`) and one table lookup + addition per window (`windowed_mul`).
//       Used by the universal setup to build the powers beta^i G.
//   EvaluationDomain::ifft over group elements (`UniversalParams::lagrange_basis`, polycommit/kzg10/data_structures.rs:68-72:
//       `domain.ifft(powers_of_beta_g as projective)`): radix-2 transform whose butterflies add / subtract points and
//       multiply them by Fr twiddles (a full scalar multiplication each), then the 1/n scaling (domain.rs:177-192).
//
// Points travel between kernels as XYZZ records (192 B); the API converts from / to the reference's Jacobian memory image.
#pragma once
#include "ec.hip.h"
#include "ff.hip.h"
#include "msm.hip.h"  // quad_add, quad_bcast, select_field / select_point

namespace sv {

// ---- scalar multiplication by a 256-bit canonical integer (MSB-first double-and-add; scalars are public) ----------
__device__ inline g1_xyzz_t g1_mul_words(const g1_xyzz_t& p, const uint32_t* k) {
    g1_xyzz_t acc = g1_xyzz_t::inf();
    int top = 255;
    while (top >= 0 && !((k[top >> 5] >> (top & 31)) & 1)) top--;
    for (int bit = top; bit >= 0; bit--) {
        acc = acc.dbl();
        if ((k[bit >> 5] >> (bit & 31)) & 1) acc.add(p);
    }
    return acc;
}
__device__ __forceinline__ g1_xyzz_t g1_neg(const g1_xyzz_t& p) { return {p.x, p.y.neg(), p.zz, p.zzz}; }

// Jacobian memory image (144 B, R = 2^384) <-> XYZZ record
static __global__ void g1_jac_to_xyzz_kernel(const uint32_t* __restrict__ in, g1_xyzz_mem_t* __restrict__ out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = in + 36 * i;
    const g1_jac_t j = {fq_t::from_raw_words(src), fq_t::from_raw_words(src + 12), fq_t::from_raw_words(src + 24)};
    g1_store_xyzz(&out[i], g1_xyzz_t::from_jacobian(j));
}
static __global__ void g1_xyzz_to_jac_kernel(const g1_xyzz_mem_t* __restrict__ in, uint32_t* __restrict__ out, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const g1_jac_t j = g1_load_xyzz(&in[i]).to_jacobian();
    uint32_t* dst = out + 36 * i;
    j.x.to_raw_words(dst);
    j.y.to_raw_words(dst + 12);
    j.z.to_raw_words(dst + 24);
}

// ---- FixedBase ----------------------------------------------------------------------------------------------------
static constexpr int FIXED_WINDOW = 8;                                  // bits per window of the device table
static constexpr int FIXED_OUTER = (253 + FIXED_WINDOW - 1) / FIXED_WINDOW;  // windows over the 253-bit scalar field
// table[outer][inner] = inner * 2^(FIXED_WINDOW * outer) * g, affine (fixed_base.rs:42-68); thread (outer, inner)
static __global__ void __launch_bounds__(256) g1_fixed_table_kernel(g1_aff_mem_t g_mem, g1_aff_mem_t* __restrict__ table) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint32_t)FIXED_OUTER << FIXED_WINDOW) return;
    const uint32_t outer = t >> FIXED_WINDOW, inner = t & ((1u << FIXED_WINDOW) - 1);
    g1_aff_t q = g1_aff_t::inf();
    const g1_aff_t g = g1_load_aff(&g_mem);
    if (inner && !g.is_inf()) {
        g1_xyzz_t base = g1_xyzz_t::from_affine(g);
        for (uint32_t d = 0; d < outer * FIXED_WINDOW; d++) base = base.dbl();
        const g1_xyzz_t m = base.mul_small(inner);
        if (!m.is_inf()) {  // to affine: x = X / ZZ, y = Y / ZZZ
            const fq_t izzz = m.zzz.inverse();
            const fq_t izz = izzz.sqr() * m.zz.sqr();  // ZZ^3 = ZZZ^2, so 1/ZZ = ZZ^2 / ZZZ^2
            q.x = m.x * izz;
            q.y = m.y * izzz;
        }
    }
    store_aff<fq_t>(&table[t], q);
}
// out_i = sum_outer table[outer][digit_outer(v_i)] (fixed_base.rs:70-97); v_i are Fr elements in Montgomery form
static __global__ void __launch_bounds__(256) g1_fixed_msm_kernel(const g1_aff_mem_t* __restrict__ table, const fr_mem_t* __restrict__ scalars, size_t n,
                                                           uint32_t* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[9];
    {
        fr_t c32 = fr_t::zero();
        c32.v[0] = 32;  // memory Montgomery -> canonical integer (see ntt.hip.h fr_to_bigint_kernel)
        (fr_t::load(&scalars[i]) * c32).pack(k);
        k[8] = 0;
    }
    g1_xyzz_t acc = g1_xyzz_t::inf();
    for (int outer = 0; outer < FIXED_OUTER; outer++) {
        const int bit = outer * FIXED_WINDOW, wi = bit >> 5, sh = bit & 31;
        const uint32_t d = (uint32_t)((((uint64_t)k[wi] | ((uint64_t)k[wi + 1] << 32)) >> sh) & ((1u << FIXED_WINDOW) - 1));
        if (d) acc.add_affine(g1_load_aff(&table[((uint32_t)outer << FIXED_WINDOW) + d]));
    }
    const g1_jac_t j = acc.to_jacobian();
    uint32_t* dst = out + 36 * i;
    j.x.to_raw_words(dst);
    j.y.to_raw_words(dst + 12);
    j.z.to_raw_words(dst + 24);
}

// ---- group NTT ----------------------------------------------------------------------------------------------------
// One decimation-in-frequency stage over XYZZ points, in place: for the pair (i, i + half) of a block of 2 * half points
//     a' = a + b,   b' = (a - b) * tw[j * stride],  j = index inside the half block, tw[k] = root^k as canonical integers.
// log2(n) stages leave the result in bit-reversed order (g1_bitrev_kernel restores natural order).
static __global__ void __launch_bounds__(64) g1_ntt_stage_kernel(g1_xyzz_mem_t* __restrict__ pts, size_t n, size_t half, const fr_mem_t* __restrict__ tw,
                                                          size_t stride) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= n / 2) return;
    const size_t j = t % half, blk = t / half;
    const size_t ia = blk * 2 * half + j, ib = ia + half;
    const g1_xyzz_t a = g1_load_xyzz(&pts[ia]), b = g1_load_xyzz(&pts[ib]);
    g1_xyzz_t s = a;
    s.add(b);
    g1_xyzz_t d = a;
    d.add(g1_neg(b));
    if (j) {  // tw[0] = 1
        uint32_t k[8];
        const uint4* q = (const uint4*)&tw[j * stride];
        const uint4 lo = q[0], hi = q[1];
        k[0] = lo.x, k[1] = lo.y, k[2] = lo.z, k[3] = lo.w, k[4] = hi.x, k[5] = hi.y, k[6] = hi.z, k[7] = hi.w;
        d = g1_mul_words(d, k);
    }
    g1_store_xyzz(&pts[ia], s);
    g1_store_xyzz(&pts[ib], d);
}
// ---- the same stage with FOUR lanes per butterfly (round 6) ------------------------------------------------------------------------
// One butterfly = two additions and a 253-bit scalar multiplication, ~2.2 million dependent instructions on one lane; at n = 2^16 a stage of the
// one-thread form above is 512 waves on 1 024 SIMDs - half the chip idle, the stage as long as ONE chain (latency-bound: a lone wave issues an
// instruction every ~4.5 cycles).  Here the four lanes of a DPP quad share every point operation of their butterfly: quad_add (msm.hip.h: the 14
// products of add-2008-s in four rounds of one product per lane) and quad_dbl below (dbl-2008-s-1 in three rounds), and the twiddle is consumed two
// bits at a time against {P, 2P, 3P} held in registers: per 2-bit window 2 x 3 + 4 = 10 product rounds instead of 2 x (9 + 14) = 46 dependent
// products.  Four times the lanes fill the chip, the chain is ~4x shorter.  Same group elements (results are compared after to_affine).
template <class F>
__device__ __forceinline__ void quad_dbl(xyzz_t<F>& p) {
    const uint32_t r = threadIdx.x & 3;
    const F u = p.y.dbl();
    F a = select_field(r == 0, u, p.x);
    const F m1 = a * a;  // V = U^2 | XX = X^2 | - | -
    const F v = quad_bcast<0>(m1), xx = quad_bcast<1>(m1);
    const F m = xx.dbl() + xx;
    a = select_field(r == 0, u, select_field(r == 1, p.x, m));
    F b = select_field(r < 2, v, m);
    const F m2 = a * b;  // W = U V | S = X V | M^2 | (M^2)
    const F w = quad_bcast<0>(m2), s = quad_bcast<1>(m2);
    const F x3 = quad_bcast<2>(m2) - s.dbl();
    a = select_field(r == 0, m, select_field(r == 2, v, w));  // M | W | V | W
    b = select_field(r < 2, select_field(r == 0, s - x3, p.y), select_field(r == 2, p.zz, p.zzz));
    const F m3 = a * b;  // M (S - X3) | W Y | V ZZ | W ZZZ
    p.x = x3;
    p.y = quad_bcast<0>(m3) - quad_bcast<1>(m3);
    p.zz = quad_bcast<2>(m3);
    p.zzz = quad_bcast<3>(m3);
}
// k * p for a canonical integer k < 2^254 in eight words (consumed), by the four lanes of a quad together; the quad's lanes hold identical (p, k)
// (force-inlined: as an out-of-line function - operands passed by reference through the caller's scratch frame - the first launch of the stage kernel in a
// fresh process never returned on gfx950 / ROCm 7.2, while the same launch after kernels with larger frames had run was fine; inlined, the kernel has no calls)
__device__ __forceinline__ g1_xyzz_t g1_quad_mul_words(const g1_xyzz_t& p, uint32_t (&k)[8]) {
    g1_xyzz_t t2 = p;
    quad_dbl(t2);
    g1_xyzz_t t3 = t2;
    quad_add(t3, p);
    g1_xyzz_t acc = g1_xyzz_t::inf();
#pragma unroll
    for (int i = 7; i > 0; i--) k[i] = (k[i] << 2) | (k[i - 1] >> 30);  // bits 255, 254 are zero (k < r < 2^253): start at the window of bits 253, 252
    k[0] <<= 2;
#pragma unroll 1
    for (int wdw = 0; wdw < 127; wdw++) {
        quad_dbl(acc);
        quad_dbl(acc);
        const uint32_t d = k[7] >> 30;
#pragma unroll
        for (int i = 7; i > 0; i--) k[i] = (k[i] << 2) | (k[i - 1] >> 30);
        k[0] <<= 2;
        const g1_xyzz_t t = select_point(d == 1, p, select_point(d == 2, t2, t3));
        g1_xyzz_t sum = acc;
        quad_add(sum, t);
        acc = select_point(d == 0, acc, sum);
    }
    return acc;
}
__device__ __forceinline__ void g1_load_words8(uint32_t (&k)[8], const fr_mem_t* src) {
    const uint4* q = (const uint4*)src;
    const uint4 lo = q[0], hi = q[1];
    k[0] = lo.x, k[1] = lo.y, k[2] = lo.z, k[3] = lo.w, k[4] = hi.x, k[5] = hi.y, k[6] = hi.z, k[7] = hi.w;
}
// thread t: butterfly t / 4.  n / 2 butterflies, n >= 32 (every lane of every wave owns a butterfly: the quad operations are wave-wide).
static __global__ void __launch_bounds__(64) g1_ntt_stage_quad_kernel(g1_xyzz_mem_t* __restrict__ pts, size_t n, size_t half, const fr_mem_t* __restrict__ tw,
                                                               size_t stride) {
    const size_t t = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 2;
    const size_t j = t % half, blk = t / half;
    const size_t ia = blk * 2 * half + j, ib = ia + half;
    const g1_xyzz_t a = g1_load_xyzz(&pts[ia]), b = g1_load_xyzz(&pts[ib]);
    g1_xyzz_t s = a;
    quad_add(s, b);
    g1_xyzz_t d = a;
    quad_add(d, g1_neg(b));
    // tw[0] = 1 multiplies like any other twiddle (a wave-uniform skip would need every butterfly of the wave at j = 0) - except in the last stage, where
    // EVERY butterfly has j = 0: no multiplication at all (kernel-uniform)
    if (half > 1) {
        uint32_t k[8];
        g1_load_words8(k, &tw[j * stride]);
        d = g1_quad_mul_words(d, k);
    }
    if ((threadIdx.x & 3) == 0) {
        g1_store_xyzz(&pts[ia], s);
        g1_store_xyzz(&pts[ib], d);
    }
}
// pts[i] <- k * pts[i], four lanes per point (n >= 16)
static __global__ void __launch_bounds__(64) g1_scale_quad_kernel(g1_xyzz_mem_t* __restrict__ pts, size_t n, fr_mem_t k_int) {
    const size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 2;
    uint32_t k[8];
    g1_load_words8(k, &k_int);
    const g1_xyzz_t r = g1_quad_mul_words(g1_load_xyzz(&pts[i]), k);
    if ((threadIdx.x & 3) == 0) g1_store_xyzz(&pts[i], r);
}
static __global__ void g1_bitrev_kernel(g1_xyzz_mem_t* __restrict__ pts, size_t n, int lg) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    size_t r = 0;
    for (int b = 0; b < lg; b++) r |= ((i >> b) & 1) << (lg - 1 - b);
    if (r > i) {
        const g1_xyzz_mem_t x = pts[i], y = pts[r];
        pts[i] = y;
        pts[r] = x;
    }
}
// pts[i] <- k * pts[i] (the 1/n of the inverse transform)
static __global__ void __launch_bounds__(64) g1_scale_kernel(g1_xyzz_mem_t* __restrict__ pts, size_t n, fr_mem_t k_int) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k[8];
    const uint32_t* kw = (const uint32_t*)&k_int;
    for (int w = 0; w < 8; w++) k[w] = kw[w];
    g1_store_xyzz(&pts[i], g1_mul_words(g1_load_xyzz(&pts[i]), k));
}

}  // namespace sv
