// poly.hip.h - the prover-round Fr vector kernels that sit between the NTTs and the MSMs (SURVEY.md §8 N2 and the
// `open` half of row a15).  All of them are O(n) streaming passes over 32-byte Fr elements in the reference's memory
// form (a * 2^256 mod r), so that data produced by an NTT can feed a commitment without leaving HBM:
//
//   fr_vec_op_kernel            a+b, a-b, a*b, a*b-c, a*s, a-s, a+b*s, s-a       second.rs:104-122 (rowcheck), kzg10/mod.rs:295-300
//   fr_horner_up/down_kernel    suffix Horner sums h_i = sum_{k>=i} a_k z^(k-i)  -> p / (X - z) and p(z)
//                                                                                 kzg10/mod.rs:213-236, dense.rs:98-114
//   fr_batch_inverse_kernel     v_i <- coeff / v_i, zeros stay zero              fields/src/lib.rs:66-129
//   fr_distribute_powers_kernel v_i <- v_i * c * g^i                             fft/domain.rs:224-254
//   fr_fold_vanishing_kernel    quotient / remainder by X^D - 1                  dense.rs:161-169 (divide_by_vanishing_poly)
//   fr_mul_vanishing_kernel     p * (X^D - 1)                                    dense.rs:153-159
//
// Representation note (ff.hip.h): the raw memory limbs of a, read as an internal value, are the internal Montgomery form of
// a * 2^-5 ("shifted").  Sums and differences of shifted values are shifted values; the product of a TRUE internal
// value (x.from_mem_mont()) with a shifted value is the shifted product.  So a kernel converts only its broadcast
// operand (z, g, c, coeff) and streams the vectors untouched; only a product of two vector elements needs one fix-up.
#pragma once
#include "ff.hip.h"

namespace sv {

enum { FR_OP_ADD = 0, FR_OP_SUB = 1, FR_OP_MUL = 2, FR_OP_MUL_SUB = 3, FR_OP_SCALE = 4, FR_OP_SUB_SCALAR = 5, FR_OP_AXPY = 6, FR_OP_RSUB_SCALAR = 7 };

// Strided batches (lock-step proving: the same pass over vector v of every proof of a batch): blockIdx.y selects the vector, every
// vector operand of batch member y starts `stride` elements after that of member y - 1.  A single call is a batch of one.
static __global__ void fr_vec_op_kernel(int op, fr_mem_t* out, const fr_mem_t* a, const fr_mem_t* b, const fr_mem_t* c, fr_mem_t s_mem, size_t n, size_t stride) {
    {
        const size_t off = (size_t)blockIdx.y * stride;
        out += off, a += off;
        if (b) b += off;
        if (c) c += off;
    }
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    const fr_t s_shift = fr_t::load(&s_mem);
    const fr_t s = (op == FR_OP_SCALE || op == FR_OP_AXPY) ? s_shift.from_mem_mont() : s_shift;
    for (; i < n; i += st) {
        const fr_t x = fr_t::load(&a[i]);
        fr_t r;
        switch (op) {
            case FR_OP_ADD: r = x + fr_t::load(&b[i]); break;
            case FR_OP_SUB: r = x - fr_t::load(&b[i]); break;
            case FR_OP_MUL: r = (x * fr_t::load(&b[i])).from_mem_mont(); break;
            case FR_OP_MUL_SUB: r = (x * fr_t::load(&b[i])).from_mem_mont() - fr_t::load(&c[i]); break;
            case FR_OP_SCALE: r = x * s; break;
            case FR_OP_SUB_SCALAR: r = x - s; break;
            case FR_OP_AXPY: r = x + fr_t::load(&b[i]) * s; break;
            default: r = s - x; break;
        }
        r.store(&out[i]);
    }
}

// ---- suffix Horner sums --------------------------------------------------------------------------------------------
// h_i = sum_{k >= i} a_k m^(k-i) satisfies h_i = a_i + m h_(i+1): a linear recurrence, parallelised over chunks of
// POLY_CHUNK consecutive coefficients.  `up` evaluates each chunk on its own (cv_t = chunk polynomial at m); the chunk
// values form the same problem with multiplier m^POLY_CHUNK (solved recursively, api.hip); `down` replays each chunk
// from its incoming carry h_(end of chunk) and writes every h_i.  A thread walks its chunk downwards, so consecutive
// reads of one thread share a cache line.
static constexpr int POLY_CHUNK = 32;

// mult[k] = m^(POLY_CHUNK^k), memory form, k < levels (single thread: a handful of multiplications)
static __global__ void fr_horner_multipliers_kernel(fr_mem_t m_mem, fr_mem_t* mult, int levels) {
    if (blockIdx.x | threadIdx.x) return;
    fr_t m = fr_t::load(&m_mem).from_mem_mont();
    for (int k = 0; k < levels; k++) {
        m.to_mem_mont().store(&mult[k]);
        m = m.pow_u64(POLY_CHUNK);
    }
}
static __global__ void fr_horner_up_kernel(const fr_mem_t* __restrict__ in, size_t n, const fr_mem_t* __restrict__ m_mem, fr_mem_t* __restrict__ cv,
                                    size_t T, size_t in_stride, size_t cv_stride) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= T) return;
    in += (size_t)blockIdx.y * in_stride;
    cv += (size_t)blockIdx.y * cv_stride;
    const fr_t m = fr_t::load(m_mem).from_mem_mont();
    const size_t lo = t * POLY_CHUNK;
    const size_t hi = (lo + POLY_CHUNK < n) ? lo + POLY_CHUNK : n;
    fr_t acc = fr_t::zero();
    for (size_t i = hi; i-- > lo;) acc = fr_t::load(&in[i]) + m * acc;
    acc.store(&cv[t]);
}
// carry[t + 1] = h at the first index of chunk t + 1 (nullptr when there is a single chunk).  Writes out[i - shift] = h_i
// for i >= shift and *first = h_0 when shift == 1 (quotient by X - m: q_(i-1) = h_i, remainder = h_0).
// in == out is allowed when shift == 0.
static __global__ void fr_horner_down_kernel(const fr_mem_t* in, size_t n, const fr_mem_t* __restrict__ m_mem, const fr_mem_t* __restrict__ carry,
                                      size_t T, fr_mem_t* out, int shift, fr_mem_t* first, size_t in_stride, size_t carry_stride, size_t out_stride) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= T) return;
    in += (size_t)blockIdx.y * in_stride;
    if (carry) carry += (size_t)blockIdx.y * carry_stride;
    out += (size_t)blockIdx.y * out_stride;
    if (first) first += blockIdx.y;
    const fr_t m = fr_t::load(m_mem).from_mem_mont();
    const size_t lo = t * POLY_CHUNK;
    const size_t hi = (lo + POLY_CHUNK < n) ? lo + POLY_CHUNK : n;
    fr_t acc = (carry && t + 1 < T) ? fr_t::load(&carry[t + 1]) : fr_t::zero();
    for (size_t i = hi; i-- > lo;) {
        acc = fr_t::load(&in[i]) + m * acc;
        if (i >= (size_t)shift)
            acc.store(&out[i - shift]);
        else if (first)
            acc.store(first);
    }
}

// ---- the same sums with a scan inside every workgroup (round 5) --------------------------------------------------------
// The chunk recursion above is four dependent levels of 32 serial products each for the 2^17 coefficients of an opening, run by
// 4 096 threads: eight launches of 50 - 100 us, 1.3 ms of the 9.5 ms of kernels of one proof (profiles/r05_proof1_timeline.md).
// Here a workgroup of 256 threads owns 256 C consecutive coefficients (C = 8 up to 2^19 coefficients): every thread folds its C
// coefficients (C serial products), the 256 thread values are turned into their suffix sums by a Kogge-Stone scan through LDS
// (8 steps of one product, multipliers m^(C 2^j)), and a workgroup gets the carry of everything behind it as ONE product per
// thread against a table of m^(256 C j) plus a tree of additions.  Three launches (tables, up, down), 17 dependent products.
// Exact field arithmetic in another order: the same values bit for bit.
static constexpr int HORNER2_B = 256;
// tab: [0, 8) step[j] = m^(C 2^j) | [8, 8 + 257) pw[j] = m^(C j) | [265, 265 + 256) pwB[j] = m^(256 C j); internal form
static constexpr int HORNER2_TAB = 8 + 257 + 256;
static __global__ void __launch_bounds__(HORNER2_B) fr_horner2_tables_kernel(fr_mem_t m_mem, fr_mem_t* __restrict__ tab, uint32_t C) {
    const uint32_t t = threadIdx.x;
    const fr_t m = fr_t::load(&m_mem).from_mem_mont();
    const fr_t mC = m.pow_u64(C);
    const fr_t mB = mC.pow_u64(HORNER2_B);
    mC.pow_u64(t).store(&tab[8 + t]);
    mB.pow_u64(t).store(&tab[8 + 257 + t]);
    if (t == 0) mB.store(&tab[8 + 256]);
    if (t < 8) mC.pow_u64(1u << t).store(&tab[t]);
}
__device__ __forceinline__ void horner2_lds_put(uint32_t* sh, uint32_t t, const fr_t& x) {
#pragma unroll
    for (int l = 0; l < 9; l++) sh[l * HORNER2_B + t] = x.v[l];
}
__device__ __forceinline__ fr_t horner2_lds_get(const uint32_t* sh, uint32_t t) {
    fr_t x;
#pragma unroll
    for (int l = 0; l < 9; l++) x.v[l] = sh[l * HORNER2_B + t];
    return x;
}
// hs[block * 256 + t] = the suffix sum of the block's thread values from thread t on (= h at the first coefficient of thread t, the
// block taken alone); bv[block] = hs[block * 256] (the block's polynomial at m)
static __global__ void __launch_bounds__(HORNER2_B) fr_horner2_up_kernel(const fr_mem_t* __restrict__ in, size_t n, fr_mem_t m_mem, const fr_mem_t* __restrict__ tab,
                                                                  fr_mem_t* __restrict__ hs, fr_mem_t* __restrict__ bv, uint32_t C, size_t in_stride,
                                                                  size_t hs_stride, size_t bv_stride) {
    __shared__ uint32_t sh[9 * HORNER2_B];
    const uint32_t t = threadIdx.x;
    in += (size_t)blockIdx.y * in_stride;
    hs += (size_t)blockIdx.y * hs_stride;
    bv += (size_t)blockIdx.y * bv_stride;
    const fr_t m = fr_t::load(&m_mem).from_mem_mont();
    const size_t lo = ((size_t)blockIdx.x * HORNER2_B + t) * C;
    const size_t hi = lo + C < n ? lo + C : n;
    fr_t h = fr_t::zero();
    for (size_t i = hi; i > lo;) {
        i--;
        h = fr_t::load(&in[i]) + m * h;
    }
#pragma unroll 1
    for (int j = 0; j < 8; j++) {
        horner2_lds_put(sh, t, h);
        __syncthreads();
        if (t + (1u << j) < HORNER2_B) h = h + fr_t::load(&tab[j]) * horner2_lds_get(sh, t + (1u << j));
        __syncthreads();
    }
    h.store(&hs[(size_t)blockIdx.x * HORNER2_B + t]);
    if (t == 0) h.store(&bv[blockIdx.x]);
}
// out[i - shift] = h_i (i >= shift), *first = h_0 when shift == 1.  nblocks <= 256 + 1.
static __global__ void __launch_bounds__(HORNER2_B) fr_horner2_down_kernel(const fr_mem_t* in, size_t n, fr_mem_t m_mem, const fr_mem_t* __restrict__ tab,
                                                                    const fr_mem_t* __restrict__ hs, const fr_mem_t* __restrict__ bv, uint32_t C, fr_mem_t* out,
                                                                    int shift, fr_mem_t* first, size_t in_stride, size_t hs_stride, size_t bv_stride,
                                                                    size_t out_stride) {
    __shared__ uint32_t sh[9 * HORNER2_B];
    const uint32_t t = threadIdx.x, b = blockIdx.x, nblocks = gridDim.x;
    in += (size_t)blockIdx.y * in_stride;
    hs += (size_t)blockIdx.y * hs_stride;
    bv += (size_t)blockIdx.y * bv_stride;
    out += (size_t)blockIdx.y * out_stride;
    if (first) first += blockIdx.y;
    const fr_t m = fr_t::load(&m_mem).from_mem_mont();
    // carry = h at the first coefficient of block b + 1 = sum_(s > b) bv[s] m^(256 C (s - b - 1)): one term per thread, then a tree of sums
    fr_t term = fr_t::zero();
    if (b + 1 + t < nblocks) term = fr_t::load(&tab[8 + 257 + t]) * fr_t::load(&bv[b + 1 + t]);
#pragma unroll 1
    for (uint32_t off = HORNER2_B / 2; off >= 1; off >>= 1) {
        horner2_lds_put(sh, t, term);
        __syncthreads();
        if (t < off) term = term + horner2_lds_get(sh, t + off);
        __syncthreads();
    }
    horner2_lds_put(sh, t, term);
    __syncthreads();
    const fr_t carry = horner2_lds_get(sh, 0);
    const size_t lo = ((size_t)b * HORNER2_B + t) * C;
    if (lo >= n) return;
    const size_t hi = lo + C < n ? lo + C : n;
    // h behind this thread's coefficients: the block's own suffix sum from thread t + 1 on, plus the carry moved across the threads in between
    fr_t h = fr_t::load(&tab[8 + (HORNER2_B - 1 - t)]) * carry;
    if (t + 1 < HORNER2_B) h = h + fr_t::load(&hs[(size_t)b * HORNER2_B + t + 1]);
    for (size_t i = hi; i > lo;) {
        i--;
        h = fr_t::load(&in[i]) + m * h;
        if (i >= (size_t)shift)
            h.store(&out[i - shift]);
        else if (first)
            h.store(first);
    }
}

// ---- batch inversion -----------------------------------------------------------------------------------------------
// Montgomery's trick per thread over the strided set {t, t + T, t + 2T, ...} (any partition gives the same values; a
// strided one keeps every access coalesced).  Prefix products are parked in `scratch` (n elements, internal form); one
// Fermat inversion per thread is amortised over ceil(n / T) elements.
static __global__ void fr_batch_inverse_kernel(fr_mem_t* __restrict__ v, size_t n, fr_mem_t coeff_mem, fr_mem_t* __restrict__ scratch, size_t T) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= T) return;
    fr_t tmp = fr_t::one();
    bool any = false;
    for (size_t i = t; i < n; i += T) {
        const fr_t x = fr_t::load(&v[i]);
        if (!x.is_zero()) {
            tmp = tmp * x.from_mem_mont();
            any = true;
        }
        tmp.store(&scratch[i]);
    }
    if (!any) return;
    tmp = tmp.inverse() * fr_t::load(&coeff_mem).from_mem_mont();
    const size_t cnt = (n - t + T - 1) / T;
    for (size_t k = cnt; k-- > 0;) {
        const size_t i = t + k * T;
        const fr_t xs = fr_t::load(&v[i]);
        if (xs.is_zero()) continue;
        const fr_t s = k ? fr_t::load(&scratch[i - T]) : fr_t::one();
        (tmp * s).to_mem_mont().store(&v[i]);
        tmp = tmp * xs.from_mem_mont();
    }
}

// v_i <- v_i * c * g^i; thread t owns i = t, t + T, ... with running power c g^t (g^T)^k
static __global__ void fr_distribute_powers_kernel(fr_mem_t* __restrict__ v, size_t n, fr_mem_t g_mem, fr_mem_t c_mem, size_t T) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= T || t >= n) return;
    const fr_t g = fr_t::load(&g_mem).from_mem_mont();
    fr_t pw = fr_t::load(&c_mem).from_mem_mont() * g.pow_u64(t);
    const fr_t step = g.pow_u64(T);
    for (size_t i = t; i < n; i += T) {
        (fr_t::load(&v[i]) * pw).store(&v[i]);
        pw = pw * step;
    }
}
static __global__ void fr_fill_kernel(fr_mem_t* v, size_t n, fr_mem_t x) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) v[i] = x;
}
// v_i <- (v_i == x) ? one : zero   (the tau-in-domain branch of evaluate_all_lagrange_coefficients, domain.rs:265-275)
static __global__ void fr_onehot_kernel(fr_mem_t* v, size_t n, fr_mem_t x_mem, fr_mem_t one_mem) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t st = (size_t)gridDim.x * blockDim.x;
    const fr_t x = fr_t::load(&x_mem);
    fr_mem_t zero_mem;
    fr_t::zero().store(&zero_mem);
    for (; i < n; i += st) v[i] = (fr_t::load(&v[i]) == x) ? one_mem : zero_mem;
}

// ---- X^D - 1 -------------------------------------------------------------------------------------------------------
// Long division of a (len coefficients) by X^D - 1 folds the coefficient classes mod D:
//   quotient_i = sum_{k >= 1} a_(i + kD)   (i < len - D),   remainder_i = sum_{k >= 0} a_(i + kD)   (i < min(D, len)).
static __global__ void fr_fold_vanishing_kernel(const fr_mem_t* __restrict__ a, size_t len, size_t D, fr_mem_t* __restrict__ quot,
                                         fr_mem_t* __restrict__ rem, size_t stride) {
    {
        const size_t off = (size_t)blockIdx.y * stride;
        a += off, rem += off;
        if (quot) quot += off;
    }
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t qlen = len > D ? len - D : 0;
    const size_t rlen = len < D ? len : D;
    if (i >= qlen && i >= rlen) return;
    fr_t acc = fr_t::zero();
    // walk the class from the top so that the running sum at index i + D is the quotient and at i the remainder
    size_t top = i + ((len - 1 - i) / D) * D;
    for (size_t j = top; j > i; j -= D) acc = acc + fr_t::load(&a[j]);
    if (i < qlen) acc.store(&quot[i]);
    if (i < rlen) (acc + fr_t::load(&a[i])).store(&rem[i]);
}
// out (len + D elements) = a * (X^D - 1)
static __global__ void fr_mul_vanishing_kernel(const fr_mem_t* __restrict__ a, size_t len, size_t D, fr_mem_t* __restrict__ out) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= len + D) return;
    const fr_t hi = (i >= D) ? fr_t::load(&a[i - D]) : fr_t::zero();
    const fr_t lo = (i < len) ? fr_t::load(&a[i]) : fr_t::zero();
    (hi - lo).store(&out[i]);
}

}  // namespace sv
