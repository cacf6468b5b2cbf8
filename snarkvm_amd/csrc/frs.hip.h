// frs.hip.h - lazily reduced, signed-limb arithmetic in the BLS12-377 scalar field for the NTT butterflies (round 4).
//
// The butterflies of ntt_pass_kernel_v2 ran on ff.hip.h's unsigned lazy routines: a sum is a carry chain, a difference adds
// 2^s * r limb by limb and carries, a product negates / masks / adds per Montgomery column (profiles/r04_ntt_isa.md: 153
// multiply-adds + ~135 other instructions per butterfly, multiply-add share of the kernel 0.49).  The recipe that ffl.hip.h applied
// to Fq, applied to Fr:
//
//   * 9 signed 29-bit limbs, value = sum v_i 2^(29 i).  "Normalised": v_0 .. v_7 in [0, 2^29), v_8 signed.  No canonical
//     representative inside a pass: a value is any integer congruent to the element's memory form, |value| < 2^10 r.
//   * Montgomery reduction by R = 2^290 = 10 limb steps for 9-limb operands (+8 multiply-adds per product).  r < 2^253, so
//     |a| < 2^10 r and a canonical twiddle w < r give |a w| / R < 2^-26 r: a product lands in (-1.0001 r, 0.0001 r) whatever
//     its operand carried - NO conditional subtraction, and the value does not grow through products.
//   * A difference u - v is 9 independent subtractions (no carries, no "+ 2^s r"): its limbs lie in (-2^30.7, 2^30.7) - see the
//     bounds below - and it goes straight into the product.  A sum is carry-normalised (it may meet further sums before its
//     next product): limbs back in [0, 2^29), the value doubles per stage at most: < 2^9 r after the 9 stages of the longest pass.
//   * The quotient digit is taken with the opposite sign: m_k = column mod 2^29, the column gets - m_k r (r = 1 mod 2^29): one
//     AND and one arithmetic 64-bit shift per column besides its multiply-adds.
//   * Column bound (signed 64-bit accumulator): 9 products |a_i| w_j with |a_i| < 2^30.7 (a difference of a normalised sum of two
//     normalised values, < 2^30, and a normalised value) and w_j < 2^29: 9 * 2^59.7 = 2^62.9; plus <= 9 quotient terms
//     m_i r_j whose sum is < 3.26 * 2^58 (limbs 1 .. 8 of r sum to 3.26 * 2^29); plus the carry: < 2^63.
//
// Twiddles enter as canonical residues of w * 2^290 (the tables built for ff.hip.h hold w * 2^261: one product by the integer
// 2^290 when a table is filled or a per-pass constant is loaded, see ntt.hip.h).  Data stays in the reference's memory form
// throughout, as before: the transform is linear and every product carries a twiddle in Montgomery form.
#pragma once
#include "ff.hip.h"

namespace sv {

struct FrS {
    static constexpr int N = 9;
    static constexpr int STEPS = 10;  // R = 2^(29 * 10) = 2^290
    static constexpr int32_t MOD[9] = {0x00000001, 0x108c0000, 0x00000042, 0x14edfda0, 0x1b00159a, 0x068f2e1b, 0x155982d1, 0x0bd34594, 0x0012ab65};
    // 2^290 mod r as a plain integer.  (a) the twiddle "one"; (b) ff.hip.h internal form w * 2^261 -> w * 2^290: w_int * C290 (exact product)
    static constexpr uint32_t C290[9] = {0x06b6e13cu, 0x0ccffe49u, 0x1e8ae22fu, 0x03709f7cu, 0x1ec3b929u, 0x02cbcfb1u, 0x15629d63u, 0x0cded9d3u, 0x0000d09eu};
    // 2^580 mod r: internal form t * 2^261 -> t * 2^580, the closing twiddle of the pass before the last one when the last pass ends
    // with a bare reduction (ntt.hip.h `reduce_only`): (x * t 2^580 / 2^290) / 2^290 = x t
    static constexpr uint32_t C580[9] = {0x11835831u, 0x153c0998u, 0x1220f31eu, 0x1cf40af6u, 0x000bb5fau, 0x1de4b39au, 0x1bd29079u, 0x0137a8ffu, 0x000e9a98u};
};

// The compiler canonicalises the sign extension of a value it has proven non-negative (a masked limb) into a zero extension and then
// fails to match "zero-extended x sign-extended" products to ONE v_mad_i64_i32 (ffl.hip.h hides the range with empty asm statements;
// here they sit inside the pass kernel's loops and keep them from unrolling early enough: the four elements of a radix-4 group end up
// in scratch memory).  A butterfly product is safe as it stands - its left operand is a raw difference, of unknown sign - ; the one
// place where a normalised value meets a table word is the closing product of a pass, and there the value goes through hide_range():
// an exclusive-or with a zero the compiler cannot see (a device variable nobody writes).
#if defined(__HIP_DEVICE_COMPILE__)
static __device__ int32_t sv_opaque_zero;
#define SV_OPAQUE_ZERO() sv_opaque_zero
#else
#define SV_OPAQUE_ZERO() 0
#endif

struct frs_t {
    static constexpr int N = 9;
    static constexpr uint32_t MASK = (1u << 29) - 1;
    int32_t v[N];

    // a canonical value (ff.hip.h limbs, < r) is a normalised signed value as it stands
    SV_HD static frs_t from_canonical(const fr_t& a) {
        frs_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = (int32_t)a.v[i];
        return r;
    }
    // u + v, carry-normalised.  Operands normalised (or sums of few normalised values: |limbs| < 2^30); same value out.
    SV_HD static frs_t add_norm(const frs_t& a, const frs_t& b) {
        frs_t r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
            const int32_t x = a.v[i] + b.v[i] + c;
            r.v[i] = (int32_t)((uint32_t)x & MASK);
            c = x >> 29;
        }
        r.v[N - 1] = a.v[N - 1] + b.v[N - 1] + c;
        return r;
    }
    // u - v limb by limb: no carries.  For normalised operands the limbs lie in (-2^29, 2^29) (top limb: the operands' range)
    SV_HD static frs_t sub_raw(const frs_t& a, const frs_t& b) {
        frs_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = a.v[i] - b.v[i];
        return r;
    }
    // a * w / 2^290: a signed (|a_i| < 2^30.7), w canonical (limbs of a value < r read as non-negative integers).  Result
    // normalised, in (-1.0001 r, 0.0001 r) for |a| < 2^10 r.
    SV_HD static frs_t mul(const frs_t& a, const fr_t& w) {
        uint32_t m[FrS::STEPS];
        frs_t r;
        int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < N + FrS::STEPS; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) acc += (int64_t)a.v[i] * (int32_t)w.v[j];
            }
#pragma unroll
            for (int i = 0; i < FrS::STEPS; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FrS::MOD[j];
            }
            if (k < FrS::STEPS) {
                m[k] = (uint32_t)acc & MASK;  // the column minus m_k * r_0 = m_k has 29 zero low bits: the shift drops exactly them
            } else {
                r.v[k - FrS::STEPS] = (k == N + FrS::STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
            }
            acc >>= 29;  // arithmetic
        }
        return r;
    }
    // x / 2^290 alone (the last pass when the previous pass' closing table carried the missing 2^290): half the multiply-adds
    SV_HD static frs_t reduce_only(const frs_t& a) {
        uint32_t m[FrS::STEPS];
        frs_t r;
        int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < N + FrS::STEPS; k++) {
            if (k < N) acc += a.v[k];
#pragma unroll
            for (int i = 0; i < FrS::STEPS; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FrS::MOD[j];
            }
            if (k < FrS::STEPS) {
                m[k] = (uint32_t)acc & MASK;
            } else {
                r.v[k - FrS::STEPS] = (k == N + FrS::STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
            }
            acc >>= 29;
        }
        return r;
    }
    // the same value with its limb ranges hidden from the compiler (see SV_OPAQUE_ZERO)
    SV_HD frs_t hide_range() const {
        const int32_t z = SV_OPAQUE_ZERO();
        frs_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = v[i] ^ z;
        return r;
    }
    // leaving the lazy domain: the canonical representative of a normalised value within (-3 r, 2 r)
    SV_HD fr_t to_canonical() const {
        int32_t t[N];
#pragma unroll
        for (int i = 0; i < N; i++) t[i] = v[i];
#pragma unroll 1
        for (int round = 0; round < 3 && t[N - 1] < 0; round++) {  // negative <=> top limb negative (normalised)
            int32_t c = 0;
#pragma unroll
            for (int i = 0; i < N - 1; i++) {
                const int32_t x = t[i] + FrS::MOD[i] + c;
                t[i] = (int32_t)((uint32_t)x & MASK);
                c = x >> 29;
            }
            t[N - 1] += FrS::MOD[N - 1] + c;
        }
        uint32_t w[N];
#pragma unroll
        for (int i = 0; i < N; i++) w[i] = (uint32_t)t[i];
        return fr_t::cond_sub(fr_t::cond_sub(w).v);  // [0, 2 r) -> [0, r)
    }
    // a twiddle of the ff.hip.h tables (internal form w * 2^261, canonical) in the form mul() wants: w * 2^290
    SV_HD static fr_t twiddle_form(const fr_t& w_int) { return w_int * fr_t::from_table(FrS::C290); }
};

}  // namespace sv
