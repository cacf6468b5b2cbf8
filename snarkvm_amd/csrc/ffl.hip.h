// ffl.hip.h - lazily reduced, signed-limb arithmetic in the BLS12-377 base field for the bucket-accumulation hot loop.
//
// ff.hip.h keeps every value canonical (< q, limbs < 2^29): per product that costs a conditional subtraction, per addition /
// subtraction a carry chain plus a select, and per Montgomery column a negate + mask + 64-bit add + shift.  The instruction
// histogram of msm_accumulate_seg_kernel (profiles/r03_accumulate_isa.md) shows what that adds up to: 2 951 multiply-adds and
// 2 238 other VALU instructions per mixed addition.  This file removes most of the second number for the one loop that matters:
//
//   * 13 signed 29-bit limbs, value = sum v_i 2^(29 i).  "Normalised": v_0 .. v_11 in [0, 2^29), v_12 signed.  A value is any
//     integer congruent to x * 2^406 (mod q) within a few q of zero - there is no canonical representative inside the loop.
//   * Montgomery reduction by R = 2^406 = 14 limb steps for 13-limb operands (one step more than ff.hip.h: +12 multiply-adds per
//     product).  With |a|, |b| < 2^10 q the quotient a b / R is below 2^-9 q in magnitude, so a product lands in
//     (-1.002 q, 0.002 q) whatever small multiples of q its operands carried: NO conditional subtraction, and sums /
//     differences of a few values feed the next product unreduced - additions and subtractions are 13 independent limb
//     operations without carries.
//   * The quotient digit is taken with the opposite sign: m_k = column mod 2^29 and the column gets  - m_k q  (q = 1 mod 2^29, so
//     the low limb cancels).  Then (column - m_k) >> 29 == column >> 29 for an arithmetic shift: a reduction column costs one
//     AND and one 64-bit shift besides its multiply-adds (ff.hip.h: negate, AND, 64-bit add, shift).
//   * Column sums are signed 64-bit (v_mad_i64_i32).  Bounds (each routine states its own): normalised x normalised and
//     normalised x difference-of-normalised products are < 2^58 in magnitude, a column holds <= 13 of them plus <= 12 quotient
//     terms whose sum is < 6.37 * 2^58 (the limbs 1 .. 12 of q sum to 6.37 * 2^29): < 2^62.3.
//
// Base points enter this arithmetic as canonical residues of x * 2^406 (the base slots of a G1 MSM hold that form, see
// runtime.hip.h), partial sums leave it through to_exact() (one product by 2^377 and a canonicalisation: back to the
// ff.hip.h representation the tail kernels use).  Exceptional cases of the addition law (equal x coordinates) are detected
// with a low-limb filter and resolved on the exact arithmetic (ec.hip.h); see xyzz_lazy_t::madd.
#pragma once
#include "ec.hip.h"

namespace sv {

struct FqL {
    static constexpr int N = 13;
    static constexpr int STEPS = 14;  // reduction steps: R = 2^(29 * 14) = 2^406
    static constexpr int32_t MOD[13] = {0x00000001, 0x08460000, 0x00000021, 0x16ba8860, 0x14800170, 0x1117dd04, 0x0e3c7bcd,
                                        0x1e601ea2, 0x1b1a22d9, 0x03650a49, 0x118ec170, 0x0f8a21d5, 0x1ae3a461};  // q (fq.rs:111-150)
    // 2^377 mod q as a plain integer: lazy (x 2^406) -> exact internal form (x 2^377):  mul(v, K377) = v 2^377 / 2^406
    static constexpr int32_t K377[13] = {0x1fffffff, 0x17b9ffff, 0x1fffffde, 0x0945779f, 0x0b7ffe8f, 0x0ee822fb, 0x11c38432,
                                         0x019fe15d, 0x04e5dd26, 0x1c9af5b6, 0x0e713e8f, 0x1075de2a, 0x051c5b9e};
    // 2^435 mod q: exact internal form (x 2^377) -> lazy:  mul(v, K435) = x 2^377 2^435 / 2^406 = x 2^406
    static constexpr int32_t K435[13] = {0x0677f3e3, 0x01fcf72f, 0x05fc93a0, 0x08eb59fe, 0x0c1a6c3f, 0x122b8005, 0x01c7de9b,
                                         0x0b159718, 0x189a8339, 0x07649107, 0x02e4fa41, 0x13571fee, 0x04eba458};
};
// constants for the EXACT arithmetic (ff.hip.h, R = 2^377) that move a value between the two Montgomery radices
struct FqLConv {
    // exact internal x 2^377  ->  canonical residue of x 2^406:  a * C406 (Montgomery product of ff.hip.h)
    static constexpr uint32_t C406[13] = {0x19eaf730u, 0x171ffffeu, 0x0d714cf8u, 0x044e31d8u, 0x1eb6e262u, 0x0bfad163u, 0x00d46e9cu,
                                          0x10b6ddbfu, 0x13b9cbddu, 0x075782afu, 0x03bd1d8au, 0x1557cab6u, 0x15742a14u};  // 2^406 mod q
    // the reference's memory form x 2^384  ->  canonical residue of x 2^406:  a * C399
    static constexpr uint32_t C399[13] = {0x1fb3d5efu, 0x1759ffffu, 0x161ae2aeu, 0x06bd319fu, 0x15cd6eabu, 0x15c6dfc5u, 0x00a7763du,
                                          0x1e5d80e1u, 0x10d7c95fu, 0x04add573u, 0x1c80b321u, 0x02e104bau, 0x10f92f11u};  // 2^399 mod q
    // canonical residue of x 2^406  ->  exact internal x 2^377:  a * C348  (2^348: a single limb)
    static constexpr uint32_t C348[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1};
};

// The compiler canonicalises the sign extension of a value it has proven non-negative (a masked limb) into a zero extension,
// and then fails to match "zero-extended x sign-extended" products (normalised limbs against the signed top limb or against a
// difference) to ONE v_mad_i64_i32: it emits two multiply-adds and two moves per such product (~450 extra instructions per
// mixed addition).  Passing every produced limb through an empty asm statement hides the range information; no instruction
// is emitted for it.
// (An asm statement inside a loop keeps `#pragma unroll` from unrolling it early enough for the column conditions to fold, so
// the thirteen statements are written out and applied to finished values only.)
#if defined(__HIP_DEVICE_COMPILE__)
#define SV_OPAQUE_LIMB(x) asm("" : "+v"(x))
#else
#define SV_OPAQUE_LIMB(x) ((void)0)
#endif
#define SV_OPAQUE_13(a)                                                                                              \
    do {                                                                                                             \
        SV_OPAQUE_LIMB((a)[0]); SV_OPAQUE_LIMB((a)[1]); SV_OPAQUE_LIMB((a)[2]); SV_OPAQUE_LIMB((a)[3]);              \
        SV_OPAQUE_LIMB((a)[4]); SV_OPAQUE_LIMB((a)[5]); SV_OPAQUE_LIMB((a)[6]); SV_OPAQUE_LIMB((a)[7]);              \
        SV_OPAQUE_LIMB((a)[8]); SV_OPAQUE_LIMB((a)[9]); SV_OPAQUE_LIMB((a)[10]); SV_OPAQUE_LIMB((a)[11]);            \
        SV_OPAQUE_LIMB((a)[12]);                                                                                     \
    } while (0)

struct fql_t {
    static constexpr int N = 13;
    static constexpr uint32_t MASK = (1u << 29) - 1;
    int32_t v[N];

    SV_HD static fql_t zero() {
        fql_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = 0;
        return r;
    }
    // a canonical residue (limbs of an fq_t read as a plain integer) is a normalised lazy value as it stands
    SV_HD static fql_t from_limbs(const fq_t& a) {
        fql_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = (int32_t)a.v[i];
        SV_OPAQUE_13(r.v);
        return r;
    }
    // limb-wise sum / difference: no carries.  |limbs| of the result = sum of the operands' bounds (the caller tracks them)
    SV_HD fql_t operator+(const fql_t& b) const {
        fql_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = v[i] + b.v[i];
        return r;
    }
    SV_HD fql_t operator-(const fql_t& b) const {
        fql_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = v[i] - b.v[i];
        return r;
    }
    // s ? -a : a  (two plain operations per limb)
    SV_HD fql_t negate_if(bool s) const {
        const int32_t m = s ? -1 : 0;
        fql_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = (v[i] ^ m) - m;
        return r;
    }
    // carry propagation: any limbs (|v_i| < 2^31 - 2^3) -> normalised, same value
    SV_HD fql_t normalized() const {
        fql_t r;
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
            const int32_t x = v[i] + c;
            r.v[i] = (int32_t)((uint32_t)x & MASK);
            c = x >> 29;
        }
        r.v[N - 1] = v[N - 1] + c;
        SV_OPAQUE_13(r.v);
        return r;
    }

    // ---- products.  Column k of the double-width value: sum_(i+j=k) a_i b_j  -  sum_(i+j=k, j>=1) m_i q_j; m_k = column mod 2^29
    // (STEPS of them), then the column moves on by an arithmetic shift.  Result normalised, = (a b - m q) / 2^406.
    // Operand bounds: |a_i b_j| < 2^58 for every pair that meets (normalised x normalised, normalised x difference of two
    // normalised values): |column| < 13 * 2^58 + 6.37 * 2^58 + carry < 2^62.3.
    SV_HD static fql_t mul(const fql_t& a, const fql_t& b) {
        uint32_t m[FqL::STEPS];
        fql_t r;
        int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < N + FqL::STEPS; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) acc += (int64_t)a.v[i] * b.v[j];
            }
#pragma unroll
            for (int i = 0; i < FqL::STEPS; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FqL::MOD[j];
            }
            if (k < FqL::STEPS) {
                m[k] = (uint32_t)acc & MASK;  // the column minus m_k * q_0 = m_k has 29 zero low bits: the shift below drops exactly them
            } else {
                r.v[k - FqL::STEPS] = (k == N + FqL::STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
            }
            acc >>= 29;  // arithmetic
        }
        SV_OPAQUE_13(r.v);
        return r;
    }
    // a^2: off-diagonal products once, the LOWER-indexed limb doubled (the top limb of a difference can reach 3.4 * 2^29 - the
    // accumulator's x lies within (-1.1 q, 3.1 q) - and must not be doubled in 32 bits).  |a_i| < 2^29 below the top limb:
    // |2 a_i * a_j| < 2^59 (2^60.8 against the top limb, once per column), at most 6 of them + one square per column: < 2^62.7
    // with the quotient terms.
    SV_HD static fql_t sqr(const fql_t& a) {
        uint32_t m[FqL::STEPS];
        int32_t a2[N];
#pragma unroll
        for (int i = 0; i < N; i++) a2[i] = a.v[i] << 1;
        fql_t r;
        int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < N + FqL::STEPS; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j > i && j < N) acc += (int64_t)a2[i] * a.v[j];
                if (j == i) acc += (int64_t)a.v[i] * a.v[i];
            }
#pragma unroll
            for (int i = 0; i < FqL::STEPS; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FqL::MOD[j];
            }
            if (k < FqL::STEPS) {
                m[k] = (uint32_t)acc & MASK;
            } else {
                r.v[k - FqL::STEPS] = (k == N + FqL::STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
            }
            acc >>= 29;
        }
        SV_OPAQUE_13(r.v);
        return r;
    }
    // a b - c d with ONE reduction (the XYZZ addition's Y3 = R (Q - X3) - Y1 PPP).  a, b: limbs in (-2^29, 2^29); c, d normalised
    // (non-negative limbs below the top one).  Here the quotient term is ADDED (m_k = -column mod 2^29): the column stays inside
    // [-13 * 2^58 - 13 * 2^58, 13 * 2^58 + 6.37 * 2^58] = [-2^62.7, 2^62.3]; with the subtracted quotient of mul() the lower end
    // would be -32.4 * 2^58 < -2^63.  Result normalised, in (-0.002 q, 1.002 q).
    SV_HD static fql_t diff_of_products(const fql_t& a, const fql_t& b, const fql_t& c, const fql_t& d) {
        uint32_t m[FqL::STEPS];
        fql_t r;
        int64_t acc = 0;
#pragma unroll
        for (int k = 0; k < N + FqL::STEPS; k++) {
#pragma unroll
            for (int i = 0; i < N; i++) {
                const int j = k - i;
                if (j >= 0 && j < N) {
                    acc += (int64_t)a.v[i] * b.v[j];
                    acc -= (int64_t)c.v[i] * d.v[j];
                }
            }
#pragma unroll
            for (int i = 0; i < FqL::STEPS; i++) {
                const int j = k - i;
                if (j >= 1 && j < N && i < k) acc += (int64_t)(int32_t)m[i] * FqL::MOD[j];
            }
            if (k < FqL::STEPS) {
                m[k] = (0u - (uint32_t)acc) & MASK;
                acc += m[k];
            } else {
                r.v[k - FqL::STEPS] = (k == N + FqL::STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
            }
            acc >>= 29;
        }
        SV_OPAQUE_13(r.v);
        return r;
    }

    // ---- raw memory image of a normalised value: the two's-complement 384-bit integer sum v_i 2^(29 i) in 12 words (limbs 0 .. 11
    // fill bits 0 .. 347, the signed top limb bits 348 .. 379, sign-extended to 384).  How the accumulate kernel leaves its partial
    // sums: no arithmetic at the flush; g1_partials_to_exact_kernel converts them afterwards.
    SV_HD void store_raw(void* p) const {
        uint32_t w[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint32_t x = 0;
#pragma unroll
            for (int i = 0; i < N - 1; i++) {
                const int lo_bit = 29 * i - 32 * j;  // position of limb i's bit 0 inside word j
                if (lo_bit > -29 && lo_bit < 32) x |= (lo_bit >= 0) ? ((uint32_t)v[i] << lo_bit) : ((uint32_t)v[i] >> (-lo_bit));
            }
            w[j] = x;
        }
        w[10] |= (uint32_t)v[12] << 28;        // bit 348 = word 10, bit 28
        w[11] = (uint32_t)(v[12] >> 4);        // arithmetic: the sign fills the top four bits
        uint4* q = (uint4*)p;
#pragma unroll
        for (int i = 0; i < 3; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    }
    SV_HD static fql_t load_raw(const void* p) {
        uint32_t w[12];
        const uint4* q = (const uint4*)p;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const uint4 t = q[i];
            w[4 * i] = t.x, w[4 * i + 1] = t.y, w[4 * i + 2] = t.z, w[4 * i + 3] = t.w;
        }
        fql_t r;
#pragma unroll
        for (int i = 0; i < N - 1; i++) {
            const int bit = 29 * i, wi = bit / 32, sh = bit % 32;
            const uint32_t lo = w[wi], hi = (wi + 1 < 12) ? w[wi + 1] : 0u;
            r.v[i] = (int32_t)(((sh == 0) ? lo : ((lo >> sh) | (hi << (32 - sh)))) & MASK);
        }
        r.v[N - 1] = (int32_t)((w[10] >> 28) | (w[11] << 4));
        return r;
    }

    // ---- leaving the lazy domain: the canonical ff.hip.h value (internal form x 2^377, < q) of a normalised lazy value within
    // (-2^10 q, 2^10 q).  One product by 2^377 brings it into (-1.002 q, 0.002 q); then at most two additions of q.
    SV_HD fq_t to_exact() const {
        fql_t k;
#pragma unroll
        for (int i = 0; i < N; i++) k.v[i] = FqL::K377[i];
        fql_t t = mul(*this, k);
#pragma unroll 1
        for (int round = 0; round < 3 && t.v[N - 1] < 0; round++) {  // negative <=> top limb negative (normalised)
            int32_t c = 0;
#pragma unroll
            for (int i = 0; i < N - 1; i++) {
                const int32_t x = t.v[i] + FqL::MOD[i] + c;
                t.v[i] = (int32_t)((uint32_t)x & MASK);
                c = x >> 29;
            }
            t.v[N - 1] += FqL::MOD[N - 1] + c;
        }
        uint32_t w[N];
#pragma unroll
        for (int i = 0; i < N; i++) w[i] = (uint32_t)t.v[i];
        return fq_t::cond_sub(w);  // [0, 0.002 q) or [0, q) already; cond_sub covers a value in [q, 2 q)
    }
    // the exact internal form (x 2^377, canonical) -> lazy
    SV_HD static fql_t from_exact(const fq_t& a) {
        fql_t k;
#pragma unroll
        for (int i = 0; i < N; i++) k.v[i] = FqL::K435[i];
        return mul(from_limbs(a), k);
    }
};

// A base slot of a G1 MSM on this arithmetic: the 128 bytes of a g1_aff_mem_t reinterpreted as 26 limbs - x then y, canonical
// residues of the coordinates times 2^406, one 29-bit limb per word: the gather needs no unpacking (78 instructions per addition
// with the packed 2 x 12-word image) - and a flag word for the point at infinity (a 26-limb zero test otherwise).
struct alignas(128) g1_lazy_slot_t {
    uint32_t w[32];  // [0, 13): x limbs, [13, 26): y limbs, [26]: 1 = point at infinity, [27, 32): unused
    SV_HD void coords(fql_t& px, fql_t& py) const {
#pragma unroll
        for (int i = 0; i < 13; i++) px.v[i] = (int32_t)w[i], py.v[i] = (int32_t)w[13 + i];
        SV_OPAQUE_13(px.v);
        SV_OPAQUE_13(py.v);
    }
    // x406, y406: canonical residues (exact-arithmetic values whose limbs are read as plain integers)
    SV_HD static void store(g1_aff_mem_t* slot, const fq_t& x406, const fq_t& y406, bool inf) {
        uint32_t* o = (uint32_t*)slot;
        uint4* q = (uint4*)slot;
        uint32_t t[28];
#pragma unroll
        for (int i = 0; i < 13; i++) t[i] = inf ? 0u : x406.v[i], t[13 + i] = inf ? 0u : y406.v[i];
        t[26] = inf ? 1u : 0u;
        t[27] = 0;
#pragma unroll
        for (int i = 0; i < 7; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        (void)o;
    }
};
static_assert(sizeof(g1_lazy_slot_t) == sizeof(g1_aff_mem_t), "a lazy base slot overlays the exact one");

// ------------------------------------------------------------------------------------------------------------------------
// XYZZ accumulator on the lazy arithmetic.  Infinity is a flag (zz is never tested for zero inside the loop).
// ------------------------------------------------------------------------------------------------------------------------
struct alignas(16) g1_lazy_partial_t {  // raw partial sum of the lazy accumulate kernel: x, y, zz, zzz limbs (4 x 13 words)
    int32_t w[52];
};
struct xyzz_lazy_t {
    fql_t x, y, zz, zzz;  // normalised; x within (-1.1 q, 3.1 q), y within (-0.01 q, 1.01 q), zz / zzz within (-1.01 q, 0.01 q)
    bool inf;

    SV_HD static xyzz_lazy_t infinity() {
        xyzz_lazy_t r;
        r.x = r.y = r.zz = r.zzz = fql_t::zero();
        r.inf = true;
        return r;
    }
    SV_HD g1_xyzz_t to_exact() const {
        if (inf) return g1_xyzz_t::inf();
        return {x.to_exact(), y.to_exact(), zz.to_exact(), zzz.to_exact()};
    }
    // raw partial sum: the 52 limbs as they are (208 bytes, no packing - the flush runs in most iterations of a wave, see
    // msm_accumulate_lazy_kernel); infinity = all zero (zz = 0 converts to the exact zz = 0)
    SV_HD void store_raw(g1_lazy_partial_t* p) const {
        uint4* q = (uint4*)p;
        if (inf) {
#pragma unroll
            for (int i = 0; i < 13; i++) q[i] = make_uint4(0, 0, 0, 0);
            return;
        }
        int32_t t[52];
#pragma unroll
        for (int i = 0; i < 13; i++) t[i] = x.v[i], t[13 + i] = y.v[i], t[26 + i] = zz.v[i], t[39 + i] = zzz.v[i];
#pragma unroll
        for (int i = 0; i < 13; i++) q[i] = make_uint4((uint32_t)t[4 * i], (uint32_t)t[4 * i + 1], (uint32_t)t[4 * i + 2], (uint32_t)t[4 * i + 3]);
    }
    SV_HD static g1_xyzz_t exact_from_raw(const g1_lazy_partial_t* p) {
        const uint4* q = (const uint4*)p;
        int32_t t[52];
#pragma unroll
        for (int i = 0; i < 13; i++) {
            const uint4 u = q[i];
            t[4 * i] = (int32_t)u.x, t[4 * i + 1] = (int32_t)u.y, t[4 * i + 2] = (int32_t)u.z, t[4 * i + 3] = (int32_t)u.w;
        }
        fql_t c[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int i = 0; i < 13; i++) c[k].v[i] = t[13 * k + i];
        return {c[0].to_exact(), c[1].to_exact(), c[2].to_exact(), c[3].to_exact()};
    }
    SV_HD static xyzz_lazy_t from_exact(const g1_xyzz_t& p) {
        xyzz_lazy_t r;
        r.inf = p.is_inf();
        if (r.inf) return infinity();
        r.x = fql_t::from_exact(p.x);
        r.y = fql_t::from_exact(p.y);
        r.zz = fql_t::from_exact(p.zz);
        r.zzz = fql_t::from_exact(p.zzz);
        return r;
    }
    // this += (px, py) [negate: -(px, py)]; px, py: canonical residues of the affine coordinates times 2^406 (normalised limbs);
    // the caller has excluded the point at infinity.  madd-2008-s:  8 M + 2 S, Y3 as one two-product reduction.
    // Returns false when the addition is exceptional (this == +-P: the x coordinates agree) - the caller resolves it on the
    // exact arithmetic; the accumulator is unchanged then.  The test P = U2 - X1 == 0 (mod q) runs on the low limb only:
    // P is a difference of two normalised values within (-4.2 q, 1.2 q), so P = 0 (mod q) means P in {-4 q, ..., q}, and
    // q = 1 (mod 2^29) turns that into "P mod 2^29 in {-4, ..., 1}": a necessary condition that a random P meets with
    // probability 6 * 2^-29 (a false alarm only costs the slow path).
    SV_HD bool madd(const fql_t& px, const fql_t& py, bool negate) {
        if (inf) {
            x = px;
            y = py.negate_if(negate).normalized();
            // ONE = 2^406 mod q as a canonical residue
            zz = zzz = one();
            inf = false;
            return true;
        }
        const fql_t u2 = fql_t::mul(px, zz);
        const fql_t s2 = fql_t::mul(py.negate_if(negate), zzz);
        const fql_t p = u2 - x;  // limbs in (-2^29, 2^29)
        const fql_t r = s2 - y;
        if ((((uint32_t)p.v[0] + 4u) & fql_t::MASK) <= 5u) return false;
        const fql_t pp = fql_t::sqr(p);
        const fql_t ppp = fql_t::mul(p, pp);
        const fql_t q = fql_t::mul(x, pp);
        const fql_t rr = fql_t::sqr(r);
        // X3 = R^2 - PPP - 2 Q, normalised on the way (limb sums within (-3 * 2^29, 2^29))
        fql_t x3;
        {
            int32_t c = 0;
#pragma unroll
            for (int i = 0; i < fql_t::N - 1; i++) {
                const int32_t t = rr.v[i] - ppp.v[i] - 2 * q.v[i] + c;
                x3.v[i] = (int32_t)((uint32_t)t & fql_t::MASK);
                c = t >> 29;
            }
            x3.v[fql_t::N - 1] = rr.v[fql_t::N - 1] - ppp.v[fql_t::N - 1] - 2 * q.v[fql_t::N - 1] + c;
            SV_OPAQUE_13(x3.v);
        }
        y = fql_t::diff_of_products(r, q - x3, y, ppp);
        x = x3;
        zz = fql_t::mul(zz, pp);
        zzz = fql_t::mul(zzz, ppp);
        return true;
    }
    SV_HD static fql_t one() {  // 2^406 mod q
        fql_t r;
#pragma unroll
        for (int i = 0; i < fql_t::N; i++) r.v[i] = (int32_t)FqLConv::C406[i];
        return r;
    }
};

// ------------------------------------------------------------------------------------------------------------------------
// The same arithmetic behind field-like operators, for the TAIL of a G1 MSM (round 4; tuning lazy_tail): the reduce rounds, the
// bucket merge, the fold and the bit-plane kernels are ec.hip.h's generic xyzz_t<F> (add-2008-s, dbl-2008-s-1) and msm.hip.h's
// quad-cooperative addition instantiated with F = fqz_t, reading the raw partial sums of the accumulate kernel as they are (no
// conversion pass) and leaving raw bit planes that the host converts (<= ~270 points).
//   * every operator returns a NORMALISED value (sums and differences propagate their carries: 13 steps, no conditional
//     subtraction), so every product is normalised x normalised;
//   * ranges (multiples of q; the int32 top limb holds up to 4.7 q, q / 2^348 = 0.84 * 2^29): products in (-1.002, 0.002);
//     P = U2 - U1, R = S2 - S1 in (-1.004, 1.004); X3 = R^2 - PPP - 2 Q in (-1.01, 3.01); Q - X3 in (-4.02, 1.02);
//     Y3 (one two-product reduction) in (-0.002, 1.002), as the difference of two products (quad_add) in (-1.004, 1.004);
//     doubling: 3 X^2 in (-3.006, 0.006), S - X3 in (-3.01, 1.01) - all inside what xyzz_lazy_t::madd feeds the same routines;
//   * x == 0 (mod q) is decided on the low limb first (q = 1 mod 2^29: k q = k mod 2^29, |k| <= 8 here) and exactly only when that
//     filter passes (17 * 2^-29 of the random cases); the point at infinity is zz = 0 as 13 zero limbs (how the accumulate kernel
//     writes an empty partial sum, how a zeroed sink starts) or any other representative of 0.
// ------------------------------------------------------------------------------------------------------------------------
struct fqz_t {
    fql_t a;
    struct mem_t {
        uint32_t w[13];
    };
    static constexpr int N = 13;

    SV_HD static fqz_t zero() { return {fql_t::zero()}; }
    SV_HD static fqz_t one() { return {xyzz_lazy_t::one()}; }
    SV_HD static fqz_t load(const mem_t* p) {
        fqz_t r;
#pragma unroll
        for (int i = 0; i < N; i++) r.a.v[i] = (int32_t)p->w[i];
        return r;
    }
    SV_HD void store(mem_t* p) const {
#pragma unroll
        for (int i = 0; i < N; i++) p->w[i] = (uint32_t)a.v[i];
    }
    SV_HD fqz_t operator*(const fqz_t& b) const { return {fql_t::mul(a, b.a)}; }
    SV_HD fqz_t sqr() const { return {fql_t::sqr(a)}; }
    SV_HD fqz_t operator+(const fqz_t& b) const { return {(a + b.a).normalized()}; }
    SV_HD fqz_t operator-(const fqz_t& b) const { return {(a - b.a).normalized()}; }
    SV_HD fqz_t dbl() const { return {(a + a).normalized()}; }
    SV_HD fqz_t neg() const { return {(fql_t::zero() - a).normalized()}; }
    SV_HD static fqz_t diff_of_products(const fqz_t& x, const fqz_t& y, const fqz_t& z, const fqz_t& w) {
        return {fql_t::diff_of_products(x.a, y.a, z.a, w.a)};
    }
    SV_HD fq_t to_exact() const { return a.to_exact(); }
    SV_HD bool is_zero() const {
        int32_t any = 0;
#pragma unroll
        for (int i = 0; i < N; i++) any |= a.v[i];
        if (any == 0) return true;
        if ((((uint32_t)a.v[0] + 8u) & fql_t::MASK) > 16u) return false;
        return a.to_exact().is_zero();
    }
};

}  // namespace sv
