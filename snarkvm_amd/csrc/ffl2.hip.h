// ffl2.hip.h - lazily reduced Fq2 = Fq[u] / (u^2 + 5) on the signed limbs of ffl.hip.h, for the G2 bucket-accumulation loop (round 4).
//
// The G2 accumulate kernel ran on canonical arithmetic: 18 279 instructions per mixed addition for 7 488 multiply-adds, 256 VGPRs +
// 215 AGPRs (profiles/r03_g2.md).  Round 3 stopped at the factor 5 of the non-residue: "5 x a value within q overflows the signed top
// limb".  It does - so the factor never meets a reduced value here: it is folded into an OPERAND as a 14-limb normalised integer
// (times5: one shift-add and one carry per limb), and the product c0 = a0 b0 - a1 (5 b1) is ONE two-product reduction.
//
// Every product operand is "tight": normalised limbs (0 .. 11 in [0, 2^29), limb 12 signed) and a value within [-1, 1] q (+- 2^-26 q),
// i.e. every limb below 2^29 in magnitude.  That makes the column bounds uniform:
//   c1 = a0 b1 + a1 b0   26 products of non-negative limbs (the four top-limb products of a column may be negative), quotient
//                        SUBTRACTED (ffl.hip.h mul): column within [-(4 + 6.4) 2^58, 26 * 2^58] < 2^62.8.  Result in (-q - e, e).
//   c0 = a0 b0 - a1 B    B = 5 b1 (14 limbs), quotient ADDED (ffl.hip.h diff_of_products): column within
//                        [-13.1 * 2^58, (13.1 + 6.4) 2^58].  Result in (-e, q + e).
// with e < 2^-26 q (|a0 b0 - 5 a1 b1| / 2^406 < 7.3 q^2 / 2^406).  Sums and differences of such values are not tight; the addition law
// below normalises each one with the multiple of q that brings it back (a constant, a sign-dependent constant, or - for X3, five
// terms wide - a multiple chosen from the top limb), one carry pass each.  Value ranges, in units of q (D = (-e, 1 + e): a c0-type
// product, S = (-1 - e, e): a c1-type product):
//   U2 = x2 ZZ, S2 = +-y2 ZZZ      (D, S)        bases canonical; -y2 enters as q - y2, never as negated limbs
//   P = U2 - X1, R = S2 - Y1       c0: D - [0,1] = [-1,1];  c1: S - [0,1] + q = [-1,1]
//   PP, PPP, Q, RR, ZZ3, ZZZ3      (D, S)
//   X3 = RR - PPP - 2 Q            c0 in [-3,1], c1 in [-1,3]  -> + k q, k from the top limb -> [0,1]
//   Y3 = R (Q - X3) - Y1 PPP       two Fq2 products (a four-product column would not fit 64 bits); D - D, S - S = [-1,1] -> + q if negative -> [0,1]
// so X and Y of the accumulator stay within [0,1], ZZ and ZZZ are raw products, and every operand above is tight.
//
// Exceptional cases (x coordinates agree): P = 0 in Fq2 means both components are in {-q, 0, q}, i.e. low limbs in {-1, 0, 1}
// (q = 1 mod 2^29): a two-limb filter, false alarms at 9 * 2^-58 per addition; the caller resolves them on the exact arithmetic.
// Host twin: snarkvm_hip_selftest_fq2_lazy (tests/test_host_arith.py).
#pragma once
#include "ffl.hip.h"

namespace sv {

struct fq2l_t {
    fql_t c0, c1;
};

namespace fq2l {

static constexpr int N = 13;
static constexpr int STEPS = FqL::STEPS;  // 14: R = 2^406
static constexpr uint32_t MASK = fql_t::MASK;

// 5 b as 14 normalised limbs (limb 13: the small signed overflow).  b tight.
SV_HD void times5(const fql_t& b, int32_t* o) {
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        const uint32_t x = 5u * (uint32_t)b.v[i] + c;  // < 5 * 2^29 + 5 < 2^32
        o[i] = (int32_t)(x & MASK);
        c = x >> 29;
    }
    const int64_t t = 5 * (int64_t)b.v[N - 1] + (int64_t)c;
    o[N - 1] = (int32_t)((uint32_t)t & MASK);
    o[N] = (int32_t)(t >> 29);
}
// (a0 b0 - a1 B) / 2^406, B of 14 limbs (times5).  Quotient added.  Result normalised, in (-e, q + e).
SV_HD fql_t mul_diff(const fql_t& a0, const fql_t& b0, const fql_t& a1, const int32_t* B) {
    uint32_t m[STEPS];
    int32_t na1[N];  // the subtracted products as multiply-adds of the negated operand (a 64-bit subtraction per product otherwise)
#pragma unroll
    for (int i = 0; i < N; i++) na1[i] = -a1.v[i];
    fql_t r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N + STEPS; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int j = k - i;
            if (j >= 0 && j < N) acc += (int64_t)a0.v[i] * b0.v[j];
            if (j >= 0 && j <= N) acc += (int64_t)na1[i] * B[j];
        }
#pragma unroll
        for (int i = 0; i < STEPS; i++) {
            const int j = k - i;
            if (j >= 1 && j < N && i < k) acc += (int64_t)(int32_t)m[i] * FqL::MOD[j];
        }
        if (k < STEPS) {
            m[k] = (0u - (uint32_t)acc) & MASK;
            acc += m[k];
        } else {
            r.v[k - STEPS] = (k == N + STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
        }
        acc >>= 29;  // arithmetic
    }
    SV_OPAQUE_13(r.v);
    return r;
}
// (a0 b1 + a1 b0) / 2^406.  Quotient subtracted.  Result normalised, in (-q - e, e).
SV_HD fql_t mul_sum(const fql_t& a0, const fql_t& b1, const fql_t& a1, const fql_t& b0) {
    uint32_t m[STEPS];
    fql_t r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N + STEPS; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int j = k - i;
            if (j >= 0 && j < N) {
                acc += (int64_t)a0.v[i] * b1.v[j];
                acc += (int64_t)a1.v[i] * b0.v[j];
            }
        }
#pragma unroll
        for (int i = 0; i < STEPS; i++) {
            const int j = k - i;
            if (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FqL::MOD[j];
        }
        if (k < STEPS) {
            m[k] = (uint32_t)acc & MASK;
        } else {
            r.v[k - STEPS] = (k == N + STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
        }
        acc >>= 29;
    }
    SV_OPAQUE_13(r.v);
    return r;
}
// (a0^2 - a1 A) / 2^406, A = 5 a1 (14 limbs): the square's off-diagonal products once against the doubled lower-indexed limb.
SV_HD fql_t sqr_diff(const fql_t& a0, const fql_t& a1, const int32_t* A) {
    uint32_t m[STEPS];
    int32_t d[N], na1[N];
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = a0.v[i] << 1, na1[i] = -a1.v[i];
    fql_t r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N + STEPS; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int j = k - i;
            if (j > i && j < N) acc += (int64_t)d[i] * a0.v[j];
            if (j == i) acc += (int64_t)a0.v[i] * a0.v[i];
            if (j >= 0 && j <= N) acc += (int64_t)na1[i] * A[j];
        }
#pragma unroll
        for (int i = 0; i < STEPS; i++) {
            const int j = k - i;
            if (j >= 1 && j < N && i < k) acc += (int64_t)(int32_t)m[i] * FqL::MOD[j];
        }
        if (k < STEPS) {
            m[k] = (0u - (uint32_t)acc) & MASK;
            acc += m[k];
        } else {
            r.v[k - STEPS] = (k == N + STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
        }
        acc >>= 29;
    }
    SV_OPAQUE_13(r.v);
    return r;
}
// 2 a0 a1 / 2^406 (every limb of a0 doubled: tight operands keep the top limb below 2^29).  Quotient subtracted: (-q - e, e).
SV_HD fql_t mul_twice(const fql_t& a0, const fql_t& a1) {
    uint32_t m[STEPS];
    int32_t d[N];
#pragma unroll
    for (int i = 0; i < N; i++) d[i] = a0.v[i] << 1;
    fql_t r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N + STEPS; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int j = k - i;
            if (j >= 0 && j < N) acc += (int64_t)d[i] * a1.v[j];
        }
#pragma unroll
        for (int i = 0; i < STEPS; i++) {
            const int j = k - i;
            if (j >= 1 && j < N && i < k) acc -= (int64_t)(int32_t)m[i] * FqL::MOD[j];
        }
        if (k < STEPS) {
            m[k] = (uint32_t)acc & MASK;
        } else {
            r.v[k - STEPS] = (k == N + STEPS - 1) ? (int32_t)acc : (int32_t)((uint32_t)acc & MASK);
        }
        acc >>= 29;
    }
    SV_OPAQUE_13(r.v);
    return r;
}
SV_HD fq2l_t mul(const fq2l_t& a, const fq2l_t& b) {
    int32_t B[N + 1];
    times5(b.c1, B);
    return {mul_diff(a.c0, b.c0, a.c1, B), mul_sum(a.c0, b.c1, a.c1, b.c0)};
}
SV_HD fq2l_t sqr(const fq2l_t& a) {
    int32_t A[N + 1];
    times5(a.c1, A);
    return {sqr_diff(a.c0, a.c1, A), mul_twice(a.c0, a.c1)};
}
// a - b + add * q, carry-normalised.  add: 0, 1, or -1 for "q if the difference is negative" (decided on the top limbs: a carry
// from below can only matter within 2^-28 q of zero, where either choice keeps the value tight).
SV_HD fql_t sub_norm(const fql_t& a, const fql_t& b, int add) {
    int32_t mask = add > 0 ? -1 : 0;
    if (add < 0) mask = (a.v[N - 1] - b.v[N - 1]) < 0 ? -1 : 0;
    fql_t r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        const int32_t x = a.v[i] - b.v[i] + (FqL::MOD[i] & mask) + c;
        r.v[i] = (int32_t)((uint32_t)x & MASK);
        c = x >> 29;
    }
    r.v[N - 1] = a.v[N - 1] - b.v[N - 1] + (FqL::MOD[N - 1] & mask) + c;
    SV_OPAQUE_13(r.v);
    return r;
}
// rr - ppp - 2 qq + k q with the k that brings the value into [0, q): k = -floor(top / q_top) from the un-normalised top limb
// (|top| < 4 q_top: exact in float; a carry from below moves the result by < 2^-28 q).  Operands normalised.
SV_HD fql_t x3_norm(const fql_t& rr, const fql_t& ppp, const fql_t& qq) {
    const int32_t top = rr.v[N - 1] - ppp.v[N - 1] - 2 * qq.v[N - 1];
    const float kf = floorf((float)top * (1.0f / (float)FqL::MOD[N - 1]));
    const int32_t k = -(int32_t)kf;
    fql_t r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        const int64_t x = (int64_t)(rr.v[i] - ppp.v[i] - 2 * qq.v[i]) + (int64_t)k * FqL::MOD[i] + c;
        r.v[i] = (int32_t)((uint32_t)x & MASK);
        c = x >> 29;
    }
    r.v[N - 1] = (int32_t)((int64_t)top + (int64_t)k * FqL::MOD[N - 1] + c);
    SV_OPAQUE_13(r.v);
    return r;
}
// q - y for a canonical y (the negated base coordinate as a NON-NEGATIVE normalised value), or y itself
SV_HD fql_t cond_neg_canonical(const fql_t& y, bool neg) {
    const int32_t m = neg ? -1 : 0;
    fql_t r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        const int32_t x = ((y.v[i] ^ m) - m) + (FqL::MOD[i] & m) + c;
        r.v[i] = (int32_t)((uint32_t)x & MASK);
        c = x >> 29;
    }
    r.v[N - 1] = ((y.v[N - 1] ^ m) - m) + (FqL::MOD[N - 1] & m) + c;
    SV_OPAQUE_13(r.v);
    return r;
}

}  // namespace fq2l

// A base slot of a G2 MSM on this arithmetic: the 256 bytes of an aff_mem_t<fq2_t> reinterpreted as four groups of 16 words - x.c0, x.c1,
// y.c0, y.c1: 13 limbs each (canonical residues of the coordinate components times 2^406, one 29-bit limb per word) and three words of
// padding, so that a lane of the pair kernel (ffl2p.hip.h) reads ITS component of a coordinate as four aligned 16-byte loads - and a flag
// word for the point at infinity in the first group's padding.
struct alignas(128) g2_lazy_slot_t {
    static constexpr int GROUP = 16, INF_WORD = 15;
    uint32_t w[64];  // [16 g, 16 g + 13): limbs of component g (x.c0, x.c1, y.c0, y.c1); w[15]: 1 = point at infinity
    SV_HD void coords(fq2l_t& px, fq2l_t& py) const {
#pragma unroll
        for (int i = 0; i < 13; i++) {
            px.c0.v[i] = (int32_t)w[i], px.c1.v[i] = (int32_t)w[GROUP + i];
            py.c0.v[i] = (int32_t)w[2 * GROUP + i], py.c1.v[i] = (int32_t)w[3 * GROUP + i];
        }
        SV_OPAQUE_13(px.c0.v);
        SV_OPAQUE_13(px.c1.v);
        SV_OPAQUE_13(py.c0.v);
        SV_OPAQUE_13(py.c1.v);
    }
    // component `comp` (0: c0, 1: c1) of x and y
    SV_HD void component(int comp, fql_t& px, fql_t& py) const {
        const uint4* q = (const uint4*)w;
        uint32_t t[32];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint4 a = q[4 * comp + k], b = q[8 + 4 * comp + k];
            t[4 * k] = a.x, t[4 * k + 1] = a.y, t[4 * k + 2] = a.z, t[4 * k + 3] = a.w;
            t[16 + 4 * k] = b.x, t[16 + 4 * k + 1] = b.y, t[16 + 4 * k + 2] = b.z, t[16 + 4 * k + 3] = b.w;
        }
#pragma unroll
        for (int i = 0; i < 13; i++) px.v[i] = (int32_t)t[i], py.v[i] = (int32_t)t[16 + i];
        SV_OPAQUE_13(px.v);
        SV_OPAQUE_13(py.v);
    }
    // x406, y406: canonical residues of coordinate * 2^406 (exact-arithmetic values whose limbs are read as plain integers)
    SV_HD static void store(aff_mem_t<fq2_t>* slot, const fq2_t& x406, const fq2_t& y406, bool inf) {
        uint4* q = (uint4*)slot;
        uint32_t t[64];
#pragma unroll
        for (int i = 0; i < 64; i++) t[i] = 0;
#pragma unroll
        for (int i = 0; i < 13; i++) {
            t[i] = inf ? 0u : x406.c0.v[i], t[GROUP + i] = inf ? 0u : x406.c1.v[i];
            t[2 * GROUP + i] = inf ? 0u : y406.c0.v[i], t[3 * GROUP + i] = inf ? 0u : y406.c1.v[i];
        }
        t[INF_WORD] = inf ? 1u : 0u;
#pragma unroll
        for (int i = 0; i < 16; i++) q[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
    }
};
static_assert(sizeof(g2_lazy_slot_t) == sizeof(aff_mem_t<fq2_t>), "a lazy G2 base slot overlays the exact one");

struct alignas(16) g2_lazy_partial_t {  // raw partial sum of the lazy G2 accumulate kernel: x, y, zz, zzz as (c0, c1) limb vectors (8 x 13 words)
    int32_t w[104];
};
// canonical residue of v * 2^406 as a lazy value, from the exact internal form
SV_HD fql_t fql_canonical_from_exact(const fq_t& a) { return fql_t::from_limbs(a * fq_t::from_table(FqLConv::C406)); }

struct xyzz_lazy2_t {
    fq2l_t x, y, zz, zzz;  // x, y components within [0, 1] q; zz, zzz raw products: c0 in (-e, 1 + e) q, c1 in (-1 - e, e) q
    bool inf;

    SV_HD static xyzz_lazy2_t infinity() {
        xyzz_lazy2_t r;
        r.x.c0 = r.x.c1 = r.y.c0 = r.y.c1 = r.zz.c0 = r.zz.c1 = r.zzz.c0 = r.zzz.c1 = fql_t::zero();
        r.inf = true;
        return r;
    }
    SV_HD xyzz_t<fq2_t> to_exact() const {
        if (inf) return xyzz_t<fq2_t>::inf();
        return {{x.c0.to_exact(), x.c1.to_exact()}, {y.c0.to_exact(), y.c1.to_exact()}, {zz.c0.to_exact(), zz.c1.to_exact()}, {zzz.c0.to_exact(), zzz.c1.to_exact()}};
    }
    SV_HD static xyzz_lazy2_t from_exact(const xyzz_t<fq2_t>& p) {
        if (p.is_inf()) return infinity();
        xyzz_lazy2_t r;
        r.inf = false;
        r.x = {fql_canonical_from_exact(p.x.c0), fql_canonical_from_exact(p.x.c1)};
        r.y = {fql_canonical_from_exact(p.y.c0), fql_canonical_from_exact(p.y.c1)};
        r.zz = {fql_canonical_from_exact(p.zz.c0), fql_canonical_from_exact(p.zz.c1)};
        r.zzz = {fql_canonical_from_exact(p.zzz.c0), fql_canonical_from_exact(p.zzz.c1)};
        return r;
    }
    // raw partial sum: the 104 limbs as they are; infinity = all zero (zz = 0 converts to the exact zz = 0)
    SV_HD void store_raw(g2_lazy_partial_t* p) const {
        uint4* q = (uint4*)p;
        if (inf) {
#pragma unroll
            for (int i = 0; i < 26; i++) q[i] = make_uint4(0, 0, 0, 0);
            return;
        }
        int32_t t[104];
#pragma unroll
        for (int i = 0; i < 13; i++) {
            t[i] = x.c0.v[i], t[13 + i] = x.c1.v[i], t[26 + i] = y.c0.v[i], t[39 + i] = y.c1.v[i];
            t[52 + i] = zz.c0.v[i], t[65 + i] = zz.c1.v[i], t[78 + i] = zzz.c0.v[i], t[91 + i] = zzz.c1.v[i];
        }
#pragma unroll
        for (int i = 0; i < 26; i++) q[i] = make_uint4((uint32_t)t[4 * i], (uint32_t)t[4 * i + 1], (uint32_t)t[4 * i + 2], (uint32_t)t[4 * i + 3]);
    }
    SV_HD static xyzz_t<fq2_t> exact_from_raw(const g2_lazy_partial_t* p) {
        const uint4* q = (const uint4*)p;
        int32_t t[104];
#pragma unroll
        for (int i = 0; i < 26; i++) {
            const uint4 u = q[i];
            t[4 * i] = (int32_t)u.x, t[4 * i + 1] = (int32_t)u.y, t[4 * i + 2] = (int32_t)u.z, t[4 * i + 3] = (int32_t)u.w;
        }
        fq_t e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            fql_t c;
#pragma unroll
            for (int i = 0; i < 13; i++) c.v[i] = t[13 * k + i];
            e[k] = c.to_exact();
        }
        return {{e[0], e[1]}, {e[2], e[3]}, {e[4], e[5]}, {e[6], e[7]}};
    }
    SV_HD static fq2l_t one() {  // (2^406 mod q, 0)
        fq2l_t r;
#pragma unroll
        for (int i = 0; i < 13; i++) r.c0.v[i] = (int32_t)FqLConv::C406[i], r.c1.v[i] = 0;
        return r;
    }
    // this += (px, py) [negate: -(px, py)]; px, py: canonical residues of the affine coordinates times 2^406; the caller has excluded the
    // point at infinity.  Returns false when the addition is exceptional (the caller resolves it on the exact arithmetic).
    SV_HD bool madd(const fq2l_t& px, const fq2l_t& py, bool negate) {
        const fq2l_t ny = {fq2l::cond_neg_canonical(py.c0, negate), fq2l::cond_neg_canonical(py.c1, negate)};
        if (inf) {
            x = px;
            y = ny;
            zz = zzz = one();
            inf = false;
            return true;
        }
        const fq2l_t u2 = fq2l::mul(px, zz);
        const fq2l_t s2 = fq2l::mul(ny, zzz);
        const fq2l_t p = {fq2l::sub_norm(u2.c0, x.c0, 0), fq2l::sub_norm(u2.c1, x.c1, 1)};
        const fq2l_t r = {fq2l::sub_norm(s2.c0, y.c0, 0), fq2l::sub_norm(s2.c1, y.c1, 1)};
        if ((((uint32_t)p.c0.v[0] + 1u) & fql_t::MASK) <= 2u && (((uint32_t)p.c1.v[0] + 1u) & fql_t::MASK) <= 2u) return false;
        const fq2l_t pp = fq2l::sqr(p);
        const fq2l_t ppp = fq2l::mul(p, pp);
        const fq2l_t q = fq2l::mul(x, pp);
        const fq2l_t rr = fq2l::sqr(r);
        const fq2l_t x3 = {fq2l::x3_norm(rr.c0, ppp.c0, q.c0), fq2l::x3_norm(rr.c1, ppp.c1, q.c1)};
        const fq2l_t d = {fq2l::sub_norm(q.c0, x3.c0, 0), fq2l::sub_norm(q.c1, x3.c1, 1)};
        const fq2l_t a = fq2l::mul(r, d);
        const fq2l_t b = fq2l::mul(y, ppp);
        y = {fq2l::sub_norm(a.c0, b.c0, -1), fq2l::sub_norm(a.c1, b.c1, -1)};
        x = x3;
        zz = fq2l::mul(zz, pp);
        zzz = fq2l::mul(zzz, ppp);
        return true;
    }
};

}  // namespace sv
