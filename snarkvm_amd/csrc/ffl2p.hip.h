// ffl2p.hip.h - Fq2 = Fq[u] / (u^2 + 5) split over a LANE PAIR, for the G2 bucket-accumulation loop (round 5).
//
// ffl2.hip.h holds both components of every Fq2 value in one lane: the accumulator, the base point and the live temporaries of the
// addition law are 398 VGPRs + 142 AGPRs (profiles/r04_g2.md) - one wave per SIMD and ~410 register-file moves per addition.  Here the
// EVEN lane of a pair holds the c0 component of every value and the ODD lane the c1 component (fields/src/fp2.rs:404-410):
//     c0 = a0 b0 - a1 (5 b1)        on the even lane
//     c1 = a0 b1 + a1 b0            on the odd lane
// Both are ONE two-product column sum with the quotient subtracted (ffl.hip.h mul), i.e. the two lanes run the SAME instruction stream:
//     r = X * own_b + Y * recv_b,   (X, Y) = even: (own_a, recv_a)   odd: (recv_a, own_a)
// where a lane SENDS its partner   even: ( a0,  b0 with a zero 14th limb)   odd: (-a1 as negated limbs,  5 b1 as 14 normalised limbs)
// through DPP quad permutations (quad_perm [1,0,3,2]: a register move inside the VALU, no LDS).  What a lane keeps alive: four
// 13-limb components of the accumulator plus the operands of one product - ~200 VGPRs, two waves per SIMD, no AGPR traffic.
// Squares are products (the pair has no cheaper square): 10 products per mixed addition, 519 multiply-adds each on each lane.
//
// Ranges (units of q; e < 2^-26): every product is (-1 - e, e) on BOTH lanes now (even lane: |a0 b0 - 5 a1 b1| < 6 q^2, minus m q with
// m < 2^406); product operands are "tight" (|value| <= 1 + 2 e: every limb below 2^29 in magnitude).  Column bounds: even lane
// [-(14 + 6.4 + 2) 2^58, 13 * 2^58], odd lane [-(4 + 6.4) 2^58, 26 * 2^58]: inside 64 bits.  The addition law normalises every sum or
// difference with the multiple of q that makes it tight again (ffl2.hip.h sub_norm / norm_k); X and Y of the accumulator stay within
// [0, 1 + 2 e], ZZ and ZZZ are raw products.
//
// Exceptional additions never leave this arithmetic: U2 - X1 = 0 in Fq2 is decided exactly (normalised values have ONE limb image per
// integer: compare with -q, 0, q) behind a low-limb filter, and resolved as the reference does (short_weierstrass_jacobian/affine.rs:
// 232-246 doubling / cancellation): R = 0 -> mdbl-2008-s-1 of the base point (ec.hip.h dbl_affine: the same representative the exact
// kernel produces), else the point at infinity.
//
// The routines are templates over an exchange policy: xp_dev (one lane per instance, DPP) on the device, xp_host (both lanes of a pair
// as arrays of two) on the host - snarkvm_hip_selftest_fq2_pair runs the identical source against the exact arithmetic.
#pragma once
#include "ffl2.hip.h"

namespace sv {
namespace fq2p {

static constexpr int N = 13;
static constexpr uint32_t MASK = fql_t::MASK;

struct xp_dev {
    static constexpr int NL = 1;
    __device__ __forceinline__ static bool odd(int) { return (threadIdx.x & 1u) != 0; }
    template <int K>
    __device__ __forceinline__ static void swap(const int32_t (&send)[1][K], int32_t (&recv)[1][K]) {
#pragma unroll
        for (int i = 0; i < K; i++) {
            int32_t v = __builtin_amdgcn_mov_dpp(send[0][i], 0xB1, 0xf, 0xf, true);  // quad_perm [1, 0, 3, 2]: the partner's register
            SV_OPAQUE_LIMB(v);  // keep the move a move (msm.hip.h: a DPP operand folded into its consumer misbehaved on gfx950 / ROCm 7.2)
            recv[0][i] = v;
        }
    }
    __device__ __forceinline__ static bool both(const bool (&f)[1]) {
        int o = __builtin_amdgcn_mov_dpp((int)f[0], 0xB1, 0xf, 0xf, true);
        SV_OPAQUE_LIMB(o);
        return f[0] && o != 0;
    }
};
struct xp_host {
    static constexpr int NL = 2;
    static bool odd(int l) { return l == 1; }
    template <int K>
    static void swap(const int32_t (&send)[2][K], int32_t (&recv)[2][K]) {
        for (int i = 0; i < K; i++) recv[0][i] = send[1][i], recv[1][i] = send[0][i];
    }
    static bool both(const bool (&f)[2]) { return f[0] && f[1]; }
};

// ---- per-lane pieces (pure: host and device) ------------------------------------------------------------------------------------------
struct opa_t {  // the a-operand as the column sum wants it: X meets the lane's own b limbs, Y the limbs its partner sent
    int32_t X[N], Y[N];
};
struct opb_t {
    fql_t own;
    int32_t recv[N + 1];
};
SV_HD void send_a(const fql_t& a, bool odd, int32_t* o) {  // even: a0   odd: -a1
    const int32_t m = odd ? -1 : 0;
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = (a.v[i] ^ m) - m;
}
SV_HD void send_b(const fql_t& b, bool odd, int32_t* o) {  // even: (b0, 0)   odd: 5 b1 as 14 normalised limbs
    int32_t t5[N + 1];
    fq2l::times5(b, t5);
#pragma unroll
    for (int i = 0; i < N; i++) o[i] = odd ? t5[i] : b.v[i];
    o[N] = odd ? t5[N] : 0;
}
// this lane's component of a * b, quotient subtracted: normalised, in (-q - e, e)
SV_HD fql_t mul_core(const opa_t& a, const opb_t& b) {
    uint32_t m[FqL::STEPS];
    fql_t r;
    int64_t acc = 0;
#pragma unroll
    for (int k = 0; k < N + FqL::STEPS; k++) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int j = k - i;
            if (j >= 0 && j < N) acc += (int64_t)a.X[i] * b.own.v[j];
            if (j >= 0 && j <= N) acc += (int64_t)a.Y[i] * b.recv[j];
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
        acc >>= 29;  // arithmetic
    }
    SV_OPAQUE_13(r.v);
    return r;
}
// t (13 raw limb sums, |value| < 8 q, |t_i| < 2^31) + k q with the k that brings the value into [0, q): k from the top limb (exact in
// float: |top| < 2^33; what the lower limbs carry moves the result by < 2^-26 q).  Normalised.
SV_HD fql_t norm_k(const int32_t* t) {
    const float kf = floorf((float)t[N - 1] * (1.0f / (float)FqL::MOD[N - 1]));
    const int32_t k = -(int32_t)kf;
    fql_t r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        const int64_t x = (int64_t)t[i] + (int64_t)k * FqL::MOD[i] + c;
        r.v[i] = (int32_t)((uint32_t)x & MASK);
        c = x >> 29;
    }
    r.v[N - 1] = (int32_t)((int64_t)t[N - 1] + (int64_t)k * FqL::MOD[N - 1] + c);
    SV_OPAQUE_13(r.v);
    return r;
}
// v = 0 (mod q) for a normalised v within (-2 q, 2 q): v is -q, 0 or q, and a normalised value has one limb image
SV_HD bool is_zero_mod_q(const fql_t& v) {
    int32_t z = 0, p = 0, n = 0;  // differences to 0, q, -q
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        z |= v.v[i];
        p |= v.v[i] ^ FqL::MOD[i];
        // -q normalised: limbs of (2^377-ish two's complement): computed on the fly as the carry chain of 0 - q
        const int32_t x = -FqL::MOD[i] + c;
        const int32_t nq = (i < N - 1) ? (int32_t)((uint32_t)x & MASK) : x;
        c = x >> 29;
        n |= v.v[i] ^ nq;
    }
    return z == 0 || p == 0 || n == 0;
}
SV_HD fql_t one_component(bool odd) {  // (2^406 mod q, 0)
    fql_t r;
#pragma unroll
    for (int i = 0; i < N; i++) r.v[i] = odd ? 0 : (int32_t)FqLConv::C406[i];
    return r;
}

// ---- the pair: NL lanes side by side (device: this lane; host twin: both) -----------------------------------------------------------------
template <class XP>
struct pair_ops {
    static constexpr int NL = XP::NL;
    typedef fql_t val_t[XP::NL];
    SV_HD static void prep_a(const val_t& v, opa_t (&o)[XP::NL]) {
        int32_t s[XP::NL][N], r[XP::NL][N];
#pragma unroll
        for (int l = 0; l < NL; l++) send_a(v[l], XP::odd(l), s[l]);
        XP::template swap<N>(s, r);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            const bool odd = XP::odd(l);  // even: (own, received) = (a0, -a1)   odd: (received, own) = (a0, a1)
#pragma unroll
            for (int i = 0; i < N; i++) o[l].X[i] = odd ? r[l][i] : v[l].v[i], o[l].Y[i] = odd ? v[l].v[i] : r[l][i];
        }
    }
    SV_HD static void prep_b(const val_t& v, opb_t (&o)[XP::NL]) {
        int32_t s[XP::NL][N + 1], r[XP::NL][N + 1];
#pragma unroll
        for (int l = 0; l < NL; l++) send_b(v[l], XP::odd(l), s[l]);
        XP::template swap<N + 1>(s, r);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            o[l].own = v[l];
#pragma unroll
            for (int i = 0; i <= N; i++) o[l].recv[i] = r[l][i];
        }
    }
    SV_HD static void mul(const opa_t (&a)[XP::NL], const opb_t (&b)[XP::NL], val_t& out) {
#pragma unroll
        for (int l = 0; l < NL; l++) out[l] = mul_core(a[l], b[l]);
    }
};

// XYZZ accumulator of a lane pair.  `inf` is pair-uniform.
template <class XP>
struct xyzz_pair_t {
    typedef pair_ops<XP> ops;
    static constexpr int NL = XP::NL;
    fql_t x[XP::NL], y[XP::NL], zz[XP::NL], zzz[XP::NL];
    bool inf;

    // this += (px, py) [negate: -(px, py)]: this lane's components of the canonical residues of the affine coordinates times 2^406; the
    // caller has excluded the point at infinity.  Every case of the addition law is resolved here.
    SV_HD void madd(const fql_t (&px)[XP::NL], const fql_t (&py)[XP::NL], bool negate) {
        fql_t ny[XP::NL];
#pragma unroll
        for (int l = 0; l < NL; l++) ny[l] = fq2l::cond_neg_canonical(py[l], negate);
        if (inf) {
#pragma unroll
            for (int l = 0; l < NL; l++) x[l] = px[l], y[l] = ny[l], zz[l] = zzz[l] = one_component(XP::odd(l));
            inf = false;
            return;
        }
        opa_t A[XP::NL];
        opb_t B[XP::NL];
        fql_t u2[XP::NL], s2[XP::NL], p[XP::NL], r[XP::NL];
        ops::prep_a(zz, A);
        ops::prep_b(px, B);
        ops::mul(A, B, u2);  // U2 = ZZ1 x2
        ops::prep_a(zzz, A);
        ops::prep_b(ny, B);
        ops::mul(A, B, s2);  // S2 = ZZZ1 (+-y2)
        bool low[XP::NL];
#pragma unroll
        for (int l = 0; l < NL; l++) {
            p[l] = fq2l::sub_norm(u2[l], x[l], 1);  // (-1 - e, e) - [0, 1] + 1
            r[l] = fq2l::sub_norm(s2[l], y[l], 1);
            low[l] = (((uint32_t)p[l].v[0] + 1u) & MASK) <= 2u;  // -q, 0, q = -1, 0, 1 (mod 2^29)
        }
        if (XP::both(low)) {
            bool pz[XP::NL], rz[XP::NL];
#pragma unroll
            for (int l = 0; l < NL; l++) pz[l] = is_zero_mod_q(p[l]), rz[l] = is_zero_mod_q(r[l]);
            if (XP::both(pz)) {
                if (XP::both(rz)) {
                    // out of line, on copies (an accumulator whose address escapes would live in scratch for the whole loop): the
                    // doubling's eleven temporaries stay out of the register allocation of the loop body
                    xyzz_pair_t tmp;
                    fql_t cx[XP::NL], cy[XP::NL];
#pragma unroll
                    for (int l = 0; l < NL; l++) cx[l] = px[l], cy[l] = ny[l];
                    mdbl_ool(&tmp, cx, cy);
#pragma unroll
                    for (int l = 0; l < NL; l++) x[l] = tmp.x[l], y[l] = tmp.y[l], zz[l] = tmp.zz[l], zzz[l] = tmp.zzz[l];
                } else {
                    inf = true;
                }
                return;
            }
        }
        // (order: every value is consumed as early as the formulas allow - ZZ and ZZZ are updated in place as soon as PP / PPP exist - so
        // that the live set stays below 256 registers: accumulator 52, two prepared operands 53, b and X3 26, the column sum ~45)
        fql_t pp[XP::NL], ppp[XP::NL], q[XP::NL], rr[XP::NL], x3[XP::NL], d[XP::NL], a[XP::NL], b[XP::NL];
        ops::prep_a(p, A);
        ops::prep_b(p, B);
        ops::mul(A, B, pp);  // PP = P^2
        ops::prep_b(pp, B);
        ops::mul(A, B, ppp);  // PPP = P PP
        ops::prep_a(x, A);
        ops::mul(A, B, q);  // Q = X1 PP
        ops::prep_a(zz, A);
        ops::mul(A, B, zz);  // ZZ3 = ZZ1 PP
        ops::prep_b(ppp, B);
        ops::prep_a(zzz, A);
        ops::mul(A, B, zzz);  // ZZZ3 = ZZZ1 PPP
        ops::prep_a(y, A);
        ops::mul(A, B, b);  // Y1 PPP
        ops::prep_a(r, A);
        ops::prep_b(r, B);
        ops::mul(A, B, rr);  // R^2
#pragma unroll
        for (int l = 0; l < NL; l++) {
            int32_t t[N];
#pragma unroll
            for (int i = 0; i < N; i++) t[i] = rr[l].v[i] - ppp[l].v[i] - 2 * q[l].v[i];
            x3[l] = norm_k(t);                       // X3 = R^2 - PPP - 2 Q  -> [0, 1)
            d[l] = fq2l::sub_norm(q[l], x3[l], 1);  // Q - X3
        }
        ops::prep_b(d, B);
        ops::mul(A, B, a);  // R (Q - X3)
#pragma unroll
        for (int l = 0; l < NL; l++) {
            y[l] = fq2l::sub_norm(a[l], b[l], -1);  // + q if negative -> [0, 1 + 2 e]
            x[l] = x3[l];
        }
    }
    static __host__ __device__ __noinline__ void mdbl_ool(xyzz_pair_t* out, const fql_t* px, const fql_t* ny) {
        fql_t cx[XP::NL], cy[XP::NL];
#pragma unroll
        for (int l = 0; l < NL; l++) cx[l] = px[l], cy[l] = ny[l];
        out->mdbl(cx, cy);
    }
    // this = 2 (px, ny): mdbl-2008-s-1 (ec.hip.h dbl_affine).  U = 2 y enters as 2 y - q: tight.
    SV_HD void mdbl(const fql_t (&px)[XP::NL], const fql_t (&ny)[XP::NL]) {
        opa_t A[XP::NL];
        opb_t B[XP::NL];
        fql_t u[XP::NL], v[XP::NL], w[XP::NL], s[XP::NL], xx[XP::NL], m[XP::NL], mm[XP::NL], x3[XP::NL], d[XP::NL], a[XP::NL], b[XP::NL];
#pragma unroll
        for (int l = 0; l < NL; l++) {
            int32_t t[N];
#pragma unroll
            for (int i = 0; i < N; i++) t[i] = 2 * ny[l].v[i] - FqL::MOD[i];
            int32_t c = 0;
#pragma unroll
            for (int i = 0; i < N - 1; i++) {
                const int32_t xv = t[i] + c;
                u[l].v[i] = (int32_t)((uint32_t)xv & MASK);
                c = xv >> 29;
            }
            u[l].v[N - 1] = t[N - 1] + c;
            SV_OPAQUE_13(u[l].v);
        }
        ops::prep_a(u, A);
        ops::prep_b(u, B);
        ops::mul(A, B, v);  // V = U^2
        ops::prep_b(v, B);
        ops::mul(A, B, w);  // W = U V
        ops::prep_a(px, A);
        ops::mul(A, B, s);  // S = X V
        ops::prep_b(px, B);
        ops::mul(A, B, xx);  // X^2
#pragma unroll
        for (int l = 0; l < NL; l++) {
            int32_t t[N];
#pragma unroll
            for (int i = 0; i < N; i++) t[i] = 3 * xx[l].v[i];
            m[l] = norm_k(t);  // M = 3 X^2 -> [0, 1)
        }
        ops::prep_a(m, A);
        ops::prep_b(m, B);
        ops::mul(A, B, mm);
#pragma unroll
        for (int l = 0; l < NL; l++) {
            int32_t t[N];
#pragma unroll
            for (int i = 0; i < N; i++) t[i] = mm[l].v[i] - 2 * s[l].v[i];
            x3[l] = norm_k(t);  // X3 = M^2 - 2 S
            d[l] = fq2l::sub_norm(s[l], x3[l], 1);
        }
        ops::prep_b(d, B);
        ops::mul(A, B, a);  // M (S - X3)
        ops::prep_a(w, A);
        ops::prep_b(ny, B);
        ops::mul(A, B, b);  // W Y
#pragma unroll
        for (int l = 0; l < NL; l++) {
            x[l] = x3[l];
            y[l] = fq2l::sub_norm(a[l], b[l], -1);
            zz[l] = v[l];
            zzz[l] = w[l];
        }
        inf = false;
    }
};

}  // namespace fq2p

// raw partial sum of the pair kernel: [coordinate x, y, zz, zzz][component c0, c1][16 words: 13 limbs + padding] - every lane writes its
// four components as 16-byte stores; all zero = the point at infinity
struct alignas(16) g2_pair_partial_t {
    int32_t w[128];
};

}  // namespace sv
