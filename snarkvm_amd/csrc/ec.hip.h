// ec.hip.h - short-Weierstrass (a = 0) point arithmetic for the MSM kernels, templated on the base
// field so that G1 (Fq) and G2 (Fq2) share one implementation.
//
// Device-side representations
//   aff_t<F>   {x, y}            bases as stored in HBM by the MSM engine; the point at infinity is the
//                                sentinel (0, 0), which is not on y^2 = x^3 + b for b != 0.
//   xyzz_t<F>  {x, y, zz, zzz}   bucket accumulator, x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2; infinity <=> zz == 0.
//   jac_t<F>   {x, y, z}         the reference's `Projective` (Jacobian, projective.rs:37-41): ABI result type.
//
// The reference accumulates buckets in batched-affine (G1, batched.rs) or Jacobian (standard.rs)
// coordinates; any complete addition law yields the same group element, and parity is defined on the
// affine-normalised result (variable_base/mod.rs:96-105), so XYZZ (cheapest mixed addition: 8M + 2S,
// https://hyperelliptic.org/EFD/g1p/auto-shortw-xyzz.html) is used here.  All exceptional cases of
// affine.rs:224-273 / projective.rs:222-291 are handled: either operand infinity, P + P, P + (-P).
#pragma once
#include "ff.hip.h"

namespace sv {

template <class F>
struct aff_t {
    F x, y;
    SV_HD bool is_inf() const { return x.is_zero() && y.is_zero(); }
    SV_HD static aff_t inf() { return {F::zero(), F::zero()}; }
};

template <class F>
struct jac_t {
    F x, y, z;
    SV_HD bool is_inf() const { return z.is_zero(); }
    // projective.rs:302-339 (double_in_place, a == 0 branch): 2M + 5S, the cheapest doubling for long chains
    SV_HD jac_t dbl() const {
        if (is_inf()) return *this;
        F a = x.sqr();
        F b = y.sqr();
        F c = b.sqr();
        F d = ((x + b).sqr() - a - c).dbl();
        F e = a.dbl() + a;
        F f = e.sqr();
        jac_t r;
        r.z = (y * z).dbl();
        r.x = f - d.dbl();
        r.y = e * (d - r.x) - c.dbl().dbl().dbl();
        return r;
    }
};

template <class F>
struct xyzz_t {
    F x, y, zz, zzz;

    SV_HD static xyzz_t inf() { return {F::zero(), F::one(), F::zero(), F::zero()}; }
    SV_HD bool is_inf() const { return zz.is_zero(); }

    SV_HD static xyzz_t from_affine(const aff_t<F>& p) {
        if (p.is_inf()) return inf();
        return {p.x, p.y, F::one(), F::one()};
    }
    // doubling of an affine point (mdbl-2008-s-1), a = 0
    SV_HD static xyzz_t dbl_affine(const aff_t<F>& p) {
        F u = p.y.dbl();
        F v = u.sqr();
        F w = u * v;
        F s = p.x * v;
        F xx = p.x.sqr();
        F m = xx.dbl() + xx;
        xyzz_t r;
        r.x = m.sqr() - s.dbl();
        r.y = m * (s - r.x) - w * p.y;
        r.zz = v;
        r.zzz = w;
        return r;
    }
    // dbl-2008-s-1, a = 0
    SV_HD xyzz_t dbl() const {
        if (is_inf()) return *this;
        F u = y.dbl();
        F v = u.sqr();
        F w = u * v;
        F s = x * v;
        F xx = x.sqr();
        F m = xx.dbl() + xx;
        xyzz_t r;
        r.x = m.sqr() - s.dbl();
        r.y = m * (s - r.x) - w * y;
        r.zz = v * zz;
        r.zzz = w * zzz;
        return r;
    }
    // The doubling inside the general addition (P + P: one pair in 2^377 for random operands) is cold code.  For Fq2 it is kept out
    // of line: the tail kernels hold a dozen addition sites each, and ~15 000 instructions of inlined doubling per site made the
    // Fq2 unit 14 of the build's 14 minutes (two 365 000-instruction kernels); measured on one box the out-of-line form costs
    // the G2 tail nothing (0.92 vs 0.915 ms at 2^16).  The mixed addition of the accumulate loop keeps its doubling inlined
    // (out of line: -5 % there), and so does everything over Fq (out of line: the 2^16 tail 0.184 vs 0.167 ms) unless the
    // A/B switch -DSV_COLD_OOL asks otherwise (`python -m snarkvm_amd.build --ool`).
    // The callee works on copies (an object whose address escapes would live in scratch for the caller).
#if defined(SV_TU_TAIL)  // the tail units hold no device-function call at all: their only full additions sit in the small fix kernels (msm.hip.h)
    static constexpr bool COLD_DBL_OOL = false;
#else
    static constexpr bool COLD_DBL_OOL = sizeof(F) > 64;
#endif
    static SV_COLD void cold_dbl_affine(xyzz_t* out, const aff_t<F>* p) { *out = dbl_affine(*p); }
    static SV_COLD void cold_dbl(xyzz_t* out, const xyzz_t* a) { *out = a->dbl(); }
    static __host__ __device__ __noinline__ void cold_dbl_ool(xyzz_t* out, const xyzz_t* a) { *out = a->dbl(); }
    // this += p  (madd-2008-s); `negate` adds -p instead (signed-digit buckets)
    SV_HD void add_affine(const aff_t<F>& p_in, bool negate = false) {
        if (p_in.is_inf()) return;
        aff_t<F> p = p_in;
        if (negate) p.y = p.y.neg();
        if (is_inf()) {
            *this = {p.x, p.y, F::one(), F::one()};
            return;
        }
        F u2 = p.x * zz;
        F s2 = p.y * zzz;
        F pp_ = u2 - x;  // P
        F r = s2 - y;    // R
        if (pp_.is_zero()) {
            if (r.is_zero()) {
                const aff_t<F> pc = p;
                xyzz_t d;
                cold_dbl_affine(&d, &pc);
                *this = d;
            } else {
                *this = inf();
            }
            return;
        }
        F pp = pp_.sqr();
        F ppp = pp_ * pp;
        F q = x * pp;
        F x3 = r.sqr() - ppp - q.dbl();
        y = F::diff_of_products(r, q - x3, y, ppp);  // one Montgomery reduction for both products
        x = x3;
        zz = zz * pp;
        zzz = zzz * ppp;
    }
    // this += o  (add-2008-s)
    SV_HD void add(const xyzz_t& o) { add_impl<false>(o, nullptr); }
    // this += o without the exceptional cases: equal x coordinates (P + P, P - P) set `dbl` and leave *this as it was.  For the kernels that keep the
    // exceptional law out of their code and have the flagged outputs recomputed by a second, plain kernel (msm.hip.h: the Fq2 fold / bit planes).
    SV_HD void add_flag(const xyzz_t& o, bool& dbl) { add_impl<true>(o, &dbl); }
    template <bool FLAG>
    SV_HD void add_impl(const xyzz_t& o, bool* dbl) {
        if (o.is_inf()) return;
        if (is_inf()) {
            *this = o;
            return;
        }
        F u1 = x * o.zz;
        F u2 = o.x * zz;
        F s1 = y * o.zzz;
        F s2 = o.y * zzz;
        F pp_ = u2 - u1;
        F r = s2 - s1;
        if (pp_.is_zero()) {
            if constexpr (FLAG) {
                *dbl = true;
            } else if (r.is_zero()) {
                const xyzz_t self = *this;
                xyzz_t d;
                if constexpr (COLD_DBL_OOL)
                    cold_dbl_ool(&d, &self);
                else
                    cold_dbl(&d, &self);
                *this = d;
            } else {
                *this = inf();
            }
            return;
        }
        F pp = pp_.sqr();
        F ppp = pp_ * pp;
        F q = u1 * pp;
        F x3 = r.sqr() - ppp - q.dbl();
        y = F::diff_of_products(r, q - x3, s1, ppp);
        x = x3;
        zz = zz * o.zz * pp;
        zzz = zzz * o.zzz * ppp;
    }
    // k * this for a small unsigned k (MSB-first double-and-add)
    SV_HD xyzz_t mul_small(uint32_t k) const {
        xyzz_t acc = inf();
        for (int bit = 31; bit >= 0; bit--) {
            acc = acc.dbl();
            if ((k >> bit) & 1) acc.add(*this);
        }
        return acc;
    }
    // Jacobian representative: with z = ZZZ/ZZ, take Z' = ZZZ = z^3: X' = x Z'^2 = X ZZ^2, Y' = y Z'^3 = Y ZZZ^2.
    // Infinity maps to the reference's canonical Projective::zero() = (0, 1, 0) (projective.rs:49-54).
    SV_HD jac_t<F> to_jacobian() const {
        if (is_inf()) return {F::zero(), F::one(), F::zero()};
        return {x * zz.sqr(), y * zzz.sqr(), zzz};
    }
    // Jacobian (X, Y, Z) -> XYZZ: zz = Z^2, zzz = Z^3
    SV_HD static xyzz_t from_jacobian(const jac_t<F>& j) {
        if (j.is_inf()) return inf();
        F zz = j.z.sqr();
        return {j.x, j.y, zz, zz * j.z};
    }
};

}  // namespace sv

// ------------------------------------------------------------------------------------------
// Memory images used by the MSM engine (internal Montgomery form, packed words per coordinate)
// ------------------------------------------------------------------------------------------
namespace sv {
#ifndef SV_BASE_ALIGN
#define SV_BASE_ALIGN 128  // one 128-byte slot per base: a gathered point never straddles two 128-byte blocks (-3.9 % accumulate time)
#endif
template <class F>
struct alignas(SV_BASE_ALIGN) aff_mem_t {  // device-native base (G1: 96 B payload, G2: 192 B); infinity = all zero
    typename F::mem_t x, y;
};
template <class F>
struct alignas(16) xyzz_mem_t {  // bucket / partial sum (G1: 192 B, G2: 384 B)
    typename F::mem_t x, y, zz, zzz;
};
template <class F>
struct alignas(16) jac_mem_t {  // the reference's Projective memory image (G1: 144 B, G2: 288 B)
    typename F::mem_t x, y, z;
};
template <class F>
SV_HD aff_t<F> load_aff(const aff_mem_t<F>* p) {
    return {F::load(&p->x), F::load(&p->y)};
}
template <class F>
SV_HD void store_aff(aff_mem_t<F>* p, const aff_t<F>& a) {
    a.x.store(&p->x);
    a.y.store(&p->y);
}
template <class F>
SV_HD xyzz_t<F> load_xyzz(const xyzz_mem_t<F>* p) {
    return {F::load(&p->x), F::load(&p->y), F::load(&p->zz), F::load(&p->zzz)};
}
template <class F>
SV_HD void store_xyzz(xyzz_mem_t<F>* p, const xyzz_t<F>& a) {
    a.x.store(&p->x);
    a.y.store(&p->y);
    a.zz.store(&p->zz);
    a.zzz.store(&p->zzz);
}

typedef aff_mem_t<fq_t> g1_aff_mem_t;
typedef xyzz_mem_t<fq_t> g1_xyzz_mem_t;
typedef aff_t<fq_t> g1_aff_t;
typedef xyzz_t<fq_t> g1_xyzz_t;
typedef jac_t<fq_t> g1_jac_t;
typedef jac_mem_t<fq_t> g1_jac_out_t;
static_assert(sizeof(g1_aff_mem_t) == (SV_BASE_ALIGN > 16 ? 128 : 96) && sizeof(g1_xyzz_mem_t) == 192 && sizeof(g1_jac_out_t) == 144, "G1 images");
static_assert(sizeof(jac_mem_t<fq2_t>) == 288, "G2Projective image");
SV_HD g1_aff_t g1_load_aff(const g1_aff_mem_t* p) { return load_aff<fq_t>(p); }
SV_HD g1_xyzz_t g1_load_xyzz(const g1_xyzz_mem_t* p) { return load_xyzz<fq_t>(p); }
SV_HD void g1_store_xyzz(g1_xyzz_mem_t* p, const g1_xyzz_t& a) { store_xyzz<fq_t>(p, a); }
}  // namespace sv
