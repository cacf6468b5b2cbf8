// hex2.hip.h - an Fq2 XYZZ addition (add-2008-s) computed by the SIXTEEN lanes of a DPP row together: the levels of a G2 tail tree
// (msm.hip.h::block_sum) where a wave has fewer than eight additions left.
//
// Why.  The G2 tail (fold + bit planes of the standard::msm path, algorithms/src/msm/variable_base/standard.rs:79-105, over
// fields/src/fp2.rs:404-410) is a chain of dependent additions, one tree level after the other.  quad_add (msm.hip.h) already deals the 14
// field products of an addition to the four lanes of a quad - four rounds of ONE Fq2 product per lane, ~2 240 dependent instructions each.  From
// the level on where a wave holds <= 4 distinct additions, 48 of its 64 lanes repeat work others do.  An Fq2 product is four independent Fq
// products (a0 b0, a1 b1, a0 b1, a1 b0), so a round of the quad schedule is SIXTEEN independent Fq products: lane l = 4 q + p of a row computes
// sub-product p of the schedule's product q - one Fq Montgomery product (~560 instructions) per lane and round instead of four.
//
// Choreography of one round (the same for the four rounds; only the operand selection differs):
//   1. every lane picks the Fq2 operands (a, b) of ITS slot q from values all sixteen lanes hold identically, and multiplies the components of
//      its sub-product p:   p = 0: a.c0 b.c0   p = 1: a.c1 b.c1   p = 2: a.c0 b.c1   p = 3: a.c1 b.c0
//   2. pair exchange inside the quad (lane p <-> p ^ 1, one `quad_perm:[1,0,3,2]` per limb); lanes 0, 1 form c0 = P0 - 5 P1 (u^2 = -5,
//      curves/src/bls12_377/fq2.rs:27-93), lanes 2, 3 form c1 = P2 + P3
//   3. gather: every lane reads c0 of slot q' from lane 4 q' and c1 from lane 4 q' + 2 of its row (DPP `row_newbcast`, one VALU move per limb), for the slots of the round that carry a product - afterwards all sixteen lanes hold the round's Fq2
//      results, identically.
// The values are canonical Montgomery residues (ff.hip.h: every operation returns the reduced representative), so the sum is BIT-IDENTICAL to
// quad_add's and to xyzz_t<fq2_t>::add's: same formula, same field values, one limb image per value.
//
// The rounds are written once, over an exchange policy `Ex`: hex_dev (this file) runs them on a GPU row, hex_host (api_g2.hip,
// snarkvm_hip_selftest_g2_hex) runs the SAME source over sixteen simulated lanes on the CPU - each lane with its own copy of everything, the
// exchanges indexing the other lanes' copies - and compares every coordinate with the exact addition.
#pragma once
#include "ec.hip.h"

namespace sv {

struct hex_lane_t {
    xyzz_t<fq2_t> A, B;  // the operands: identical on the sixteen lanes
    fq_t prod, other, comb;
    fq2_t m[4];          // the Fq2 products of the current round, gathered
    fq2_t U1, S1, P, R, PP, ZZ12, RR, ZZZ12, PPP, Q, ZZ3, X3;
    xyzz_t<fq2_t> res;
};

SV_HD fq_t hex_pick(bool c, const fq_t& a, const fq_t& b) {  // c ? a : b, limb by limb (no control flow: the lanes of a row differ in c)
    fq_t r;
#pragma unroll
    for (int i = 0; i < fq_t::N; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
SV_HD fq2_t hex_pick(bool c, const fq2_t& a, const fq2_t& b) { return {hex_pick(c, a.c0, b.c0), hex_pick(c, a.c1, b.c1)}; }
// the operands of slot q out of four candidates
SV_HD fq2_t hex_slot(int q, const fq2_t& s0, const fq2_t& s1, const fq2_t& s2, const fq2_t& s3) {
    return hex_pick(q < 2, hex_pick(q == 0, s0, s1), hex_pick(q == 2, s2, s3));
}
// step 1: this lane's Fq sub-product of a * b
SV_HD fq_t hex_sub_product(int p, const fq2_t& a, const fq2_t& b) {
    const fq_t x = hex_pick((p & 1) != 0, a.c1, a.c0);
    const fq_t y = hex_pick(p == 1 || p == 2, b.c1, b.c0);
    return x * y;
}
// step 2 (after the pair exchange): c0 on lanes p < 2, c1 on lanes p >= 2
SV_HD fq_t hex_combine(int p, const fq_t& prod, const fq_t& other) {
    const fq_t p1 = hex_pick(p == 0, other, prod);  // lanes 0 / 1: P1 = a.c1 b.c1 (lane 1's product)
    const fq_t p0 = hex_pick(p == 0, prod, other);
    const fq_t c0 = p0 - fq2_t::mul5(p1);
    const fq_t c1 = prod + other;
    return hex_pick(p < 2, c0, c1);
}

// One round: operand selection `sel(q, state) -> (a, b)`, sub-products, combine, gather of the slots in `mask`.
template <class Ex, class Sel>
SV_HD void hex_round(Ex& ex, unsigned mask, Sel sel) {
    ex.each([&](int lane, hex_lane_t& s) {
        fq2_t a, b;
        sel(lane >> 2, s, a, b);
        s.prod = hex_sub_product(lane & 3, a, b);
    });
    ex.pair_exchange();
    ex.each([&](int lane, hex_lane_t& s) { s.comb = hex_combine(lane & 3, s.prod, s.other); });
    ex.gather(mask);
}

// Rounds 1: U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1, then P = U2 - U1, R = S2 - S1 on every lane.
template <class Ex>
SV_HD void hex_add_round1(Ex& ex) {
    hex_round(ex, 0xF, [](int q, const hex_lane_t& s, fq2_t& a, fq2_t& b) {
        a = hex_slot(q, s.A.x, s.B.x, s.A.y, s.B.y);
        b = hex_slot(q, s.B.zz, s.A.zz, s.B.zzz, s.A.zzz);
    });
    ex.each([&](int, hex_lane_t& s) {
        s.U1 = s.m[0];
        s.S1 = s.m[2];
        s.P = s.m[1] - s.m[0];
        s.R = s.m[3] - s.m[2];
    });
}
// Rounds 2-4 (the quad schedule of msm.hip.h::quad_add): the result lands in s.res on every lane.
//   2   PP = P^2 | ZZ1 ZZ2 | R^2 | ZZZ1 ZZZ2
//   3   PPP = P PP | Q = U1 PP | ZZ3 = ZZ1 ZZ2 PP | -            X3 = R^2 - PPP - 2 Q
//   4   ZZZ3 = ZZZ1 ZZZ2 PPP | - | R (Q - X3) | S1 PPP            Y3 = m[2] - m[3]
template <class Ex>
SV_HD void hex_add_rounds234(Ex& ex) {
    hex_round(ex, 0xF, [](int q, const hex_lane_t& s, fq2_t& a, fq2_t& b) {
        a = hex_slot(q, s.P, s.A.zz, s.R, s.A.zzz);
        b = hex_slot(q, s.P, s.B.zz, s.R, s.B.zzz);
    });
    ex.each([&](int, hex_lane_t& s) {
        s.PP = s.m[0];
        s.ZZ12 = s.m[1];
        s.RR = s.m[2];
        s.ZZZ12 = s.m[3];
    });
    hex_round(ex, 0x7, [](int q, const hex_lane_t& s, fq2_t& a, fq2_t& b) {
        a = hex_slot(q, s.P, s.U1, s.ZZ12, s.P);  // slot 3 idles (repeats slot 0's product; not gathered)
        b = s.PP;
    });
    ex.each([&](int, hex_lane_t& s) {
        s.PPP = s.m[0];
        s.Q = s.m[1];
        s.ZZ3 = s.m[2];
        s.X3 = s.RR - s.PPP - s.Q.dbl();
    });
    hex_round(ex, 0xD, [](int q, const hex_lane_t& s, fq2_t& a, fq2_t& b) {
        a = hex_slot(q, s.ZZZ12, s.ZZZ12, s.R, s.S1);  // slot 1 idles
        b = hex_slot(q, s.PPP, s.PPP, s.Q - s.X3, s.PPP);
    });
    ex.each([&](int, hex_lane_t& s) {
        s.res.x = s.X3;
        s.res.y = s.m[2] - s.m[3];
        s.res.zz = s.ZZ3;
        s.res.zzz = s.m[0];
    });
}

// ---- sixteen simulated lanes on the CPU (snarkvm_hip_selftest_g2_hex): every lane owns a full copy of the state, the exchanges index the others'
struct hex_host {
    hex_lane_t s[16];
    template <class Fn>
    void each(Fn f) {
        for (int l = 0; l < 16; l++) f(l, s[l]);
    }
    void pair_exchange() {
        for (int l = 0; l < 16; l++) s[l].other = s[l ^ 1].prod;
    }
    void gather(unsigned mask) {
        for (int l = 0; l < 16; l++)
            for (int q = 0; q < 4; q++)
                if ((mask >> q) & 1) s[l].m[q] = {s[4 * q].comb, s[4 * q + 2].comb};
    }
};
// The whole addition as the device runs it, on the simulated row: 0 = every lane ends with exactly `want` (coordinate by coordinate), else the
// 1-based lane that differs (or 17: the lanes disagree about the equal-x fallback).
static inline int hex_add_host_check(const xyzz_t<fq2_t>& a, const xyzz_t<fq2_t>& b, const xyzz_t<fq2_t>& want) {
    hex_host* ex = new hex_host();
    for (int l = 0; l < 16; l++) ex->s[l].A = a, ex->s[l].B = b;
    const bool inf1 = a.is_inf(), inf2 = b.is_inf();
    hex_add_round1(*ex);
    int bad = 0;
    bool any_same = false, all_same = true;
    for (int l = 0; l < 16; l++) {
        const bool same_x = !inf1 && !inf2 && ex->s[l].P.is_zero();
        any_same = any_same || same_x;
        all_same = all_same && same_x;
    }
    if (any_same != all_same) bad = 17;
    if (!bad && !any_same) {  // (the fallback IS the exact law: nothing to compare)
        hex_add_rounds234(*ex);
        for (int l = 0; l < 16 && !bad; l++) {
            const xyzz_t<fq2_t>& r = ex->s[l].res;
            const xyzz_t<fq2_t> got = inf2 ? a : (inf1 ? b : r);
            if (!(got.x == want.x && got.y == want.y && got.zz == want.zz && got.zzz == want.zzz)) bad = l + 1;
        }
    }
    delete ex;
    return bad;
}

#if defined(__HIPCC__)
// ---- the GPU row ---------------------------------------------------------------------------------------------------------------
// DPP `row_newbcast:N` (gfx90a+): every lane of a row reads lane N of ITS row inside the VALU - no LDS round trip, no s_waitcnt.  (The first
// form of this file gathered with ds_bpermute: 364 of them per addition, each batch followed by an exposed LDS latency - measured, the sixteen-lane
// addition then cost what the four-lane one costs, profiles/r06_g2_tail.md; that variant is gone.)  tools/exp/dpp_bcast.hip checks the control's semantics
// on the hardware.
template <int LANE>
__device__ __forceinline__ fq_t hex_row_bcast(const fq_t& a) {
    fq_t r;
#pragma unroll
    for (int i = 0; i < fq_t::N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0x150 + LANE, 0xf, 0xf, true);
    return r;
}
struct hex_dev {
    hex_lane_t& s;
    template <class Fn>
    __device__ __forceinline__ void each(Fn f) {
        f((int)(threadIdx.x & 15), s);
    }
    __device__ __forceinline__ void pair_exchange() {
#pragma unroll
        for (int i = 0; i < fq_t::N; i++) s.other.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)s.prod.v[i], 0xB1, 0xf, 0xf, true);  // quad_perm:[1,0,3,2]
    }
    template <int Q>
    __device__ __forceinline__ void gather_slot(unsigned mask) {
        if ((mask >> Q) & 1) {
            s.m[Q].c0 = hex_row_bcast<4 * Q>(s.comb);
            s.m[Q].c1 = hex_row_bcast<4 * Q + 2>(s.comb);
        }
    }
    __device__ __forceinline__ void gather(unsigned mask) {
        gather_slot<0>(mask);
        gather_slot<1>(mask);
        gather_slot<2>(mask);
        gather_slot<3>(mask);
    }
};
// acc += o.  Precondition: the sixteen lanes of every row hold bit-identical (acc, o); postcondition: they hold the bit-identical sum.
// Every lane of the wave must call it (the gathers are wave operations).
// Equal x coordinates (P = U2 - U1 = 0; every lane of the row sees it after round 1) anywhere in the wave set `dbl` and abandon the addition: the kernel marks its
// output and msm_*_fix_kernel (msm.hip.h) recomputes it with the plain law.  No device-function call, no second copy of the law in the kernel.
__device__ __forceinline__ void hex_add(xyzz_t<fq2_t>& acc, const xyzz_t<fq2_t>& o, bool& dbl) {
    hex_lane_t s;
    s.A = acc;
    s.B = o;
    hex_dev ex{s};
    const bool inf1 = acc.is_inf(), inf2 = o.is_inf();
    hex_add_round1(ex);
    const bool same_x = !inf1 && !inf2 && s.P.is_zero();
    if (__ballot(same_x) != 0) {  // wave-uniform
        dbl = true;               // (every lane of the wave: the flag is per output anyway)
        return;
    }
    hex_add_rounds234(ex);
    acc.x = hex_pick(inf2, acc.x, hex_pick(inf1, o.x, s.res.x));
    acc.y = hex_pick(inf2, acc.y, hex_pick(inf1, o.y, s.res.y));
    acc.zz = hex_pick(inf2, acc.zz, hex_pick(inf1, o.zz, s.res.zz));
    acc.zzz = hex_pick(inf2, acc.zzz, hex_pick(inf1, o.zzz, s.res.zzz));
}
#endif

}  // namespace sv
