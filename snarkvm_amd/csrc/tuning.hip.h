// tuning.hip.h - the ONE place where the backend reads launch-geometry / algorithm-variant switches.
//
// A prover library should not change its arithmetic or its launch geometry because some variable happens to be set in the
// environment of the process that loaded it (round-3 review: ~25 getenv sites in the product path).  Everything that is an
// experiment knob now lives in one struct, filled ONCE from ONE variable:
//
//     SNARKVM_HIP_TUNING="key=value,key=value,..."        (unknown keys are an error reported on stderr and ignored)
//
// Every key is an A/B switch whose two sides return bit-identical results (tests/test_gpu_multidevice.py runs the suite's
// MSM / NTT checks under each of them); the defaults are the measured winners and are what DESIGN.md describes.  Operational
// configuration - not tuning - keeps its own, documented variables: SNARKVM_HIP_DEVICES (device set), SNARKVM_HIP_BASE_CACHE /
// _MB (the opt-in base cache of snarkvm_msm), SNARKVM_HIP_NTT_TW_MB (HBM cap of the closing-twiddle cache), SNARKVM_HIP_TRACE
// (chunk timeline on stderr).
//
//   key             default  meaning
//   lazy            1        G1 accumulation on the signed-limb arithmetic of ffl.hip.h (0: exact kernel)
//   lazy_tail       1        1: G1 reduce rounds, bucket merge, fold and bit planes on the lazy arithmetic too (ffl.hip.h::fqz_t), no conversion pass.
//                            Measured at 2^24: the conversion pass goes (-0.52 ms) but the tail kernels do not get faster (reduce 0.89 vs 0.87 ms,
//                            fold + bit planes 1.54 vs 1.46 ms: every sum / difference is a 13-step carry chain in a latency-bound kernel, and the
//                            fold / bit-plane kernels spill 808 B at 256 registers): 30.71 vs 30.88 ms per step, 197.4 vs 196.6 proofs/s.  Round 5: ON - where it
//                            pays is a single proof (no conversion pass in any of its six commitment rounds: 8.49 vs 8.88 ms per proof,
//                            profiles/r05_proof1_timeline.md)
//   lazy2           1        G2 accumulation on the signed-limb Fq2 arithmetic (0: exact kernel)
//   fused           1        wide windows: scalar read fused with the level-1 partition (0: stand-alone digit matrix)
//   hist            2        scalar-read kernel variant (1: 512 threads x 4 scalars, one LDS histogram - round 3; 2: 1 024 threads x 2
//                            scalars, four private histogram copies)
//   prefetch        2        base gather software pipeline: 0 never, 1 single-round grids, 2 always
//   acc_lds         98304    dynamic LDS request that keeps a second accumulate workgroup off a CU (single-round grids); 0: off
//   acc_one_wg      0        1: one accumulate workgroup per CU for multi-round grids too
//   reduce_rounds   1        fixed reduce rounds of a multi-round MSM (0 .. 8), each shrinking a bucket's partial sums by the group size (seg2, 16).
//                            Round 4: one round of 16 instead of two of 8 - for uniform scalars the second round only copied (0.84 -> 0.57 ms of
//                            reduce time at 2^24, 29.84 -> 29.51 ms per step; one round of 64: 0.99 ms); what an all-equal scalar vector leaves in
//                            one bucket (2^24: 16 384 partial sums instead of 4 096) goes to the flattened-list fold: ~3 ms more on that input
//   fold_flat       1        flattened-list fold for every MSM (0: per-bucket lists for big ones)
//   fuse_batch      1        small instances of a batch travel as fused multi-instance groups
//   fuse_max_k      64       instances per fused group
//   fuse_reduce     -1       reduce rounds of a fused group before its fold (0: none; 1: one for every group: +1.6 % on the lock-step proof replay and a bound
//                            on what an all-equal scalar vector can leave in one bucket; -1: one round for groups of >= 8 instances only).  Round 5 chose 1 on
//                            a proof with every round enqueued at once (7.29 ms against 7.45 with 0, 7.40 with -1).  Round 6, the proof in TRANSCRIPT order (a
//                            round's group of 1 - 4 instances is alone on the critical path: its reduce round is 0.11 ms in front of a fold that takes the
//                            partial sums as they are), three interleaved runs on one box: 1: 8.16 - 8.19 ms per proof, witness-like pool 7.54 - 7.62;
//                            0: 8.01 - 8.09 / 6.98 - 6.99; -1: 8.04 / 7.05 - 7.12; everything at once 6.71 - 6.81 with 1, 6.81 - 6.89 with -1; lock step
//                            (groups of 64) 207.4 against 206.0 proofs/s - hence -1.
//   coalesce        1        concurrent callers of small registered MSMs are fused by an in-library dispatcher
//   coalesce_us     40       how long a dispatcher waits for further callers when others are inside the library
//   lanes           0        lanes a batch cycles through (0: 8 below 2^20 pairs, 3 above)
//   msm_chunk_lg    20       pairs per upload / compute chunk of snarkvm_msm (host bases)
//   scalar_chunk_lg 22       pairs per scalar chunk of a host-scalar MSM over registered bases (equal chunks: several devices, or scalar_geo=0)
//   scalar_geo      4        one device, >= 2^22 pairs: two or three scalar chunks of sizes 1 : g [: g^2] instead of equal ones (0: off)
//   taper           1        snarkvm_msm: the last chunk is cut again (1/2, 1/4, 1/4) and all chunks share one bucket sink and one tail; the scalar
//                            chunks of a host-scalar MSM over registered bases share one sink as well
//   ramp            3        snarkvm_msm: the FIRST chunk is cut into 1/2^r, 1/2^r, 1/2^(r-1), ..., 1/2 (0: off) so that the GPU starts after a short upload
//   ring_lanes      3        lanes (streams with their own staging buffers) the chunks of one snarkvm_msm call cycle through
//   seg             0        accumulate segment length override (0: planner)
//   seg2            0        reduce-round group size override (0: 16)
//   fold_l          0        planner L override
//   scan1           1        single-launch scan for mid-size counter arrays
//   ntt_min_tiles   256      workgroups a small-transform pass is spread over
//   ntt_full_tw     1        materialised closing-twiddle tables
//   ntt_fold        1        2^261 [/ n] folded into the table of the pass before the last
//   ntt_signed      0        1: NTT butterflies on the signed limbs of frs.hip.h (measured: 2.115 vs 2.062 ms of kernels at 2^24 - not faster)
//   ntt_batch       1        snarkvm_hip_ntt_device_batch: one launch per pass for all vectors of a (direction, type) group
//   xcd             1        scatter kernels of the radix partition: XCD x walks a contiguous tile range (msm_sort.hip.h::xcd_tile)
//   fold_threads2   128      G2: threads per output of a small fold (its kernels run one wave per SIMD: 256-thread workgroups = one per CU)
//   coalesce_slots  2        dispatchers of the coalescer that may be inside the library at once (per handle)
//   group_quad      1        group NTT (setup): four lanes per butterfly, 2-bit windowed twiddle multiplication on quad-cooperative point operations; 0: one lane per butterfly, bit-serial
//   hex2            1        G2 tail trees: from exchange distance 8 on, every addition on the sixteen lanes of a DPP row (hex2.hip.h, gathers by DPP row
//                            broadcast; the ds_bpermute form of round 6 measured no gain and is gone); 0: four lanes (quad_add) throughout
//   tail_quads      13       fold / bit planes: quad-strided accumulation in front of the trees (no plain one-lane addition), bit mask: 1 = G2 bit planes, 2 = G2 fold (measured slower: off), 4 = G1 bit planes, 8 = G1 fold
//   fold_small2     256      G2: threads per output of a fold of <= 256 workgroups (one turn of the chip; 128 / 64: round 5's halved workgroup - an A/B and bisection switch)
//   fold_mid        128      G1: threads per output of a fold of 513 .. 1 024 workgroups (a fused group of 3 - 4 proof-sized instances); 64: one wave per output
//   aux_cus         0        experiment: N > 0 creates the upper half of the lanes (where a scope puts its further MSM streams) with a compute-unit mask of N CUs.
//                            One proof in transcript order, whose only further-stream MSM is the independent G2 one: 8.06 ms with 0, 7.83 - 7.90 with 128, 7.88 - 7.95
//                            with 192, 8.54 with 64, 9.8 - 10.1 with 32 (two alternating runs, one box) - a background MSM confined to half the chip disturbs the
//                            critical stream less.  Not the default and not an ABI flag: every mode that puts commitment ROUNDS on those lanes is 1.5 - 5x slower
//                            under the mask, and the reference's prover issues no such MSM (profiles/r06_summary.md).
//   pair2           1        G2 accumulation on a lane pair (ffl2p.hip.h: c0 on the even lane, c1 on the odd lane; 0: both components in one lane, ffl2.hip.h)
//   horner2         1        p / (X - z): three launches with a scan inside every workgroup (0: the four-level chunk recursion of round 3)
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace sv {

struct tuning_t {
    int lazy = 1, lazy2 = 1, fused = 1, hist = 2, prefetch = 2;
    long acc_lds = 96 * 1024;
    int acc_one_wg = 0, reduce_rounds = 1, fold_flat = 1, fuse_batch = 1, fuse_max_k = 64, fuse_reduce = -1, coalesce = 1, coalesce_us = 40, lanes = 0;
    int msm_chunk_lg = 20, scalar_chunk_lg = 22, taper = 1, ring_lanes = 3, seg = 0, seg2 = 0, fold_l = 0, scan1 = 1;
    int ntt_min_tiles = 256, ntt_full_tw = 1, ntt_fold = 1, ntt_signed = 0, ntt_batch = 1;
    int xcd = 1, fold_threads2 = 128, coalesce_slots = 2, ramp = 3, scalar_geo = 4, lazy_tail = 1, horner2 = 1, pair2 = 1, hex2 = 1, group_quad = 1, tail_quads = 13, aux_low_prio = 0, fold_small2 = 256, fold_mid = 128, aux_cus = 0;

    bool set(const char* key, long v) {
#define SV_TUNE_KEY(name)                  \
    if (!strcmp(key, #name)) {             \
        name = (decltype(name))v;          \
        return true;                       \
    }
        SV_TUNE_KEY(lazy) SV_TUNE_KEY(lazy2) SV_TUNE_KEY(fused) SV_TUNE_KEY(hist) SV_TUNE_KEY(prefetch) SV_TUNE_KEY(acc_lds)
        SV_TUNE_KEY(acc_one_wg) SV_TUNE_KEY(reduce_rounds) SV_TUNE_KEY(fold_flat) SV_TUNE_KEY(fuse_batch) SV_TUNE_KEY(fuse_max_k) SV_TUNE_KEY(fuse_reduce) SV_TUNE_KEY(coalesce)
        SV_TUNE_KEY(coalesce_us) SV_TUNE_KEY(lanes) SV_TUNE_KEY(msm_chunk_lg) SV_TUNE_KEY(scalar_chunk_lg) SV_TUNE_KEY(taper) SV_TUNE_KEY(ring_lanes) SV_TUNE_KEY(seg) SV_TUNE_KEY(seg2)
        SV_TUNE_KEY(fold_l) SV_TUNE_KEY(scan1) SV_TUNE_KEY(ntt_min_tiles) SV_TUNE_KEY(ntt_full_tw) SV_TUNE_KEY(ntt_fold) SV_TUNE_KEY(ntt_signed)
        SV_TUNE_KEY(ntt_batch) SV_TUNE_KEY(xcd) SV_TUNE_KEY(fold_threads2) SV_TUNE_KEY(coalesce_slots) SV_TUNE_KEY(ramp) SV_TUNE_KEY(scalar_geo) SV_TUNE_KEY(lazy_tail) SV_TUNE_KEY(horner2) SV_TUNE_KEY(pair2) SV_TUNE_KEY(hex2) SV_TUNE_KEY(group_quad) SV_TUNE_KEY(tail_quads) SV_TUNE_KEY(aux_low_prio) SV_TUNE_KEY(fold_small2) SV_TUNE_KEY(fold_mid) SV_TUNE_KEY(aux_cus)
#undef SV_TUNE_KEY
        return false;
    }
    void parse(const char* s) {
        while (s && *s) {
            while (*s == ',' || *s == ' ') s++;
            const char* eq = strchr(s, '=');
            const char* end = strchr(s, ',');
            if (!end) end = s + strlen(s);
            if (!*s) break;
            char key[32] = {0};
            if (!eq || eq > end || (size_t)(eq - s) >= sizeof key) {
                fprintf(stderr, "[snarkvm_hip] SNARKVM_HIP_TUNING: malformed item at \"%s\" (ignored)\n", s);
            } else {
                memcpy(key, s, (size_t)(eq - s));
                char* num_end = nullptr;
                const long v = strtol(eq + 1, &num_end, 0);
                if (num_end == eq + 1 || !set(key, v)) fprintf(stderr, "[snarkvm_hip] SNARKVM_HIP_TUNING: unknown key or bad value \"%s\" (ignored)\n", key);
            }
            s = end;
        }
    }
    // values a typo must not turn into a hang or a crash (coalesce_slots = 0: every coalescible MSM would wait for a dispatcher forever)
    void clamp() {
        auto fix = [](const char* key, auto& v, long lo, long hi) {
            if ((long)v < lo || (long)v > hi) {
                const long was = (long)v;
                v = (decltype(v + 0))((long)v < lo ? lo : hi);
                fprintf(stderr, "[snarkvm_hip] SNARKVM_HIP_TUNING: %s=%ld is out of range [%ld, %ld]; using %ld\n", key, was, lo, hi, (long)v);
            }
        };
        fix("coalesce_slots", coalesce_slots, 1, 8);
        fix("coalesce_us", coalesce_us, 0, 100000);
        fix("fuse_max_k", fuse_max_k, 2, 256);
        fix("fuse_reduce", fuse_reduce, -1, 4);
        fix("reduce_rounds", reduce_rounds, 0, 8);
        fix("lanes", lanes, 0, 8);
        fix("ring_lanes", ring_lanes, 2, 8);
        fix("ntt_min_tiles", ntt_min_tiles, 1, 1 << 20);
        fix("hist", hist, 1, 2);
        fix("prefetch", prefetch, 0, 2);
        fix("acc_lds", acc_lds, 0, 160 * 1024);
        fix("msm_chunk_lg", msm_chunk_lg, 16, 30);
        fix("scalar_chunk_lg", scalar_chunk_lg, 18, 30);
        fix("ramp", ramp, 0, 6);
        fix("scalar_geo", scalar_geo, 0, 16);
        fix("seg", seg, 0, 4096);
        fix("seg2", seg2, 0, 256);
        fix("fold_l", fold_l, 0, 64);
    }
};
// parsed on first use, once per process (the variable is not consulted again)
inline const tuning_t& tuning() {
    static const tuning_t t = [] {
        tuning_t v;
        v.parse(getenv("SNARKVM_HIP_TUNING"));
        v.clamp();
        return v;
    }();
    return t;
}

}  // namespace sv
