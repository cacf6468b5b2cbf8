// msm.hip.h - Pippenger variable-base MSM over BLS12-377 G1 for gfx950.
//
// Replaces (behaviour, not code): sppark's msm_t::invoke as called from
// algorithms/cuda/cuda/snarkvm.cu:249-311, and the CPU algorithms VariableBase::msm /
// batched::msm / standard::msm (algorithms/src/msm/variable_base/{mod,batched,standard}.rs).
// Result = sum_i scalars[i] * bases[i] as a Jacobian point; the representative differs from the
// reference's, the affine normalisation is identical (that is what the reference's tests compare,
// variable_base/mod.rs:96-105,116-117).
//
// Pipeline (all on one stream of one lane; runtime.hip.h::msm_run enqueues it, DESIGN.md 3.2 has the sizes):
//   1 scalar read   wide windows (c > 16, registered tables): radix_hist1_fused_kernel (msm_sort.hip.h) reads every 32-byte
//               scalar once, recodes it in registers into signed c-bit digits with the bias trick
//               (s' = s + sum_w 2^(c-1+cw); digit_w = ((s' >> cw) & (2^c-1)) - 2^(c-1)) and counts level-1 bins in LDS -
//               no digit matrix exists.  c <= 16: msm_digits_kernel writes u16 digits [rows][n].
//   2-4 sort    LDS-staged radix partition of the (index | sign, bucket) entries: two levels, three for wide windows
//               (msm_sort.hip.h); output is bucket-major `sorted` + bucket offsets `boff`.
//   5 accumulate  msm_accumulate_seg_kernel: balanced segments - thread t owns S consecutive sorted entries, gathers the
//               bases, XYZZ mixed additions, one partial sum per bucket touched.  Integer-ALU bound.
//   6 reduce    msm_reduce_kernel rounds (only for > 2^22 digit entries): partial sums per bucket -> <= tail_partials.
//   7 fold + bit planes  msm_fold_kernel (two-axis row / column sums of a window), msm_bitplane_kernel (sum_i (i+1) P_i as
//               sum_j 2^j S_j); 8 the Horner chain over <= ~270 bit-plane sums runs on the host (runtime.hip.h msm_accum_t).
// Digit zero is skipped (batched.rs:350: bucket index wraps to u32::MAX and is ignored).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "ec.hip.h"
#include "ffl.hip.h"
#include "ffl2.hip.h"
#include "hex2.hip.h"
#include "tuning.hip.h"

namespace sv {

struct msm_plan_t {
    size_t n;
    int c;             // window width (bits)
    int W;             // bucket windows (Horner chain = c * (W - 1) doublings)
    int J;             // precomputed base tables: table j holds 2^(c*W*j) * P, so digit row j*W + w lands in window w
    int Wd;            // digit rows per scalar = W * J
    uint32_t nb;       // buckets per window = 2^(c-1)
    uint32_t nbt;      // W * nb
    uint32_t chunk;    // scalars per histogram chunk
    uint32_t nchunks;
    uint32_t S, S2;    // points per accumulate thread; partials per reduce thread
    uint32_t L;        // buckets per bucket-reduction thread
    int rounds;        // reduce rounds
    uint32_t bias[10]; // sum_w 2^(c-1+cw), 320-bit
};

// Window width of a table-less MSM (W = ceil(254 / c) windows).  The top window holds only 254 - c (W - 1) scalar bits (plus
// the recoding carry), so for most c its few buckets receive n / 2 .. n / 8 entries each; two shapes keep the tail balanced:
//   c <= 11 (n <= 2^17): no fold - one 256-thread workgroup per (window, bit) walks all partial sums of its window
//            (msm_bitplane_kernel<F, false>), whatever their distribution over the buckets;
//   c = 16 (beyond): 14 bits in the top window - every window is equally full (and fewer digit rows than c = 12 .. 15).
static inline int msm_pick_c(size_t n) {
    if (n > ((size_t)1 << 17)) return 16;
    int lg = 0;
    while (((size_t)1 << lg) < n) lg++;
    int c = lg - 4;
    if (c < 2) c = 2;
    if (c > 11) c = 11;
    return c;
}
// tables == 1: W = ceil(254 / c) windows.  tables == J > 1 (registered bases with precomputed 2^(part * j) multiples,
// part = table_bits, default 256 / J): the scalar bits split into J parts of `part` bits; c must divide the part width.
// c > 16 ("wide" windows, one bucket window per table: c == part) needs J * c >= 254 and runs the three-level sort and the
// two-axis bucket fold; it pays when every bucket still receives tens of points (J * n >> 2^(c-1)).
static constexpr int MSM_C_MAX = 23;
static constexpr int MSM_BIAS_BITS = 288;  // digit rows * window bits never exceed this (bias[10] = 320 bits, one word of headroom for the carry)
static inline msm_plan_t msm_make_plan(size_t n, int c_override = 0, int tables = 1, int table_bits = 0) {
    msm_plan_t p;
    p.n = n;
    p.J = tables < 1 ? 1 : tables;
    if (p.J == 1) {
        p.c = c_override ? c_override : msm_pick_c(n);
        if (p.c > 16) p.c = 16;
        p.W = (254 + p.c - 1) / p.c;
    } else {
        const int part = table_bits > 0 ? table_bits : 256 / p.J;
        if (c_override) {
            p.c = c_override;
            if (p.c > part) p.c = part;
            while (part % p.c) p.c--;  // largest divisor of the part width not above the request
            if (p.c > 16 && p.c != part) {  // wide windows exist only as one window per table
                p.c = 16;
                while (part % p.c) p.c--;
            }
            if (p.c < 2) p.c = part <= 16 ? part : 2;
        } else if (part <= 16) {
            p.c = n >= 4096 ? 16 : 8;
            if (p.c > part) p.c = part;
            while (part % p.c) p.c--;
            if (p.c < 2) p.c = part;  // a prime part width (13 x 20 tables ...) has no smaller window: one window per table set
        } else {
            // wide tables: the divisor d of the part width (d <= 16, or the whole part) with the least estimated work:
            // J * (part / d) * n bucket additions + ~8 addition-equivalents of sort / fold overhead per bucket
            double best = 0;
            p.c = 0;
            for (int d = 2; d <= part && d <= MSM_C_MAX; d++) {
                if (part % d || (d > 16 && d != part)) continue;
                const double cost = (double)p.J * (part / d) * (double)n + (double)(part / d) * (double)((size_t)1 << (d - 1)) * 8.0;
                if (p.c == 0 || cost < best) {
                    best = cost;
                    p.c = d;
                }
            }
            if (p.c == 0) p.c = 1;
        }
        p.W = part / p.c;
    }
    p.Wd = p.W * p.J;
    p.nb = 1u << (p.c - 1);
    p.nbt = (uint32_t)p.W * p.nb;
    // scalars per (chunk, window) workgroup: keep ~2^18 entries per private scatter region whatever the table count
    p.chunk = (1u << 18) / (uint32_t)p.J;
    p.chunk &= ~7u;  // multiple of 8: 16-byte digit loads
    if (p.chunk > n) p.chunk = (uint32_t)(n ? n : 1);
    p.nchunks = (uint32_t)((n + p.chunk - 1) / p.chunk);
    const int env_S = tuning().seg, env_S2 = tuning().seg2, env_L = tuning().fold_l;  // experiment overrides (tuning.hip.h), 0 = planner
    // points per accumulate thread.  Long segments amortise the partial-sum flushes (every thread leaves one partial sum per
    // bucket it touches and the tail pays two general additions for each), but the grid should be whole rounds of one wave
    // per SIMD (2^16 threads: one resident wave already keeps the multiplier ~95 % busy, tools/ecbench.hip): a grid of 1 088
    // waves on 1 024 SIMDs runs for two rounds.  So: the fewest rounds k that keep S <= 64 (128 for the biggest MSMs), then
    // the S that fills them - S = 16 at 2^16 x 16 tables, 64 from 2^18 on, 18 for 70 000 points.
    {
        const size_t E = (size_t)p.Wd * n;
        const size_t smax = E >= ((size_t)1 << 27) ? 128 : 64, round = (size_t)1 << 16;
        const size_t k = (E + smax * round - 1) / (smax * round);
        const size_t sfill = k ? (E + k * round - 1) / (k * round) : 4;
        p.S = (uint32_t)(sfill < 4 ? 4 : sfill);
    }
    if (env_S > 0) p.S = env_S;
    p.S2 = env_S2 > 1 ? env_S2 : 16;
    p.L = env_L > 0 ? (uint32_t)env_L : 8;
    if (p.L > p.nb) p.L = p.nb;
    while (p.nb % p.L) p.L--;
    p.rounds = 0;
    size_t m = ((size_t)p.J * n + p.S - 1) / p.S;
    while (m > 1) {
        m = (m + p.S2 - 1) / p.S2;
        p.rounds++;
    }
    for (int i = 0; i < 10; i++) p.bias[i] = 0;
    for (int w = 0; w < p.Wd; w++) {
        int bit = p.c - 1 + p.c * w;
        if (bit >= MSM_BIAS_BITS) break;  // unreachable for geometries check_tables() admits; never write past bias[]
        p.bias[bit / 32] |= 1u << (bit % 32);  // bits are distinct: no carries while building the constant
    }
    return p;
}

// ------------------------------------------------------------------------------------------
// generic exclusive scan (u32), tiles of 2048 per 256-thread block, recursive over block sums
// ------------------------------------------------------------------------------------------
static constexpr int SCAN_TILE = 2048;
static __global__ void scan_tile_kernel(const uint32_t* in, uint32_t* out, uint32_t* block_sums, size_t n) {
    __shared__ uint32_t sh[256];
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        s += v[i];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan of per-thread sums
        uint32_t t = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0u;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = sh[threadIdx.x] - s;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 255 && block_sums) block_sums[blockIdx.x] = sh[255];
}
static __global__ void scan_add_kernel(uint32_t* out, const uint32_t* block_offsets, size_t n) {
    const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * 8;
    const uint32_t add = block_offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < 8; i++)
        if (base + i < n) out[base + i] += add;
}
static inline size_t scan_tmp_elems(size_t n) {
    size_t tot = 0;
    while (n > 1) {
        n = (n + SCAN_TILE - 1) / SCAN_TILE;
        tot += n + 1;
        if (n == 1) break;
    }
    return tot + 8;
}
// Mid-size scans (2 049 .. 33 792 counters: the bucket tables of a small MSM) in ONE launch: a 1 024-thread workgroup stages the
// array in LDS (<= 132 KB), every thread scans its own run of `per` words (per odd: conflict-free strides), the thread totals
// are scanned by shuffles, and the result goes back through LDS.  Three launches (tiles, tile sums, add) otherwise.
static constexpr uint32_t SCAN_ONE_THREADS = 1024, SCAN_ONE_MAX = 33 * SCAN_ONE_THREADS;
static __global__ void __launch_bounds__(1024) scan_one_block_kernel(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t per) {
    extern __shared__ uint32_t scan_sm[];
    __shared__ uint32_t wave_tot[16];
    const uint32_t t = threadIdx.x;
    for (uint32_t i = t; i < n; i += SCAN_ONE_THREADS) scan_sm[i] = in[i];
    __syncthreads();
    const uint32_t lo = t * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += scan_sm[i];
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)incl, off);
        if ((t & 63) >= (uint32_t)off) incl += u;
    }
    if ((t & 63) == 63) wave_tot[t >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < (t >> 6); w++) base += wave_tot[w];
    uint32_t run = base + incl - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t v = scan_sm[i];
        scan_sm[i] = run;
        run += v;
    }
    __syncthreads();
    for (uint32_t i = t; i < n; i += SCAN_ONE_THREADS) out[i] = scan_sm[i];
}
// out may alias in
static inline void exclusive_scan_u32(hipStream_t st, const uint32_t* in, uint32_t* out, size_t n, uint32_t* tmp) {
    if (n == 0) return;
    const int one_launch = tuning().scan1;  // A/B switch
    if (one_launch && n > (size_t)SCAN_TILE && n <= (size_t)SCAN_ONE_MAX) {
        const uint32_t per = (uint32_t)((n + SCAN_ONE_THREADS - 1) / SCAN_ONE_THREADS) | 1u;
        hipLaunchKernelGGL(scan_one_block_kernel, dim3(1), dim3(SCAN_ONE_THREADS), (size_t)n * 4, st, in, out, (uint32_t)n, per);
        return;
    }
    const size_t blocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(scan_tile_kernel, dim3((unsigned)blocks), dim3(256), 0, st, in, out, tmp, n);
    if (blocks > 1) {
        exclusive_scan_u32(st, tmp, tmp, blocks, tmp + blocks + 1);
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, tmp, n);
    }
}

// ------------------------------------------------------------------------------------------
// 0. base conversion: Rust `Affine` (x, y Montgomery R = 2^384, infinity flag; stride bytes) -> g1_aff_mem_t
// ------------------------------------------------------------------------------------------
// form406 (G1 only): the slot becomes a g1_lazy_slot_t (ffl.hip.h) - the unpacked canonical residues of x * 2^406, y * 2^406, the
// operand form of the lazily reduced accumulate arithmetic - instead of the exact internal form x * 2^377: the same single
// product per coordinate, another constant.
template <class F>
SV_HD void store_base_slot(aff_mem_t<F>* slot, const uint32_t* xw, const uint32_t* yw, bool inf, int form406) {
    aff_t<F> a = aff_t<F>::inf();
    if (!inf) {
        a.x = F::from_raw_words(xw);
        a.y = F::from_raw_words(yw);
    }
    store_aff<F>(slot, a);
}
template <>
SV_HD void store_base_slot<fq_t>(aff_mem_t<fq_t>* slot, const uint32_t* xw, const uint32_t* yw, bool inf, int form406) {
    if (form406) {
        const fq_t c = fq_t::from_table(FqLConv::C399);  // memory form x 2^384 -> x 2^406
        g1_lazy_slot_t::store(slot, fq_t::unpack(xw) * c, fq_t::unpack(yw) * c, inf);
        return;
    }
    g1_aff_t a = g1_aff_t::inf();
    if (!inf) {
        a.x = fq_t::from_raw_words(xw);
        a.y = fq_t::from_raw_words(yw);
    }
    store_aff<fq_t>(slot, a);
}
template <>
SV_HD void store_base_slot<fq2_t>(aff_mem_t<fq2_t>* slot, const uint32_t* xw, const uint32_t* yw, bool inf, int form406) {
    if (form406) {
        const fq_t c = fq_t::from_table(FqLConv::C399);  // memory form x 2^384 -> x 2^406, per component
        g2_lazy_slot_t::store(slot, {fq_t::unpack(xw) * c, fq_t::unpack(xw + 12) * c}, {fq_t::unpack(yw) * c, fq_t::unpack(yw + 12) * c}, inf);
        return;
    }
    aff_t<fq2_t> a = aff_t<fq2_t>::inf();
    if (!inf) {
        a.x = fq2_t::from_raw_words(xw);
        a.y = fq2_t::from_raw_words(yw);
    }
    store_aff<fq2_t>(slot, a);
}
// in-place: exact slot -> g2_lazy_slot_t (the last step of a G2 registration)
static __global__ void g2_bases_to_form406_kernel(aff_mem_t<fq2_t>* slots, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fq_t c = fq_t::from_table(FqLConv::C406);
    const aff_t<fq2_t> a = load_aff<fq2_t>(&slots[i]);
    g2_lazy_slot_t::store(&slots[i], {a.x.c0 * c, a.x.c1 * c}, {a.y.c0 * c, a.y.c1 * c}, a.is_inf());
}
// in-place: exact slot -> g1_lazy_slot_t (the last step of a G1 registration, after the tables have been derived from one another)
static __global__ void g1_bases_to_form406_kernel(g1_aff_mem_t* slots, size_t n) {
    const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fq_t c = fq_t::from_table(FqLConv::C406);
    const g1_aff_t a = g1_load_aff(&slots[i]);
    g1_lazy_slot_t::store(&slots[i], a.x * c, a.y * c, a.is_inf());
}
template <class F>
__global__ void convert_bases_kernel(const uint8_t* in, size_t stride, size_t n, aff_mem_t<F>* out, int form406) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = (const uint32_t*)(in + i * stride);  // stride is a multiple of 8 (Rust layout)
    constexpr int MW = F::MEM_WORDS;
    uint32_t xw[MW], yw[MW];
#pragma unroll
    for (int k = 0; k < MW; k++) {
        xw[k] = src[k];
        yw[k] = src[MW + k];
    }
    const uint32_t inf = src[2 * MW] & 0xffu;
    store_base_slot<F>(&out[i], xw, yw, inf != 0, form406);
}

// ------------------------------------------------------------------------------------------
// 1. digits (the scalar-read phase)
// ------------------------------------------------------------------------------------------
struct msm_digit_params_t {
    uint32_t bias[10];
    int c, W;
    size_t n;
    int montgomery;  // scalars are Fr elements in Montgomery form: fuse Fr::to_bigint (kzg10/mod.rs:469-474) into the read
};
template <class DT>  // uint16_t for c <= 16, uint32_t for wider windows
__global__ void __launch_bounds__(256) msm_digits_kernel(const uint4* __restrict__ scalars, DT* __restrict__ digits,
                                                         msm_digit_params_t p) {
    // 256 scalars per block iteration: coalesced 16-byte loads into LDS, then one scalar per thread
    __shared__ uint4 stage[512];
    const size_t nblk = (p.n + 255) / 256;
    for (size_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const size_t first = blk * 256;
        const size_t cnt = (p.n - first < 256) ? (p.n - first) : 256;
        for (int k = threadIdx.x; k < 512; k += 256)
            if ((size_t)(k >> 1) < cnt) stage[k] = scalars[first * 2 + k];
        __syncthreads();
        if (threadIdx.x < cnt) {
            const uint4 lo = stage[2 * threadIdx.x], hi = stage[2 * threadIdx.x + 1];
            uint32_t s[11] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w, 0u, 0u, 0u};
            if (p.montgomery) {
                // a*2^256 read as internal a*2^-5 (see ff.hip.h): one Montgomery product by the integer 2^5 gives a
                fr_t c32 = fr_t::zero();
                c32.v[0] = 32;
                (fr_t::unpack(s) * c32).pack(s);
            }
            uint64_t carry = 0;
#pragma unroll
            for (int k = 0; k < 10; k++) {
                carry += (uint64_t)s[k] + p.bias[k];
                s[k] = (uint32_t)carry;
                carry >>= 32;
            }
            const uint32_t mask = (1u << p.c) - 1;
            const size_t i = first + threadIdx.x;
            for (int w = 0; w < p.W; w++) {
                const int bit = p.c * w, wi = bit >> 5, sh = bit & 31;
                uint64_t two = (uint64_t)s[wi] | ((uint64_t)s[wi + 1] << 32);
                digits[(size_t)w * p.n + i] = (DT)((uint32_t)(two >> sh) & mask);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------
// 5./6. accumulate + reduce rounds
// ------------------------------------------------------------------------------------------
// cnt_out[k] = ceil(cnt_in[k] / S)   (level 0: cnt_in = bucket sizes)
static __global__ void msm_alloc_kernel(const uint32_t* cnt_in, uint32_t* cnt_out, uint32_t nbt, uint32_t S) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nbt) return;
    cnt_out[k] = (k == nbt) ? 0u : (cnt_in[k] + S - 1) / S;
}
// largest k in [0, nbt) with start[k] <= t   (start is non-decreasing, start[nbt] = total > t)
__device__ __forceinline__ uint32_t find_bucket(const uint32_t* start, uint32_t nbt, uint32_t t) {
    uint32_t lo = 0, hi = nbt;  // invariant: start[lo] <= t < start[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= t)
            lo = mid;
        else
            hi = mid;
    }
    return lo;
}
template <class F>
__global__ void __launch_bounds__(256) msm_reduce_kernel(const xyzz_mem_t<F>* __restrict__ in,
                                                         const uint32_t* __restrict__ in_start,
                                                         const uint32_t* __restrict__ in_cnt,
                                                         const uint32_t* __restrict__ out_start,
                                                         xyzz_mem_t<F>* __restrict__ out, uint32_t nbt, uint32_t S2) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= out_start[nbt]) return;
    const uint32_t k = find_bucket(out_start, nbt, t);
    const uint32_t j = t - out_start[k];
    const uint32_t lo = in_start[k] + j * S2;
    uint32_t hi = lo + S2;
    const uint32_t end = in_start[k] + in_cnt[k];
    if (hi > end) hi = end;
    xyzz_t<F> acc = load_xyzz<F>(&in[lo]);
    for (uint32_t pos = lo + 1; pos < hi; pos++) acc.add(load_xyzz<F>(&in[pos]));
    store_xyzz<F>(&out[t], acc);
}

// acc[k * L + slot] += the partial sums of bucket k (the bucket sink of a chunked MSM, runtime.hip.h::msm_bucket_sink_t)
template <class F>
__global__ void __launch_bounds__(256) msm_bucket_merge_kernel(const xyzz_mem_t<F>* __restrict__ part, const uint32_t* __restrict__ start,
                                                               const uint32_t* __restrict__ cnt, xyzz_mem_t<F>* __restrict__ acc, uint32_t nbt, uint32_t L,
                                                               uint32_t slot) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nbt) return;
    const uint32_t c = cnt[k];
    if (!c) return;
    const uint32_t lo = start[k];
    xyzz_mem_t<F>* dst = &acc[(size_t)k * L + slot];
    xyzz_t<F> a = load_xyzz<F>(dst);
    for (uint32_t i = 0; i < c; i++) a.add(load_xyzz<F>(&part[lo + i]));
    store_xyzz<F>(dst, a);
}
// the lists a tail kernel reads from a sink: bucket k owns the L consecutive slots [k * L, (k + 1) * L)
static __global__ void msm_sink_lists_kernel(uint32_t* __restrict__ start, uint32_t* __restrict__ cnt, uint32_t nbt, uint32_t L) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nbt) return;
    start[k] = k * L;
    cnt[k] = k < nbt ? L : 0u;
}

// ------------------------------------------------------------------------------------------
// 7.-9. tail: weighted bucket sum  sum_b (b + 1) B_b  per window, then the combination of the windows.
//
// Everything here is a short chain of DEPENDENT group operations (one wave runs one XYZZ addition in ~10-13 us,
// tools/ecbench.hip), so the tail is organised to minimise that depth, not the operation count:
//   7a fold      (windows with >= 2^11 buckets) bucket index b = hi * 2^m + lo:
//                   sum_b (b + 1) B_b = 2^m * sum_hi hi * H_hi + sum_lo (lo + 1) * L_lo,  H_hi = sum_lo B_(hi,lo),  L_lo = sum_hi B_(hi,lo)
//                one workgroup per row / column sum - 256 threads (a few serial additions per lane, then an 8-level tree) when
//                the MSM is small and latency-bound, one wave (less tree overhead per output) for the 2^21 buckets of a big one;
//                the leftover partial sums of the accumulate phase (<= tail_partials per bucket) are consumed here directly;
//   7b bit planes  sum_i (i + 1) P_i = sum_j 2^j S_j with S_j = sum over the entries whose weight has bit j set: one
//                workgroup per (tail window, bit) sums its subset as a tree - no running sums, no double-and-add chains;
//   9  the remaining sum_p 2^p (planes at bit position p) is a Horner chain of <= ~270 doublings with wave-uniform data: that
//      is host work, like the reference's host-side collapse of the per-GPU results (algorithms/cuda/cuda/snarkvm.cu:290-295)
//      - runtime.hip.h::msm_finish_host runs it with this same arithmetic compiled for the host (~1 us per operation on one
//      core instead of ~10 us on one GPU lane).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ fq_t shfl_xor_field(const fq_t& a, int mask) {
    fq_t r;
#pragma unroll
    for (int i = 0; i < fq_t::N; i++) r.v[i] = (uint32_t)__shfl_xor((int)a.v[i], mask);
    return r;
}
__device__ __forceinline__ fq2_t shfl_xor_field(const fq2_t& a, int mask) { return {shfl_xor_field(a.c0, mask), shfl_xor_field(a.c1, mask)}; }
__device__ __forceinline__ fqz_t shfl_xor_field(const fqz_t& a, int mask) {
    fqz_t r;
#pragma unroll
    for (int i = 0; i < fqz_t::N; i++) r.a.v[i] = __shfl_xor(a.a.v[i], mask);
    SV_OPAQUE_13(r.a.v);  // see quad_perm_field below
    return r;
}
template <class F>
__device__ __forceinline__ xyzz_t<F> shfl_xor_point(const xyzz_t<F>& a, int mask) {
    return {shfl_xor_field(a.x, mask), shfl_xor_field(a.y, mask), shfl_xor_field(a.zz, mask), shfl_xor_field(a.zzz, mask)};
}
// DPP quad permutation (lanes 4k .. 4k + 3 read each other's registers inside the VALU: no LDS traffic); CTRL = the four
// source lanes, two bits each.  quad_bcast<L>: every lane of the quad reads lane L.
template <int CTRL>
__device__ __forceinline__ fq_t quad_perm_field(const fq_t& a) {
    fq_t r;
#pragma unroll
    for (int i = 0; i < fq_t::N; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], CTRL, 0xf, 0xf, true);
    return r;
}
template <int CTRL>
__device__ __forceinline__ fq2_t quad_perm_field(const fq2_t& a) {
    return {quad_perm_field<CTRL>(a.c0), quad_perm_field<CTRL>(a.c1)};
}
// The permuted limbs pass through empty asm statements (ffl.hip.h: SV_OPAQUE_13) before anything consumes them.  Without that the
// compiler folds the DPP move into the consuming instruction (v_subrev_u32_dpp ... quad_perm:[3,3,3,3] in Y3 = bcast<2>(m4) -
// bcast<3>(m4) of quad_add), and on gfx950 / ROCm 7.2 that folded form computed with the lane's OWN limb when its other operand had
// just been written by a DPP move: Y3 = m4[2] - m4[own lane] on three lanes of every quad (tools/exp/lazytail_dev.hip prints it).
// The exact arithmetic never hits the pattern: its subtractions are borrow chains, not single limb operations.
template <int CTRL>
__device__ __forceinline__ fqz_t quad_perm_field(const fqz_t& a) {
    fqz_t r;
#pragma unroll
    for (int i = 0; i < fqz_t::N; i++) r.a.v[i] = __builtin_amdgcn_mov_dpp(a.a.v[i], CTRL, 0xf, 0xf, true);
    SV_OPAQUE_13(r.a.v);
    return r;
}
template <int L, class F>
__device__ __forceinline__ F quad_bcast(const F& a) {
    return quad_perm_field<L * 0x55>(a);
}
__device__ __forceinline__ fq_t select_field(bool c, const fq_t& a, const fq_t& b) {  // c ? a : b
    fq_t r;
#pragma unroll
    for (int i = 0; i < fq_t::N; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}
__device__ __forceinline__ fq2_t select_field(bool c, const fq2_t& a, const fq2_t& b) {
    return {select_field(c, a.c0, b.c0), select_field(c, a.c1, b.c1)};
}
__device__ __forceinline__ fqz_t select_field(bool c, const fqz_t& a, const fqz_t& b) {
    fqz_t r;
#pragma unroll
    for (int i = 0; i < fqz_t::N; i++) r.a.v[i] = c ? a.a.v[i] : b.a.v[i];
    return r;
}
template <class F>
__device__ __forceinline__ xyzz_t<F> select_point(bool c, const xyzz_t<F>& a, const xyzz_t<F>& b) {
    return {select_field(c, a.x, b.x), select_field(c, a.y, b.y), select_field(c, a.zz, b.zz), select_field(c, a.zzz, b.zzz)};
}

// acc += o (add-2008-s) computed by the FOUR lanes of a DPP quad together.  Precondition: the four lanes hold bit-identical
// (acc, o); postcondition: they hold the bit-identical sum.  A serial tail addition is latency-bound (one wave issues one
// instruction per ~4.5 cycles whatever the other 63 lanes do: 12.8 us per addition, tools/ecbench.hip), and in a tree
// reduction the lanes hold redundant copies anyway - so the 14 field products of the formula are dealt to the four lanes as
// four rounds of one product each:
//   round 1   U1 = X1 ZZ2 | U2 = X2 ZZ1 | S1 = Y1 ZZZ2 | S2 = Y2 ZZZ1      then  P = U2 - U1 (lanes 0, 1), R = S2 - S1 (lanes 2, 3)
//   round 2   PP = P^2    | ZZ1 ZZ2     | R^2          | ZZZ1 ZZZ2
//   round 3   PPP = P PP  | Q = U1 PP   | ZZ3 = ZZ1 ZZ2 PP | -
//   round 4   ZZZ3 = ZZZ1 ZZZ2 PPP | -  | R (Q - X3)   | S1 PPP            with X3 = R^2 - PPP - 2 Q
// Operands move between the lanes as DPP quad permutations.  Infinity operands are a final select; P = +-Q (U1 == U2) in any
// quad sends the whole wave through the plain formula (wave-uniform branch; every lane of a quad takes the same path).
// Fq2 (TAIL_FLAGGED): the fallback is not in the kernel.  Equal x coordinates anywhere in the wave set `dbl` and abandon the addition - the kernel marks its output
// and msm_*_fix_kernel recomputes it with the plain law.
// The Fq2 tail kernels therefore hold neither a second copy of the addition law per site (~48 000 instructions each) nor a device-function call: round 6
// measured calls inside these ~260 000-instruction kernels coming back with live registers of the caller overwritten, with the compiler's interprocedural
// register allocation on AND off, from one source revision to the next (tests/test_gpu_parity.py::test_msm_of_repeated_points_repeats_and_matches_the_oracle).
template <class F>
struct TAIL_FLAGGED {
    static constexpr bool value = sizeof(F) > 64;
};
template <class F>
__device__ __forceinline__ void tail_add(xyzz_t<F>& acc, const xyzz_t<F>& o, bool& dbl) {  // the plain one-lane addition of a tail kernel
    if constexpr (TAIL_FLAGGED<F>::value)
        acc.add_flag(o, dbl);
    else
        acc.add(o);
}
template <class F>
__device__ __forceinline__ void quad_add(xyzz_t<F>& acc, const xyzz_t<F>& o, bool& dbl) {
    const uint32_t r = threadIdx.x & 3;
    const bool inf1 = acc.is_inf(), inf2 = o.is_inf();
    F a = select_field(r < 2, select_field(r == 0, acc.x, o.x), select_field(r == 2, acc.y, o.y));
    F b = select_field(r < 2, select_field(r == 0, o.zz, acc.zz), select_field(r == 2, o.zzz, acc.zzz));
    const F m1 = a * b;                          // U1 | U2 | S1 | S2
    const F t = quad_perm_field<0xB1>(m1);       // U2 | U1 | S2 | S1   (lanes swapped inside pairs)
    const F d = select_field((r & 1) != 0, m1, t) - select_field((r & 1) != 0, t, m1);  // P | P | R | R
    const bool same_x = !inf1 && !inf2 && r < 2 && d.is_zero();
    if (__ballot(same_x) != 0) {  // wave-uniform
        if constexpr (TAIL_FLAGGED<F>::value)
            dbl = true;  // (every lane of the wave: the flag is per output anyway)
        else
            acc.add(o);  // Fq: the plain law, inlined
        return;
    }
    a = select_field((r & 1) == 0, d, select_field(r == 1, acc.zz, acc.zzz));
    b = select_field((r & 1) == 0, d, select_field(r == 1, o.zz, o.zzz));
    const F m2 = a * b;                          // PP | ZZ1 ZZ2 | R^2 | ZZZ1 ZZZ2
    const F pp = quad_bcast<0>(m2);
    a = select_field(r == 1, t, select_field(r == 2, quad_bcast<1>(m2), d));
    const F m3 = a * pp;                         // PPP | Q | ZZ3 | (unused)
    const F ppp = quad_bcast<0>(m3), q = quad_bcast<1>(m3);
    const F x3 = m2 - ppp - q.dbl();             // X3 on lane 2
    a = select_field(r == 0, quad_bcast<3>(m2), select_field(r == 2, d, t));
    b = select_field(r == 2, q - x3, ppp);
    const F m4 = a * b;                          // ZZZ3 | (unused) | R (Q - X3) | S1 PPP
    xyzz_t<F> res;
    res.x = quad_bcast<2>(x3);
    res.y = quad_bcast<2>(m4) - quad_bcast<3>(m4);
    res.zz = quad_bcast<2>(m3);
    res.zzz = quad_bcast<0>(m4);
    acc = select_point(inf2, acc, select_point(inf1, o, res));
}
template <class F>
__device__ __forceinline__ void quad_add(xyzz_t<F>& acc, const xyzz_t<F>& o) {  // callers outside the tail kernels (group.hip.h: G1 only)
    static_assert(!TAIL_FLAGGED<F>::value, "the Fq2 form reports P + P instead of computing it");
    bool unused = false;
    quad_add(acc, o, unused);
}
// Sum of `acc` over the lanes of a workgroup of 64, 256 or 1024 threads; the result is valid in thread 0.  Level 1 (lane
// pairs) is a plain addition with the operands in the same order on both lanes, so that from level 2 on the four lanes of a
// quad hold bit-identical operands and share every addition (quad_add); the wave totals go through LDS to wave 0, whose quads
// add them pairwise and finish with the same exchange levels.  `sh`: one point per wave.
// hex != 0 (Fq2 only; tuning hex2): from the level on where a wave holds <= 4 distinct additions - exchange distance 8 - every addition is shared by the
// SIXTEEN lanes of a DPP row (hex2.hip.h::hex_add: one Fq product per lane and round instead of one Fq2 product), which needs the sixteen lanes to hold
// bit-identical operands: the distance-4 level (and every later one) then adds its two operands in the same order on both sides.  Same sums, bit for bit.
// from_quads: the four lanes of every quad already hold ONE bit-identical value (a quad-strided accumulation, tail_quad_accumulate below): the two
// intra-quad levels - the only ones with a plain, one-lane addition - do not exist.
template <class F>
__device__ __forceinline__ void block_sum(xyzz_t<F>& acc, xyzz_mem_t<F>* sh, int hex, bool from_quads, bool& dbl) {
    if (!from_quads) {
        const xyzz_t<F> o = shfl_xor_point(acc, 1);
        const bool odd = (threadIdx.x & 1) != 0;
        xyzz_t<F> lo = select_point(odd, o, acc);
        tail_add(lo, select_point(odd, acc, o), dbl);
        acc = lo;
    }
    if (!from_quads) {
        const xyzz_t<F> o = shfl_xor_point(acc, 2);
        const bool hi = (threadIdx.x & 2) != 0;
        xyzz_t<F> lo = select_point(hi, o, acc);
        quad_add(lo, select_point(hi, acc, o), dbl);
        acc = lo;
    }
    if constexpr (std::is_same<F, fq2_t>::value) {
        if (hex) {
            {
                const xyzz_t<F> o = shfl_xor_point(acc, 4);
                const bool hi = (threadIdx.x & 4) != 0;
                xyzz_t<F> lo = select_point(hi, o, acc);
                quad_add(lo, select_point(hi, acc, o), dbl);
                acc = lo;
            }
#pragma unroll 1
            for (int off = 8; off < 64; off <<= 1) {
                const xyzz_t<F> o = shfl_xor_point(acc, off);
                const bool hi = (threadIdx.x & off) != 0;
                xyzz_t<F> lo = select_point(hi, o, acc);
                hex_add(lo, select_point(hi, acc, o), dbl);
                acc = lo;
            }
            if (blockDim.x == 64) return;
            const uint32_t wv = threadIdx.x >> 6, nw = blockDim.x >> 6;  // 2 or 4 waves (<= 256 threads: four rows of wave 0 take the <= 2 pairs)
            if ((threadIdx.x & 63) == 0) store_xyzz<F>(&sh[wv], acc);
            __syncthreads();
            if (threadIdx.x < 64) {
                const uint32_t pair = (threadIdx.x >> 4) & (nw / 2 - 1);  // row -> the pair of wave totals it adds
                acc = load_xyzz<F>(&sh[2 * pair]);
                hex_add(acc, load_xyzz<F>(&sh[2 * pair + 1]), dbl);
#pragma unroll 1
                for (uint32_t off = 16; off < 8 * nw; off <<= 1) {
                    const xyzz_t<F> o = shfl_xor_point(acc, (int)off);
                    const bool hi = (threadIdx.x & off) != 0;
                    xyzz_t<F> lo = select_point(hi, o, acc);
                    hex_add(lo, select_point(hi, acc, o), dbl);
                    acc = lo;
                }
            }
            return;
        }
    }
#pragma unroll 1
    for (int off = 4; off < 64; off <<= 1) quad_add(acc, shfl_xor_point(acc, off), dbl);
    if (blockDim.x == 64) return;
    const uint32_t wv = threadIdx.x >> 6, nw = blockDim.x >> 6;  // 4 or 16 waves
    if ((threadIdx.x & 63) == 0) store_xyzz<F>(&sh[wv], acc);
    __syncthreads();
    if (threadIdx.x < 64) {
        const uint32_t pair = (threadIdx.x >> 2) & (nw / 2 - 1);  // quad -> the pair of wave totals it adds
        acc = load_xyzz<F>(&sh[2 * pair]);
        quad_add(acc, load_xyzz<F>(&sh[2 * pair + 1]), dbl);
#pragma unroll 1
        for (uint32_t off = 4; off < 2 * nw; off <<= 1) quad_add(acc, shfl_xor_point(acc, (int)off), dbl);
    }
}
// The end of a tail kernel: thread 0 stores the workgroup's sum; Fq2: and whether ANY addition on the way met equal x coordinates - the output is then meaningless and
// msm_*_fix_kernel (below) recomputes it.  Every thread of the workgroup must arrive.
template <class F>
__device__ __forceinline__ void tail_store(xyzz_mem_t<F>* out, uint32_t* flags, size_t slot, const xyzz_t<F>& acc, bool dbl) {
    if constexpr (TAIL_FLAGGED<F>::value) {
        const int any = __syncthreads_or(dbl ? 1 : 0);
        if (threadIdx.x == 0) flags[slot] = (uint32_t)any;
    }
    if (threadIdx.x == 0) store_xyzz<F>(&out[slot], acc);
}
// Register budget of the tail kernels: the G1 kernels need ~270 registers without a bound and would then run ONE 256-thread
// workgroup per CU - the 384 workgroups of a 2^15-bucket fold would take two rounds on 256 CUs; two waves per SIMD (256
// registers) put them all on the chip at once.  The Fq2 kernels keep the full file.
template <class F>
struct TAIL_WAVES {
    static constexpr int value = sizeof(F) <= 64 ? 2 : 1;
};
// In-place running sums of s[1 .. n] (s[0] = 0) by a workgroup of 64 or 256 threads: s[i] = s[1] + ... + s[i].  tmp: 256 words.
__device__ __forceinline__ void block_running_sums(uint32_t* s, uint32_t n, uint32_t* tmp) {
    const uint32_t B = blockDim.x, t = threadIdx.x;
    const uint32_t per = (n + B - 1) / B;
    const uint32_t lo = 1 + t * per, hi = lo + per < n + 1 ? lo + per : n + 1;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += s[i];
    tmp[t] = sum;
    __syncthreads();
    if (t < 64) {  // exclusive scan of the B thread sums by one wave: B / 64 per lane, then a shuffle scan
        const uint32_t e = B >> 6;
        uint32_t v[4], own = 0;
        for (uint32_t q = 0; q < e; q++) {
            v[q] = tmp[t * e + q];
            own += v[q];
        }
        uint32_t incl = own;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = (uint32_t)__shfl_up((int)incl, off);
            if (t >= (uint32_t)off) incl += u;
        }
        uint32_t run = incl - own;
        for (uint32_t q = 0; q < e; q++) {
            tmp[t * e + q] = run;
            run += v[q];
        }
    }
    __syncthreads();
    uint32_t run = tmp[t];
    for (uint32_t i = lo; i < hi; i++) {
        run += s[i];
        s[i] = run;
    }
    __syncthreads();
}
// the segment of position p in the offsets off[0 .. nseg] (off[0] = 0 <= p < off[nseg]): the largest i with off[i] <= p
// (empty segments share their offset with the next one and are never returned)
__device__ __forceinline__ uint32_t find_segment(const uint32_t* off, uint32_t nseg, uint32_t p) {
    uint32_t lo = 0, hi = nseg;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}
static constexpr uint32_t TAIL_MAX_SEG = 2048;  // buckets per fold column (2^hb, hb <= 11) / per unfolded window (2^10)
// 7a. grid (2^m + 2^hb, W): workgroups [0, 2^m) of a window fold columns (fixed lo -> L_lo, slot lo), workgroups
// [2^m, 2^m + 2^hb) fold rows (fixed hi -> H_hi, slot 2^m + hi - 1; H_0 has weight 0 and no slot).  Bucket k = w * nb + b
// holds cnt[k] partial sums at sums[start[k] ...].  out: per window 2^(m+1) dense slots (slot 2^(m+1) - 1 is unused).
// The partial sums of a row are one contiguous range; those of a column (2^hb buckets at stride 2^m) are flattened through
// running sums of their counts in LDS.  Either way lane t takes positions t, t + B, ... of the list: the work of a workgroup
// is (partial sums of its row / column) / B whatever their distribution over the buckets - a bucket that received half of the
// scalars (a witness full of ones) costs its row and its column a few more additions, not a reduce round with a host
// read-back in front of it.
// FLAT = false (big MSMs after their reduce rounds: <= TAIL_PARTIALS per bucket, thousands of buckets per column): lane ->
// buckets i, i + B, ... of the column, no LDS staging.
template <class F, bool FLAT>
__global__ void __launch_bounds__(256, TAIL_WAVES<F>::value) msm_fold_kernel(const xyzz_mem_t<F>* __restrict__ sums, const uint32_t* __restrict__ start,
                                                       const uint32_t* __restrict__ cnt, xyzz_mem_t<F>* __restrict__ out, int m, int hb, int hex, int quads,
                                                       uint32_t* __restrict__ flags) {
    __shared__ xyzz_mem_t<F> sh[16];
    __shared__ uint32_t s_off[FLAT ? TAIL_MAX_SEG + 1 : 1], s_start[FLAT ? TAIL_MAX_SEG : 1], s_tmp[FLAT ? 256 : 1];
    const uint32_t nlo = 1u << m, nhi = 1u << hb;
    const uint32_t w = blockIdx.y;
    const uint32_t kbase = w << (m + hb);
    // rows first: a row spans 2^m buckets, a column 2^hb <= 2^m - with an odd number of index bits a row is twice the work of a column, and
    // the workgroups that start last should be the short ones (2^24 x 22-bit windows: 3 072 workgroups for 2 048 resident waves)
    const bool column = blockIdx.x >= nhi;
    const uint32_t fixed = column ? blockIdx.x - nhi : blockIdx.x;
    if (!column && fixed == 0) return;
    const size_t slot = ((size_t)w << (m + 1)) + (column ? fixed : nlo + fixed - 1);
    bool dbl = false;  // Fq2: some addition met equal x coordinates (tail_store)
    if (!FLAT) {
        xyzz_t<F> acc = xyzz_t<F>::inf();
        // one loop for both shapes: a column is nhi buckets at stride 2^m, a row one contiguous range of partial sums
        const uint32_t k0 = kbase + (fixed << m);
        const uint32_t nouter = column ? nhi : 1u;
        for (uint32_t i = column ? threadIdx.x : 0u; i < nouter; i += column ? blockDim.x : 1u) {
            const uint32_t k = kbase + (i << m) + fixed;
            const uint32_t q0 = column ? start[k] : start[k0] + threadIdx.x;
            const uint32_t q1 = column ? q0 + cnt[k] : start[k0 + nlo - 1] + cnt[k0 + nlo - 1];
            for (uint32_t q = q0; q < q1; q += column ? 1u : blockDim.x) tail_add(acc, load_xyzz<F>(&sums[q]), dbl);
        }
        block_sum<F>(acc, sh, hex, false, dbl);
        tail_store<F>(out, flags, slot, acc, dbl);
        return;
    }
    uint32_t nseg;
    if (column) {
        for (uint32_t i = threadIdx.x; i < nhi; i += blockDim.x) {
            const uint32_t k = kbase + (i << m) + fixed;
            s_start[i] = start[k];
            s_off[i + 1] = cnt[k];
        }
        if (threadIdx.x == 0) s_off[0] = 0;
        __syncthreads();
        block_running_sums(s_off, nhi, s_tmp);
        nseg = nhi;
    } else {
        const uint32_t k0 = kbase + (fixed << m);
        if (threadIdx.x == 0) {
            s_start[0] = start[k0];
            s_off[0] = 0;
            s_off[1] = start[k0 + nlo - 1] + cnt[k0 + nlo - 1] - start[k0];
        }
        __syncthreads();
        nseg = 1;
    }
    const uint32_t total = s_off[nseg];
    xyzz_t<F> acc = xyzz_t<F>::inf();
    if (quads) {
        // quad-strided: every QUAD takes positions q, q + Q, ... of the list - its four lanes load the same partial sum - and adds them with the
        // quad-cooperative law.  No lane ever runs a plain addition alone (14 dependent field products on one lane: for Fq2 ~125 us, three times a
        // quad-cooperative one), and the tree starts at the quads.  The trip count is block-uniform: quad_add is a wave-wide operation.
        const uint32_t nq = blockDim.x >> 2, qid = threadIdx.x >> 2;
        const uint32_t iters = (total + nq - 1) / nq;
#pragma unroll 1
        for (uint32_t it = 0; it < iters; it++) {
            const uint32_t p = qid + it * nq;
            xyzz_t<F> x = xyzz_t<F>::inf();
            if (p < total) {
                const uint32_t i = find_segment(s_off, nseg, p);
                x = load_xyzz<F>(&sums[s_start[i] + (p - s_off[i])]);
            }
            if (it == 0)
                acc = x;
            else
                quad_add(acc, x, dbl);
        }
        block_sum<F>(acc, sh, hex, true, dbl);
    } else {
        for (uint32_t p = threadIdx.x; p < total; p += blockDim.x) {
            const uint32_t i = find_segment(s_off, nseg, p);
            tail_add(acc, load_xyzz<F>(&sums[s_start[i] + (p - s_off[i])]), dbl);
        }
        block_sum<F>(acc, sh, hex, false, dbl);
    }
    tail_store<F>(out, flags, slot, acc, dbl);
}
// The fold output `slot` again, for the outputs msm_fold_kernel flagged (Fq2: an addition met equal x coordinates - P + P or P - P): ONE wave per output and the plain law with its doubling, two
// sites.  Same grid as the fold; an unflagged workgroup returns at once (the common case: ~3 us per launch).  Equal partial sums are what a tiled or repeated base
// vector produces (the reference's own MSM benches tile a handful of points, benches/msm/variable_base.rs:29-32) - rare per output, not per run.
template <class F>
__global__ void __launch_bounds__(64) msm_fold_fix_kernel(const xyzz_mem_t<F>* __restrict__ sums, const uint32_t* __restrict__ start, const uint32_t* __restrict__ cnt,
                                                          xyzz_mem_t<F>* __restrict__ out, int m, int hb, const uint32_t* __restrict__ flags) {
    const uint32_t nlo = 1u << m, nhi = 1u << hb;
    const uint32_t w = blockIdx.y;
    const uint32_t kbase = w << (m + hb);
    const bool column = blockIdx.x >= nhi;
    const uint32_t fixed = column ? blockIdx.x - nhi : blockIdx.x;
    if (!column && fixed == 0) return;
    const size_t slot = ((size_t)w << (m + 1)) + (column ? fixed : nlo + fixed - 1);
    if (!flags[slot]) return;
    xyzz_t<F> acc = xyzz_t<F>::inf();
    const uint32_t k0 = kbase + (fixed << m);
    const uint32_t nouter = column ? nhi : 1u;
    for (uint32_t i = column ? threadIdx.x : 0u; i < nouter; i += column ? 64u : 1u) {
        const uint32_t k = kbase + (i << m) + fixed;
        const uint32_t q0 = column ? start[k] : start[k0] + threadIdx.x;
        const uint32_t q1 = column ? q0 + cnt[k] : start[k0 + nlo - 1] + cnt[k0 + nlo - 1];
#pragma unroll 1
        for (uint32_t q = q0; q < q1; q += column ? 1u : 64u) acc.add(load_xyzz<F>(&sums[q]));
    }
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) acc.add(shfl_xor_point(acc, off));
    if (threadIdx.x == 0) store_xyzz<F>(&out[slot], acc);
}
// 7b. grid (nbits, tail windows).  Tail window tw holds N entries, entry i has weight i + 1:
//   DENSE (after a fold): tw = 2 * w + sub; sub 0 = the L sums (N = 2^m), sub 1 = the H sums (N = 2^hb - 1); entry i is
//          sums[(w << (m + 1)) + (sub << m) + i];
//   else (small windows, nb <= 1024): tw = w, N = nb, entry i = the cnt[k] partial sums of bucket k = w * nb + i - one
//          contiguous range for the whole window, walked position by position like a fold row (the bucket of a position comes
//          from the window's `start` values in LDS).
// planes[tw * nbits + j] = sum of the entries of tw whose weight has bit j set.
template <class F, bool DENSE>
__global__ void __launch_bounds__(256, TAIL_WAVES<F>::value) msm_bitplane_kernel(const xyzz_mem_t<F>* __restrict__ sums, const uint32_t* __restrict__ start,
                                                           const uint32_t* __restrict__ cnt, xyzz_mem_t<F>* __restrict__ planes, uint32_t nb,
                                                           int m, int hb, int hex, int quads, uint32_t* __restrict__ flags) {
    __shared__ xyzz_mem_t<F> sh[16];
    __shared__ uint32_t s_off[DENSE ? 1 : TAIL_MAX_SEG + 1];
    const uint32_t j = blockIdx.x, tw = blockIdx.y, nbits = gridDim.x;
    const size_t slot = (size_t)tw * nbits + j;
    xyzz_t<F> acc = xyzz_t<F>::inf();
    bool dbl = false;  // Fq2: some addition met equal x coordinates (tail_store)
    if (DENSE) {
        const uint32_t w = tw >> 1, sub = tw & 1;
        const uint32_t N = sub ? (1u << hb) - 1 : (1u << m);
        const size_t base = ((size_t)w << (m + 1)) + ((size_t)sub << m);
        if (quads) {  // quad-strided, as in the fold: entry i belongs to quad i mod Q
            const uint32_t nq = blockDim.x >> 2, qid = threadIdx.x >> 2;
            const uint32_t iters = (N + nq - 1) / nq;
#pragma unroll 1
            for (uint32_t it = 0; it < iters; it++) {
                const uint32_t i = qid + it * nq;
                xyzz_t<F> x = xyzz_t<F>::inf();
                if (i < N && (((i + 1) >> j) & 1)) x = load_xyzz<F>(&sums[base + i]);
                if (it == 0)
                    acc = x;
                else
                    quad_add(acc, x, dbl);
            }
            block_sum<F>(acc, sh, hex, true, dbl);
            tail_store<F>(planes, flags, slot, acc, dbl);
            return;
        }
        for (uint32_t i = threadIdx.x; i < N; i += blockDim.x)
            if (((i + 1) >> j) & 1) tail_add(acc, load_xyzz<F>(&sums[base + i]), dbl);
    } else {
        const size_t base = (size_t)tw * nb;
        const uint32_t p0 = start[base];
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) s_off[i] = start[base + i] - p0;
        if (threadIdx.x == 0) s_off[nb] = start[base + nb - 1] + cnt[base + nb - 1] - p0;
        __syncthreads();
        const uint32_t total = s_off[nb];
        for (uint32_t p = threadIdx.x; p < total; p += blockDim.x) {
            const uint32_t i = find_segment(s_off, nb, p);
            if (((i + 1) >> j) & 1) tail_add(acc, load_xyzz<F>(&sums[p0 + p]), dbl);
        }
    }
    block_sum<F>(acc, sh, hex, false, dbl);
    tail_store<F>(planes, flags, slot, acc, dbl);
}
// The bit plane `slot` again, for the planes msm_bitplane_kernel flagged (see msm_fold_fix_kernel): one wave, the plain law.
template <class F, bool DENSE>
__global__ void __launch_bounds__(64) msm_bitplane_fix_kernel(const xyzz_mem_t<F>* __restrict__ sums, const uint32_t* __restrict__ start, const uint32_t* __restrict__ cnt,
                                                              xyzz_mem_t<F>* __restrict__ planes, uint32_t nb, int m, int hb, const uint32_t* __restrict__ flags) {
    __shared__ uint32_t s_off[DENSE ? 1 : TAIL_MAX_SEG + 1];
    const uint32_t j = blockIdx.x, tw = blockIdx.y, nbits = gridDim.x;
    const size_t slot = (size_t)tw * nbits + j;
    if (!flags[slot]) return;
    xyzz_t<F> acc = xyzz_t<F>::inf();
    // one loop for both shapes: positions p of a list of `total` entries, entry p has weight wgt(p) and lives at sums[first + p]
    size_t first;
    uint32_t total;
    if (DENSE) {
        const uint32_t w = tw >> 1, sub = tw & 1;
        total = sub ? (1u << hb) - 1 : (1u << m);
        first = ((size_t)w << (m + 1)) + ((size_t)sub << m);
    } else {
        const size_t base = (size_t)tw * nb;
        const uint32_t p0 = start[base];
        for (uint32_t i = threadIdx.x; i < nb; i += 64u) s_off[i] = start[base + i] - p0;
        if (threadIdx.x == 0) s_off[nb] = start[base + nb - 1] + cnt[base + nb - 1] - p0;
        __syncthreads();
        total = s_off[nb];
        first = p0;
    }
#pragma unroll 1
    for (uint32_t p = threadIdx.x; p < total; p += 64u) {
        const uint32_t i = DENSE ? p : find_segment(s_off, nb, p);
        if (((i + 1) >> j) & 1) acc.add(load_xyzz<F>(&sums[first + p]));
    }
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) acc.add(shfl_xor_point(acc, off));
    if (threadIdx.x == 0) store_xyzz<F>(&planes[slot], acc);
}

// The fold / bit-plane kernels are the largest functions of the library - every level of block_sum is an inlined cooperative addition - and the long pole of the build.
// Their instantiations live in translation units of their own (csrc/tail_g1.hip, csrc/tail_g2.hip, csrc/tail_g2_planes.hip, csrc/tail_g2_fix.hip), compiled in parallel with the units that LAUNCH them; everywhere
// else they are only declared (a launch references the kernel's host-side handle, an ordinary external symbol).
#define SV_TAIL_FOLD_KERNELS(PREFIX, F)                                                                                                                             \
    PREFIX template __global__ void msm_fold_kernel<F, true>(const xyzz_mem_t<F>*, const uint32_t*, const uint32_t*, xyzz_mem_t<F>*, int, int, int, int, uint32_t*);          \
    PREFIX template __global__ void msm_fold_kernel<F, false>(const xyzz_mem_t<F>*, const uint32_t*, const uint32_t*, xyzz_mem_t<F>*, int, int, int, int, uint32_t*);
#define SV_TAIL_PLANE_KERNELS(PREFIX, F)                                                                                                                            \
    PREFIX template __global__ void msm_bitplane_kernel<F, true>(const xyzz_mem_t<F>*, const uint32_t*, const uint32_t*, xyzz_mem_t<F>*, uint32_t, int, int, int, int, uint32_t*); \
    PREFIX template __global__ void msm_bitplane_kernel<F, false>(const xyzz_mem_t<F>*, const uint32_t*, const uint32_t*, xyzz_mem_t<F>*, uint32_t, int, int, int, int, uint32_t*);
#define SV_TAIL_KERNELS(PREFIX, F) SV_TAIL_FOLD_KERNELS(PREFIX, F) SV_TAIL_PLANE_KERNELS(PREFIX, F)
#define SV_TAIL_FIX_KERNELS(PREFIX, F)                                                                                                                              \
    PREFIX template __global__ void msm_fold_fix_kernel<F>(const xyzz_mem_t<F>*, const uint32_t*, const uint32_t*, xyzz_mem_t<F>*, int, int, const uint32_t*);          \
    PREFIX template __global__ void msm_bitplane_fix_kernel<F, true>(const xyzz_mem_t<F>*, const uint32_t*, const uint32_t*, xyzz_mem_t<F>*, uint32_t, int, int, const uint32_t*); \
    PREFIX template __global__ void msm_bitplane_fix_kernel<F, false>(const xyzz_mem_t<F>*, const uint32_t*, const uint32_t*, xyzz_mem_t<F>*, uint32_t, int, int, const uint32_t*);
#ifndef SV_TU_TAIL
SV_TAIL_KERNELS(extern, fq_t)
SV_TAIL_KERNELS(extern, fqz_t)
SV_TAIL_KERNELS(extern, fq2_t)
SV_TAIL_FIX_KERNELS(extern, fq2_t)
#endif

// ------------------------------------------------------------------------------------------
// Base tables for registered (static) bases: next[i] = 2^shift * prev[i], affine.  Lets one bucket window serve
// `tables` digit rows, which cuts the serial Horner chain and the bucket reduction by `tables`.
// ------------------------------------------------------------------------------------------
// Thread t owns the `run` consecutive points [t * run, ...) (run <= PRE_RUN, 1 for small vectors): `shift` Jacobian doublings each, then ONE field inversion
// for the whole run (Montgomery's trick over the Z coordinates: prefix products forward, back-substitution backward) instead
// of a 570-product Fermat inversion per point - the inversions were 3/4 of the registration time.  scratch: 4 field elements
// per point of the slab being processed (X, Y, Z, prefix product).
static constexpr int PRE_RUN = 16;
template <class F>
__global__ void __launch_bounds__(256) precompute_table_kernel(const aff_mem_t<F>* __restrict__ prev, aff_mem_t<F>* __restrict__ next,
                                                               size_t n, int shift, int run, typename F::mem_t* __restrict__ scratch) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t first = t * (size_t)run;
    if (first >= n) return;
    const size_t cnt = n - first < (size_t)run ? n - first : (size_t)run;
    typename F::mem_t* sx = scratch + 4 * first;
    F prod = F::one();
    for (size_t i = 0; i < cnt; i++) {
        const aff_t<F> p = load_aff<F>(&prev[first + i]);
        jac_t<F> j = {p.x, p.y, p.is_inf() ? F::zero() : F::one()};
        for (int d = 0; d < shift; d++) j = j.dbl();
        if (!j.is_inf()) prod = prod * j.z;  // points at infinity stay out of the product
        j.x.store(&sx[4 * i]);
        j.y.store(&sx[4 * i + 1]);
        j.z.store(&sx[4 * i + 2]);
        prod.store(&sx[4 * i + 3]);
    }
    F inv = prod.inverse();  // of the product of the non-zero Z of the run (one() when there is none)
    for (size_t i = cnt; i-- > 0;) {
        const F z = F::load(&sx[4 * i + 2]);
        aff_t<F> q = aff_t<F>::inf();
        if (!z.is_zero()) {
            const F before = i == 0 ? F::one() : F::load(&sx[4 * i - 1]);  // prefix product of the points before i
            const F zi = inv * before;
            inv = inv * z;
            const F zi2 = zi.sqr();
            q.x = F::load(&sx[4 * i]) * zi2;
            q.y = F::load(&sx[4 * i + 1]) * (zi2 * zi);
        }
        store_aff<F>(&next[first + i], q);
    }
}
static constexpr size_t PRE_SLAB = (size_t)1 << 20;  // points per launch: bounds the scratch at 4 field elements x 2^20

// ------------------------------------------------------------------------------------------
// Projective -> Affine (affine.rs:331-353 `From<Projective> for Affine`; batch form projective.rs:172-219)
// in: Jacobian memory images (144 B), out: Rust G1Affine (104 B).  One Fermat inversion per point.
// ------------------------------------------------------------------------------------------
static __global__ void g1_to_affine_kernel(const uint32_t* in, uint32_t* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* src = in + 36 * i;
    uint32_t* dst = out + 26 * i;
    const fq_t z = fq_t::from_raw_words(src + 24);
    if (z.is_zero()) {  // Affine::zero() = (0, 1, infinity) (affine.rs:55-60)
        for (int k = 0; k < 12; k++) dst[k] = 0;
        fq_t::one().to_raw_words(dst + 12);
        dst[24] = 1;
        dst[25] = 0;
        return;
    }
    const fq_t x = fq_t::from_raw_words(src), y = fq_t::from_raw_words(src + 12);
    const fq_t zi = z.inverse();
    const fq_t zi2 = zi.sqr();
    (x * zi2).to_raw_words(dst);
    (y * (zi2 * zi)).to_raw_words(dst + 12);
    dst[24] = 0;
    dst[25] = 0;
}

// ------------------------------------------------------------------------------------------
// Sum of a few Jacobian points (the per-device partial results of a point-range-split MSM: replaces the host `dadd`
// loop of algorithms/cuda/cuda/snarkvm.cu:290-295).  One workgroup; lane t adds points t, t + 64, ..., then an LDS tree.
// ------------------------------------------------------------------------------------------
static __global__ void __launch_bounds__(64) g1_sum_kernel(const uint32_t* in, size_t n, uint32_t* out) {
    __shared__ g1_xyzz_mem_t part[64];
    g1_xyzz_t acc = g1_xyzz_t::inf();
    for (size_t i = threadIdx.x; i < n; i += 64) {
        const uint32_t* src = in + 36 * i;
        g1_jac_t j = {fq_t::from_raw_words(src), fq_t::from_raw_words(src + 12), fq_t::from_raw_words(src + 24)};
        acc.add(g1_xyzz_t::from_jacobian(j));
    }
    g1_store_xyzz(&part[threadIdx.x], acc);
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            g1_xyzz_t a = g1_load_xyzz(&part[threadIdx.x]);
            a.add(g1_load_xyzz(&part[threadIdx.x + s]));
            g1_store_xyzz(&part[threadIdx.x], a);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const g1_jac_t r = g1_load_xyzz(&part[0]).to_jacobian();
        r.x.to_raw_words(out);
        r.y.to_raw_words(out + 12);
        r.z.to_raw_words(out + 24);
    }
}

// ------------------------------------------------------------------------------------------
// Synthetic base generation (benchmark / test utility): out[i] = (start + i) * G in the Rust layout
// ------------------------------------------------------------------------------------------
static constexpr int GEN_RUN = 32;
static __global__ void __launch_bounds__(256) g1_generate_bases_kernel(g1_aff_mem_t gen, uint64_t start, size_t n, uint8_t* out,
                                                                size_t stride, g1_xyzz_mem_t* scratch_pts,
                                                                fq_mem_t* scratch_prod) {
    const size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t first = t * GEN_RUN;
    if (first >= n) return;
    const size_t cnt = (n - first < (size_t)GEN_RUN) ? (n - first) : (size_t)GEN_RUN;
    const g1_aff_t g = g1_load_aff(&gen);
    // (start + first) * G by double-and-add
    const uint64_t k = start + first;
    g1_xyzz_t cur = g1_xyzz_t::inf();
    for (int bit = 63; bit >= 0; bit--) {
        cur = cur.dbl();
        if ((k >> bit) & 1) cur.add_affine(g);
    }
    // running addition; prefix products of zzz for one shared inversion (Montgomery's trick).
    // Multiples of G below the group order are never infinity, so every zzz is invertible.
    fq_t prod = fq_t::one();
    for (size_t i = 0; i < cnt; i++) {
        g1_store_xyzz(&scratch_pts[first + i], cur);
        prod = prod * cur.zzz;
        prod.store(&scratch_prod[first + i]);
        cur.add_affine(g);
    }
    fq_t inv = prod.inverse();
    for (size_t i = cnt; i-- > 0;) {
        const g1_xyzz_t pt = g1_load_xyzz(&scratch_pts[first + i]);
        const fq_t zzz_inv = (i == 0) ? inv : inv * fq_t::load(&scratch_prod[first + i - 1]);
        inv = inv * pt.zzz;
        const fq_t zz_inv = zzz_inv.sqr() * pt.zz.sqr();  // zz^3 = zzz^2  =>  1/zz = zz^2 / zzz^2
        uint32_t w[26];
        (pt.x * zz_inv).to_mem_mont().pack(w);
        (pt.y * zzz_inv).to_mem_mont().pack(w + 12);
        w[24] = 0;  // infinity = false + padding
        w[25] = 0;
        uint32_t* dst = (uint32_t*)(out + (first + i) * stride);
        for (int q = 0; q < 26; q++) dst[q] = w[q];
    }
}

}  // namespace sv
