// msm_sort.hip.h - bucket sort of the MSM digit matrix by a multi-level, LDS-staged radix partition.
//
// Why: the first sort scattered 4-byte entries with per-bucket cursors straight into HBM.  On MI355X the L2 is
// write-through for such stores, so every entry left the chip as its own 32-byte sector: rocprofv3 WRITE_SIZE showed
// 8.4 GiB written for 1 GiB of entries (profiles/r01_rocprofv3_pmc_hbm_bytes_final.txt).  Here every workgroup first
// groups the entries of its tile in LDS and then writes whole runs, so the stores of a wave instruction are contiguous:
//
//   level 1   tile = TILE digits of one digit row; key = top HB bits of the bucket index (<= 256 bins per window)
//             -> v[] (virtual index | sign<<31, 4 B) and rem[] (the remaining RB low bucket bits, 1 or 2 B),
//                grouped by (window, bin)
//   level k   tile = TILE items of one segment of the previous level; key = the next <= 7 bits of the remainder
//             -> v'[] (+ rem'[] while bits remain), grouped by (segment, key); seg_start'[] = first position of each group
//   Windows up to c = 16 take two levels (8 + 7 bits), wider windows (c <= 23, registered bases with one bucket window)
//   three (8 + 7 + 7).  After the last level the entries are bucket-major: bucket k owns sorted[boff[k], boff[k+1]).
//
// Each level is histogram -> exclusive scan -> staged scatter; all positions are deterministic functions of the
// histograms except the order inside a (tile, bin) run (LDS cursor order), which does not change any bucket's content.
#pragma once
#include "ffl.hip.h"
#include "ffl2.hip.h"
#include "ffl2p.hip.h"
#include "msm.hip.h"

namespace sv {

static constexpr int SORT_TILE = 8192;     // items per workgroup tile
static constexpr int SORT_THREADS = 256;   // 32 items per thread

// Fused multi-instance MSM (runtime.hip.h::msm_run, `multi`): K independent MSMs over slices of ONE registered base vector
// (the commitments of a prover round, sonic_pc/mod.rs:186-245) share one launch sequence.  The instances are laid side by side
// in a padded concatenation (every instance starts on a multiple of SORT_TILE positions; the padding holds zero digits, which
// touch no bucket), the instance id is the top key of the radix partition (bucket window = instance: K x 2^(c-1) buckets), and
// an entry's virtual index is the SLOT of its base in the handle's table array (table * h.n + base index), so everything after
// level 1 - the further sort levels, the accumulate grid, the fold and the bit planes - runs unchanged on K "windows".
struct msm_inst_t {
    const uint4* scalars;  // device pointer: n scalars of 32 B
    uint32_t n;            // scalars of the instance
    uint32_t n0;           // scalars [0, n0) pair with bases off0 + i, the others with off1 + (i - n0) (KZG10's two base ranges)
    uint32_t off0, off1;   // base indices relative to the registered vector
    uint32_t pstart;       // first position of the instance in the padded concatenation (a multiple of SORT_TILE)
    uint32_t ptiles;       // padded length / SORT_TILE
};
struct msm_radix_params_t {
    size_t n;              // scalars (multi: padded positions of all instances)
    int c, W, J;           // window bits, bucket windows, base tables (digit row j*W + w feeds window w)
    int HB, LB;            // bucket index = (bin << LB) | rem, bin < 2^HB (level-1 key), rem < 2^LB (<= 7: one more level, <= 14: two)
    uint32_t nb;           // 2^(c-1)
    uint32_t tiles_per_row;  // ceil(n / SORT_TILE)
    uint32_t TPW;          // level-1 tiles per window = J * tiles_per_row
    const msm_inst_t* inst = nullptr;  // multi: K + 1 entries (the last one is a sentinel with pstart = n); window = instance, W == 1
    uint32_t ninst = 0;
    // An entry's virtual index is the SLOT of its base relative to one base pointer B (the accumulate kernel reads B[v], no
    // arithmetic): digit row of table j, scalar i  ->  j * vstride + (i < vn0 ? vr0 + i : vr1 + (i - vn0)).  (vn0, vr0, vr1) come
    // from the instance table in a multi-instance run.
    uint32_t vstride = 0, vn0 = 0, vr0 = 0, vr1 = 0;
    uint32_t xcd = 0;      // scatter kernels: tile = xcd_tile(blockIdx.x, tiles) instead of blockIdx.x
};
// Workgroup -> tile map of the scatter kernels (round 4).  The dispatcher places workgroup b on XCD b % 8 (observed, MI355X_MICROARCH.md
// "Workgroup dispatch"; a speed assumption only), so with tile = b the runs that tiles t and t + 1 write side by side into the same
// 128-byte lines (16 entries = 64 B of v1 per (row, bin) and tile at level 1) dirty those lines in two different L2s, neither of which
// ever sees a whole line: profiles/r04_pmc_traffic.json counted 1.90 GB written for 1.21 GB of entries.  Here XCD x walks the
// contiguous tile range [x T / 8, (x + 1) T / 8): neighbouring runs meet in one L2 and leave it as whole lines.  Bijective for any T.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t b, uint32_t T) {
    const uint32_t q = T >> 3, r = T & 7u, x = b & 7u;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
}
// One level-1 tile: digits [lo, hi) of one digit row feeding bucket window w; its counters live at cbase + bin * TPW + tw.
struct l1_tile_t {
    uint32_t w, tw, TPW, j;
    size_t cbase;     // first counter of window w
    size_t row;       // first digit of the tile's row (index into the digit matrix)
    size_t lo, hi;
    uint32_t inst_idx;  // multi: the instance
};
__device__ __forceinline__ l1_tile_t l1_decode_tile(const msm_radix_params_t& p, uint32_t g) {
    l1_tile_t t;
    const uint32_t B1 = 1u << p.HB;
    if (!p.inst) {
        t.w = g / p.TPW;
        t.tw = g - t.w * p.TPW;
        t.TPW = p.TPW;
        t.j = t.tw / p.tiles_per_row;
        const uint32_t tt = t.tw - t.j * p.tiles_per_row;
        t.cbase = (size_t)t.w * B1 * p.TPW;
        t.row = (size_t)(t.j * p.W + t.w) * p.n;
        t.lo = (size_t)tt * SORT_TILE;
        t.hi = (t.lo + SORT_TILE < p.n) ? t.lo + SORT_TILE : p.n;
        t.inst_idx = 0;
        return t;
    }
    // multi: the tiles of instance k are [J * pstart_k / TILE, J * pstart_(k+1) / TILE), table-major inside the instance
    const uint32_t J = (uint32_t)p.J;
    uint32_t lo = 0, hi = p.ninst;  // invariant: J * pstart[lo] / TILE <= g < J * pstart[hi] / TILE
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (J * (p.inst[mid].pstart / SORT_TILE) <= g) lo = mid; else hi = mid;
    }
    const msm_inst_t in = p.inst[lo];
    const uint32_t tw0 = J * (in.pstart / SORT_TILE);
    t.w = lo;
    t.inst_idx = lo;
    t.tw = g - tw0;
    t.TPW = J * in.ptiles;
    t.j = t.tw / in.ptiles;
    const uint32_t tt = t.tw - t.j * in.ptiles;
    t.cbase = (size_t)B1 * tw0;
    t.row = (size_t)t.j * p.n;
    t.lo = (size_t)in.pstart + (size_t)tt * SORT_TILE;
    t.hi = t.lo + SORT_TILE;  // padded: always a whole tile
    return t;
}

// decode one digit: returns false for digit zero; else bucket index b (0-based) and the sign
__device__ __forceinline__ bool digit_bucket(uint32_t u, int half, uint32_t& b, uint32_t& neg) {
    const int dv = (int)u - half;
    if (dv == 0) return false;
    neg = dv < 0 ? 0x80000000u : 0u;
    b = (uint32_t)((dv < 0 ? -dv : dv) - 1);
    return true;
}

// Visit the digits (u16 or u32) of scalars [lo, hi) of one row; 16-byte loads when the row is 16-byte aligned.
template <class DT, class Fn>
__device__ __forceinline__ void for_each_digit_t(const DT* __restrict__ d, size_t n, size_t lo, size_t hi, Fn fn) {
    constexpr int DPV = 16 / (int)sizeof(DT);
    if ((n & (DPV - 1)) == 0 && (lo & (DPV - 1)) == 0 && ((hi - lo) & (DPV - 1)) == 0) {
        const uint4* d4 = (const uint4*)(d + lo);
        const size_t nvec = (hi - lo) / DPV;
        for (size_t v = threadIdx.x; v < nvec; v += blockDim.x) {
            const uint4 q = d4[v];
            const uint32_t wds[4] = {q.x, q.y, q.z, q.w};
            const size_t i = lo + v * DPV;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (sizeof(DT) == 2) {
                    fn(wds[k] & 0xffffu, i + 2 * k);
                    fn(wds[k] >> 16, i + 2 * k + 1);
                } else {
                    fn(wds[k], i + k);
                }
            }
        }
    } else {
        for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) fn((uint32_t)d[i], i);
    }
}

// ---- level 1 histogram: counts1[(w * B1 + bin) * TPW + tw]
template <class DT>
__global__ void __launch_bounds__(SORT_THREADS) radix_hist1_kernel(const DT* __restrict__ digits, uint32_t* __restrict__ counts1,
                                                                   msm_radix_params_t p) {
    __shared__ uint32_t hist[256];
    const uint32_t B1 = 1u << p.HB;
    const l1_tile_t tl = l1_decode_tile(p, blockIdx.x);
    for (uint32_t i = threadIdx.x; i < B1; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const int half = 1 << (p.c - 1);
    for_each_digit_t<DT>(digits + tl.row, p.n, tl.lo, tl.hi, [&](uint32_t u, size_t) {
        uint32_t b, neg;
        if (digit_bucket(u, half, b, neg)) atomicAdd(&hist[b >> p.LB], 1u);
    });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < B1; i += blockDim.x) counts1[tl.cbase + (size_t)i * tl.TPW + tl.tw] = hist[i];
}

// ---- level 1 scatter: stage the tile in LDS grouped by bin, then write whole runs
// Exclusive scan over the SORT_THREADS values held one per thread (wave-level shuffles + 4 wave totals in LDS).
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_tot) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += t;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wv; k++) base += wave_tot[k];
    return base + inc - v;
}
template <class DT, class RT>
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter1_kernel(const DT* __restrict__ digits,
                                                                      const uint32_t* __restrict__ counts1,
                                                                      const uint32_t* __restrict__ off1, uint32_t* __restrict__ v1,
                                                                      RT* __restrict__ l1, msm_radix_params_t p) {
    __shared__ uint32_t lcount[256], lstart[256], cursor[256], gbase[256], wave_tot[4];
    __shared__ uint32_t sv_[SORT_TILE];
    __shared__ RT sl_[SORT_TILE];
    __shared__ uint8_t sbin_[SORT_TILE];
    const uint32_t B1 = 1u << p.HB;
    const l1_tile_t tl = l1_decode_tile(p, p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x);
    // this tile's histogram was computed by radix_hist1_kernel; its global run starts are off1[...]
    const size_t lo = tl.lo, hi = tl.hi;
    const int half = 1 << (p.c - 1);
    const DT* row = digits + tl.row;
    // full aligned tile: issue this thread's 16-byte digit loads before anything else
    constexpr int DPV = 16 / (int)sizeof(DT);  // digits per 16-byte load
    const bool vec = (p.n & (DPV - 1)) == 0 && hi - lo == SORT_TILE;
    constexpr int NV = SORT_TILE / DPV / SORT_THREADS;
    uint4 dq[NV];
    if (vec) {
        const uint4* d4 = (const uint4*)(row + lo);
#pragma unroll
        for (int k = 0; k < NV; k++) dq[k] = d4[threadIdx.x + k * SORT_THREADS];
    }
    {
        const uint32_t i = threadIdx.x;  // SORT_THREADS == 256 >= B1: one bin per thread
        const uint32_t cnt = (i < B1) ? counts1[tl.cbase + (size_t)i * tl.TPW + tl.tw] : 0u;
        gbase[i] = (i < B1) ? off1[tl.cbase + (size_t)i * tl.TPW + tl.tw] : 0u;
        lcount[i] = cnt;
        const uint32_t start = block_excl_scan(cnt, wave_tot);
        lstart[i] = start;
        cursor[i] = start;
    }
    __syncthreads();
    // virtual index of digit position i = the slot of its base (msm_radix_params_t)
    uint32_t pstart = 0, vn0 = p.vn0, vr0 = p.vr0, vr1 = p.vr1;
    if (p.inst) {
        const msm_inst_t in = p.inst[tl.inst_idx];
        pstart = in.pstart, vn0 = in.n0, vr0 = in.off0, vr1 = in.off1;
    }
    const uint32_t voff = tl.j * p.vstride;
    const uint32_t lmask = (1u << p.LB) - 1;
    auto place = [&](uint32_t u, size_t i) {
        uint32_t b, neg;
        if (digit_bucket(u, half, b, neg)) {
            const uint32_t bin = b >> p.LB;
            const uint32_t pos = atomicAdd(&cursor[bin], 1u);
            const uint32_t idx = (uint32_t)i - pstart;
            const uint32_t vi = idx < vn0 ? vr0 + idx : vr1 + (idx - vn0);
            sv_[pos] = (voff + vi) | neg;
            sl_[pos] = (RT)(b & lmask);
            sbin_[pos] = (uint8_t)bin;
        }
    };
    if (vec) {
#pragma unroll
        for (int k = 0; k < NV; k++) {
            const uint32_t wds[4] = {dq[k].x, dq[k].y, dq[k].z, dq[k].w};
            const size_t i = lo + (size_t)(threadIdx.x + k * SORT_THREADS) * DPV;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                if (sizeof(DT) == 2) {
                    place(wds[m] & 0xffffu, i + 2 * m);
                    place(wds[m] >> 16, i + 2 * m + 1);
                } else {
                    place(wds[m], i + m);
                }
            }
        }
    } else {
        for_each_digit_t<DT>(row, p.n, lo, hi, place);
    }
    __syncthreads();
    const uint32_t total = lstart[B1 - 1] + lcount[B1 - 1];
    for (uint32_t pos = threadIdx.x; pos < total; pos += blockDim.x) {
        const uint32_t bin = sbin_[pos];
        const size_t dst = (size_t)gbase[bin] + (pos - lstart[bin]);
        v1[dst] = sv_[pos];
        l1[dst] = sl_[pos];
    }
}

// ---- level 1 fused with the scalar read (wide windows: the 2^24 regime) ---------------------------------------------------
// The stand-alone digit kernel reads 32 B per scalar and writes the J x n digit matrix (48 B per scalar at J = 12), which the
// level-1 histogram and scatter then read once each: 176 B of HBM traffic per scalar before a single entry is in place, and
// a "scalar-read phase" that spends most of its time writing.  Here the digits never exist in memory: the histogram kernel IS
// the scalar-read phase (a read-only stream of the scalars, digits recoded in registers, counts in LDS, one contiguous row of
// counters written per workgroup), and the scatter kernel reads the scalars a second time, keeps the recoded words of its
// scalars in registers and places FUSED_G digit rows at a time through LDS (the (v, remainder) staging of a tile would not fit
// for all rows at once).  64 B read + 72 B written per scalar instead of 248 B.
//   tile t            = FUSED_TILE consecutive scalars x all digit rows; key = row * B1 + bin
//   cnt[t][key]       tile-major, so a workgroup reads / writes one contiguous row of KEYS counters
//   offsets           entries are grouped by q = (window, bin); inside a group by table j, then by tile.  The exclusive scan in
//                     that order runs hierarchically: per-chunk column sums (FUSED_CHUNK tiles) laid out [(q, j)][chunk], one
//                     flat exclusive scan over them, then a per-chunk pass that turns them into off[t][key].
static constexpr int FUSED_TILE = 2048;    // scalars per workgroup
static constexpr int FUSED_THREADS = 512;  // 4 scalars per thread
static constexpr int FUSED_G = 4;          // digit rows staged per round of the scatter
static constexpr int FUSED_SPT = FUSED_TILE / FUSED_THREADS;
static constexpr int FUSED_CHUNK = 64;     // tiles per chunk of the offset scan

// 32-byte scalar (two 16-byte halves) -> recoded words s' = scalar (+ Fr::to_bigint) + bias; digit row r = (s' >> c r) & (2^c - 1)
__device__ __forceinline__ void recode_scalar(const uint4& lo, const uint4& hi, const msm_digit_params_t& dp, uint32_t* s /* 11 words */) {
    s[0] = lo.x, s[1] = lo.y, s[2] = lo.z, s[3] = lo.w, s[4] = hi.x, s[5] = hi.y, s[6] = hi.z, s[7] = hi.w, s[8] = 0, s[9] = 0, s[10] = 0;
    if (dp.montgomery) {
        fr_t c32 = fr_t::zero();
        c32.v[0] = 32;  // a * 2^256 read as internal a * 2^-5 (ff.hip.h): one Montgomery product by the integer 2^5 gives a
        (fr_t::unpack(s) * c32).pack(s);
    }
    uint64_t carry = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        carry += (uint64_t)s[k] + dp.bias[k];
        s[k] = (uint32_t)carry;
        carry >>= 32;
    }
}
// C and r are compile-time (the callers unroll over r): the word index and the shift fold to constants, so the recoded words
// stay in registers (a run-time index would push them to scratch memory)
template <int C>
__device__ __forceinline__ uint32_t recoded_digit(const uint32_t* s, int r) {
    const int bit = C * r, wi = bit >> 5, sh = bit & 31;
    // v_alignbit_b32: low 32 bits of (s[wi + 1] : s[wi]) >> sh - kept as an intrinsic so that the compiler does not turn the
    // 64-bit shift into an unaligned load from a stack copy of s[]
    const uint32_t x = sh ? __builtin_amdgcn_alignbit(s[wi + 1], s[wi], (uint32_t)sh) : s[wi];
    return x & ((1u << C) - 1);
}
static constexpr int FUSED_MAX_ROWS = 16;  // 288 digit bits / 17-bit windows
// Coalesced read of 2 * FUSED_THREADS consecutive scalars (half a tile) into `stage` (4 * FUSED_THREADS uint4 = 32 KB): every
// thread issues its four 16-byte loads before anything waits on them.
__device__ __forceinline__ void fused_stage_half(const uint4* __restrict__ scalars, size_t first, size_t n, uint4* stage) {
    const size_t cnt = first >= n ? 0 : (n - first < (size_t)(2 * FUSED_THREADS) ? n - first : (size_t)(2 * FUSED_THREADS));
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint32_t idx = threadIdx.x + k * FUSED_THREADS;
        v[k] = ((size_t)(idx >> 1) < cnt) ? scalars[first * 2 + idx] : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) stage[threadIdx.x + k * FUSED_THREADS] = v[k];
}
// dynamic LDS: rows * B1 counters.  The scalar read needs no staging here: every thread issues the eight 16-byte loads of its
// four scalars up front (lane stride 32 B: the two loads of a scalar use the two halves of the same cache lines, so every
// line is fetched from HBM once) and nothing but the LDS counters sits between the loads and the end of the kernel.
template <int C>
__global__ void __launch_bounds__(FUSED_THREADS) radix_hist1_fused_kernel(const uint4* __restrict__ scalars, uint32_t* __restrict__ cnt,
                                                                   msm_radix_params_t p, msm_digit_params_t dp) {
    extern __shared__ uint32_t fused_hist[];
    const uint32_t B1 = 1u << p.HB;
    const uint32_t rows = (uint32_t)dp.W;  // digit rows per scalar (= W * J)
    const uint32_t keys = rows * B1;
    uint32_t* hist = fused_hist;
    const uint32_t t = blockIdx.x;
    uint4 lo[FUSED_SPT], hi[FUSED_SPT];
#pragma unroll
    for (int q = 0; q < FUSED_SPT; q++) {
        const size_t i = (size_t)t * FUSED_TILE + (size_t)q * FUSED_THREADS + threadIdx.x;
        if (i < p.n) {
            lo[q] = scalars[2 * i];
            hi[q] = scalars[2 * i + 1];
        } else {
            lo[q] = hi[q] = make_uint4(0, 0, 0, 0);
        }
    }
    for (uint32_t i = threadIdx.x; i < keys; i += FUSED_THREADS) hist[i] = 0;
    __syncthreads();
    const int half = 1 << (p.c - 1);
#pragma unroll
    for (int q = 0; q < FUSED_SPT; q++) {
        if ((size_t)t * FUSED_TILE + (size_t)q * FUSED_THREADS + threadIdx.x >= p.n) continue;
        uint32_t s[11];
        recode_scalar(lo[q], hi[q], dp, s);
#pragma unroll
        for (int r = 0; r < FUSED_MAX_ROWS; r++) {
            uint32_t b, neg;
            if (C * r < MSM_BIAS_BITS && (uint32_t)r < rows && digit_bucket(recoded_digit<C>(s, r), half, b, neg)) atomicAdd(&hist[r * B1 + (b >> p.LB)], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < keys; i += FUSED_THREADS) cnt[(size_t)t * keys + i] = hist[i];
}
// The same counts from a WIDER workgroup with REPLICATED histograms (round 4; tuning hist=2, the default).  tools/exp/histbench.hip
// took the one-tile kernel above apart on 2^24 scalars: reading alone takes 85 - 88 us (6.1 - 6.3 TB/s), reading + recoding + digit
// extraction 93 us, the full kernel 124 - 127 us - the LDS atomics are what is left, and they serialise when lanes of a wave hit
// the same counter (64 lanes into 128 bins of a row).  Four private copies of the histogram (copy = lane & 3, interleaved so that
// different copies never share a bank) cut those collisions fourfold; 1 024 threads x 2 scalars instead of 512 x 4 keep twice the
// waves per tile in flight with half the registers each.  Measured: 102.8 us = 5.2 TB/s = 0.65 of the 8 TB/s spec (one copy:
// 126 us, two: 105, eight: 121; 512 x 4 with four copies: 115; a streaming variant that kept the next tile's loads in flight while
// counting - fewer, fatter workgroups - was slower than the round-3 kernel: 153 us).  Same output: one row of counters per tile.
static constexpr int HISTW_THREADS = 1024;
static constexpr int HISTW_SPT = FUSED_TILE / HISTW_THREADS;  // 2
static constexpr int HISTW_COPIES = 4;
template <int C>
__global__ void __launch_bounds__(HISTW_THREADS) radix_hist1_wide_kernel(const uint4* __restrict__ scalars, uint32_t* __restrict__ cnt, msm_radix_params_t p,
                                                                  msm_digit_params_t dp) {
    extern __shared__ uint32_t fused_hist[];  // keys * HISTW_COPIES counters, copy-interleaved
    const uint32_t B1 = 1u << p.HB;
    const uint32_t rows = (uint32_t)dp.W;
    const uint32_t keys = rows * B1;
    const uint32_t t = blockIdx.x;
    uint4 lo[HISTW_SPT], hi[HISTW_SPT];
#pragma unroll
    for (int q = 0; q < HISTW_SPT; q++) {
        const size_t i = (size_t)t * FUSED_TILE + (size_t)q * HISTW_THREADS + threadIdx.x;
        if (i < p.n) {
            lo[q] = scalars[2 * i];
            hi[q] = scalars[2 * i + 1];
        } else {
            lo[q] = hi[q] = make_uint4(0, 0, 0, 0);
        }
    }
    for (uint32_t i = threadIdx.x; i < keys * HISTW_COPIES; i += HISTW_THREADS) fused_hist[i] = 0;
    __syncthreads();
    const int half = 1 << (p.c - 1);
    const uint32_t copy = threadIdx.x & (HISTW_COPIES - 1);
#pragma unroll
    for (int q = 0; q < HISTW_SPT; q++) {
        if ((size_t)t * FUSED_TILE + (size_t)q * HISTW_THREADS + threadIdx.x >= p.n) continue;
        uint32_t s[11];
        recode_scalar(lo[q], hi[q], dp, s);
#pragma unroll
        for (int r = 0; r < FUSED_MAX_ROWS; r++) {
            uint32_t b, neg;
            if (C * r < MSM_BIAS_BITS && (uint32_t)r < rows && digit_bucket(recoded_digit<C>(s, r), half, b, neg))
                atomicAdd(&fused_hist[(r * B1 + (b >> p.LB)) * HISTW_COPIES + copy], 1u);
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < keys; i += HISTW_THREADS) {
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < HISTW_COPIES; k++) v += fused_hist[i * HISTW_COPIES + k];
        cnt[(size_t)t * keys + i] = v;
    }
}
// key = (j * W + w) * B1 + bin  ->  position of (q = w * B1 + bin, j) in the scan order
__device__ __forceinline__ uint32_t fused_group_index(uint32_t key, uint32_t B1, uint32_t W, uint32_t J) {
    const uint32_t r = key / B1, bin = key - r * B1;
    const uint32_t w = r % W, j = r / W;
    return (w * B1 + bin) * J + j;
}
// chunk column sums: csum[(q * J + j) * nchunks + chunk] = sum over the chunk's tiles of cnt[t][key]
static __global__ void __launch_bounds__(FUSED_THREADS) fused_chunk_sums_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ csum, uint32_t ntiles,
                                                                         uint32_t nchunks, uint32_t keys, uint32_t B1, uint32_t W, uint32_t J) {
    // grid (chunks, key blocks): 128 chunks alone would leave half of the CUs without a workgroup (78 us for 50 MB of counters)
    const uint32_t ch = blockIdx.x;
    const uint32_t t0 = ch * FUSED_CHUNK, t1 = (t0 + FUSED_CHUNK < ntiles) ? t0 + FUSED_CHUNK : ntiles;
    for (uint32_t key = blockIdx.y * FUSED_THREADS + threadIdx.x; key < keys; key += gridDim.y * FUSED_THREADS) {
        uint32_t s = 0;
#pragma unroll 8
        for (uint32_t t = t0; t < t1; t++) s += cnt[(size_t)t * keys + key];
        csum[(size_t)fused_group_index(key, B1, W, J) * nchunks + ch] = s;
    }
}
// off[t][key] = scanned chunk offset + prefix of the chunk's earlier tiles; chunk 0 also publishes the bin starts of level 2
static __global__ void __launch_bounds__(FUSED_THREADS) fused_tile_offsets_kernel(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ choff,
                                                                           const uint32_t* __restrict__ csum, uint32_t* __restrict__ off,
                                                                           uint32_t* __restrict__ binstart, uint32_t ntiles, uint32_t nchunks, uint32_t keys,
                                                                           uint32_t B1, uint32_t W, uint32_t J) {
    const uint32_t ch = blockIdx.x;
    const uint32_t t0 = ch * FUSED_CHUNK, t1 = (t0 + FUSED_CHUNK < ntiles) ? t0 + FUSED_CHUNK : ntiles;
    for (uint32_t key = blockIdx.y * FUSED_THREADS + threadIdx.x; key < keys; key += gridDim.y * FUSED_THREADS) {
        const uint32_t g = fused_group_index(key, B1, W, J);
        uint32_t run = choff[(size_t)g * nchunks + ch];
        if (ch == 0 && g % J == 0) binstart[g / J] = run;
#pragma unroll 8
        for (uint32_t t = t0; t < t1; t++) {
            const size_t idx = (size_t)t * keys + key;
            const uint32_t c = cnt[idx];
            off[idx] = run;
            run += c;
        }
    }
    if (ch == 0 && blockIdx.y == 0 && threadIdx.x == 0) {  // one past the last group: the total number of entries
        const size_t last = (size_t)W * B1 * J * nchunks - 1;
        binstart[W * B1] = choff[last] + csum[last];
    }
}
// Exclusive scan over FUSED_THREADS values held one per thread (8 waves)
__device__ __forceinline__ uint32_t block512_excl_scan(uint32_t v, uint32_t* wave_tot) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(inc, d);
        if (lane >= (uint32_t)d) inc += t;
    }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t k = 0; k < wv; k++) base += wave_tot[k];
    return base + inc - v;
}
// requires FUSED_G * B1 <= FUSED_THREADS (B1 <= 128: the wide path keys level 1 on 7 bits); uint16_t remainders (LB = 14)
template <int C>
__global__ void __launch_bounds__(FUSED_THREADS) radix_scatter1_fused_kernel(const uint4* __restrict__ scalars, const uint32_t* __restrict__ cnt,
                                                                      const uint32_t* __restrict__ off, uint32_t* __restrict__ v1,
                                                                      uint16_t* __restrict__ l1, msm_radix_params_t p, msm_digit_params_t dp) {
    __shared__ uint32_t sv_[FUSED_G * FUSED_TILE];  // also the staging area of the scalar read (4 * FUSED_THREADS uint4 = 32 KB)
    __shared__ uint16_t sl_[FUSED_G * FUSED_TILE];
    __shared__ uint16_t skey_[FUSED_G * FUSED_TILE];
    __shared__ uint32_t lcount[FUSED_THREADS], lstart[FUSED_THREADS], cursor[FUSED_THREADS], gbase[FUSED_THREADS], wave_tot[8];
    const uint32_t B1 = 1u << p.HB;
    const uint32_t rows = (uint32_t)dp.W;
    const uint32_t keys = rows * B1;
    const uint32_t t = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : blockIdx.x;
    const int half = 1 << (p.c - 1);
    const uint32_t lmask = (1u << p.LB) - 1;
    // ---- read the tile's scalars (coalesced through LDS) and keep their recoded words in registers
    uint32_t sw[FUSED_SPT][10];
    uint4* stage = (uint4*)sv_;
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
        const size_t first = (size_t)t * FUSED_TILE + (size_t)hf * (2 * FUSED_THREADS);
        __syncthreads();
        fused_stage_half(scalars, first, p.n, stage);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 2; sub++) {
            const uint32_t li = sub * FUSED_THREADS + threadIdx.x;
            uint32_t s[11];
            recode_scalar(stage[2 * li], stage[2 * li + 1], dp, s);  // slots past n hold zeros: recoded, never placed
#pragma unroll
            for (int k = 0; k < 10; k++) sw[hf * 2 + sub][k] = s[k];
        }
    }
    // ---- FUSED_G digit rows per round: group in LDS by (row, bin), write whole runs
#pragma unroll
    for (int round = 0; round < FUSED_MAX_ROWS / FUSED_G; round++) {
        const uint32_t r0 = (uint32_t)round * FUSED_G;
        if (r0 >= rows) break;
        __syncthreads();  // staging area / previous round's runs consumed
        {
            const uint32_t lk = threadIdx.x;  // one (row in round, bin) per thread
            const uint32_t rl = lk / B1, bin = lk - rl * B1, r = r0 + rl;
            uint32_t c = 0, gb = 0;
            if (rl < FUSED_G && r < rows) {
                const size_t idx = (size_t)t * keys + (size_t)r * B1 + bin;
                c = cnt[idx];
                gb = off[idx];
            }
            lcount[lk] = c;
            gbase[lk] = gb;
            const uint32_t start = block512_excl_scan(c, wave_tot);
            lstart[lk] = start;
            cursor[lk] = start;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < FUSED_SPT; q++) {
            const size_t i64 = (size_t)t * FUSED_TILE + (size_t)(q >> 1) * (2 * FUSED_THREADS) + (size_t)(q & 1) * FUSED_THREADS + threadIdx.x;
            if (i64 >= p.n) continue;
            const uint32_t i = (uint32_t)i64;
#pragma unroll
            for (int rl = 0; rl < FUSED_G; rl++) {
                const int r = round * FUSED_G + rl;  // compile-time after unrolling
                uint32_t b, neg;
                if (C * r < MSM_BIAS_BITS && (uint32_t)r < rows && digit_bucket(recoded_digit<C>(sw[q], r), half, b, neg)) {
                    const uint32_t lk = (uint32_t)rl * B1 + (b >> p.LB);
                    const uint32_t pos = atomicAdd(&cursor[lk], 1u);
                    const uint32_t j = (uint32_t)r / (uint32_t)p.W;
                    sv_[pos] = (j * p.vstride + (i < p.vn0 ? p.vr0 + i : p.vr1 + (i - p.vn0))) | neg;
                    sl_[pos] = (uint16_t)(b & lmask);
                    skey_[pos] = (uint16_t)lk;
                }
            }
        }
        __syncthreads();
        const uint32_t last = FUSED_G * B1 - 1;
        const uint32_t total = lstart[last] + lcount[last];
#pragma unroll 4
        for (uint32_t pos = threadIdx.x; pos < total; pos += FUSED_THREADS) {
            const uint32_t lk = skey_[pos];
            const size_t dst = (size_t)gbase[lk] + (pos - lstart[lk]);
            v1[dst] = sv_[pos];
            l1[dst] = sl_[pos];
        }
    }
}

// ---- level 2 tiling: bin q = w * B1 + bin covers [binstart[q], binstart[q+1]) of v1/l1 and gets ceil(size / TILE) tiles
static __global__ void radix_bin_layout_kernel(const uint32_t* __restrict__ off1, const uint32_t* __restrict__ counts1, size_t ncounts1,
                                        uint32_t* __restrict__ binstart, uint32_t nbins, uint32_t TPW) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nbins) return;
    binstart[q] = (q < nbins) ? off1[(size_t)q * TPW] : off1[ncounts1 - 1] + counts1[ncounts1 - 1];
}
// multi: bin q = k * B1 + bin of instance k starts at off1[B1 * J * pstart_k / TILE + bin * J * ptiles_k]
static __global__ void radix_bin_layout_multi_kernel(const uint32_t* __restrict__ off1, const uint32_t* __restrict__ counts1, size_t ncounts1,
                                              uint32_t* __restrict__ binstart, uint32_t nbins, msm_radix_params_t p) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nbins) return;
    if (q == nbins) {
        binstart[q] = off1[ncounts1 - 1] + counts1[ncounts1 - 1];
        return;
    }
    const uint32_t B1 = 1u << p.HB, k = q >> p.HB, bin = q & (B1 - 1);
    const msm_inst_t in = p.inst[k];
    const uint32_t J = (uint32_t)p.J;
    binstart[q] = off1[(size_t)B1 * J * (in.pstart / SORT_TILE) + (size_t)bin * J * in.ptiles];
}
// multi: the scalar-read phase of a fused batch.  One thread per padded position: recode the scalar into its J digit rows
// (u16, row-major over the padded positions); padding positions get the zero digit.  A block of 256 positions lies inside one
// instance (instances start on multiples of SORT_TILE), so the instance search is block-uniform.
static __global__ void __launch_bounds__(256) msm_digits_multi_kernel(const msm_inst_t* __restrict__ inst, uint32_t ninst,
                                                               uint16_t* __restrict__ digits, msm_digit_params_t p) {
    const uint32_t g = blockIdx.x * 256u + threadIdx.x;  // p.n = padded positions, a multiple of 256
    const uint32_t g0 = blockIdx.x * 256u;
    uint32_t lo = 0, hi = ninst;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (inst[mid].pstart <= g0) lo = mid; else hi = mid;
    }
    const msm_inst_t in = inst[lo];
    const uint32_t idx = g - in.pstart;
    const uint32_t half = 1u << (p.c - 1);
    if (idx >= in.n) {
        for (int w = 0; w < p.W; w++) digits[(size_t)w * p.n + g] = (uint16_t)half;
        return;
    }
    const uint4 q0 = in.scalars[2 * (size_t)idx], q1 = in.scalars[2 * (size_t)idx + 1];
    uint32_t s[11] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, 0u, 0u, 0u};
    if (p.montgomery) {
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
    for (int w = 0; w < p.W; w++) {
        const int bit = p.c * w, wi = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)s[wi] | ((uint64_t)s[wi + 1] << 32);
        digits[(size_t)w * p.n + g] = (uint16_t)((uint32_t)(two >> sh) & mask);
    }
}
static __global__ void radix_bin_tiles_kernel(const uint32_t* __restrict__ binstart, uint32_t* __restrict__ ntiles, uint32_t nbins) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nbins) return;
    ntiles[q] = (q == nbins) ? 0u : (binstart[q + 1] - binstart[q] + SORT_TILE - 1) / SORT_TILE;
}

// ---- level k histogram: counts2[tile2 * B2 + key], key = (rem >> shift) & (B2 - 1) (B2 = 2^LB keys; `shift` bits stay for later levels)
template <class RIN>
__global__ void __launch_bounds__(SORT_THREADS) radix_hist2_kernel(const RIN* __restrict__ l1, const uint32_t* __restrict__ binstart,
                                                                   const uint32_t* __restrict__ tile2_start, uint32_t* __restrict__ counts2,
                                                                   uint32_t nbins, int LB, int shift) {
    // four private copies of the histogram (copy = lane & 3, interleaved): 64 lanes into <= 128 keys serialise on equal keys, as in the
    // level-1 kernel (radix_hist1_wide_kernel); all of a thread's loads are issued before its first atomic
    __shared__ uint32_t hist[128 * 4];
    const uint32_t t2 = blockIdx.x;
    if (t2 >= tile2_start[nbins]) return;
    const uint32_t B2 = 1u << LB;
    const uint32_t q = find_bucket(tile2_start, nbins, t2);
    const uint32_t lt = t2 - tile2_start[q];
    for (uint32_t i = threadIdx.x; i < 4 * B2; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t lo = binstart[q] + lt * SORT_TILE;
    uint32_t hi = lo + SORT_TILE;
    if (hi > binstart[q + 1]) hi = binstart[q + 1];
    constexpr int PER = SORT_TILE / SORT_THREADS;
    uint32_t ll[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint32_t i = lo + threadIdx.x + (uint32_t)k * SORT_THREADS;
        ll[k] = i < hi ? (uint32_t)l1[i] : 0xffffffffu;
    }
    const uint32_t copy = threadIdx.x & 3u;
#pragma unroll
    for (int k = 0; k < PER; k++)
        if (ll[k] != 0xffffffffu) atomicAdd(&hist[(((ll[k] >> shift) & (B2 - 1)) << 2) | copy], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < B2; i += blockDim.x) counts2[(size_t)t2 * B2 + i] = hist[4 * i] + hist[4 * i + 1] + hist[4 * i + 2] + hist[4 * i + 3];
}
// ---- per bucket k = q * B2 + low: exclusive prefix of its counts over the tiles of bin q + bucket size
static __global__ void radix_colscan2_kernel(const uint32_t* __restrict__ counts2, uint32_t* __restrict__ off2,
                                      const uint32_t* __restrict__ tile2_start, uint32_t* __restrict__ bsize, uint32_t nbins, int LB,
                                      uint32_t* __restrict__ max_size) {
    __shared__ uint32_t blk_max;
    if (threadIdx.x == 0) blk_max = 0;
    __syncthreads();
    const uint32_t B2 = 1u << LB;
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nbt = nbins * B2;
    uint32_t run = 0;
    if (k < nbt) {
        const uint32_t q = k >> LB, low = k & (B2 - 1);
        const uint32_t t0 = tile2_start[q], t1 = tile2_start[q + 1];
        for (uint32_t t2 = t0; t2 < t1; t2++) {
            const size_t idx = (size_t)t2 * B2 + low;
            off2[idx] = run;
            run += counts2[idx];
        }
    }
    if (k <= nbt) bsize[k] = run;
    if (run) atomicMax(&blk_max, run);
    __syncthreads();
    // thousands of workgroups: only those that can still raise the maximum touch the global atomic
    if (threadIdx.x == 0 && blk_max > __atomic_load_n(max_size, __ATOMIC_RELAXED)) atomicMax(max_size, blk_max);
}
// Same result, one workgroup per SEGMENT (few segments with many tiles each: the per-key loop over ~100 tiles is split
// over 8 slices of 128 key lanes).  blockDim.x must be 1024.
static __global__ void __launch_bounds__(1024) radix_colscan2_seg_kernel(const uint32_t* __restrict__ counts2, uint32_t* __restrict__ off2,
                                                                  const uint32_t* __restrict__ tile2_start, uint32_t* __restrict__ bsize,
                                                                  uint32_t nbins, int LB, uint32_t* __restrict__ max_size) {
    __shared__ uint32_t part[8][128];
    __shared__ uint32_t blk_max;
    const uint32_t B2 = 1u << LB;
    const uint32_t q = blockIdx.x;
    const uint32_t key = threadIdx.x & 127, sl = threadIdx.x >> 7;
    if (threadIdx.x == 0) blk_max = 0;
    const uint32_t t0 = tile2_start[q], t1 = tile2_start[q + 1];
    const uint32_t per = (t1 - t0 + 7) / 8;
    uint32_t a = t0 + sl * per, b = a + per;
    if (a > t1) a = t1;
    if (b > t1) b = t1;
    uint32_t s = 0;
    if (key < B2) {
#pragma unroll 8
        for (uint32_t t2 = a; t2 < b; t2++) s += counts2[(size_t)t2 * B2 + key];  // independent loads: keep several in flight
    }
    part[sl][key] = s;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t i = 0; i < 8; i++) {
        const uint32_t v = part[i][key];
        if (i < sl) run += v;
        total += v;
    }
    if (key < B2) {
#pragma unroll 8
        for (uint32_t t2 = a; t2 < b; t2++) {
            const size_t idx = (size_t)t2 * B2 + key;
            const uint32_t cnt = counts2[idx];
            off2[idx] = run;
            run += cnt;
        }
        if (sl == 0) {
            bsize[(size_t)q * B2 + key] = total;
            if (total) atomicMax(&blk_max, total);
        }
    }
    if (q == 0 && threadIdx.x == 0) bsize[(size_t)nbins * B2] = 0;
    __syncthreads();
    if (threadIdx.x == 0 && blk_max > __atomic_load_n(max_size, __ATOMIC_RELAXED)) atomicMax(max_size, blk_max);
}
// ---- level k scatter: out[boff[k] + off2[tile][key] + rank inside (tile, key)], k = q * B2 + key; the remainder bits below
// `shift` travel along in rem_out while levels remain (rem_out == nullptr on the last level)
template <class RIN, class ROUT>
__global__ void __launch_bounds__(SORT_THREADS) radix_scatter2_kernel(const uint32_t* __restrict__ v1, const RIN* __restrict__ l1,
                                                                      const uint32_t* __restrict__ binstart,
                                                                      const uint32_t* __restrict__ tile2_start,
                                                                      const uint32_t* __restrict__ counts2,
                                                                      const uint32_t* __restrict__ off2, const uint32_t* __restrict__ boff,
                                                                      uint32_t* __restrict__ sorted, ROUT* __restrict__ rem_out, uint32_t nbins,
                                                                      int LB, int shift, uint32_t xcd) {
    __shared__ uint32_t lcount[128], lstart[128], cursor[128], gbase[128], wave_tot[4];
    __shared__ uint32_t sv_[SORT_TILE];
    __shared__ uint8_t slow_[SORT_TILE];
    __shared__ ROUT srem_[SORT_TILE];
    const uint32_t ntiles2 = tile2_start[nbins];  // the grid is an upper bound computed on the host
    if (blockIdx.x >= ntiles2) return;
    const uint32_t t2 = xcd ? xcd_tile(blockIdx.x, ntiles2) : blockIdx.x;
    const uint32_t B2 = 1u << LB;
    const uint32_t q = find_bucket(tile2_start, nbins, t2);
    const uint32_t lt = t2 - tile2_start[q];
    const uint32_t lo = binstart[q] + lt * SORT_TILE;
    uint32_t hi = lo + SORT_TILE;
    if (hi > binstart[q + 1]) hi = binstart[q + 1];
    {
        const uint32_t i = threadIdx.x;  // one key per thread (B2 <= 128 < SORT_THREADS)
        const uint32_t cnt = (i < B2) ? counts2[(size_t)t2 * B2 + i] : 0u;
        const uint32_t start = block_excl_scan(cnt, wave_tot);
        if (i < 128) {
            lcount[i] = cnt;
            gbase[i] = (i < B2) ? boff[(q << LB) | i] + off2[(size_t)t2 * B2 + i] : 0u;
            lstart[i] = start;
            cursor[i] = start;
        }
    }
    __syncthreads();
    {
        // all of a thread's global loads are issued before the first LDS atomic (32 items per thread)
        constexpr int PER = SORT_TILE / SORT_THREADS;
        uint32_t lv[PER];
        uint32_t ll[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const uint32_t i = lo + threadIdx.x + (uint32_t)k * SORT_THREADS;
            const bool ok = i < hi;
            lv[k] = ok ? v1[i] : 0u;
            ll[k] = ok ? (uint32_t)l1[i] : 0xffffffffu;
        }
        const uint32_t rmask = (1u << shift) - 1;
#pragma unroll
        for (int k = 0; k < PER; k++) {
            if (ll[k] != 0xffffffffu) {
                const uint32_t key = (ll[k] >> shift) & (B2 - 1);
                const uint32_t pos = atomicAdd(&cursor[key], 1u);
                sv_[pos] = lv[k];
                slow_[pos] = (uint8_t)key;
                if (rem_out) srem_[pos] = (ROUT)(ll[k] & rmask);
            }
        }
    }
    __syncthreads();
    const uint32_t total = hi - lo;
#pragma unroll 4
    for (uint32_t pos = threadIdx.x; pos < total; pos += blockDim.x) {
        const uint32_t key = slow_[pos];
        const size_t dst = (size_t)gbase[key] + (pos - lstart[key]);
        sorted[dst] = sv_[pos];
        if (rem_out) rem_out[dst] = srem_[pos];
    }
}

// ---- balanced ("segmented") accumulate: thread t owns the S consecutive entries sorted[tS, (t+1)S) whatever buckets they
// belong to, and flushes one partial sum per bucket it touches.  Every lane of a wave performs exactly S mixed additions,
// so small or uneven buckets (wide windows: ~100 entries each) cost no SIMD idle time.  Bucket k receives one partial from
// each of the threads tS in [boff[k] / S, (boff[k+1] - 1) / S]: cnt[k] = that count, slot = start[k] + (t - boff[k] / S),
// the (cnt, start, partial) triple the reduce rounds consume.
static __global__ void msm_alloc_seg_kernel(const uint32_t* __restrict__ boff, uint32_t* __restrict__ cnt, uint32_t nbt, uint32_t S) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > nbt) return;
    uint32_t c = 0;
    if (k < nbt) {
        const uint32_t lo = boff[k], hi = boff[k + 1];
        if (hi > lo) c = (hi - 1) / S - lo / S + 1;
    }
    cnt[k] = c;
}
// PREFETCH (single-round MSMs: one wave per SIMD, nothing else hides the base gather): the slot of entry pos + 1 is requested
// before the addition of entry pos starts, so its ~2 us of HBM latency run under the ~9 us of arithmetic.  Big MSMs keep two
// resident waves per SIMD instead (the extra 24 registers of the prefetched slot were measured: no gain there).
template <class F, int MINW, bool PREFETCH>  // MINW: waves per SIMD asked of the register allocator (3 -> <= 168 VGPRs for G1)
__global__ void __launch_bounds__(256, MINW) msm_accumulate_seg_kernel(const aff_mem_t<F>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                                 const uint32_t* __restrict__ boff, const uint32_t* __restrict__ start,
                                                                 xyzz_mem_t<F>* __restrict__ partial, uint32_t nbt, uint32_t S, uint32_t debug_idx_mask) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = boff[nbt];
    const uint64_t lo64 = (uint64_t)t * S;
    if (lo64 >= total) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (total - lo < S) ? total : lo + S;
    uint32_t k = find_bucket(boff, nbt, lo);  // the non-empty bucket that contains entry `lo`
    uint32_t kend = boff[k + 1];
    // bucket bookkeeping one bucket ahead (see msm_accumulate_lazy_kernel): no dependent loads at the flush
    uint32_t kend2 = boff[k + 2 <= nbt ? k + 2 : nbt];
    uint32_t start_k = start[k];
    uint32_t part_off = t - boff[k] / S;  // only the thread's first bucket can have started in an earlier thread
    xyzz_t<F> acc = xyzz_t<F>::inf();
    // entry = slot of the base relative to `bases` (31 bits) | sign << 31  (mask: timing experiments of profiling builds only)
    auto slot_of = [&](uint32_t e) -> const aff_mem_t<F>* { return &bases[(e & 0x7fffffffu) & debug_idx_mask]; };
    uint32_t e_next = sorted[lo];
    aff_mem_t<F> raw_next;
    if (PREFETCH) raw_next = *slot_of(e_next);
    for (uint32_t pos = lo; pos < hi; pos++) {
        if (pos >= kend) {  // bucket k ends inside this segment: flush and move to the bucket of `pos`
            store_xyzz<F>(&partial[start_k + part_off], acc);
            acc = xyzz_t<F>::inf();
            part_off = 0;
            k++;
            kend = kend2;
            while (pos >= kend) {
                k++;
                kend = boff[k + 1];
            }
            kend2 = boff[k + 2 <= nbt ? k + 2 : nbt];
            start_k = start[k];
        }
        const uint32_t e = e_next;
        aff_mem_t<F> raw;
        if (PREFETCH) {
            raw = raw_next;
            if (pos + 1 < hi) {
                e_next = sorted[pos + 1];
                raw_next = *slot_of(e_next);
            }
        } else {
            raw = *slot_of(e);
            if (pos + 1 < hi) e_next = sorted[pos + 1];
        }
        acc.add_affine(load_aff<F>(&raw), (e >> 31) != 0);
    }
    store_xyzz<F>(&partial[start_k + part_off], acc);
}

// raw partial sums of msm_accumulate_lazy_kernel (52 signed limbs, congruent to coordinate * 2^406) -> the exact representation
// the tail kernels read (canonical, internal form): four products per partial sum, every lane busy
static __global__ void __launch_bounds__(256) g1_partials_to_exact_kernel(const g1_lazy_partial_t* __restrict__ raw, g1_xyzz_mem_t* __restrict__ partial,
                                                                   const uint32_t* __restrict__ start, uint32_t nbt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= start[nbt]) return;  // start[nbt] = number of partial sums
    g1_store_xyzz(&partial[i], xyzz_lazy_t::exact_from_raw(&raw[i]));
}
// the cold path of the lazy kernel, out of line: acc += +-(px, py) on the exact arithmetic (doubling, cancellation, or a false alarm
// of the low-limb filter).  Keeping ~9 000 instructions and their constants out of the kernel body relieves its scalar registers.
static __device__ __noinline__ void lazy_exceptional_add(xyzz_lazy_t* acc, const fql_t* px, const fql_t* py, bool neg) {
    g1_xyzz_t ex = acc->to_exact();
    const fq_t c348 = fq_t::from_table(FqLConv::C348);
    fq_t bx, by;
#pragma unroll
    for (int i = 0; i < 13; i++) bx.v[i] = (uint32_t)px->v[i], by.v[i] = (uint32_t)py->v[i];
    ex.add_affine({bx * c348, by * c348}, neg);
    *acc = xyzz_lazy_t::from_exact(ex);
}
// ---- the same kernel on the lazily reduced arithmetic of ffl.hip.h (G1).  The base slots hold canonical residues of the
// coordinates times 2^406 as unpacked limbs (g1_lazy_slot_t; runtime.hip.h::bases_to_lazy_form / convert_bases form406); the accumulator lives in signed limbs
// without a canonical form; partial sums are flushed raw and converted to the exact representation by a dense pass afterwards.  The addition law's exceptional cases (the filter of xyzz_lazy_t::madd) are resolved on
// the exact arithmetic: cold code.  Per addition: 3 046 multiply-adds + ~900 other instructions (exact kernel: 2 951 + 2 238).
template <bool PREFETCH>
__global__ void __launch_bounds__(256, 1) msm_accumulate_lazy_kernel(const g1_aff_mem_t* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                              const uint32_t* __restrict__ boff, const uint32_t* __restrict__ start,
                                                              g1_lazy_partial_t* __restrict__ partial, uint32_t nbt, uint32_t S, uint32_t debug_idx_mask) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = boff[nbt];
    const uint64_t lo64 = (uint64_t)t * S;
    if (lo64 >= total) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (total - lo < S) ? total : lo + S;
    uint32_t k = find_bucket(boff, nbt, lo);  // the non-empty bucket that contains entry `lo`
    uint32_t kend = boff[k + 1];
    // The bucket bookkeeping runs one bucket ahead: the end of the NEXT bucket (boff[k + 2]) and the partial-sum slot of the
    // CURRENT one (start[k]) are requested when a lane enters bucket k, so the flush at the bucket's end - reached by some lane
    // of a wave in most iterations - finds both in registers instead of waiting for two dependent loads from 8 MB arrays
    // (boff / start: one entry per bucket) while the wave's arithmetic stands still.  boff has nbt + 1 entries: the index is
    // clamped (a clamped value is only ever compared after the segment's last entry).
    uint32_t kend2 = boff[k + 2 <= nbt ? k + 2 : nbt];
    uint32_t start_k = start[k];
    // slot of this thread's partial sum inside bucket k: start[k] + (t - first thread of the bucket).  Only the thread's FIRST bucket
    // can have started in an earlier thread; every later one starts inside this segment, i.e. its first thread is t (no division
    // inside the loop)
    uint32_t part_off = t - boff[k] / S;
    xyzz_lazy_t acc = xyzz_lazy_t::infinity();
    auto slot_of = [&](uint32_t e) -> const g1_lazy_slot_t* { return (const g1_lazy_slot_t*)&bases[(e & 0x7fffffffu) & debug_idx_mask]; };
    // Software pipeline of the gather.  PREFETCH (one wave per SIMD, nothing else hides latency): two stages - the INDEX of entry
    // pos + 2 and the BASE of entry pos + 1 are requested before the addition of entry pos starts, so neither the index load nor
    // the dependent base load (index -> address) is ever waited for with an idle SIMD.  Otherwise (two resident waves): the index
    // one iteration ahead, the base at its use.
    uint32_t e_cur = sorted[lo];
    uint32_t e_n1 = lo + 1 < hi ? sorted[lo + 1] : 0u;
    g1_lazy_slot_t raw_next;
    if (PREFETCH) raw_next = *slot_of(e_cur);
    for (uint32_t pos = lo;; pos++) {
        const bool end = pos >= hi;
        if (end || pos >= kend) {
            // bucket k ends here: flush its partial sum - RAW (signed limbs, no arithmetic): the lanes of a wave reach their bucket
            // boundaries in different iterations, so whatever the flush costs, the wave pays it in most iterations (at 2^24: 1.7
            // flushes per 64 additions per lane = some lane flushes in 81 % of the iterations); g1_partials_to_exact_kernel
            // converts all partial sums in one dense pass afterwards
            acc.store_raw(&partial[start_k + part_off]);
            if (end) break;
            part_off = 0;
            acc.inf = true;  // the coordinates of an empty accumulator are never read
            k++;
            kend = kend2;
            while (pos >= kend) {  // empty buckets in between (rare: the sort leaves none for uniform scalars)
                k++;
                kend = boff[k + 1];
            }
            kend2 = boff[k + 2 <= nbt ? k + 2 : nbt];
            start_k = start[k];
        }
        const uint32_t e = e_cur;
        g1_lazy_slot_t raw;
        if (PREFETCH) {
            raw = raw_next;
            if (pos + 1 < hi) raw_next = *slot_of(e_n1);  // e_n1 arrived during the previous addition
        } else {
            raw = *slot_of(e);
        }
        e_cur = e_n1;
        if (pos + 2 < hi) e_n1 = sorted[pos + 2];
        if (raw.w[26]) continue;  // the point at infinity
        const bool neg = (e >> 31) != 0;
        fql_t px, py;
        raw.coords(px, py);
        if (!acc.madd(px, py, neg)) {
            // acc == +-P (or a false alarm of the low-limb filter, 6 * 2^-29 per addition): resolved out of line on the exact arithmetic.
            // The callee gets COPIES: an accumulator whose address escaped would live in scratch memory for the whole loop.
            xyzz_lazy_t tmp = acc;
            const fql_t tx = px, ty = py;
            lazy_exceptional_add(&tmp, &tx, &ty, neg);
            acc = tmp;
        }
    }
}

// ---- G2: the same kernel on the lazily reduced Fq2 arithmetic of ffl2.hip.h (round 4).  Base slots: g2_lazy_slot_t (canonical residues
// of the four coordinate components times 2^406, unpacked); partial sums leave raw (104 limbs) and g2_partials_to_exact_kernel converts
// them for the tail kernels; exceptional additions are resolved out of line on the exact arithmetic.
static __global__ void __launch_bounds__(256) g2_partials_to_exact_kernel(const g2_lazy_partial_t* __restrict__ raw, xyzz_mem_t<fq2_t>* __restrict__ partial,
                                                                   const uint32_t* __restrict__ start, uint32_t nbt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= start[nbt]) return;
    store_xyzz<fq2_t>(&partial[i], xyzz_lazy2_t::exact_from_raw(&raw[i]));
}
// (inlined: an out-of-line call from this 400-register kernel never returns on gfx950 / ROCm 7.2 - tools/exp/g2lazy_dev.hip reproduces it in
// 60 lines, with the call the second addition of a doubled point hangs, inlined it matches the host chain; the G1 kernel's 248 registers
// keep its call working.  The cold code costs the hot loop nothing measurable: the two blocks of the addition keep their instruction counts.)
static __device__ __forceinline__ void lazy2_exceptional_add(xyzz_lazy2_t* acc, const fq2l_t* px, const fq2l_t* py, bool neg) {
    xyzz_t<fq2_t> ex = acc->to_exact();
    const fq_t c348 = fq_t::from_table(FqLConv::C348);
    fq_t t[4];
#pragma unroll
    for (int i = 0; i < 13; i++) {
        t[0].v[i] = (uint32_t)px->c0.v[i], t[1].v[i] = (uint32_t)px->c1.v[i];
        t[2].v[i] = (uint32_t)py->c0.v[i], t[3].v[i] = (uint32_t)py->c1.v[i];
    }
    ex.add_affine({{t[0] * c348, t[1] * c348}, {t[2] * c348, t[3] * c348}}, neg);
    *acc = xyzz_lazy2_t::from_exact(ex);
}
template <bool PREFETCH>
__global__ void __launch_bounds__(256, 1) msm_accumulate_lazy2_kernel(const aff_mem_t<fq2_t>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                               const uint32_t* __restrict__ boff, const uint32_t* __restrict__ start,
                                                               g2_lazy_partial_t* __restrict__ partial, uint32_t nbt, uint32_t S, uint32_t debug_idx_mask) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = boff[nbt];
    const uint64_t lo64 = (uint64_t)t * S;
    if (lo64 >= total) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (total - lo < S) ? total : lo + S;
    uint32_t k = find_bucket(boff, nbt, lo);
    uint32_t kend = boff[k + 1];
    uint32_t kend2 = boff[k + 2 <= nbt ? k + 2 : nbt];  // bucket bookkeeping one bucket ahead (see msm_accumulate_lazy_kernel)
    uint32_t start_k = start[k];
    uint32_t part_off = t - boff[k] / S;
    xyzz_lazy2_t acc = xyzz_lazy2_t::infinity();
    auto slot_of = [&](uint32_t e) -> const g2_lazy_slot_t* { return (const g2_lazy_slot_t*)&bases[(e & 0x7fffffffu) & debug_idx_mask]; };
    uint32_t e_cur = sorted[lo];
    uint32_t e_n1 = lo + 1 < hi ? sorted[lo + 1] : 0u;
    for (uint32_t pos = lo;; pos++) {
        const bool end = pos >= hi;
        if (end || pos >= kend) {
            acc.store_raw(&partial[start_k + part_off]);
            if (end) break;
            part_off = 0;
            acc.inf = true;
            k++;
            kend = kend2;
            while (pos >= kend) {
                k++;
                kend = boff[k + 1];
            }
            kend2 = boff[k + 2 <= nbt ? k + 2 : nbt];
            start_k = start[k];
        }
        const uint32_t e = e_cur;
        const g2_lazy_slot_t* sp = slot_of(e);
        e_cur = e_n1;
        if (pos + 2 < hi) e_n1 = sorted[pos + 2];
        if (sp->w[g2_lazy_slot_t::INF_WORD]) continue;  // the point at infinity
        const bool neg = (e >> 31) != 0;
        fq2l_t px, py;
        sp->coords(px, py);
        if (!acc.madd(px, py, neg)) {
            xyzz_lazy2_t tmp = acc;
            const fq2l_t tx = px, ty = py;
            lazy2_exceptional_add(&tmp, &tx, &ty, neg);
            acc = tmp;
        }
    }
    (void)PREFETCH;  // the 52-limb slot is read at its use: a second resident slot would not fit the register file
}

// ---- G2 on a lane pair (round 5; ffl2p.hip.h): lanes 2 t and 2 t + 1 walk segment t together, the even lane holding the c0 component of
// every value and the odd lane the c1 component; every case of the addition law stays in the pair arithmetic.  Partial sums leave as
// g2_pair_partial_t (each lane stores its four components) and g2_pair_partials_to_exact_kernel converts them - one thread per coordinate
// component - into the exact representation the tail kernels read.
static __global__ void __launch_bounds__(256) g2_pair_partials_to_exact_kernel(const g2_pair_partial_t* __restrict__ raw, xyzz_mem_t<fq2_t>* __restrict__ partial,
                                                                        const uint32_t* __restrict__ start, uint32_t nbt) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = g >> 3, k = g & 7;  // partial sum, component (2 * coordinate + c0 / c1)
    if (i >= start[nbt]) return;
    const uint4* q = (const uint4*)&raw[i].w[16 * k];
    uint32_t t[16];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint4 u = q[j];
        t[4 * j] = u.x, t[4 * j + 1] = u.y, t[4 * j + 2] = u.z, t[4 * j + 3] = u.w;
    }
    fql_t c;
#pragma unroll
    for (int j = 0; j < 13; j++) c.v[j] = (int32_t)t[j];
    c.to_exact().store(&((fq_mem_t*)&partial[i])[k]);
}
template <bool PREFETCH>  // (a template so that only the unit that launches it - api_g2.hip - compiles it)
__global__ void __launch_bounds__(256, 2) msm_accumulate_pair2_kernel(const aff_mem_t<fq2_t>* __restrict__ bases, const uint32_t* __restrict__ sorted,
                                                               const uint32_t* __restrict__ boff, const uint32_t* __restrict__ start,
                                                               g2_pair_partial_t* __restrict__ partial, uint32_t nbt, uint32_t S, uint32_t debug_idx_mask) {
    const uint32_t gt = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = gt >> 1;            // the segment of this lane pair
    const int comp = (int)(gt & 1u);       // 0: c0 (even lane), 1: c1 (odd lane)
    const uint32_t total = boff[nbt];
    const uint64_t lo64 = (uint64_t)t * S;
    if (lo64 >= total) return;  // both lanes of a pair leave together
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (total - lo < S) ? total : lo + S;
    uint32_t k = find_bucket(boff, nbt, lo);
    uint32_t kend = boff[k + 1];
    uint32_t start_k = start[k];  // (no look-ahead of the bucket bookkeeping here: two resident waves hide the loads, and the registers are needed)
    uint32_t part_off = t - boff[k] / S;
    fq2p::xyzz_pair_t<fq2p::xp_dev> acc;
    acc.inf = true;
    auto slot_of = [&](uint32_t e) -> const g2_lazy_slot_t* { return (const g2_lazy_slot_t*)&bases[(e & 0x7fffffffu) & debug_idx_mask]; };
    for (uint32_t pos = lo;; pos++) {
        const bool end = pos >= hi;
        if (end || pos >= kend) {
            uint4* q = (uint4*)&partial[start_k + part_off].w[16 * comp];  // component (2 * coordinate + comp) at words 32 * coordinate + 16 * comp
            if (acc.inf) {
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++)
#pragma unroll
                    for (int j = 0; j < 4; j++) q[8 * c4 + j] = make_uint4(0, 0, 0, 0);
            } else {
                const fql_t* cs[4] = {&acc.x[0], &acc.y[0], &acc.zz[0], &acc.zzz[0]};
#pragma unroll
                for (int c4 = 0; c4 < 4; c4++) {
                    const int32_t* v = cs[c4]->v;
                    q[8 * c4 + 0] = make_uint4((uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]);
                    q[8 * c4 + 1] = make_uint4((uint32_t)v[4], (uint32_t)v[5], (uint32_t)v[6], (uint32_t)v[7]);
                    q[8 * c4 + 2] = make_uint4((uint32_t)v[8], (uint32_t)v[9], (uint32_t)v[10], (uint32_t)v[11]);
                    q[8 * c4 + 3] = make_uint4((uint32_t)v[12], 0u, 0u, 0u);
                }
            }
            if (end) break;
            part_off = 0;
            acc.inf = true;
            do {
                k++;
                kend = boff[k + 1];
            } while (pos >= kend);
            start_k = start[k];
        }
        const uint32_t e = sorted[pos];
        const g2_lazy_slot_t* sp = slot_of(e);
        if (sp->w[g2_lazy_slot_t::INF_WORD]) continue;  // the point at infinity (both lanes read the same word)
        fql_t px[1], py[1];
        sp->component(comp, px[0], py[0]);
        acc.madd(px, py, (e >> 31) != 0);
    }
}

}  // namespace sv
