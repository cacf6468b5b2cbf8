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
#include "msm.hip.h"

namespace sv {

static constexpr int SORT_TILE = 8192;     // items per workgroup tile
static constexpr int SORT_THREADS = 256;   // 32 items per thread

struct msm_radix_params_t {
    size_t n;              // scalars
    int c, W, J;           // window bits, bucket windows, base tables (digit row j*W + w feeds window w)
    int HB, LB;            // bucket index = (bin << LB) | rem, bin < 2^HB (level-1 key), rem < 2^LB (<= 7: one more level, <= 14: two)
    uint32_t nb;           // 2^(c-1)
    uint32_t tiles_per_row;  // ceil(n / SORT_TILE)
    uint32_t TPW;          // level-1 tiles per window = J * tiles_per_row
};

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
    const uint32_t g = blockIdx.x;            // global level-1 tile
    const uint32_t w = g / p.TPW, tw = g - w * p.TPW;
    const uint32_t j = tw / p.tiles_per_row, t = tw - j * p.tiles_per_row;
    for (uint32_t i = threadIdx.x; i < B1; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const size_t lo = (size_t)t * SORT_TILE;
    const size_t hi = (lo + SORT_TILE < p.n) ? lo + SORT_TILE : p.n;
    const int half = 1 << (p.c - 1);
    for_each_digit_t<DT>(digits + (size_t)(j * p.W + w) * p.n, p.n, lo, hi, [&](uint32_t u, size_t) {
        uint32_t b, neg;
        if (digit_bucket(u, half, b, neg)) atomicAdd(&hist[b >> p.LB], 1u);
    });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < B1; i += blockDim.x) counts1[((size_t)w * B1 + i) * p.TPW + tw] = hist[i];
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
    const uint32_t g = blockIdx.x;
    const uint32_t w = g / p.TPW, tw = g - w * p.TPW;
    const uint32_t j = tw / p.tiles_per_row, t = tw - j * p.tiles_per_row;
    // this tile's histogram was computed by radix_hist1_kernel; its global run starts are off1[...]
    const size_t lo = (size_t)t * SORT_TILE;
    const size_t hi = (lo + SORT_TILE < p.n) ? lo + SORT_TILE : p.n;
    const int half = 1 << (p.c - 1);
    const DT* row = digits + (size_t)(j * p.W + w) * p.n;
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
        const uint32_t cnt = (i < B1) ? counts1[((size_t)w * B1 + i) * p.TPW + tw] : 0u;
        gbase[i] = (i < B1) ? off1[((size_t)w * B1 + i) * p.TPW + tw] : 0u;
        lcount[i] = cnt;
        const uint32_t start = block_excl_scan(cnt, wave_tot);
        lstart[i] = start;
        cursor[i] = start;
    }
    __syncthreads();
    const uint32_t voff = (uint32_t)((size_t)j * p.n);
    const uint32_t lmask = (1u << p.LB) - 1;
    auto place = [&](uint32_t u, size_t i) {
        uint32_t b, neg;
        if (digit_bucket(u, half, b, neg)) {
            const uint32_t bin = b >> p.LB;
            const uint32_t pos = atomicAdd(&cursor[bin], 1u);
            sv_[pos] = (voff + (uint32_t)i) | neg;
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

// ---- level 2 tiling: bin q = w * B1 + bin covers [binstart[q], binstart[q+1]) of v1/l1 and gets ceil(size / TILE) tiles
static __global__ void radix_bin_layout_kernel(const uint32_t* __restrict__ off1, const uint32_t* __restrict__ counts1, size_t ncounts1,
                                        uint32_t* __restrict__ binstart, uint32_t nbins, uint32_t TPW) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > nbins) return;
    binstart[q] = (q < nbins) ? off1[(size_t)q * TPW] : off1[ncounts1 - 1] + counts1[ncounts1 - 1];
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
    __shared__ uint32_t hist[128];
    const uint32_t t2 = blockIdx.x;
    if (t2 >= tile2_start[nbins]) return;
    const uint32_t B2 = 1u << LB;
    const uint32_t q = find_bucket(tile2_start, nbins, t2);
    const uint32_t lt = t2 - tile2_start[q];
    for (uint32_t i = threadIdx.x; i < B2; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    const uint32_t lo = binstart[q] + lt * SORT_TILE;
    uint32_t hi = lo + SORT_TILE;
    if (hi > binstart[q + 1]) hi = binstart[q + 1];
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) atomicAdd(&hist[((uint32_t)l1[i] >> shift) & (B2 - 1)], 1u);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < B2; i += blockDim.x) counts2[(size_t)t2 * B2 + i] = hist[i];
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
    if (key < B2)
        for (uint32_t t2 = a; t2 < b; t2++) s += counts2[(size_t)t2 * B2 + key];
    part[sl][key] = s;
    __syncthreads();
    uint32_t run = 0, total = 0;
    for (uint32_t i = 0; i < 8; i++) {
        const uint32_t v = part[i][key];
        if (i < sl) run += v;
        total += v;
    }
    if (key < B2) {
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
                                                                      int LB, int shift) {
    __shared__ uint32_t lcount[128], lstart[128], cursor[128], gbase[128], wave_tot[4];
    __shared__ uint32_t sv_[SORT_TILE];
    __shared__ uint8_t slow_[SORT_TILE];
    __shared__ ROUT srem_[SORT_TILE];
    const uint32_t t2 = blockIdx.x;
    if (t2 >= tile2_start[nbins]) return;
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
template <class F, int MINW>  // MINW: waves per SIMD asked of the register allocator (3 -> <= 168 VGPRs for G1)
__global__ void __launch_bounds__(256, MINW) msm_accumulate_seg_kernel(const aff_mem_t<F>* __restrict__ bases,
                                                                 const aff_mem_t<F>* __restrict__ bases1, uint32_t n0,
                                                                 const uint32_t* __restrict__ sorted, const uint32_t* __restrict__ boff,
                                                                 const uint32_t* __restrict__ start, xyzz_mem_t<F>* __restrict__ partial,
                                                                 uint32_t nbt, uint32_t S, uint32_t n, size_t table_stride,
                                                                 uint32_t debug_idx_mask) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t total = boff[nbt];
    const uint64_t lo64 = (uint64_t)t * S;
    if (lo64 >= total) return;
    const uint32_t lo = (uint32_t)lo64;
    const uint32_t hi = (total - lo < S) ? total : lo + S;
    uint32_t k = find_bucket(boff, nbt, lo);  // the non-empty bucket that contains entry `lo`
    uint32_t kend = boff[k + 1];
    xyzz_t<F> acc = xyzz_t<F>::inf();
    for (uint32_t pos = lo; pos < hi; pos++) {
        if (pos >= kend) {  // bucket k ends inside this segment: flush and move to the bucket of `pos`
            store_xyzz<F>(&partial[start[k] + (t - boff[k] / S)], acc);
            acc = xyzz_t<F>::inf();
            do {
                k++;
                kend = boff[k + 1];
            } while (pos >= kend);
        }
        const uint32_t e = sorted[pos];
        const uint32_t v = e & 0x7fffffffu;  // virtual index = table * n + scalar index
        const uint32_t tbl = v / n;
        const uint32_t idx = (v - tbl * n) & debug_idx_mask;  // bases come in up to two segments (mask: timing experiments only)
        const aff_mem_t<F> raw = *((idx < n0 ? &bases[idx] : &bases1[idx - n0]) + (size_t)tbl * table_stride);
        acc.add_affine(load_aff<F>(&raw), (e >> 31) != 0);
    }
    store_xyzz<F>(&partial[start[k] + (t - boff[k] / S)], acc);
}

}  // namespace sv
