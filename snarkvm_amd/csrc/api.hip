// api.hip - C ABI (include/snarkvm_hip.h) and host runtime of the gfx950 MSM / NTT backend.
//
// Host runtime = what algorithms/cuda/cuda/snarkvm.cu:73-312 (snarkvm_t) and snarkvm_api.cu:23-84 are in the
// reference: a lazily constructed per-process context (device arenas, stream, twiddle tables), staging of the
// caller's host buffers, error reporting as RustError, serialisation of concurrent callers.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/snarkvm_hip.h"
#include "ec.cuh"
#include "ff.cuh"
#include "msm.cuh"
#include "msm_sort.cuh"
#include "ntt.cuh"
#include "group.cuh"
#include "poly.cuh"
#include "serde.cuh"

using namespace sv;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static RustError ok() { return RustError{0, nullptr}; }
static RustError fail(int code, const std::string& msg) {
    char* m = (char*)malloc(msg.size() + 1);
    if (m) memcpy(m, msg.c_str(), msg.size() + 1);
    return RustError{code ? code : 1, m};
}
struct hip_failure {
    hipError_t e;
    const char* what;
    int line;
};
#define HIP_TRY(x)                                             \
    do {                                                       \
        hipError_t _e = (x);                                   \
        if (_e != hipSuccess) throw hip_failure{_e, #x, __LINE__}; \
    } while (0)
static RustError from_failure(const hip_failure& f) {
    char buf[512];
    snprintf(buf, sizeof buf, "snarkvm_hip: %s failed at api.hip:%d: %s", f.what, f.line, hipGetErrorString(f.e));
    return fail((int)f.e, buf);
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct dev_buf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        if (p) HIP_TRY(hipFree(p));
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        HIP_TRY(hipMalloc(&p, want));
        cap = want;
    }
    template <class T>
    T* as() const {
        return (T*)p;
    }
};

struct msm_ws_t {
    hipStream_t stream = nullptr;
    dev_buf scalars, digits, counts, offsets, scan_tmp, sorted, boff, cnt_a, cnt_b, start_a, start_b, part_a, part_b, contrib, wsum, result;
    dev_buf rv1, rl1, rcounts2, roff2, rbinstart, rntiles, rtstart, rbsize;  // radix-partition sort (msm_sort.cuh)
    dev_buf rv2, rl2, rmid_size, rmid_boff;                                   // its middle level (wide windows)
    dev_buf fold_sums, fold_idx;                                              // two-axis bucket fold (wide windows)
};

struct phase_rec {
    const char* name;
    hipEvent_t e0, e1;
    double ms;
};

struct context_t {
    std::mutex mu;
    bool ready = false;
    int device = 0;
    hipStream_t stream = nullptr;
    ntt_tables_t tb{};
    dev_buf tables_mem;
    // NTT staging
    dev_buf ntt_data, ntt_scratch, ntt_acc;
    dev_buf serde_status;  // one u32 of SERDE_* bits (serde.cuh)
    dev_buf poly[5];  // staging / scratch of the prover-round vector kernels (poly.cuh)
    // MSM workspaces: lane 0 runs on the main stream; lanes 1.. are used by the batch API so that the latency-bound
    // tail of one MSM (bucket reduction, Horner) overlaps the throughput-bound accumulation of the next
    static constexpr int LANES = 8;  // streams + workspaces available to the batch API
    msm_ws_t lane[LANES];
    // lanes a batch actually cycles through: more lanes hide more of the latency-bound tail of small MSMs, fewer keep the
    // workspace footprint of big ones down (a 2^24 lane holds ~4 GB)
    static int batch_lanes(size_t npoints) {
        static const int env = getenv("SNARKVM_HIP_LANES") ? atoi(getenv("SNARKVM_HIP_LANES")) : 0;
        int l = env > 0 ? env : (npoints >= ((size_t)1 << 20) ? 3 : LANES);  // measured: 8 lanes +7 % below 2^20, no gain above
        return l < 1 ? 1 : (l > LANES ? LANES : l);
    }
    dev_buf bases_tmp, scalars_tmp, gen_pts, gen_prod;
    void* batch_pinned = nullptr;
    size_t batch_pinned_cap = 0;
    // profiling
    bool profiling = false;
    std::vector<phase_rec> phases;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;

    void init() {
        if (ready) return;
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) throw hip_failure{e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount (no MI355X visible)", __LINE__};
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        lane[0].stream = stream;
        for (int l = 1; l < LANES; l++) HIP_TRY(hipStreamCreateWithFlags(&lane[l].stream, hipStreamNonBlocking));
        // tables: 4 x (lo + hi) x 4096 + 2 x 128 + 25 + 4, 32 B each
        const size_t entries = 8 * NTT_TW_SIZE + 256 + 32 + 8;
        tables_mem.ensure(entries * sizeof(fr_mem_t));
        fr_mem_t* base = tables_mem.as<fr_mem_t>();
        size_t off = 0;
        auto take = [&](size_t n) {
            fr_mem_t* r = base + off;
            off += n;
            return r;
        };
        for (int d = 0; d < 2; d++) {
            tb.pow_lo[d] = take(NTT_TW_SIZE);
            tb.pow_hi[d] = take(NTT_TW_SIZE);
            tb.g_lo[d] = take(NTT_TW_SIZE);
            tb.g_hi[d] = take(NTT_TW_SIZE);
            tb.local[d] = take(128);
        }
        tb.size_inv = take(32);
        tb.consts = take(8);
        HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel_v2, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)msm_hist_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)msm_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
        HIP_TRY(hipFuncSetAttribute((const void*)msm_locoff_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024));
        hipLaunchKernelGGL(ntt_setup_consts, dim3(1), dim3(64), 0, stream, tb);
        hipLaunchKernelGGL(ntt_fill_tables, dim3(NTT_TW_SIZE / 256), dim3(256), 0, stream, tb);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(stream));
        ready = true;
    }
    // ---- profiling helpers
    hipEvent_t new_event() {
        if (events_used == event_pool.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            event_pool.push_back(e);
        }
        return event_pool[events_used++];
    }
    void begin_call() {
        phases.clear();
        events_used = 0;
    }
    void phase_begin(const char* name) {
        if (!profiling) return;
        phase_rec r{name, new_event(), new_event(), 0.0};
        HIP_TRY(hipEventRecord(r.e0, stream));
        phases.push_back(r);
    }
    void phase_end() {
        if (!profiling) return;
        HIP_TRY(hipEventRecord(phases.back().e1, stream));
    }
    void end_call() {
        if (!profiling) return;
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& r : phases) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, r.e0, r.e1));
            r.ms = ms;
        }
    }
};
static context_t g_ctx;

struct snarkvm_hip_bases {
    g1_aff_mem_t* d = nullptr;  // tables * n entries: table j at d + j * n holds 2^(256 / tables * j) * P_i
    size_t n = 0;
    int tables = 1;
    int table_bits = 256;  // table j = 2^(table_bits * j) * P
};

// ------------------------------------------------------------------------------------------------
// MSM driver
// ------------------------------------------------------------------------------------------------
static const uint64_t FQ_R[6] = {202099033278250856ull,  5854854902718660529ull, 11492539364873682930ull,
                                 8885205928937022213ull, 5545221690922665192ull, 39800542322357402ull};  // fq.rs:134-141
// Projective::zero() = (0, 1, 0) in Montgomery form (projective.rs:49-54); Fq2 one = (R, 0)
template <class F>
static void write_infinity(void* out) {
    const size_t fb = sizeof(typename F::mem_t);
    memset(out, 0, 3 * fb);
    memcpy((uint8_t*)out + fb, FQ_R, 48);
}

// d_bases: converted device bases; d_scalars: device scalars (32 B each); result written to host `out` (144 B)
template <class F>
static void msm_run(context_t& ctx, const aff_mem_t<F>* d_bases, const uint4* d_scalars, size_t n, void* out, int window_bits,
                    const aff_mem_t<F>* d_bases1 = nullptr, size_t n0 = ~(size_t)0, int scalars_montgomery = 0, int tables = 1,
                    size_t table_stride = 0, int lane_idx = 0, bool sync = true, int table_bits = 0) {
    msm_ws_t& c = ctx.lane[lane_idx];
    // per-phase HIP events only on the synchronous single-MSM path (lane 0)
    auto phase_begin = [&](const char* name) { if (lane_idx == 0 && sync) ctx.phase_begin(name); };
    auto phase_end = [&]() { if (lane_idx == 0 && sync) ctx.phase_end(); };
    if (n0 > n) n0 = n;
    if (n == 0) {
        write_infinity<F>(out);
        return;
    }
    if (n >= ((size_t)1 << 31)) throw hip_failure{hipErrorInvalidValue, "msm: npoints must be < 2^31", __LINE__};
    const msm_plan_t pl = msm_make_plan(n, window_bits, tables, table_bits);
    const bool wide = pl.c > 16;  // u32 digits, three-level sort, two-axis bucket fold
    if ((size_t)pl.Wd * n >= ((size_t)1 << 32)) throw hip_failure{hipErrorInvalidValue, "msm: windows * npoints must be < 2^32", __LINE__};
    if ((size_t)pl.J * n >= ((size_t)1 << 31)) throw hip_failure{hipErrorInvalidValue, "msm: tables * npoints must be < 2^31", __LINE__};
    hipStream_t st = c.stream;
    constexpr unsigned WS_THREADS = sizeof(xyzz_mem_t<F>) > 192 ? 128 : 256;  // window-sum LDS tile <= 48 KiB
    const size_t E_max = (size_t)pl.Wd * n;
    const uint32_t nbt = pl.nbt;

    c.digits.ensure(E_max * (wide ? sizeof(uint32_t) : sizeof(uint16_t)));
    c.scan_tmp.ensure((scan_tmp_elems((size_t)nbt + 1)) * 4);
    c.boff.ensure(((size_t)nbt + 2) * 4);
    c.cnt_a.ensure(((size_t)nbt + 1) * 4);
    c.cnt_b.ensure(((size_t)nbt + 1) * 4);
    c.start_a.ensure(((size_t)nbt + 1) * 4);
    c.start_b.ensure(((size_t)nbt + 1) * 4);
    // thread-count bounds per level: T_(r+1) <= T_r / S2 + nbt + 1 (fixed point ~ nbt * 64/63), plus slack
    const size_t slack = (size_t)nbt / 32 + 64;
    const size_t T0_max = E_max / pl.S + nbt + 1 + slack;
    const size_t T1_max = T0_max / pl.S2 + nbt + 1 + slack;
    c.part_a.ensure(T0_max * sizeof(xyzz_mem_t<F>));
    c.part_b.ensure(T1_max * sizeof(xyzz_mem_t<F>));
    // tail geometry: a wide window is first folded into two windows of 2^fold_m entries (msm_fold_kernel)
    const int K = pl.c - 1;
    const int fold_m = (K + 1) / 2, fold_hb = K - fold_m;
    static const int fold_min_k = getenv("SNARKVM_HIP_FOLD_MIN_K") ? atoi(getenv("SNARKVM_HIP_FOLD_MIN_K")) : 11;
    const bool fold = pl.W == 1 && (wide || K >= fold_min_k);  // also shortens the latency-bound tail of 16-bit windows
    const uint32_t tail_nb = fold ? (1u << fold_m) : pl.nb;
    const int tail_W = fold ? 2 : pl.W;
    const int tail_c = fold ? fold_m : pl.c;
    uint32_t tail_L = fold ? (pl.L < 4 ? pl.L : 4) : pl.L;
    if (tail_L > tail_nb) tail_L = tail_nb;
    while (tail_nb % tail_L) tail_L--;
    const uint32_t J = tail_nb / tail_L;
    c.contrib.ensure((size_t)tail_W * J * sizeof(xyzz_mem_t<F>));
    c.wsum.ensure((size_t)tail_W * sizeof(xyzz_mem_t<F>));
    c.result.ensure(sizeof(jac_mem_t<F>));

    // 1. digits
    phase_begin("msm_digits");
    {
        msm_digit_params_t dp;
        memcpy(dp.bias, pl.bias, sizeof dp.bias);
        dp.c = pl.c;
        dp.W = pl.Wd;
        dp.n = n;
        dp.montgomery = scalars_montgomery;
        size_t blocks = (n + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        if (wide)
            hipLaunchKernelGGL((msm_digits_kernel<uint32_t>), dim3((unsigned)blocks), dim3(256), 0, st, d_scalars, c.digits.as<uint32_t>(), dp);
        else
            hipLaunchKernelGGL((msm_digits_kernel<uint16_t>), dim3((unsigned)blocks), dim3(256), 0, st, d_scalars, c.digits.as<uint16_t>(), dp);
    }
    phase_end();
    static const int sort_mode = getenv("SNARKVM_HIP_SORT") ? atoi(getenv("SNARKVM_HIP_SORT")) : 1;  // 1 = radix partition, 0 = chunk-major
    int rounds = 0;
    if (sort_mode == 1 || wide) {
        // ---- 2.-4. LDS-staged radix partition (msm_sort.cuh) -> bucket-major `sorted` + boff; two levels, three when wide
        msm_radix_params_t rp;
        rp.n = n;
        rp.c = pl.c;
        rp.W = pl.W;
        rp.J = pl.J;
        const int LBL = K < 7 ? K : 7;  // key bits of the last level
        rp.LB = wide ? 14 : LBL;        // bits left below the level-1 key
        rp.HB = K - rp.LB;
        rp.nb = pl.nb;
        rp.tiles_per_row = (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
        rp.TPW = (uint32_t)pl.J * rp.tiles_per_row;
        const uint32_t B1 = 1u << rp.HB;
        const uint32_t nbins = (uint32_t)pl.W * B1;
        const size_t ncounts1 = (size_t)nbins * rp.TPW;
        const size_t tiles1 = (size_t)pl.W * rp.TPW;
        const uint32_t nseg_last = wide ? nbins << 7 : nbins;  // segments feeding the last level
        const size_t tiles2_max = E_max / SORT_TILE + nseg_last + 1;
        c.counts.ensure(ncounts1 * 4);
        c.offsets.ensure(ncounts1 * 4);
        c.scan_tmp.ensure(scan_tmp_elems(ncounts1 > (size_t)nbt + 2 ? ncounts1 : (size_t)nbt + 2) * 4);
        c.rv1.ensure(E_max * 4);
        c.rl1.ensure(E_max * (wide ? 2 : 1));
        c.rcounts2.ensure(tiles2_max * 128 * 4);
        c.roff2.ensure(tiles2_max * 128 * 4);
        c.rbinstart.ensure(((size_t)nbins + 2) * 4);
        c.rntiles.ensure(((size_t)nseg_last + 2) * 4);
        c.rtstart.ensure(((size_t)nseg_last + 2) * 4);
        c.rbsize.ensure(((size_t)nbt + 3) * 4);
        c.sorted.ensure(E_max * 4);
        uint32_t* counts1 = c.counts.as<uint32_t>();
        uint32_t* off1 = c.offsets.as<uint32_t>();
        uint32_t* bsize = c.rbsize.as<uint32_t>();
        uint32_t* d_max = bsize + nbt + 1;
        uint32_t* boffp = c.boff.as<uint32_t>();
        phase_begin("msm_sort_level1");
        if (wide) {
            hipLaunchKernelGGL((radix_hist1_kernel<uint32_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint32_t>(), counts1, rp);
            exclusive_scan_u32(st, counts1, off1, ncounts1, c.scan_tmp.as<uint32_t>());
            hipLaunchKernelGGL((radix_scatter1_kernel<uint32_t, uint16_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint32_t>(),
                               counts1, off1, c.rv1.as<uint32_t>(), c.rl1.as<uint16_t>(), rp);
        } else {
            hipLaunchKernelGGL((radix_hist1_kernel<uint16_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint16_t>(), counts1, rp);
            exclusive_scan_u32(st, counts1, off1, ncounts1, c.scan_tmp.as<uint32_t>());
            hipLaunchKernelGGL((radix_scatter1_kernel<uint16_t, uint8_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint16_t>(),
                               counts1, off1, c.rv1.as<uint32_t>(), c.rl1.as<uint8_t>(), rp);
        }
        hipLaunchKernelGGL(radix_bin_layout_kernel, dim3((nbins + 1 + 255) / 256), dim3(256), 0, st, off1, counts1, ncounts1, c.rbinstart.as<uint32_t>(),
                           nbins, rp.TPW);
        phase_end();
        // one further level: items (v_in, rem_in) grouped in `nseg` segments -> grouped by (segment, next `bits` key bits)
        auto tile_segments = [&](const uint32_t* seg_start, uint32_t nseg) {
            hipLaunchKernelGGL(radix_bin_tiles_kernel, dim3((nseg + 1 + 255) / 256), dim3(256), 0, st, seg_start, c.rntiles.as<uint32_t>(), nseg);
            exclusive_scan_u32(st, c.rntiles.as<uint32_t>(), c.rtstart.as<uint32_t>(), (size_t)nseg + 1, c.scan_tmp.as<uint32_t>());
        };
        // per (segment, key): exclusive prefix of the tile counts + group sizes; few big segments -> one workgroup per segment
        auto colscan = [&](uint32_t* sizes, uint32_t nsegs, int bits, uint32_t* dmax) {
            if (nsegs <= 4096)
                hipLaunchKernelGGL(radix_colscan2_seg_kernel, dim3(nsegs), dim3(1024), 0, st, c.rcounts2.as<uint32_t>(), c.roff2.as<uint32_t>(),
                                   c.rtstart.as<uint32_t>(), sizes, nsegs, bits, dmax);
            else
                hipLaunchKernelGGL(radix_colscan2_kernel, dim3(((nsegs << bits) + 1 + 255) / 256), dim3(256), 0, st, c.rcounts2.as<uint32_t>(),
                                   c.roff2.as<uint32_t>(), c.rtstart.as<uint32_t>(), sizes, nsegs, bits, dmax);
        };
        const uint32_t* seg_start = c.rbinstart.as<uint32_t>();
        uint32_t nseg = nbins;
        const uint32_t* v_in = c.rv1.as<uint32_t>();
        if (wide) {
            phase_begin("msm_sort_level2");
            const uint32_t ngroups = nseg << 7;
            const size_t tmax = E_max / SORT_TILE + nseg + 1;
            c.rv2.ensure(E_max * 4);
            c.rl2.ensure(E_max);
            c.rmid_size.ensure(((size_t)ngroups + 3) * 4);
            c.rmid_boff.ensure(((size_t)ngroups + 3) * 4);
            c.scan_tmp.ensure(scan_tmp_elems((size_t)ngroups + 2) * 4);
            uint32_t* msize = c.rmid_size.as<uint32_t>();
            uint32_t* mboff = c.rmid_boff.as<uint32_t>();
            tile_segments(seg_start, nseg);
            hipLaunchKernelGGL((radix_hist2_kernel<uint16_t>), dim3((unsigned)tmax), dim3(SORT_THREADS), 0, st, c.rl1.as<uint16_t>(), seg_start,
                               c.rtstart.as<uint32_t>(), c.rcounts2.as<uint32_t>(), nseg, 7, 7);
            HIP_TRY(hipMemsetAsync(msize + ngroups + 1, 0, 4, st));
            colscan(msize, nseg, 7, msize + ngroups + 1);
            exclusive_scan_u32(st, msize, mboff, (size_t)ngroups + 1, c.scan_tmp.as<uint32_t>());
            hipLaunchKernelGGL((radix_scatter2_kernel<uint16_t, uint8_t>), dim3((unsigned)tmax), dim3(SORT_THREADS), 0, st, v_in, c.rl1.as<uint16_t>(),
                               seg_start, c.rtstart.as<uint32_t>(), c.rcounts2.as<uint32_t>(), c.roff2.as<uint32_t>(), mboff, c.rv2.as<uint32_t>(),
                               c.rl2.as<uint8_t>(), nseg, 7, 7);
            phase_end();
            seg_start = mboff;
            nseg = ngroups;
            v_in = c.rv2.as<uint32_t>();
        }
        phase_begin(wide ? "msm_sort_level3" : "msm_sort_level2");
        tile_segments(seg_start, nseg);
        const uint8_t* rem_last = wide ? c.rl2.as<uint8_t>() : c.rl1.as<uint8_t>();
        hipLaunchKernelGGL((radix_hist2_kernel<uint8_t>), dim3((unsigned)tiles2_max), dim3(SORT_THREADS), 0, st, rem_last, seg_start,
                           c.rtstart.as<uint32_t>(), c.rcounts2.as<uint32_t>(), nseg, LBL, 0);
        HIP_TRY(hipMemsetAsync(d_max, 0, 4, st));
        colscan(bsize, nseg, LBL, d_max);
        exclusive_scan_u32(st, bsize, boffp, (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
        hipLaunchKernelGGL((radix_scatter2_kernel<uint8_t, uint8_t>), dim3((unsigned)tiles2_max), dim3(SORT_THREADS), 0, st, v_in, rem_last, seg_start,
                           c.rtstart.as<uint32_t>(), c.rcounts2.as<uint32_t>(), c.roff2.as<uint32_t>(), boffp, c.sorted.as<uint32_t>(),
                           (uint8_t*)nullptr, nseg, LBL, 0);
        phase_end();
        static const int seg_mode = getenv("SNARKVM_HIP_SEG") ? atoi(getenv("SNARKVM_HIP_SEG")) : 1;  // 1 = balanced segments (default)
        uint32_t max_bucket = 0;  // the number of reduce rounds follows the largest bucket (4-byte read-back)
        HIP_TRY(hipMemcpyAsync(&max_bucket, d_max, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // ---- 5. accumulate
        phase_begin("msm_accumulate");
        if (seg_mode) {
            // a bucket of s entries is touched by at most (s - 1) / S + 2 segment threads
            // (the tail kernels add up to TAIL_PARTIALS leftover partials per bucket themselves: one reduce round less)
            static const size_t tail_partials = getenv("SNARKVM_HIP_TAILP") ? (size_t)atoi(getenv("SNARKVM_HIP_TAILP")) : 4;
            for (size_t m = max_bucket ? ((size_t)max_bucket - 1) / pl.S + 2 : 0; m > tail_partials; m = (m + pl.S2 - 1) / pl.S2) rounds++;
            hipLaunchKernelGGL(msm_alloc_seg_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, boffp, c.cnt_a.as<uint32_t>(), nbt, pl.S);
            exclusive_scan_u32(st, c.cnt_a.as<uint32_t>(), c.start_a.as<uint32_t>(), (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
            const size_t nthreads = (E_max + pl.S - 1) / pl.S;
            static const int acc_minw = getenv("SNARKVM_HIP_ACC_MINW") ? atoi(getenv("SNARKVM_HIP_ACC_MINW")) : 1;
            // timing experiment only (wrong results): restrict the gather to the first 2^k bases to separate ALU time from HBM gather time
            static const uint32_t dbg_mask = getenv("SNARKVM_HIP_DEBUG_IDX_MASK") ? (uint32_t)strtoul(getenv("SNARKVM_HIP_DEBUG_IDX_MASK"), nullptr, 0) : 0xffffffffu;
            if (acc_minw >= 3 && sizeof(typename F::mem_t) == 48)
                hipLaunchKernelGGL((msm_accumulate_seg_kernel<F, 3>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, d_bases,
                                   d_bases1 ? d_bases1 : d_bases, (uint32_t)n0, c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(),
                                   c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S, (uint32_t)n, table_stride, dbg_mask);
            else
                hipLaunchKernelGGL((msm_accumulate_seg_kernel<F, 1>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, d_bases,
                                   d_bases1 ? d_bases1 : d_bases, (uint32_t)n0, c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(),
                                   c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S, (uint32_t)n, table_stride, dbg_mask);
        } else {
            for (size_t m = ((size_t)max_bucket + pl.S - 1) / pl.S; m > 1; m = (m + pl.S2 - 1) / pl.S2) rounds++;
            hipLaunchKernelGGL(msm_alloc_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, bsize, c.cnt_a.as<uint32_t>(), nbt, pl.S);
            exclusive_scan_u32(st, c.cnt_a.as<uint32_t>(), c.start_a.as<uint32_t>(), (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
            hipLaunchKernelGGL((msm_accumulate_bm_kernel<F>), dim3((unsigned)((T0_max + 255) / 256)), dim3(256), 0, st, d_bases,
                               d_bases1 ? d_bases1 : d_bases, (uint32_t)n0, c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(),
                               c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S, (uint32_t)n, table_stride);
        }
        phase_end();
    } else {
    // 2.-4. counting sort by (window, bucket), chunk-major layout
    {
        const size_t ncounts = (size_t)nbt * pl.nchunks;
        c.counts.ensure(ncounts * 4);
        c.offsets.ensure(ncounts * 4);
        c.scan_tmp.ensure((scan_tmp_elems(ncounts > nbt + 1 ? ncounts : nbt + 1)) * 4);
        c.sorted.ensure((size_t)pl.W * pl.nchunks * pl.chunk * pl.J * 4);
    }
    msm_sort_params_t sp;
    sp.n = n;
    sp.chunk = pl.chunk;
    sp.nchunks = pl.nchunks;
    sp.nb = pl.nb;
    sp.c = pl.c;
    sp.W = pl.W;
    sp.J = pl.J;
    const size_t lds = (size_t)pl.nb * 4;
    uint32_t* rank = c.counts.as<uint32_t>();      // counts, turned into ranks in place
    uint32_t* loc_off = c.offsets.as<uint32_t>();  // offset of each bucket inside its (window, chunk) region
    uint32_t* bsize = c.boff.as<uint32_t>();       // bucket sizes
    phase_begin("msm_histogram");
    hipLaunchKernelGGL(msm_hist_kernel, dim3(pl.nchunks, pl.W), dim3(1024), lds, st, c.digits.as<uint16_t>(), rank, sp);
    phase_end();
    phase_begin("msm_bucket_rank");
    uint32_t* d_max = bsize + nbt + 1;  // one extra word behind the sizes
    HIP_TRY(hipMemsetAsync(d_max, 0, 4, st));
    hipLaunchKernelGGL(msm_bucket_rank_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, rank, bsize, pl.nb, pl.nchunks, nbt, d_max);
    phase_end();
    phase_begin("msm_scatter");
    hipLaunchKernelGGL(msm_locoff_kernel, dim3(pl.nchunks, pl.W), dim3(1024), lds + 4096, st, rank, bsize, loc_off, sp);
    static const int env_passes = getenv("SNARKVM_HIP_SCATTER_PASSES") ? atoi(getenv("SNARKVM_HIP_SCATTER_PASSES")) : 0;
    uint32_t npass = env_passes > 0 ? (uint32_t)env_passes : (pl.nb >= 8192 ? 2u : 1u);  // measured: 1: 4.13, 2: 3.92, 4: 5.21, 8: 4.94 ms (2^24)
    while (pl.nb % npass) npass--;
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(pl.nchunks, pl.W, npass), dim3(1024), lds / npass, st, c.digits.as<uint16_t>(), loc_off,
                       c.sorted.as<uint32_t>(), sp, npass);
    phase_end();
    // the largest bucket decides how many reduce rounds are needed (4-byte read-back; worst-case sizing would run
    // up to 7 mostly idle rounds)
    uint32_t max_bucket = 0;
    HIP_TRY(hipMemcpyAsync(&max_bucket, d_max, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (size_t m = ((size_t)max_bucket + pl.S - 1) / pl.S; m > 1; m = (m + pl.S2 - 1) / pl.S2) rounds++;
    // 5. accumulate
    phase_begin("msm_accumulate");
    hipLaunchKernelGGL(msm_alloc_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, bsize, c.cnt_a.as<uint32_t>(), nbt, pl.S);
    exclusive_scan_u32(st, c.cnt_a.as<uint32_t>(), c.start_a.as<uint32_t>(), (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
    {
        static const int acc_waves = getenv("SNARKVM_HIP_ACC_WAVES") ? atoi(getenv("SNARKVM_HIP_ACC_WAVES")) : 3;
        const dim3 grid((unsigned)((T0_max + 255) / 256));
        // timing experiment only (wrong results): restrict the gather to the first 2^k bases to separate ALU time from HBM gather time
        static const uint32_t dbg_mask = getenv("SNARKVM_HIP_DEBUG_IDX_MASK") ? (uint32_t)strtoul(getenv("SNARKVM_HIP_DEBUG_IDX_MASK"), nullptr, 0) : 0xffffffffu;
        // variant: 1 = plain loop (default), 2 = software-prefetched gather, 4 = force <= 128 VGPRs (spills; measured slower)
        if (acc_waves >= 4 && sizeof(typename F::mem_t) == 48)
            hipLaunchKernelGGL((msm_accumulate_kernel<F, 4>), grid, dim3(256), 0, st, d_bases, d_bases1 ? d_bases1 : d_bases, (uint32_t)n0,
                               c.sorted.as<uint32_t>(), rank, loc_off, bsize, c.start_a.as<uint32_t>(), c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S,
                               pl.nb, pl.nchunks, pl.chunk * (uint32_t)pl.J, (uint32_t)n, table_stride, dbg_mask);
        else if (acc_waves == 2 && sizeof(typename F::mem_t) == 48)
            hipLaunchKernelGGL((msm_accumulate_kernel<F, 2>), grid, dim3(256), 0, st, d_bases, d_bases1 ? d_bases1 : d_bases, (uint32_t)n0,
                               c.sorted.as<uint32_t>(), rank, loc_off, bsize, c.start_a.as<uint32_t>(), c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S,
                               pl.nb, pl.nchunks, pl.chunk * (uint32_t)pl.J, (uint32_t)n, table_stride, dbg_mask);
        else
            hipLaunchKernelGGL((msm_accumulate_kernel<F, 1>), grid, dim3(256), 0, st, d_bases, d_bases1 ? d_bases1 : d_bases, (uint32_t)n0,
                               c.sorted.as<uint32_t>(), rank, loc_off, bsize, c.start_a.as<uint32_t>(), c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S,
                               pl.nb, pl.nchunks, pl.chunk * (uint32_t)pl.J, (uint32_t)n, table_stride, dbg_mask);
    }
    phase_end();
    }
    // 6. reduce rounds: (cnt_a, start_a, part_a) -> (cnt_b, start_b, part_b) -> ...
    phase_begin("msm_reduce_partials");
    uint32_t *cnt_in = c.cnt_a.as<uint32_t>(), *cnt_out = c.cnt_b.as<uint32_t>();
    uint32_t *start_in = c.start_a.as<uint32_t>(), *start_out = c.start_b.as<uint32_t>();
    xyzz_mem_t<F> *pin = c.part_a.as<xyzz_mem_t<F>>(), *pout = c.part_b.as<xyzz_mem_t<F>>();
    size_t T_in_max = T0_max;
    for (int r = 0; r < rounds; r++) {
        size_t T_out_max = T_in_max / pl.S2 + nbt + 1;
        if (T_out_max > T1_max) T_out_max = T1_max;  // both ping-pong buffers hold >= T1_max partials
        hipLaunchKernelGGL(msm_alloc_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, cnt_in, cnt_out, nbt, pl.S2);
        exclusive_scan_u32(st, cnt_out, start_out, (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
        hipLaunchKernelGGL((msm_reduce_kernel<F>), dim3((unsigned)((T_out_max + 255) / 256)), dim3(256), 0, st, pin, start_in, cnt_in, start_out, pout,
                           nbt, pl.S2);
        std::swap(cnt_in, cnt_out);
        std::swap(start_in, start_out);
        std::swap(pin, pout);
        T_in_max = T_out_max;
    }
    phase_end();
    // 7.-9. bucket reduction, window sums, Horner
    phase_begin("msm_bucket_reduce");
    const xyzz_mem_t<F>* tail_sums = pin;
    const uint32_t *tail_start = start_in, *tail_cnt = cnt_in;
    if (fold) {
        const uint32_t slots = 2u << fold_m;
        c.fold_sums.ensure((size_t)slots * sizeof(xyzz_mem_t<F>));
        c.fold_idx.ensure((size_t)slots * 8);
        uint32_t* fstart = c.fold_idx.as<uint32_t>();
        uint32_t* fcnt = fstart + slots;
        static const int fold_wg = getenv("SNARKVM_HIP_FOLD_WG") ? atoi(getenv("SNARKVM_HIP_FOLD_WG")) : 64;  // 64: one wave per output
        if (fold_wg == 64)
            hipLaunchKernelGGL((msm_fold_wave_kernel<F>), dim3((1u << fold_m) + (1u << fold_hb)), dim3(64), 0, st, pin, start_in, cnt_in,
                               c.fold_sums.as<xyzz_mem_t<F>>(), fstart, fcnt, fold_m, fold_hb);
        else
            hipLaunchKernelGGL((msm_fold_kernel<F>), dim3((1u << fold_m) + (1u << fold_hb)), dim3(WS_THREADS), WS_THREADS * sizeof(xyzz_mem_t<F>), st,
                               pin, start_in, cnt_in, c.fold_sums.as<xyzz_mem_t<F>>(), fstart, fcnt, fold_m, fold_hb);
        tail_sums = c.fold_sums.as<xyzz_mem_t<F>>();
        tail_start = fstart;
        tail_cnt = fcnt;
    }
    const uint32_t total_threads = (uint32_t)tail_W * J;
    hipLaunchKernelGGL((msm_bucket_reduce_kernel<F>), dim3((total_threads + 255) / 256), dim3(256), 0, st, tail_sums, tail_start, tail_cnt,
                       c.contrib.as<xyzz_mem_t<F>>(), tail_nb, tail_L, total_threads);
    hipLaunchKernelGGL((msm_window_sum_kernel<F>), dim3(tail_W), dim3(WS_THREADS), WS_THREADS * sizeof(xyzz_mem_t<F>), st, c.contrib.as<xyzz_mem_t<F>>(), c.wsum.as<xyzz_mem_t<F>>(), J);
    phase_end();
    phase_begin("msm_final_horner");
    hipLaunchKernelGGL((msm_final_kernel<F>), dim3(1), dim3(64), 0, st, c.wsum.as<xyzz_mem_t<F>>(), c.result.as<jac_mem_t<F>>(), tail_W, tail_c);
    phase_end();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c.result.p, sizeof(jac_mem_t<F>), hipMemcpyDeviceToHost, st));  // `out` is pinned when !sync
    if (sync) HIP_TRY(hipStreamSynchronize(st));
}

template <class F>
static void convert_bases(context_t& c, const uint8_t* d_in, size_t stride, size_t n, aff_mem_t<F>* d_out) {
    if (!n) return;
    hipLaunchKernelGGL((convert_bases_kernel<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, d_in, stride, n, d_out);
    HIP_TRY(hipGetLastError());
}

// Plain FFI MSM (host pointers): stage, convert, run.  G1: F = fq_t (stride >= 104), G2: F = fq2_t (stride >= 200).
template <class F>
static void msm_host(context_t& c, void* out, const void* points, size_t npoints, const void* scalars, size_t stride) {
    if (npoints == 0) {
        write_infinity<F>(out);
        return;
    }
    const size_t min_stride = 2 * sizeof(typename F::mem_t) + 8;
    if (stride < min_stride || (stride & 7)) throw hip_failure{hipErrorInvalidValue, "msm: bad ffi_affine_sz for this curve", __LINE__};
    const size_t aff_bytes = (npoints * sizeof(aff_mem_t<F>) + 255) & ~(size_t)255;
    c.bases_tmp.ensure(aff_bytes + npoints * stride);
    c.scalars_tmp.ensure(npoints * 32);
    uint8_t* raw = c.bases_tmp.as<uint8_t>() + aff_bytes;
    c.phase_begin("msm_h2d");
    HIP_TRY(hipMemcpyAsync(raw, points, npoints * stride, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(c.scalars_tmp.p, scalars, npoints * 32, hipMemcpyHostToDevice, c.stream));
    c.phase_end();
    c.phase_begin("msm_convert_bases");
    convert_bases<F>(c, raw, stride, npoints, c.bases_tmp.as<aff_mem_t<F>>());
    c.phase_end();
    msm_run<F>(c, c.bases_tmp.as<aff_mem_t<F>>(), c.scalars_tmp.as<uint4>(), npoints, out, 0);
}

// ------------------------------------------------------------------------------------------------
// exported functions
// ------------------------------------------------------------------------------------------------
#define API_BEGIN                                  \
    std::lock_guard<std::mutex> _lk(g_ctx.mu);     \
    try {                                          \
        g_ctx.init();                              \
        g_ctx.begin_call();
#define API_END                                    \
    g_ctx.end_call();                              \
    return ok();                                   \
    }                                              \
    catch (const hip_failure& f) {                 \
        return from_failure(f);                    \
    }                                              \
    catch (const std::exception& e) {              \
        return fail(1, std::string("snarkvm_hip: ") + e.what()); \
    }                                              \
    catch (...) {                                  \
        return fail(1, "snarkvm_hip: unknown error"); \
    }

// ---- test-hook helpers (C++ linkage)
template <class F>
SV_HD void field_op(int op, const uint32_t* a, const uint32_t* b, uint32_t* out) {
    // operands are memory-form Montgomery residues: convert to internal, operate, convert back
    F x = F::unpack(a).from_mem_mont();
    F y = F::unpack(b).from_mem_mont();
    F r;
    switch (op) {
        case 0: r = x + y; break;
        case 1: r = x - y; break;
        case 2: r = x * y; break;
        case 3: r = x.sqr(); break;
        case 4: r = x.inverse(); break;
        case 5: r = x.neg(); break;
        case 6: r = F::unpack(a).int_to_mont(); break;                  // from_bigint: integer -> Montgomery
        case 7: (x.mont_to_int()).pack(out); return;                    // to_bigint: Montgomery -> integer
        case 9: r = F::diff_of_products(x, y, y, x + y); break;  // x*y - y*(x+y) with one reduction
        case 8: {  // lazy-arithmetic chain used by the NTT butterflies (Fr only): ((a + b) - b + 2r) * b == a * b
            if (F::N != 9) { r = x * y; break; }
            uint32_t kp[F::N];
            F::mod_shl(kp, 1);
            F t = F::add_lazy(x, y);         // < 2r
            t = F::sub_lazy(t, y, kp);       // < 4r
            r = t.mul_lazy(y).reduce_lazy();
            break;
        }
        default: r = F::zero();
    }
    r.to_mem_mont().pack(out);
}
__global__ void devtest_field_kernel(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (field == 0)
        field_op<fr_t>(op, a + 8 * i, b + 8 * i, out + 8 * i);
    else
        field_op<fq_t>(op, a + 12 * i, b + 12 * i, out + 12 * i);
}

extern "C" {

int snarkvm_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int snarkvm_hip_batch_lanes(size_t npoints) { return context_t::batch_lanes(npoints); }
RustError snarkvm_hip_set_device(int device) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    if (g_ctx.ready && g_ctx.device != device) return fail(1, "snarkvm_hip_set_device: context already initialised on another device");
    g_ctx.device = device;
    return ok();
}
void snarkvm_hip_set_profiling(int enabled) {
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    g_ctx.profiling = enabled != 0;
}
int snarkvm_hip_get_phase_count(void) { return (int)g_ctx.phases.size(); }
const char* snarkvm_hip_get_phase_name(int i) { return (i >= 0 && i < (int)g_ctx.phases.size()) ? g_ctx.phases[i].name : ""; }
double snarkvm_hip_get_phase_ms(int i) { return (i >= 0 && i < (int)g_ctx.phases.size()) ? g_ctx.phases[i].ms : 0.0; }

RustError snarkvm_hip_synchronize(void) {
    API_BEGIN
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    API_END
}

// ---- NTT -------------------------------------------------------------------------------------
static void check_ntt_args(uint32_t lg, int order, int dir, int type) {
    if (lg > (uint32_t)NTT_LG_MAX) throw hip_failure{hipErrorMemoryAllocation, "ntt: lg_domain_size > 24 is not supported by this backend", __LINE__};
    if (order < 0 || order > 3 || dir < 0 || dir > 1 || type < 0 || type > 1) throw hip_failure{hipErrorInvalidValue, "ntt: bad enum value", __LINE__};
}
RustError snarkvm_ntt(void* inout, uint32_t lg, enum NTTInputOutputOrder order, enum NTTDirection dir, enum NTTType type) {
    API_BEGIN
    check_ntt_args(lg, (int)order, (int)dir, (int)type);
    const size_t bytes = sizeof(fr_mem_t) << lg;
    g_ctx.ntt_data.ensure(bytes);
    g_ctx.ntt_scratch.ensure(bytes);
    g_ctx.phase_begin("ntt_h2d");
    HIP_TRY(hipMemcpyAsync(g_ctx.ntt_data.p, inout, bytes, hipMemcpyHostToDevice, g_ctx.stream));
    g_ctx.phase_end();
    g_ctx.phase_begin("ntt_kernels");
    ntt_run(g_ctx.stream, g_ctx.tb, g_ctx.ntt_data.as<fr_mem_t>(), g_ctx.ntt_scratch.as<fr_mem_t>(), (int)lg, (int)order, (int)dir, (int)type);
    g_ctx.phase_end();
    HIP_TRY(hipGetLastError());
    g_ctx.phase_begin("ntt_d2h");
    HIP_TRY(hipMemcpyAsync(inout, g_ctx.ntt_data.p, bytes, hipMemcpyDeviceToHost, g_ctx.stream));
    g_ctx.phase_end();
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    API_END
}
RustError snarkvm_hip_ntt_device(void* d_inout, uint32_t lg, int order, int dir, int type) {
    API_BEGIN
    check_ntt_args(lg, order, dir, type);
    g_ctx.ntt_scratch.ensure(sizeof(fr_mem_t) << lg);
    g_ctx.phase_begin("ntt_kernels");
    ntt_run(g_ctx.stream, g_ctx.tb, (fr_mem_t*)d_inout, g_ctx.ntt_scratch.as<fr_mem_t>(), (int)lg, order, dir, type);
    g_ctx.phase_end();
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    API_END
}

// ---- polymul -----------------------------------------------------------------------------------
RustError snarkvm_polymul(void* out, size_t pcount, const void* polynomials, const void* plens, size_t ecount, const void* evaluations,
                          const void* elens, uint32_t lg) {
    // corner cases of snarkvm.cu:196-210 first (no device needed for the copy)
    const fr_mem_t* const* polys = (const fr_mem_t* const*)polynomials;
    const fr_mem_t* const* evals = (const fr_mem_t* const*)evaluations;
    const size_t* pl = (const size_t*)plens;
    const size_t* el = (const size_t*)elens;
    if (pcount + ecount == 0) return ok();
    if (pcount + ecount == 1 && pcount == 1) {
        memcpy(out, polys[0], sizeof(fr_mem_t) * pl[0]);
        return ok();
    }
    API_BEGIN
    check_ntt_args(lg, 0, 0, 0);
    const size_t n = (size_t)1 << lg;
    const size_t bytes = sizeof(fr_mem_t) * n;
    for (size_t k = 0; k < pcount; k++)
        if (pl[k] > n) throw hip_failure{hipErrorInvalidValue, "polymul: polynomial longer than the domain", __LINE__};
    for (size_t k = 0; k < ecount; k++)
        if (el[k] != n) throw hip_failure{hipErrorInvalidValue, "polymul: evaluation vector length != domain size", __LINE__};
    g_ctx.ntt_data.ensure(bytes);
    g_ctx.ntt_scratch.ensure(bytes);
    g_ctx.ntt_acc.ensure(bytes);
    hipStream_t st = g_ctx.stream;
    fr_mem_t* data = g_ctx.ntt_data.as<fr_mem_t>();
    fr_mem_t* acc = g_ctx.ntt_acc.as<fr_mem_t>();
    if (pcount + ecount == 1) {  // a single evaluation vector: zero-pad + inverse NTT (snarkvm.cu:203-208)
        HIP_TRY(hipMemsetAsync(data, 0, bytes, st));
        HIP_TRY(hipMemcpyAsync(data, evals[0], sizeof(fr_mem_t) * el[0], hipMemcpyHostToDevice, st));
        ntt_run(st, g_ctx.tb, data, g_ctx.ntt_scratch.as<fr_mem_t>(), (int)lg, NTT_NN, NTT_INVERSE, NTT_STANDARD);
        HIP_TRY(hipMemcpyAsync(out, data, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    } else {
        const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        for (size_t k = 0; k < pcount + ecount; k++) {
            fr_mem_t* dst = (k == 0) ? acc : data;
            if (k < pcount) {
                HIP_TRY(hipMemcpyAsync(dst, polys[k], sizeof(fr_mem_t) * pl[k], hipMemcpyHostToDevice, st));
                if (pl[k] < n) HIP_TRY(hipMemsetAsync(dst + pl[k], 0, sizeof(fr_mem_t) * (n - pl[k]), st));
                ntt_run(st, g_ctx.tb, dst, g_ctx.ntt_scratch.as<fr_mem_t>(), (int)lg, NTT_NN, NTT_FORWARD, NTT_STANDARD);
            } else {
                HIP_TRY(hipMemcpyAsync(dst, evals[k - pcount], bytes, hipMemcpyHostToDevice, st));
            }
            if (k > 0) hipLaunchKernelGGL(fr_pointwise_mul_kernel, dim3(blocks), dim3(256), 0, st, acc, acc, data, n, 1);
        }
        ntt_run(st, g_ctx.tb, acc, g_ctx.ntt_scratch.as<fr_mem_t>(), (int)lg, NTT_NN, NTT_INVERSE, NTT_STANDARD);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, acc, bytes, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    API_END
}

// ---- MSM ---------------------------------------------------------------------------------------
// ---- optional base cache behind the unmodified FFI -------------------------------------------------
// The reference's callers pass slices of ONE long-lived vector (`powers_of_beta_g[lz .. lz + len]`, kzg10/mod.rs:117-119)
// and its GPU path re-uploads them on every call.  With SNARKVM_HIP_BASE_CACHE=<tables> (1, 2, 4, 8 or 16; unset = off) a
// call whose base range lies inside a range seen before reuses the device copy (with `tables` precomputed multiples):
// no upload, no conversion, no Horner chain.  A hit is verified against raw copies of every CACHE_STEP-th point of the
// slice; a mismatch drops the entry.  Host pointers are only compared, never dereferenced outside the call that passed
// them.  At most CACHE_MAX ranges are kept (least recently used goes first).
static void register_bases_impl(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device, int tables,
                                int table_bits);
struct base_cache_entry {
    const uint8_t* host = nullptr;
    size_t n = 0, stride = 0;
    snarkvm_hip_bases* h = nullptr;
    std::vector<uint8_t> samples;  // 97 bytes (x, y, infinity) of points 0, CACHE_STEP, 2 * CACHE_STEP, ...
    uint64_t last_use = 0;
};
static constexpr size_t CACHE_STEP = 4096, CACHE_MAX = 4;
static std::vector<base_cache_entry> g_base_cache;
static uint64_t g_cache_tick = 0;
static int base_cache_tables() {
    static const int t = getenv("SNARKVM_HIP_BASE_CACHE") ? atoi(getenv("SNARKVM_HIP_BASE_CACHE")) : 0;
    return (t == 1 || t == 2 || t == 4 || t == 8 || t == 16) ? t : 0;
}
static void base_cache_drop(size_t i) {
    if (g_base_cache[i].h) {
        if (g_base_cache[i].h->d) (void)hipFree(g_base_cache[i].h->d);
        delete g_base_cache[i].h;
    }
    g_base_cache.erase(g_base_cache.begin() + (long)i);
}
// registered handle + offset covering [points, points + npoints * stride), registering the range on a miss
static const snarkvm_hip_bases* base_cache_lookup(const void* points, size_t npoints, size_t stride, size_t& offset) {
    const uint8_t* p = (const uint8_t*)points;
    for (size_t i = 0; i < g_base_cache.size(); i++) {
        base_cache_entry& e = g_base_cache[i];
        if (e.stride != stride || p < e.host || p + npoints * stride > e.host + e.n * stride || (size_t)(p - e.host) % stride) continue;
        const size_t off = (size_t)(p - e.host) / stride;
        bool same = true;
        for (size_t k = (off + CACHE_STEP - 1) / CACHE_STEP; k * CACHE_STEP < off + npoints && same; k++)
            same = memcmp(&e.samples[k * 97], e.host + k * CACHE_STEP * stride, 97) == 0;
        if (!same) {  // the memory behind a cached range changed: forget it
            base_cache_drop(i);
            break;
        }
        e.last_use = ++g_cache_tick;
        offset = off;
        return e.h;
    }
    while (g_base_cache.size() >= CACHE_MAX) {
        size_t lru = 0;
        for (size_t i = 1; i < g_base_cache.size(); i++)
            if (g_base_cache[i].last_use < g_base_cache[lru].last_use) lru = i;
        base_cache_drop(lru);
    }
    // a slice of a bigger vector may come first: ranges that the new one contains are superseded
    for (size_t i = g_base_cache.size(); i-- > 0;)
        if (g_base_cache[i].stride == stride && g_base_cache[i].host >= p && g_base_cache[i].host + g_base_cache[i].n * stride <= p + npoints * stride)
            base_cache_drop(i);
    base_cache_entry e;
    e.host = p;
    e.n = npoints;
    e.stride = stride;
    register_bases_impl(&e.h, points, npoints, stride, 0, base_cache_tables(), 0);
    for (size_t k = 0; k * CACHE_STEP < npoints; k++) e.samples.insert(e.samples.end(), p + k * CACHE_STEP * stride, p + k * CACHE_STEP * stride + 97);
    e.last_use = ++g_cache_tick;
    g_base_cache.push_back(e);
    offset = 0;
    return g_base_cache.back().h;
}

RustError snarkvm_msm(void* out, const void* points, size_t npoints, const void* scalars, size_t ffi_affine_sz) {
    API_BEGIN
    if (base_cache_tables() && npoints > 1024 && ffi_affine_sz >= 104 && !(ffi_affine_sz & 7)) {
        size_t offset = 0;
        const snarkvm_hip_bases* h = base_cache_lookup(points, npoints, ffi_affine_sz, offset);
        g_ctx.scalars_tmp.ensure(npoints * 32);
        g_ctx.phase_begin("msm_h2d");
        HIP_TRY(hipMemcpyAsync(g_ctx.scalars_tmp.p, scalars, npoints * 32, hipMemcpyHostToDevice, g_ctx.stream));
        g_ctx.phase_end();
        msm_run<fq_t>(g_ctx, h->d + offset, g_ctx.scalars_tmp.as<uint4>(), npoints, out, 0, nullptr, ~(size_t)0, 0, h->tables, h->n, 0, true, h->table_bits);
    } else {
        msm_host<fq_t>(g_ctx, out, points, npoints, scalars, ffi_affine_sz);
    }
    API_END
}
RustError snarkvm_hip_msm_g2(void* out, const void* points, size_t npoints, const void* scalars, size_t ffi_affine_sz) {
    API_BEGIN
#ifdef SV_NO_G2  // development builds only (python -m snarkvm_amd.build --fast): skips the Fq2 kernel instantiations
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    msm_host<fq2_t>(g_ctx, out, points, npoints, scalars, ffi_affine_sz);
#endif
    API_END
}

// ---- registered G2 bases (extension): same engine over fq2_t, precomputed tables remove the serial Horner chain that
// dominates a one-shot G2 MSM (240 Fq2 doublings, ~10 ms)
struct snarkvm_hip_bases_g2 {
    aff_mem_t<fq2_t>* d = nullptr;
    size_t n = 0;
    int tables = 1;
    int table_bits = 256;
};
static void check_tables(int tables, int table_bits, const char* who);
RustError snarkvm_hip_register_bases_g2(snarkvm_hip_bases_g2_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int tables,
                                        int window_bits) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!handle || (npoints && !points)) throw hip_failure{hipErrorInvalidValue, "register_bases_g2: null argument", __LINE__};
    if (ffi_affine_sz < 200 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "register_bases_g2: bad stride", __LINE__};
    check_tables(tables, window_bits, "register_bases_g2");
    snarkvm_hip_bases_g2* h = new snarkvm_hip_bases_g2();
    h->n = npoints;
    h->tables = tables;
    h->table_bits = window_bits ? window_bits : 256 / tables;
    if (npoints) {
        try {
            HIP_TRY(hipMalloc((void**)&h->d, (size_t)tables * npoints * sizeof(aff_mem_t<fq2_t>)));
            g_ctx.bases_tmp.ensure(npoints * ffi_affine_sz);
            HIP_TRY(hipMemcpyAsync(g_ctx.bases_tmp.p, points, npoints * ffi_affine_sz, hipMemcpyHostToDevice, g_ctx.stream));
            convert_bases<fq2_t>(g_ctx, g_ctx.bases_tmp.as<uint8_t>(), ffi_affine_sz, npoints, h->d);
            for (int j = 1; j < tables; j++)
                hipLaunchKernelGGL((precompute_table_kernel<fq2_t>), dim3((unsigned)((npoints + 255) / 256)), dim3(256), 0, g_ctx.stream,
                                   h->d + (size_t)(j - 1) * npoints, h->d + (size_t)j * npoints, npoints, h->table_bits);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        } catch (...) {
            if (h->d) (void)hipFree(h->d);
            delete h;
            throw;
        }
    }
    *handle = h;
#endif
    API_END
}
void snarkvm_hip_free_bases_g2(snarkvm_hip_bases_g2_t* h) {
    if (!h) return;
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    if (h->d) (void)hipFree(h->d);
    delete h;
}
RustError snarkvm_hip_msm_g2_registered(void* out, const snarkvm_hip_bases_g2_t* h, size_t offset, size_t npoints, const void* scalars,
                                        int scalars_on_device, int window_bits) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!h || offset + npoints > h->n) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: range exceeds the registered bases", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: window_bits must be 0 or 2..23", __LINE__};
    const uint4* d_sc = (const uint4*)scalars;
    if (!scalars_on_device && npoints) {
        g_ctx.scalars_tmp.ensure(npoints * 32);
        g_ctx.phase_begin("msm_h2d");
        HIP_TRY(hipMemcpyAsync(g_ctx.scalars_tmp.p, scalars, npoints * 32, hipMemcpyHostToDevice, g_ctx.stream));
        g_ctx.phase_end();
        d_sc = g_ctx.scalars_tmp.as<uint4>();
    }
    msm_run<fq2_t>(g_ctx, h->d + offset, d_sc, npoints, out, window_bits, nullptr, ~(size_t)0, 0, h->tables, h->n, 0, true, h->table_bits);
#endif
    API_END
}

// tables 1 .. J-1 of a registered base vector: table j = 2^(256 / J) * table j-1
static void precompute_tables(snarkvm_hip_bases* h) {
    for (int j = 1; j < h->tables; j++)
        hipLaunchKernelGGL((precompute_table_kernel<fq_t>), dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, g_ctx.stream,
                           h->d + (size_t)(j - 1) * h->n, h->d + (size_t)j * h->n, h->n, h->table_bits);
    HIP_TRY(hipGetLastError());
}
static void check_tables(int tables, int table_bits, const char* who) {
    const bool legacy = table_bits == 0 && (tables == 1 || tables == 2 || tables == 4 || tables == 8 || tables == 16);
    const bool windowed = table_bits >= 2 && table_bits <= MSM_C_MAX && tables >= 1 && tables <= 127 && tables * table_bits >= 254;
    if (!legacy && !windowed)
        throw std::runtime_error(std::string(who) + ": tables must be 1, 2, 4, 8 or 16, or tables * window_bits >= 254 with window_bits in 2..23");
}
static void register_bases_impl(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device, int tables,
                                int table_bits) {
    if (!handle) throw hip_failure{hipErrorInvalidValue, "register_bases: null handle", __LINE__};
    if (ffi_affine_sz < 104 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "register_bases: bad stride", __LINE__};
    check_tables(tables, table_bits, "register_bases");
    snarkvm_hip_bases* h = new snarkvm_hip_bases();
    h->n = npoints;
    h->tables = tables;
    h->table_bits = table_bits ? table_bits : 256 / tables;
    if (npoints) {
        HIP_TRY(hipMalloc((void**)&h->d, (size_t)tables * npoints * sizeof(g1_aff_mem_t)));
        const uint8_t* src = (const uint8_t*)points;
        if (!on_device) {
            g_ctx.bases_tmp.ensure(npoints * ffi_affine_sz);
            HIP_TRY(hipMemcpyAsync(g_ctx.bases_tmp.p, points, npoints * ffi_affine_sz, hipMemcpyHostToDevice, g_ctx.stream));
            src = g_ctx.bases_tmp.as<uint8_t>();
        }
        convert_bases<fq_t>(g_ctx, src, ffi_affine_sz, npoints, h->d);
        precompute_tables(h);
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    *handle = h;
}

// ---- canonical (de)serialisation of G1 points (serde.cuh) ---------------------------------------
static void serde_throw_on_status(uint32_t st, const char* who) {
    if (!st) return;
    std::string m = std::string(who) + ":";
    if (st & SERDE_BAD_FLAGS) m += " UnexpectedFlags (both flag bits set)";
    if (st & SERDE_NOT_CANONICAL) m += " coordinate >= q";
    if (st & SERDE_NOT_ON_CURVE) m += " InvalidData (point not on the curve)";
    if (st & SERDE_NOT_IN_SUBGROUP) m += " InvalidData (point not in the prime-order subgroup)";
    throw std::runtime_error(m);  // SerializationError: surfaces as RustError code 1 with this message
}
// bytes (host) -> native base slots and / or Rust-layout records (both device); returns the SERDE_* status bits
static uint32_t g1_deserialize_run(const void* bytes, size_t n, int compressed, int validate, g1_aff_mem_t* d_native, uint8_t* d_rust) {
    const size_t psz = compressed ? 48 : 96;
    g_ctx.bases_tmp.ensure(n * psz);
    g_ctx.serde_status.ensure(4);
    HIP_TRY(hipMemcpyAsync(g_ctx.bases_tmp.p, bytes, n * psz, hipMemcpyHostToDevice, g_ctx.stream));
    HIP_TRY(hipMemsetAsync(g_ctx.serde_status.p, 0, 4, g_ctx.stream));
    hipLaunchKernelGGL(g1_deserialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g_ctx.stream, g_ctx.bases_tmp.as<uint8_t>(), n, compressed,
                       validate, d_native, d_rust, g_ctx.serde_status.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    uint32_t st = 0;
    HIP_TRY(hipMemcpyAsync(&st, g_ctx.serde_status.p, 4, hipMemcpyDeviceToHost, g_ctx.stream));
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    return st;
}
RustError snarkvm_hip_register_bases_serialized(snarkvm_hip_bases_t** handle, const void* bytes, size_t npoints, int compressed, int validate,
                                                int tables) {
    API_BEGIN
    if (!handle || (npoints && !bytes)) throw hip_failure{hipErrorInvalidValue, "register_bases_serialized: null argument", __LINE__};
    check_tables(tables, 0, "register_bases_serialized");
    snarkvm_hip_bases* h = new snarkvm_hip_bases();
    h->n = npoints;
    h->tables = tables;
    h->table_bits = 256 / tables;
    if (npoints) {
        try {
            HIP_TRY(hipMalloc((void**)&h->d, (size_t)tables * npoints * sizeof(g1_aff_mem_t)));
            serde_throw_on_status(g1_deserialize_run(bytes, npoints, compressed, validate, h->d, nullptr), "register_bases_serialized");
            precompute_tables(h);
            HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        } catch (...) {
            if (h->d) (void)hipFree(h->d);
            delete h;
            throw;
        }
    }
    *handle = h;
    API_END
}
RustError snarkvm_hip_g1_deserialize(void* out_affine, const void* bytes, size_t n, int compressed, int validate) {
    API_BEGIN
    if (n) {
        if (!out_affine || !bytes) throw hip_failure{hipErrorInvalidValue, "g1_deserialize: null argument", __LINE__};
        g_ctx.poly[0].ensure(n * 104);
        const uint32_t st = g1_deserialize_run(bytes, n, compressed, validate, nullptr, g_ctx.poly[0].as<uint8_t>());
        serde_throw_on_status(st, "g1_deserialize");
        HIP_TRY(hipMemcpyAsync(out_affine, g_ctx.poly[0].p, n * 104, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}
RustError snarkvm_hip_g2_deserialize(void* out_affine, const void* bytes, size_t n, int validate) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (n) {
        if (!out_affine || !bytes) throw hip_failure{hipErrorInvalidValue, "g2_deserialize: null argument", __LINE__};
        g_ctx.bases_tmp.ensure(n * 192);
        g_ctx.poly[0].ensure(n * 200);
        g_ctx.serde_status.ensure(4);
        HIP_TRY(hipMemcpyAsync(g_ctx.bases_tmp.p, bytes, n * 192, hipMemcpyHostToDevice, g_ctx.stream));
        HIP_TRY(hipMemsetAsync(g_ctx.serde_status.p, 0, 4, g_ctx.stream));
        hipLaunchKernelGGL(g2_deserialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g_ctx.stream, g_ctx.bases_tmp.as<uint8_t>(), n, validate,
                           g_ctx.poly[0].as<uint8_t>(), g_ctx.serde_status.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        uint32_t st = 0;
        HIP_TRY(hipMemcpyAsync(&st, g_ctx.serde_status.p, 4, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipMemcpyAsync(out_affine, g_ctx.poly[0].p, n * 200, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        serde_throw_on_status(st, "g2_deserialize");
    }
#endif
    API_END
}
RustError snarkvm_hip_g2_serialize(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (n) {
        if (!out_bytes || !affine) throw hip_failure{hipErrorInvalidValue, "g2_serialize: null argument", __LINE__};
        if (ffi_affine_sz < 200 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "g2_serialize: bad stride", __LINE__};
        g_ctx.bases_tmp.ensure(n * ffi_affine_sz);
        g_ctx.poly[0].ensure(n * 192);
        HIP_TRY(hipMemcpyAsync(g_ctx.bases_tmp.p, affine, n * ffi_affine_sz, hipMemcpyHostToDevice, g_ctx.stream));
        hipLaunchKernelGGL(g2_serialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g_ctx.stream, g_ctx.bases_tmp.as<uint8_t>(), ffi_affine_sz, n,
                           g_ctx.poly[0].as<uint8_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_bytes, g_ctx.poly[0].p, n * 192, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
#endif
    API_END
}
RustError snarkvm_hip_g1_serialize(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz, int compressed) {
    API_BEGIN
    if (n) {
        if (!out_bytes || !affine) throw hip_failure{hipErrorInvalidValue, "g1_serialize: null argument", __LINE__};
        if (ffi_affine_sz < 104 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "g1_serialize: bad stride", __LINE__};
        const size_t psz = compressed ? 48 : 96;
        g_ctx.bases_tmp.ensure(n * ffi_affine_sz);
        g_ctx.poly[0].ensure(n * psz);
        HIP_TRY(hipMemcpyAsync(g_ctx.bases_tmp.p, affine, n * ffi_affine_sz, hipMemcpyHostToDevice, g_ctx.stream));
        hipLaunchKernelGGL(g1_serialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g_ctx.stream, g_ctx.bases_tmp.as<uint8_t>(), ffi_affine_sz, n,
                           compressed, g_ctx.poly[0].as<uint8_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_bytes, g_ctx.poly[0].p, n * psz, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}
RustError snarkvm_hip_register_bases(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device) {
    API_BEGIN
    register_bases_impl(handle, points, npoints, ffi_affine_sz, on_device, 1, 0);
    API_END
}
RustError snarkvm_hip_register_bases_tables(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device,
                                            int tables) {
    API_BEGIN
    register_bases_impl(handle, points, npoints, ffi_affine_sz, on_device, tables, 0);
    API_END
}
RustError snarkvm_hip_register_bases_windowed(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device,
                                              int tables, int window_bits) {
    API_BEGIN
    if (window_bits <= 0) throw hip_failure{hipErrorInvalidValue, "register_bases_windowed: window_bits must be positive", __LINE__};
    register_bases_impl(handle, points, npoints, ffi_affine_sz, on_device, tables, window_bits);
    API_END
}
void snarkvm_hip_free_bases(snarkvm_hip_bases_t* h) {
    if (!h) return;
    std::lock_guard<std::mutex> lk(g_ctx.mu);
    if (h->d) (void)hipFree(h->d);
    delete h;
}
RustError snarkvm_hip_msm_registered(void* out, const snarkvm_hip_bases_t* h, size_t offset, size_t npoints, const void* scalars,
                                     int scalars_on_device, int window_bits) {
    API_BEGIN
    if (!h || offset + npoints > h->n) throw hip_failure{hipErrorInvalidValue, "msm_registered: range exceeds the registered bases", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_registered: window_bits must be 0 or 2..23", __LINE__};
    const uint4* d_sc = (const uint4*)scalars;
    if (!scalars_on_device && npoints) {
        g_ctx.scalars_tmp.ensure(npoints * 32);
        g_ctx.phase_begin("msm_h2d");
        HIP_TRY(hipMemcpyAsync(g_ctx.scalars_tmp.p, scalars, npoints * 32, hipMemcpyHostToDevice, g_ctx.stream));
        g_ctx.phase_end();
        d_sc = g_ctx.scalars_tmp.as<uint4>();
    }
    msm_run<fq_t>(g_ctx, h->d + offset, d_sc, npoints, out, window_bits, nullptr, ~(size_t)0, 0, h->tables, h->n, 0, true, h->table_bits);
    API_END
}

RustError snarkvm_hip_msm_registered_ex(void* out, const snarkvm_hip_bases_t* h, size_t off0, size_t n0, size_t off1, size_t n1,
                                        const void* scalars, int scalars_on_device, int scalars_montgomery, int window_bits) {
    API_BEGIN
    if (!h || off0 + n0 > h->n || off1 + n1 > h->n) throw hip_failure{hipErrorInvalidValue, "msm_registered_ex: range exceeds the registered bases", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_registered_ex: window_bits must be 0 or 2..23", __LINE__};
    const size_t n = n0 + n1;
    const uint4* d_sc = (const uint4*)scalars;
    if (!scalars_on_device && n) {
        g_ctx.scalars_tmp.ensure(n * 32);
        g_ctx.phase_begin("msm_h2d");
        HIP_TRY(hipMemcpyAsync(g_ctx.scalars_tmp.p, scalars, n * 32, hipMemcpyHostToDevice, g_ctx.stream));
        g_ctx.phase_end();
        d_sc = g_ctx.scalars_tmp.as<uint4>();
    }
    msm_run<fq_t>(g_ctx, h->d + off0, d_sc, n, out, window_bits, h->d + off1, n0, scalars_montgomery, h->tables, h->n, 0, true, h->table_bits);
    API_END
}
RustError snarkvm_hip_msm_registered_batch(void* outs, const snarkvm_hip_bases_t* h, size_t count, const size_t* offsets, const size_t* npoints,
                                           const void* const* scalars, int scalars_on_device, int scalars_montgomery, int window_bits) {
    API_BEGIN
    if (!h) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: null handle", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: window_bits must be 0 or 2..23", __LINE__};
    if (count * 144 > g_ctx.batch_pinned_cap) {
        if (g_ctx.batch_pinned) HIP_TRY(hipHostFree(g_ctx.batch_pinned));
        g_ctx.batch_pinned = nullptr;
        g_ctx.batch_pinned_cap = 0;
        HIP_TRY(hipHostMalloc(&g_ctx.batch_pinned, count * 144 + 144, hipHostMallocDefault));
        g_ctx.batch_pinned_cap = count * 144 + 144;
    }
    uint8_t* stage = (uint8_t*)g_ctx.batch_pinned;
    size_t largest = 0;
    for (size_t k = 0; k < count; k++) largest = npoints[k] > largest ? npoints[k] : largest;
    const int nlanes = context_t::batch_lanes(largest);
    for (size_t k = 0; k < count; k++) {
        if (offsets[k] + npoints[k] > h->n) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: range exceeds the registered bases", __LINE__};
        const int lane = (int)(k % (size_t)nlanes);
        msm_ws_t& ws = g_ctx.lane[lane];
        const uint4* d_sc = (const uint4*)scalars[k];
        if (!scalars_on_device && npoints[k]) {
            // the lane's previous instance may still be reading its scalar buffer: stream order serialises the copy behind it
            ws.scalars.ensure(npoints[k] * 32);
            HIP_TRY(hipMemcpyAsync(ws.scalars.p, scalars[k], npoints[k] * 32, hipMemcpyHostToDevice, ws.stream));
            d_sc = ws.scalars.as<uint4>();
        }
        msm_run<fq_t>(g_ctx, h->d + offsets[k], d_sc, npoints[k], stage + 144 * k, window_bits, nullptr, ~(size_t)0, scalars_montgomery, h->tables,
                      h->n, lane, false, h->table_bits);
    }
    for (int l = 0; l < context_t::LANES; l++) HIP_TRY(hipStreamSynchronize(g_ctx.lane[l].stream));
    memcpy(outs, stage, count * 144);
    API_END
}
RustError snarkvm_hip_g1_sum(void* out, const void* in_projective, size_t n) {
    API_BEGIN
    if (!out || (n && !in_projective)) throw hip_failure{hipErrorInvalidValue, "g1_sum: null argument", __LINE__};
    if (n == 0) {
        write_infinity<fq_t>(out);
    } else {
        g_ctx.poly[0].ensure(n * 144 + 144);
        uint32_t* d_in = g_ctx.poly[0].as<uint32_t>();
        uint32_t* d_out = d_in + 36 * n;
        HIP_TRY(hipMemcpyAsync(d_in, in_projective, n * 144, hipMemcpyHostToDevice, g_ctx.stream));
        hipLaunchKernelGGL(g1_sum_kernel, dim3(1), dim3(64), 0, g_ctx.stream, (const uint32_t*)d_in, n, d_out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, d_out, 144, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}
RustError snarkvm_hip_g1_to_affine(void* out_affine, const void* in_projective, size_t n) {
    API_BEGIN
    if (n) {
        dev_buf din, dout;
        din.ensure(n * 144);
        dout.ensure(n * 104);
        HIP_TRY(hipMemcpyAsync(din.p, in_projective, n * 144, hipMemcpyHostToDevice, g_ctx.stream));
        hipLaunchKernelGGL(g1_to_affine_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, g_ctx.stream, din.as<uint32_t>(), dout.as<uint32_t>(), n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_affine, dout.p, n * 104, hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
        (void)hipFree(din.p);
        (void)hipFree(dout.p);
    }
    API_END
}

// ---- Fr vector helpers ---------------------------------------------------------------------------
RustError snarkvm_hip_fr_mul_device(void* d_out, const void* d_a, const void* d_b, size_t n) {
    API_BEGIN
    if (n) {
        const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(fr_pointwise_mul_kernel, dim3(blocks), dim3(256), 0, g_ctx.stream, (fr_mem_t*)d_out, (const fr_mem_t*)d_a, (const fr_mem_t*)d_b, n, 1);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}
RustError snarkvm_hip_fr_convert_device(void* d_out, const void* d_in, size_t n, int to_bigint) {
    API_BEGIN
    if (n) {
        const unsigned blocks = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(fr_to_bigint_kernel, dim3(blocks), dim3(256), 0, g_ctx.stream, (fr_mem_t*)d_out, (const fr_mem_t*)d_in, n, to_bigint);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}

// ---- prover-round polynomial kernels (poly.cuh) ---------------------------------------------------
static fr_mem_t fr_mem_from_host(const void* p) {
    fr_mem_t m;
    memcpy(&m, p, sizeof m);
    return m;
}
static unsigned fr_grid(size_t n, unsigned block = 256) {
    const size_t b = (n + block - 1) / block;
    return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}
// operand `p` (n elements) as a device pointer: itself, or a staged copy in ctx.poly[slot]
static fr_mem_t* fr_stage_in(int slot, const void* p, size_t n, int on_device) {
    if (on_device || !p) return (fr_mem_t*)p;
    g_ctx.poly[slot].ensure(sizeof(fr_mem_t) * (n ? n : 1));
    if (n) HIP_TRY(hipMemcpyAsync(g_ctx.poly[slot].p, p, sizeof(fr_mem_t) * n, hipMemcpyHostToDevice, g_ctx.stream));
    return g_ctx.poly[slot].as<fr_mem_t>();
}
static fr_mem_t* fr_stage_out(int slot, void* p, size_t n, int on_device) {
    if (on_device || !p) return (fr_mem_t*)p;
    g_ctx.poly[slot].ensure(sizeof(fr_mem_t) * (n ? n : 1));
    return g_ctx.poly[slot].as<fr_mem_t>();
}
static void fr_finish_out(fr_mem_t* d, void* p, size_t n, int on_device) {
    if (!on_device && p && n) HIP_TRY(hipMemcpyAsync(p, d, sizeof(fr_mem_t) * n, hipMemcpyDeviceToHost, g_ctx.stream));
}

RustError snarkvm_hip_fr_vec_op(int op, void* out, const void* a, const void* b, const void* c, const void* scalar, size_t n, int on_device) {
    API_BEGIN
    if (op < 0 || op > FR_OP_RSUB_SCALAR) throw hip_failure{hipErrorInvalidValue, "fr_vec_op: unknown op", __LINE__};
    const bool need_b = op == FR_OP_ADD || op == FR_OP_SUB || op == FR_OP_MUL || op == FR_OP_MUL_SUB || op == FR_OP_AXPY;
    const bool need_c = op == FR_OP_MUL_SUB;
    const bool need_s = op == FR_OP_SCALE || op == FR_OP_SUB_SCALAR || op == FR_OP_AXPY || op == FR_OP_RSUB_SCALAR;
    if (n && (!out || !a || (need_b && !b) || (need_c && !c) || (need_s && !scalar)))
        throw hip_failure{hipErrorInvalidValue, "fr_vec_op: missing operand", __LINE__};
    if (n) {
        fr_mem_t s{};
        if (need_s) s = fr_mem_from_host(scalar);
        const fr_mem_t* da = fr_stage_in(0, a, n, on_device);
        const fr_mem_t* db = need_b ? fr_stage_in(1, b, n, on_device) : nullptr;
        const fr_mem_t* dc = need_c ? fr_stage_in(2, c, n, on_device) : nullptr;
        fr_mem_t* dout = fr_stage_out(3, out, n, on_device);
        hipLaunchKernelGGL(fr_vec_op_kernel, dim3(fr_grid(n)), dim3(256), 0, g_ctx.stream, op, dout, da, db, dc, s, n);
        HIP_TRY(hipGetLastError());
        fr_finish_out(dout, out, n, on_device);
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}

// out[i - shift] = h_i = sum_{k >= i} in[k] m^(k - i) (and *first = h_0 when shift == 1); `out` may be null (only h_0 wanted).
// Scratch for the chunk values of every level lives in ctx.poly[4].
static void fr_suffix_horner(const fr_mem_t* d_in, size_t n, const fr_mem_t& m, fr_mem_t* d_out, int shift, fr_mem_t* d_first) {
    hipStream_t st = g_ctx.stream;
    int levels = 1;
    size_t total = 0;
    for (size_t t = n; t > 1;) {
        t = (t + POLY_CHUNK - 1) / POLY_CHUNK;
        total += t;
        levels++;
    }
    g_ctx.poly[4].ensure(sizeof(fr_mem_t) * (total + levels + 2));
    fr_mem_t* mult = g_ctx.poly[4].as<fr_mem_t>();
    fr_mem_t* cvbase = mult + levels + 1;
    hipLaunchKernelGGL(fr_horner_multipliers_kernel, dim3(1), dim3(1), 0, st, m, mult, levels);
    // up-sweep: level k holds the chunk values of level k - 1 (level 0 = the input)
    std::vector<const fr_mem_t*> in_at{d_in};
    std::vector<size_t> n_at{n};
    fr_mem_t* next = cvbase;
    while (n_at.back() > 1) {
        const size_t cur = n_at.back();
        const size_t T = (cur + POLY_CHUNK - 1) / POLY_CHUNK;
        const int k = (int)n_at.size() - 1;
        hipLaunchKernelGGL(fr_horner_up_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, in_at.back(), cur, mult + k, next, T);
        in_at.push_back(next);
        n_at.push_back(T);
        next += T;
    }
    // the single value of the top level is h_0 of every level below; down-sweep turns each level's chunk values into
    // its suffix sums in place, the input level writes to `out`
    const int top = (int)n_at.size() - 1;
    for (int k = top; k >= 0; k--) {
        const size_t cur = n_at[k];
        const size_t T = (cur + POLY_CHUNK - 1) / POLY_CHUNK;
        const fr_mem_t* carry = (k < top) ? in_at[k + 1] : nullptr;
        if (k > 0) {
            if (k == top) continue;  // one element: it already is its own suffix sum
            hipLaunchKernelGGL(fr_horner_down_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, in_at[k], cur, mult + k, carry, T,
                               (fr_mem_t*)in_at[k], 0, (fr_mem_t*)nullptr);
        } else if (d_out) {
            hipLaunchKernelGGL(fr_horner_down_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, d_in, cur, mult, carry, T, d_out, shift,
                               d_first);
        } else if (d_first) {
            // only h_0: the value of the top level, or of the lone input element
            HIP_TRY(hipMemcpyAsync(d_first, top > 0 ? in_at[top] : d_in, sizeof(fr_mem_t), hipMemcpyDeviceToDevice, st));
        }
    }
    HIP_TRY(hipGetLastError());
}

RustError snarkvm_hip_fr_divide_by_linear(void* quotient, void* remainder, const void* poly, size_t n, const void* point, int on_device) {
    API_BEGIN
    if (!point || (n && !poly)) throw hip_failure{hipErrorInvalidValue, "fr_divide_by_linear: missing operand", __LINE__};
    if (n == 0) {
        if (remainder) memset(remainder, 0, sizeof(fr_mem_t));
    } else {
        const fr_mem_t z = fr_mem_from_host(point);
        const fr_mem_t* din = fr_stage_in(0, poly, n, on_device);
        fr_mem_t* dq = (quotient && n > 1) ? fr_stage_out(1, quotient, n - 1, on_device) : nullptr;
        g_ctx.poly[2].ensure(sizeof(fr_mem_t));
        fr_mem_t* drem = g_ctx.poly[2].as<fr_mem_t>();
        fr_suffix_horner(din, n, z, dq, 1, drem);
        if (dq) fr_finish_out(dq, quotient, n - 1, on_device);
        if (remainder) HIP_TRY(hipMemcpyAsync(remainder, drem, sizeof(fr_mem_t), hipMemcpyDeviceToHost, g_ctx.stream));
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}

static void fr_batch_inverse_run(fr_mem_t* d_v, size_t n, const fr_mem_t& coeff) {
    // >= 32 elements per thread amortise the per-thread Fermat inversion; cap the thread count for huge vectors
    size_t T = (n + 31) / 32;
    if (T > (size_t)1 << 17) T = (size_t)1 << 17;
    g_ctx.poly[4].ensure(sizeof(fr_mem_t) * n);
    hipLaunchKernelGGL(fr_batch_inverse_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, g_ctx.stream, d_v, n, coeff, g_ctx.poly[4].as<fr_mem_t>(), T);
    HIP_TRY(hipGetLastError());
}
RustError snarkvm_hip_fr_batch_inversion_and_mul(void* inout, size_t n, const void* coeff, int on_device) {
    API_BEGIN
    if (n) {
        if (!inout || !coeff) throw hip_failure{hipErrorInvalidValue, "fr_batch_inversion_and_mul: missing operand", __LINE__};
        fr_mem_t* dv = fr_stage_in(0, inout, n, on_device);
        fr_batch_inverse_run(dv, n, fr_mem_from_host(coeff));
        fr_finish_out(dv, inout, n, on_device);
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}

static void fr_distribute_powers_run(fr_mem_t* d_v, size_t n, const fr_mem_t& g, const fr_mem_t& c) {
    size_t T = (n + 31) / 32;
    if (T > (size_t)1 << 17) T = (size_t)1 << 17;
    hipLaunchKernelGGL(fr_distribute_powers_kernel, dim3((unsigned)((T + 63) / 64)), dim3(64), 0, g_ctx.stream, d_v, n, g, c, T);
    HIP_TRY(hipGetLastError());
}
RustError snarkvm_hip_fr_distribute_powers(void* inout, size_t n, const void* g, const void* c, int on_device) {
    API_BEGIN
    if (n) {
        if (!inout || !g || !c) throw hip_failure{hipErrorInvalidValue, "fr_distribute_powers: missing operand", __LINE__};
        fr_mem_t* dv = fr_stage_in(0, inout, n, on_device);
        fr_distribute_powers_run(dv, n, fr_mem_from_host(g), fr_mem_from_host(c));
        fr_finish_out(dv, inout, n, on_device);
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}

// TWO_ADIC_ROOT_OF_UNITY (fr.rs:115-120), memory form - the host copy of ntt.cuh's device table
static const uint32_t FR_TWO_ADIC_ROOT_MEM_HOST[8] = {0xda3ad648u, 0xaf80da4du, 0xfc381dacu, 0x5e223adbu,
                                                      0xb2f92525u, 0x03ba0666u, 0x3befb0ceu, 0x0f906c5bu};
RustError snarkvm_hip_fr_lagrange_coefficients(void* out, uint32_t lg, const void* tau, int on_device) {
    API_BEGIN
    if (lg > 30) throw hip_failure{hipErrorInvalidValue, "fr_lagrange_coefficients: lg_domain_size > 30", __LINE__};
    if (!out || !tau) throw hip_failure{hipErrorInvalidValue, "fr_lagrange_coefficients: missing operand", __LINE__};
    const size_t n = (size_t)1 << lg;
    // scalar set-up with the same arithmetic compiled for the host (domain.rs:118-147, 258-264)
    fr_t omega = fr_t::unpack(FR_TWO_ADIC_ROOT_MEM_HOST).from_mem_mont();
    for (uint32_t i = lg; i < 47; i++) omega = omega.sqr();
    const fr_mem_t tau_mem = fr_mem_from_host(tau);
    const fr_t tau_i = fr_t::load(&tau_mem).from_mem_mont();
    const fr_t t_size = tau_i.pow_u64((uint64_t)n);
    fr_mem_t one_mem, omega_mem;
    fr_t::one().to_mem_mont().store(&one_mem);
    omega.to_mem_mont().store(&omega_mem);
    fr_mem_t* du = fr_stage_out(0, out, n, on_device);
    hipStream_t st = g_ctx.stream;
    hipLaunchKernelGGL(fr_fill_kernel, dim3(fr_grid(n)), dim3(256), 0, st, du, n, one_mem);
    fr_distribute_powers_run(du, n, omega_mem, one_mem);  // u_i = omega^i
    if (t_size == fr_t::one()) {
        hipLaunchKernelGGL(fr_onehot_kernel, dim3(fr_grid(n)), dim3(256), 0, st, du, n, tau_mem, one_mem);
    } else {
        fr_mem_t l_mem;
        ((t_size - fr_t::one()) * fr_t::from_u32((uint32_t)n).inverse()).to_mem_mont().store(&l_mem);
        hipLaunchKernelGGL(fr_vec_op_kernel, dim3(fr_grid(n)), dim3(256), 0, st, (int)FR_OP_RSUB_SCALAR, du, (const fr_mem_t*)du, (const fr_mem_t*)nullptr,
                           (const fr_mem_t*)nullptr, tau_mem, n);  // tau - omega^i
        fr_batch_inverse_run(du, n, one_mem);
        fr_distribute_powers_run(du, n, omega_mem, l_mem);  // * l * omega^i
    }
    HIP_TRY(hipGetLastError());
    fr_finish_out(du, out, n, on_device);
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    API_END
}

RustError snarkvm_hip_fr_divide_by_vanishing(void* quotient, void* remainder, const void* poly, size_t len, size_t domain_size, int on_device) {
    API_BEGIN
    if (domain_size == 0) throw hip_failure{hipErrorInvalidValue, "fr_divide_by_vanishing: empty domain", __LINE__};
    if (len) {
        if (!poly || !remainder || (len > domain_size && !quotient)) throw hip_failure{hipErrorInvalidValue, "fr_divide_by_vanishing: missing operand", __LINE__};
        const size_t qlen = len > domain_size ? len - domain_size : 0;
        const size_t rlen = len < domain_size ? len : domain_size;
        const fr_mem_t* din = fr_stage_in(0, poly, len, on_device);
        fr_mem_t* dq = qlen ? fr_stage_out(1, quotient, qlen, on_device) : nullptr;
        fr_mem_t* dr = fr_stage_out(2, remainder, rlen, on_device);
        const size_t threads = qlen > rlen ? qlen : rlen;
        hipLaunchKernelGGL(fr_fold_vanishing_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, g_ctx.stream, din, len, domain_size, dq, dr);
        HIP_TRY(hipGetLastError());
        if (qlen) fr_finish_out(dq, quotient, qlen, on_device);
        fr_finish_out(dr, remainder, rlen, on_device);
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}
RustError snarkvm_hip_fr_mul_by_vanishing(void* out, const void* poly, size_t len, size_t domain_size, int on_device) {
    API_BEGIN
    const size_t olen = len + domain_size;
    if (olen) {
        if (!out || (len && !poly)) throw hip_failure{hipErrorInvalidValue, "fr_mul_by_vanishing: missing operand", __LINE__};
        const fr_mem_t* din = fr_stage_in(0, poly, len, on_device);
        fr_mem_t* dout = fr_stage_out(1, out, olen, on_device);
        hipLaunchKernelGGL(fr_mul_vanishing_kernel, dim3((unsigned)((olen + 255) / 256)), dim3(256), 0, g_ctx.stream, din, len, domain_size, dout);
        HIP_TRY(hipGetLastError());
        fr_finish_out(dout, out, olen, on_device);
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}

// ---- setup-time group operations (group.cuh) -------------------------------------------------------
RustError snarkvm_hip_g1_fixed_base_msm(void* out_projective, const void* g_affine, const void* scalars, size_t n) {
    API_BEGIN
    if (n) {
        if (!out_projective || !g_affine || !scalars) throw hip_failure{hipErrorInvalidValue, "g1_fixed_base_msm: null argument", __LINE__};
        hipStream_t st = g_ctx.stream;
        // the base in the engine's native form, through the regular conversion kernel
        g_ctx.bases_tmp.ensure(256 + sizeof(g1_aff_mem_t));
        HIP_TRY(hipMemcpyAsync(g_ctx.bases_tmp.p, g_affine, 104, hipMemcpyHostToDevice, st));
        g1_aff_mem_t* d_g = (g1_aff_mem_t*)(g_ctx.bases_tmp.as<uint8_t>() + 256);
        convert_bases<fq_t>(g_ctx, g_ctx.bases_tmp.as<uint8_t>(), 104, 1, d_g);
        g1_aff_mem_t g_native;
        HIP_TRY(hipMemcpyAsync(&g_native, d_g, sizeof g_native, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const size_t entries = (size_t)FIXED_OUTER << FIXED_WINDOW;
        g_ctx.poly[0].ensure(entries * sizeof(g1_aff_mem_t));
        g_ctx.poly[1].ensure(n * 32);
        g_ctx.poly[2].ensure(n * 144);
        hipLaunchKernelGGL(g1_fixed_table_kernel, dim3((unsigned)((entries + 255) / 256)), dim3(256), 0, st, g_native, g_ctx.poly[0].as<g1_aff_mem_t>());
        HIP_TRY(hipMemcpyAsync(g_ctx.poly[1].p, scalars, n * 32, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(g1_fixed_msm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g_ctx.poly[0].as<g1_aff_mem_t>(),
                           g_ctx.poly[1].as<fr_mem_t>(), n, g_ctx.poly[2].as<uint32_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_projective, g_ctx.poly[2].p, n * 144, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    API_END
}
RustError snarkvm_hip_g1_group_ntt(void* inout_projective, uint32_t lg, int inverse) {
    API_BEGIN
    if (lg > 24) throw hip_failure{hipErrorInvalidValue, "g1_group_ntt: lg_domain_size > 24", __LINE__};
    if (!inout_projective) throw hip_failure{hipErrorInvalidValue, "g1_group_ntt: null argument", __LINE__};
    const size_t n = (size_t)1 << lg;
    hipStream_t st = g_ctx.stream;
    g_ctx.poly[0].ensure(n * 144);
    g_ctx.poly[1].ensure(n * sizeof(g1_xyzz_mem_t));
    g_ctx.poly[2].ensure((n / 2 + 1) * sizeof(fr_mem_t));
    uint32_t* d_jac = g_ctx.poly[0].as<uint32_t>();
    g1_xyzz_mem_t* d_pts = g_ctx.poly[1].as<g1_xyzz_mem_t>();
    fr_mem_t* d_tw = g_ctx.poly[2].as<fr_mem_t>();
    HIP_TRY(hipMemcpyAsync(d_jac, inout_projective, n * 144, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(g1_jac_to_xyzz_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const uint32_t*)d_jac, d_pts, n);
    if (lg > 0) {
        // twiddles root^k (k < n/2) as canonical integers: ones -> distribute_powers -> to_bigint, all on the device
        fr_t omega = fr_t::unpack(FR_TWO_ADIC_ROOT_MEM_HOST).from_mem_mont();
        for (uint32_t i = lg; i < 47; i++) omega = omega.sqr();  // group_gen of the 2^lg domain (fft_field.rs:75-85)
        if (inverse) omega = omega.inverse();
        fr_mem_t one_mem, root_mem;
        fr_t::one().to_mem_mont().store(&one_mem);
        omega.to_mem_mont().store(&root_mem);
        const size_t h = n / 2;
        hipLaunchKernelGGL(fr_fill_kernel, dim3(fr_grid(h)), dim3(256), 0, st, d_tw, h, one_mem);
        fr_distribute_powers_run(d_tw, h, root_mem, one_mem);
        hipLaunchKernelGGL(fr_to_bigint_kernel, dim3(fr_grid(h)), dim3(256), 0, st, d_tw, (const fr_mem_t*)d_tw, h, 1);
        for (size_t half = n / 2; half >= 1; half >>= 1)
            hipLaunchKernelGGL(g1_ntt_stage_kernel, dim3((unsigned)((n / 2 + 63) / 64)), dim3(64), 0, st, d_pts, n, half, (const fr_mem_t*)d_tw, n / (2 * half));
        hipLaunchKernelGGL(g1_bitrev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_pts, n, (int)lg);
        if (inverse) {  // * size_inv (domain.rs:190)
            fr_mem_t k_int;
            fr_t::from_u32((uint32_t)n).inverse().mont_to_int().store(&k_int);
            hipLaunchKernelGGL(g1_scale_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_pts, n, k_int);
        }
    }
    hipLaunchKernelGGL(g1_xyzz_to_jac_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, (const g1_xyzz_mem_t*)d_pts, d_jac, n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(inout_projective, d_jac, n * 144, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    API_END
}

// ---- synthetic bases ---------------------------------------------------------------------------
// G1 generator (g1.rs:219-253), memory Montgomery form, 64-bit limbs
static const uint64_t G1_GEN_X[6] = {1171681672315280277ull, 6528257384425852712ull,  7514971432460253787ull,
                                     2032708395764262463ull, 12876543207309632302ull, 107509843840671767ull};
static const uint64_t G1_GEN_Y[6] = {13572190014569192121ull, 15344828677741220784ull, 17067903700058808083ull,
                                     10342263224753415805ull, 1083990386877464092ull,  21335464879237822ull};
RustError snarkvm_hip_g1_generate_bases_device(void* d_out, uint64_t start, size_t npoints) {
    API_BEGIN
    if (npoints) {
        // convert the generator on the host with the same arithmetic
        uint32_t xw[12], yw[12];
        memcpy(xw, G1_GEN_X, 48);
        memcpy(yw, G1_GEN_Y, 48);
        g1_aff_t g{fq_t::unpack(xw).from_mem_mont(), fq_t::unpack(yw).from_mem_mont()};
        g1_aff_mem_t gm;
        g.x.pack(gm.x.w);
        g.y.pack(gm.y.w);
        g_ctx.gen_pts.ensure(npoints * sizeof(g1_xyzz_mem_t));
        g_ctx.gen_prod.ensure(npoints * sizeof(fq_mem_t));
        const size_t threads = (npoints + GEN_RUN - 1) / GEN_RUN;
        hipLaunchKernelGGL(g1_generate_bases_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, g_ctx.stream, gm, start, npoints,
                           (uint8_t*)d_out, (size_t)104, g_ctx.gen_pts.as<g1_xyzz_mem_t>(), g_ctx.gen_prod.as<fq_mem_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    }
    API_END
}

// ---- test hooks ----------------------------------------------------------------------------------
int snarkvm_hip_selftest_field(int field, int op, const void* a, const void* b, void* out, size_t n) {
    const uint32_t* A = (const uint32_t*)a;
    const uint32_t* B = (const uint32_t*)(b ? b : a);
    uint32_t* O = (uint32_t*)out;
    for (size_t i = 0; i < n; i++) {
        if (field == 0)
            field_op<fr_t>(op, A + 8 * i, B + 8 * i, O + 8 * i);
        else if (field == 1)
            field_op<fq_t>(op, A + 12 * i, B + 12 * i, O + 12 * i);
        else
            return 1;
    }
    return 0;
}
RustError snarkvm_hip_devtest_field(int field, int op, const void* a, const void* b, void* out, size_t n) {
    API_BEGIN
    if (field < 0 || field > 1) throw hip_failure{hipErrorInvalidValue, "devtest_field: field must be 0 or 1", __LINE__};
    const size_t bytes = n * (field == 0 ? 32 : 48);
    dev_buf da, db, dout;
    da.ensure(bytes);
    db.ensure(bytes);
    dout.ensure(bytes);
    HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db.p, b ? b : a, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(devtest_field_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, g_ctx.stream, field, op, da.as<uint32_t>(),
                       db.as<uint32_t>(), dout.as<uint32_t>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(g_ctx.stream));
    HIP_TRY(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    (void)hipFree(da.p);
    (void)hipFree(db.p);
    (void)hipFree(dout.p);
    API_END
}
// The MSM planner on the host (no device needed): out = {c, W, J, Wd, nb, nbt, S, S2, L, wide}.  Returns 0.
int snarkvm_hip_selftest_msm_plan(size_t n, int window_bits, int tables, int table_bits, uint32_t* out) {
    const msm_plan_t p = msm_make_plan(n, window_bits, tables, table_bits);
    const uint32_t v[10] = {(uint32_t)p.c, (uint32_t)p.W, (uint32_t)p.J, (uint32_t)p.Wd, p.nb, p.nbt, p.S, p.S2, p.L, p.c > 16 ? 1u : 0u};
    for (int i = 0; i < 10; i++) out[i] = v[i];
    // bias must place one 2^(c-1) per digit row below 320 bits
    uint32_t chk[10] = {0};
    for (int w = 0; w < p.Wd; w++) {
        const int bit = p.c - 1 + p.c * w;
        if (bit >= 320) return 1;
        chk[bit / 32] |= 1u << (bit % 32);
    }
    for (int i = 0; i < 10; i++)
        if (chk[i] != p.bias[i]) return 2;
    return 0;
}
// naive sum_i scalar_i * P_i on the host with the device point arithmetic (scalars: 256-bit, 32 B each)
int snarkvm_hip_selftest_g1_msm_naive(const void* points, size_t npoints, size_t stride, const void* scalars, void* out) {
    const uint8_t* P = (const uint8_t*)points;
    const uint32_t* S = (const uint32_t*)scalars;
    g1_xyzz_t total = g1_xyzz_t::inf();
    for (size_t i = 0; i < npoints; i++) {
        const uint32_t* src = (const uint32_t*)(P + i * stride);
        g1_aff_t a;
        if (src[24] & 0xff)
            a = g1_aff_t::inf();
        else
            a = {fq_t::unpack(src).from_mem_mont(), fq_t::unpack(src + 12).from_mem_mont()};
        g1_xyzz_t acc = g1_xyzz_t::inf();
        for (int bit = 255; bit >= 0; bit--) {
            acc = acc.dbl();
            if ((S[8 * i + bit / 32] >> (bit % 32)) & 1) acc.add_affine(a);
        }
        // route half of the additions through the xyzz+xyzz law and the negation path
        if (i & 1) {
            g1_xyzz_t neg = acc;
            neg.y = neg.y.neg();
            g1_xyzz_t t2 = total;
            t2.add(acc);
            t2.add(neg);  // + acc - acc
            t2.add(acc);
            total = t2;
        } else {
            total.add(acc);
        }
    }
    const g1_jac_t j = total.to_jacobian();
    uint32_t* o = (uint32_t*)out;
    j.x.to_mem_mont().pack(o);
    j.y.to_mem_mont().pack(o + 12);
    j.z.to_mem_mont().pack(o + 24);
    return 0;
}

}  // extern "C"
