#pragma once
// runtime.hip.h - host runtime shared by the translation units of the backend (api.hip: G1 / Fr entry points, api_g2.hip: the Fq2
// instantiations, compiled in parallel by snarkvm_amd/build.py).  Everything here is header-only (static / inline / templates)
// except the one context object, which api.hip defines.
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
#include "ec.hip.h"
#include "ff.hip.h"
#include "msm.hip.h"
#include "msm_sort.hip.h"
#include "ntt.hip.h"
#include "group.hip.h"
#include "poly.hip.h"
#include "serde.hip.h"

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
    dev_buf rv1, rl1, rcounts2, roff2, rbinstart, rntiles, rtstart, rbsize;  // radix-partition sort (msm_sort.hip.h)
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
    dev_buf serde_status;  // one u32 of SERDE_* bits (serde.hip.h)
    dev_buf poly[5];  // staging / scratch of the prover-round vector kernels (poly.hip.h)
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
extern context_t g_ctx;  // defined in api.hip

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
    int rounds = 0;
    {
        // ---- 2.-4. LDS-staged radix partition (msm_sort.hip.h) -> bucket-major `sorted` + boff; two levels, three when wide
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
        uint32_t max_bucket = 0;  // the number of reduce rounds follows the largest bucket (4-byte read-back)
        HIP_TRY(hipMemcpyAsync(&max_bucket, d_max, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        // ---- 5. accumulate
        phase_begin("msm_accumulate");
        {
            // a bucket of s entries is touched by at most (s - 1) / S + 2 segment threads
            // (the tail kernels add up to TAIL_PARTIALS leftover partials per bucket themselves: one reduce round less)
            static const size_t tail_partials = getenv("SNARKVM_HIP_TAILP") ? (size_t)atoi(getenv("SNARKVM_HIP_TAILP")) : 4;
            for (size_t m = max_bucket ? ((size_t)max_bucket - 1) / pl.S + 2 : 0; m > tail_partials; m = (m + pl.S2 - 1) / pl.S2) rounds++;
            hipLaunchKernelGGL(msm_alloc_seg_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, boffp, c.cnt_a.as<uint32_t>(), nbt, pl.S);
            exclusive_scan_u32(st, c.cnt_a.as<uint32_t>(), c.start_a.as<uint32_t>(), (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
            const size_t nthreads = (E_max + pl.S - 1) / pl.S;
            // timing experiment only (wrong results): restrict the gather to the first 2^k bases to separate ALU time from HBM gather time
            static const uint32_t dbg_mask = getenv("SNARKVM_HIP_DEBUG_IDX_MASK") ? (uint32_t)strtoul(getenv("SNARKVM_HIP_DEBUG_IDX_MASK"), nullptr, 0) : 0xffffffffu;
            // (a 3-waves-per-SIMD build of this kernel - 168 VGPRs - and a software-pipelined gather were measured: no gain)
            hipLaunchKernelGGL((msm_accumulate_seg_kernel<F, 1>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, d_bases,
                               d_bases1 ? d_bases1 : d_bases, (uint32_t)n0, c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(),
                               c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S, (uint32_t)n, table_stride, dbg_mask);
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
        hipLaunchKernelGGL((msm_fold_wave_kernel<F>), dim3((1u << fold_m) + (1u << fold_hb)), dim3(64), 0, st, pin, start_in, cnt_in,
                           c.fold_sums.as<xyzz_mem_t<F>>(), fstart, fcnt, fold_m, fold_hb);
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
// Kernels with more than 64 KB of dynamic LDS need the attribute on THEIR function object; the non-template kernels are
// static, i.e. every translation unit launches its own copy, so every unit sets the attribute once for its copies.
static void tu_kernel_attributes() {
    static bool done = false;
    if (done) return;
#ifdef SV_TU_NTT  // the unit that launches the NTT passes (api_fr.hip)
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel_v2, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
#endif
    done = true;
}
#define API_BEGIN                                  \
    std::lock_guard<std::mutex> _lk(g_ctx.mu);     \
    try {                                          \
        g_ctx.init();                              \
        tu_kernel_attributes();                    \
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
static __global__ void devtest_field_kernel(int field, int op, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (field == 0)
        field_op<fr_t>(op, a + 8 * i, b + 8 * i, out + 8 * i);
    else
        field_op<fq_t>(op, a + 12 * i, b + 12 * i, out + 12 * i);
}

// ---- helpers shared by the G1 and G2 entry points
static void check_tables(int tables, int table_bits, const char* who) {
    const bool legacy = table_bits == 0 && (tables == 1 || tables == 2 || tables == 4 || tables == 8 || tables == 16);
    // upper bound: the recoding bias holds one bit per digit row below MSM_BIAS_BITS (msm_plan_t::bias, the digit kernels' 11-word scalar)
    const bool windowed = table_bits >= 2 && table_bits <= MSM_C_MAX && tables >= 1 && tables <= 127 && tables * table_bits >= 254 &&
                          tables * table_bits <= MSM_BIAS_BITS;
    if (!legacy && !windowed)
        throw std::runtime_error(std::string(who) + ": tables must be 1, 2, 4, 8 or 16, or 254 <= tables * window_bits <= 288 with window_bits in 2..23");
}
static void serde_throw_on_status(uint32_t st, const char* who) {
    if (!st) return;
    std::string m = std::string(who) + ":";
    if (st & SERDE_BAD_FLAGS) m += " UnexpectedFlags (both flag bits set)";
    if (st & SERDE_NOT_CANONICAL) m += " coordinate >= q";
    if (st & SERDE_NOT_ON_CURVE) m += " InvalidData (point not on the curve)";
    if (st & SERDE_NOT_IN_SUBGROUP) m += " InvalidData (point not in the prime-order subgroup)";
    throw std::runtime_error(m);  // SerializationError: surfaces as RustError code 1 with this message
}

