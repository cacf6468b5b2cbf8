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
#include <time.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <exception>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
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

static double host_now_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

// ------------------------------------------------------------------------------------------------
// runtime: devices, lanes (the reference's (dev, stream) resource tokens), staging buffers
// ------------------------------------------------------------------------------------------------
// Workspace growth is the one thing in the library that synchronises the whole device behind the caller's back (hipFree waits for
// every stream): counted, so that a caller - or a test - can see that a timed region allocated nothing
// (snarkvm_hip_alloc_stats: {device allocations, device bytes, pinned allocations, pinned bytes, microseconds inside them}).
void sv_alloc_note(int slot, size_t bytes, double ms);  // api.hip: the process-wide counters (this header is compiled into four units)
// A buffer that is outgrown in the middle of a call is not freed on the spot: hipFree returns only when EVERY stream of the device is
// idle (measured: 281 ms behind ~280 ms of kernels queued on another stream, tools/tables1_cliff.py), i.e. the second instance of a
// pipelined batch would only be enqueued after the first one had finished, and a caller on another lane would stall behind this one.
// The old block goes to a list of the calling thread and is released by sv_drain_frees() when that thread's call ends (lane_t::end_call, scope flush),
// or at once when an allocation fails.  Inside a long scope an outgrown block therefore stays allocated beside its replacement until the flush.  (Round 4's "tables1" bench leg had six such frees inside its timed region; whether they were what
// the driver's 93.6 ms per step came from could not be reproduced - profiles/r05_summary.md.)
void sv_defer_free(void* p);
void sv_drain_frees();
struct alloc_timer_t {
    double t0;
    int slot;
    size_t bytes;
    alloc_timer_t(int s, size_t b) : t0(host_now_ms()), slot(s), bytes(b) {}
    ~alloc_timer_t() { sv_alloc_note(slot, bytes, host_now_ms() - t0); }
};
struct dev_buf {
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {  // on the CURRENT device (a lane guard has selected it)
        if (bytes <= cap) return;
        size_t want = bytes + bytes / 8 + 256;
        alloc_timer_t timer(0, want);
        if (p) sv_defer_free(p);
        p = nullptr;
        cap = 0;
        if (hipMalloc(&p, want) != hipSuccess) {  // out of memory with outgrown blocks still parked: release them and try once more
            (void)hipGetLastError();
            p = nullptr;
            sv_drain_frees();
            HIP_TRY(hipMalloc(&p, want));
        }
        cap = want;
    }
    template <class T>
    T* as() const {
        return (T*)p;
    }
};
struct pinned_buf {  // page-locked host staging (the reference's per-GPU pinned arena, snarkvm.cu:51,123-151)
    void* p = nullptr;
    size_t cap = 0;
    void ensure(size_t bytes) {
        if (bytes <= cap) return;
        size_t want = bytes + bytes / 8 + 4096;
        alloc_timer_t timer(2, want);
        if (p) HIP_TRY(hipHostFree(p));
        p = nullptr;
        cap = 0;
        HIP_TRY(hipHostMalloc(&p, want, hipHostMallocDefault));
        cap = want;
    }
    template <class T>
    T* as() const {
        return (T*)p;
    }
};

struct phase_rec {
    const char* name;
    hipEvent_t e0, e1;
    double ms;
};
struct device_t;

// One lane = one HIP stream of one device with every scratch buffer a call needs: what a `(dev, stream)` token of the
// reference's resource channel stands for (snarkvm.cu:84,146-150).  A caller owns a lane for the duration of one API call;
// concurrent callers (rayon workers: one commitment each, sonic_pc/mod.rs:203-245) get different lanes / devices.
struct lane_t {
    device_t* dev = nullptr;
    int index = 0;
    hipStream_t stream = nullptr;
    // MSM workspace
    dev_buf scalars, digits, counts, offsets, scan_tmp, sorted, boff, cnt_a, cnt_b, start_a, start_b, part_a, part_b, part_raw, planes;
    dev_buf rv1, rl1, rcounts2, roff2, rbinstart, rntiles, rtstart, rbsize;  // radix-partition sort (msm_sort.hip.h)
    dev_buf rv2, rl2, rmid_size, rmid_boff;                                   // its middle level (wide windows)
    dev_buf fold_sums;                                                        // two-axis bucket fold
    dev_buf tail_flags;                                                       // Fq2 tail: outputs whose tree met equal x coordinates (recomputed by the fix kernels)
    dev_buf sink_acc;                                                         // bucket sink of a chunked MSM (msm_bucket_sink_t)
    dev_buf fchunk;                                                           // chunk sums / offsets of the fused level-1 scan
    dev_buf bases_tmp, scalars_tmp, gen_pts, gen_prod;
    // NTT / polynomial staging
    dev_buf ntt_data, ntt_scratch, ntt_acc, ntt_alt;
    dev_buf serde_status;  // one u32 of SERDE_* bits (serde.hip.h)
    dev_buf poly[5];       // staging / scratch of the prover-round vector kernels (poly.hip.h)
    pinned_buf pin, pin2;  // host staging: MSM bit-plane sums / chunked uploads
    hipStream_t alt = nullptr;  // second stream of the lane: copies of operand k + 1 while operand k is transformed (polynomial.cuh:136-242)
    hipEvent_t ev[4] = {};
    std::vector<void*> tw_leases;  // twiddle tables this call pinned in the device's cache
    // deferred-synchronisation scope (snarkvm_hip_scope_begin): the owning thread's device-resident calls are only enqueued
    bool in_scope = false;
    struct deferred_out_t {
        void* dst;
        size_t off, bytes;
    };
    std::vector<deferred_out_t> deferred;  // host results parked in pin2 until the scope ends
    size_t deferred_bytes = 0;
    // a lane owned by a scope (its main lane or one of its MSM lanes): `pin` is handed out piecewise to the MSMs the scope has enqueued
    // (bit planes + instance tables stay where they are until the scope's flush has collected them), events that must outlive a call
    size_t pin_used = 0;
    std::vector<hipEvent_t> scope_events;
    size_t scope_events_used = 0;
    hipEvent_t scope_event() {
        if (scope_events_used == scope_events.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            scope_events.push_back(e);
        }
        return scope_events[scope_events_used++];
    }
    // profiling
    std::vector<phase_rec> phases;
    std::vector<hipEvent_t> event_pool;
    size_t events_used = 0;

    hipEvent_t new_event() {
        if (events_used == event_pool.size()) {
            hipEvent_t e;
            HIP_TRY(hipEventCreate(&e));
            event_pool.push_back(e);
        }
        return event_pool[events_used++];
    }
    inline ntt_ctx_t ntt_ctx(hipStream_t st = nullptr);
    inline void begin_call();
    inline void phase_begin(const char* name);
    inline void phase_end();
    inline void phase_host(const char* name, double ms);
    inline void end_call();
    // the end of a call whose results all live in device memory: wait, unless the calling thread deferred that to its scope's end
    void sync_or_defer() {
        if (!in_scope) HIP_TRY(hipStreamSynchronize(stream));
    }
    // a small host result (<= a few KB) of a call that may run inside a scope: copied now and waited for, or parked in pinned
    // memory and delivered by snarkvm_hip_scope_end
    // may_defer = false: the call itself waits for the stream (host operands), so the value must be in `dst` when it returns
    void host_result(void* dst, const void* d_src, size_t bytes, bool may_defer = true) {
        if (!in_scope || !may_defer) {
            HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, stream));
            return;
        }
        if (deferred_bytes + bytes > pin2.cap) {
            if (!deferred.empty()) flush_scope();  // staging full: deliver what is parked, then start over
            pin2.ensure(deferred_bytes + bytes > (size_t)1 << 16 ? deferred_bytes + bytes : (size_t)1 << 16);
        }
        HIP_TRY(hipMemcpyAsync(pin2.as<uint8_t>() + deferred_bytes, d_src, bytes, hipMemcpyDeviceToHost, stream));
        deferred.push_back({dst, deferred_bytes, bytes});
        deferred_bytes += (bytes + 15) & ~(size_t)15;
    }
    inline void flush_scope();
};

struct device_t {
    int logical = 0, physical = 0;
    bool ready = false;
    std::mutex init_mu;
    ntt_tables_t tb{};
    dev_buf tables_mem;
    ntt_tw_cache_t tw;
    static constexpr int LANES = 16;  // streams are created up front; a lane's buffers only when it is first used
    lane_t lane[LANES];
    // token pool
    std::mutex mu;
    std::condition_variable cv;
    uint32_t busy = 0;
    // lanes held by deferred-synchronisation scopes (for as long as the scope is open).  At most SCOPE_LANES_MAX of them: four lanes of
    // every device always belong to calls that return their lane when they return, so a thread that waits for a lane - the ninth
    // scope_begin, a call on another device from inside a scope, the per-device worker threads of a multi-GPU call - waits for
    // something that ends (round-4 review: eight scopes that each issued an MSM held all eight lanes and waited for a ninth).
    static constexpr int SCOPE_LANES_MAX = LANES - 4;
    int scope_held = 0;

    void init() {  // the calling thread has this device current
        std::lock_guard<std::mutex> lk(init_mu);
        if (ready) return;
        // The upper half of the lanes runs on LOW-priority streams (tuning aux_low_prio): a scope hands its asynchronous MSMs - the work that is NOT on the caller's
        // critical path: the independent G2 MSM of a proof, commitments whose results are only due at scope_end - to those lanes (take_for_scope searches from the
        // top), ordinary calls and the scopes' own lanes take from the bottom.  When the chip is contended the dispatcher serves the critical stream first and the
        // background MSM fills the gaps the transcript order leaves (one proof in Fiat-Shamir order idles the GPU ~1.3 ms between its rounds).  With more than
        // LANES / 2 concurrent callers ordinary calls reach those lanes too: they then simply share one priority level among themselves.
        int prio_low = 0, prio_high = 0;
        if (hipDeviceGetStreamPriorityRange(&prio_low, &prio_high) != hipSuccess) prio_low = prio_high = 0;  // (numerically: low >= high)
        for (int l = 0; l < LANES; l++) {
            lane[l].dev = this;
            lane[l].index = l;
            if (l >= LANES / 2 && tuning().aux_cus > 0) {
                // (experiment, tuning aux_cus = N: the same lanes on streams that may only use the first N compute units of the mask order - a background MSM
                // then cannot take the whole chip away from the critical stream for the length of its fold)
                uint32_t mask[8] = {0};
                for (int b = 0; b < tuning().aux_cus && b < 256; b++) mask[b >> 5] |= 1u << (b & 31);
                HIP_TRY(hipExtStreamCreateWithCUMask(&lane[l].stream, 8, mask));
            } else if (l >= LANES / 2 && tuning().aux_low_prio && prio_low != prio_high)
                HIP_TRY(hipStreamCreateWithPriority(&lane[l].stream, hipStreamNonBlocking, prio_low));
            else
                HIP_TRY(hipStreamCreateWithFlags(&lane[l].stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&lane[l].alt, hipStreamNonBlocking));
            for (auto& e : lane[l].ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        // tables: 4 x (lo + hi) x NTT_TW_SIZE + 2 x NTT_LOCAL + size_inv[27] + 4 constants, 32 B each
        const size_t entries = 8 * NTT_TW_SIZE + 2 * NTT_LOCAL + 32 + 8;
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
            tb.local[d] = take(NTT_LOCAL);
        }
        tb.size_inv = take(32);
        tb.consts = take(8);
        hipStream_t st = lane[0].stream;
        hipLaunchKernelGGL(ntt_setup_consts, dim3(1), dim3(64), 0, st, tb);
        hipLaunchKernelGGL(ntt_fill_tables, dim3(NTT_TW_SIZE / 256), dim3(256), 0, st, tb);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(st));
        ready = true;
    }
    // lowest free lanes, up to `want` (at least one: blocks until a lane is free); returns the number taken
    int take(lane_t** out, int want) {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return busy != (1u << LANES) - 1; });
        int got = 0;
        for (int l = 0; l < LANES && got < want; l++)
            if (!(busy & (1u << l))) {
                busy |= 1u << l;
                out[got++] = &lane[l];
            }
        return got;
    }
    bool try_take(lane_t** out) {
        std::lock_guard<std::mutex> lk(mu);
        for (int l = 0; l < LANES; l++)
            if (!(busy & (1u << l))) {
                busy |= 1u << l;
                *out = &lane[l];
                return true;
            }
        return false;
    }
    // n lanes (<= want) without waiting; 0 when none is free
    int try_take_n(lane_t** out, int want) {
        std::lock_guard<std::mutex> lk(mu);
        int got = 0;
        for (int l = 0; l < LANES && got < want; l++)
            if (!(busy & (1u << l))) {
                busy |= 1u << l;
                out[got++] = &lane[l];
            }
        return got;
    }
    // a lane for a scope: block = the scope's main lane (waits for a free lane AND for room under the cap); else an extra MSM lane, only if
    // one is free right now
    lane_t* take_for_scope(bool block) {
        std::unique_lock<std::mutex> lk(mu);
        auto room = [&] { return busy != (1u << LANES) - 1 && scope_held < SCOPE_LANES_MAX; };
        if (block)
            cv.wait(lk, room);
        else if (!room())
            return nullptr;
        // a scope's own lane: from the bottom (normal priority); its further MSM lanes: from the top (low-priority streams, see init)
        for (int i = 0; i < LANES; i++) {
            const int l = block ? i : LANES - 1 - i;
            if (!(busy & (1u << l))) {
                busy |= 1u << l;
                scope_held++;
                return &lane[l];
            }
        }
        return nullptr;
    }
    void give(lane_t* l, bool from_scope = false) {
        {
            std::lock_guard<std::mutex> lk(mu);
            busy &= ~(1u << l->index);
            if (from_scope) scope_held--;
        }
        cv.notify_all();  // waiters wait on different conditions (a free lane / room under the scope cap)
    }
};

// The process-wide runtime: the set of devices in use (the reference's `ngpus()` loop, snarkvm.cu:123-151), chosen by
// snarkvm_hip_set_devices / SNARKVM_HIP_DEVICES (default: every visible device).  A physical device may be listed more than
// once: each entry is an independent logical device (own streams, workspaces, base replicas) - how the multi-device paths
// are exercised on a one-GPU box.
struct runtime_t {
    std::mutex cfg_mu;
    std::vector<int> want;  // physical ids requested (empty: default)
    std::vector<std::unique_ptr<device_t>> devs;
    bool configured = false;
    std::atomic<uint32_t> rr{0};
    // profiling: phases of the most recent profiled call (any lane)
    std::atomic<bool> profiling{false};
    std::mutex prof_mu;
    std::vector<std::pair<std::string, double>> last_phases;

    void configure() {
        std::lock_guard<std::mutex> lk(cfg_mu);
        if (configured) return;
        int ndev = 0;
        hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev == 0) throw hip_failure{e == hipSuccess ? hipErrorNoDevice : e, "hipGetDeviceCount (no MI355X visible)", __LINE__};
        std::vector<int> ids = want;
        if (ids.empty()) {
            if (const char* env = getenv("SNARKVM_HIP_DEVICES")) {
                for (const char* p = env; *p;) {
                    char* end = nullptr;
                    long v = strtol(p, &end, 10);
                    if (end == p) break;
                    ids.push_back((int)v);
                    p = (*end == ',') ? end + 1 : end;
                }
            }
        }
        if (ids.empty())
            for (int d = 0; d < ndev; d++) ids.push_back(d);
        for (int id : ids)
            if (id < 0 || id >= ndev) throw hip_failure{hipErrorInvalidDevice, "device index out of range (snarkvm_hip_set_devices / SNARKVM_HIP_DEVICES)", __LINE__};
        for (size_t i = 0; i < ids.size(); i++) {
            devs.emplace_back(new device_t());
            devs.back()->logical = (int)i;
            devs.back()->physical = ids[i];
        }
        configured = true;
    }
    int ndev() {
        configure();
        return (int)devs.size();
    }
    // logical device that owns a device pointer (-1: not a device pointer of a configured device)
    int device_of(const void* ptr) {
        configure();
        hipPointerAttribute_t at;
        if (!ptr || hipPointerGetAttributes(&at, ptr) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        std::vector<int> match;
        for (auto& d : devs)
            if (d->physical == at.device) match.push_back(d->logical);
        if (match.empty()) return -1;
        return match[rr.fetch_add(1) % match.size()];
    }
};
extern runtime_t g_rt;  // defined in api.hip

// Per-(translation unit, device) kernel attributes: kernels with more than 64 KB of dynamic LDS need the attribute on THEIR
// function object; the non-template kernels are static, i.e. every unit launches its own copy.
static void tu_kernel_attributes(int logical) {
    static std::mutex mu;
    static std::vector<char> done;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)done.size() <= logical) done.resize(logical + 1, 0);
    if (done[logical]) return;
    HIP_TRY(hipFuncSetAttribute((const void*)scan_one_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(SCAN_ONE_MAX * 4)));
#ifdef SV_TU_G1  // the unit that launches the G1 MSMs (api.hip): see msm_acc_lds()
    HIP_TRY(hipFuncSetAttribute((const void*)msm_accumulate_lazy_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
#endif
#ifdef SV_TU_NTT  // the unit that launches the NTT passes (api_fr.hip)
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel_v2<false, ntt_arith_u>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel_v2<true, ntt_arith_u>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel_v2<false, ntt_arith_s>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void*)ntt_pass_kernel_v2<true, ntt_arith_s>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
#endif
    done[logical] = 1;
}

// RAII ownership of lanes.  Selecting a lane makes its device current on the calling thread (the HIP current device is per
// host thread - rayon workers!) and restores the previous one on release.
// Deferred-synchronisation scope of the calling thread (snarkvm_hip_scope_begin / _end): one lane stays bound to the thread; its
// device-resident calls borrow that lane and return after the enqueue.  Anything that takes OTHER lanes (host-buffer MSMs, another
// device) first waits for the scope's work - what it reads may have been produced inside the scope - and then runs on the scope's own
// lane (plus whatever further lanes are free right now): a thread inside a scope never waits for a lane.
// SNARKVM_HIP_SCOPE_ASYNC_MSM: MSMs over registered bases with device-resident scalars are enqueued as well - on up to SCOPE_AUX_MAX
// further lanes of the scope, in turn, each behind an event of the scope's stream - and their host finishes (the Horner chain over the
// bit planes) run when the scope is flushed.
static constexpr int SCOPE_AUX_MAX = 7;  // one proof: six commitment rounds + the G2 MSM, every one on its own stream
struct scope_pending_t {
    hipEvent_t done;               // everything the finish reads has arrived in pinned memory
    std::function<void()> finish;  // host Horner chains -> the callers' `out` buffers
    const void* tag;               // the first `out` pointer of the call that enqueued it (snarkvm_hip_scope_collect(out))
};
struct thread_scope_t {
    lane_t* lane = nullptr;
    int prev_device = -1;
    unsigned flags = 0;
    lane_t* aux[SCOPE_AUX_MAX] = {};
    int naux = 0;
    unsigned aux_rr = 0;
    bool aux_exhausted = false;  // a try for a further lane failed: not tried again in this scope
    std::vector<scope_pending_t> pending;
    thread_scope_t() = default;
    thread_scope_t(const thread_scope_t&) = default;
    thread_scope_t& operator=(const thread_scope_t&) = default;
    ~thread_scope_t();  // a thread that ends with its scope open gives the lanes back (api.hip)
};
// ONE object per thread for the whole library: defined in api.hip.  (Rounds 3-4 kept it as a function-local static of this header,
// i.e. one copy per translation unit: a scope opened by api.hip was invisible to the transforms and vector passes of api_fr.hip and
// to the G2 calls of api_g2.hip - they took another lane and waited for it call by call, unordered against the scope's stream.)
extern thread_local thread_scope_t g_tl_scope;
static thread_scope_t& tl_scope() { return g_tl_scope; }
// wait for everything the calling thread's scope has enqueued; deliver the results it owes (MSM outputs, parked host values)
static void scope_flush() {
    thread_scope_t& sc = tl_scope();
    if (!sc.lane) return;
    std::exception_ptr err;
    // in enqueue order: the host finishes of the early MSMs run while the GPU still works on the later ones
    for (scope_pending_t& p : sc.pending) {
        try {
            HIP_TRY(hipEventSynchronize(p.done));
            p.finish();
        } catch (...) {
            if (!err) err = std::current_exception();
        }
    }
    sc.pending.clear();
    try {
        sc.lane->flush_scope();
        for (int i = 0; i < sc.naux; i++) HIP_TRY(hipStreamSynchronize(sc.aux[i]->stream));
    } catch (...) {
        if (!err) err = std::current_exception();
    }
    sc.lane->pin_used = 0;
    sc.lane->scope_events_used = 0;
    for (int i = 0; i < sc.naux; i++) sc.aux[i]->pin_used = 0, sc.aux[i]->scope_events_used = 0;
    if (err) std::rethrow_exception(err);
}
// the outputs of the MSM call that was given `tag` as its (first) output - or, tag == nullptr, of every MSM the scope has enqueued so far
// (snarkvm_hip_scope_collect): the scope stays open, its own stream is not waited for, the other pending MSMs stay pending
static void scope_collect(const void* tag) {
    thread_scope_t& sc = tl_scope();
    if (!sc.lane) return;
    std::exception_ptr err;
    if (tag) {
        // While this thread would only WAIT for `tag`'s results, it delivers what has already arrived for the others (a proof in transcript order: the independent
        // G2 MSM issued first is ready two or three rounds later - its Horner chain over Fq2, ~0.15 ms of host time, then runs under a commitment round's kernels
        // instead of behind the last round at scope_end).  Only while the wanted results are not there yet: they are never delayed by more than one such finish.
        auto ready = [](const scope_pending_t& p) {
            const hipError_t e = hipEventQuery(p.done);
            if (e == hipSuccess) return true;
            (void)hipGetLastError();  // (hipErrorNotReady is an answer, not a failure: it must not surface in a later launch check)
            return false;
        };
        auto wanted_ready = [&]() {
            for (const scope_pending_t& p : sc.pending)
                if (p.tag == tag && !ready(p)) return false;
            return true;
        };
        for (size_t i = 0; i < sc.pending.size() && !wanted_ready();) {
            if (sc.pending[i].tag != tag && ready(sc.pending[i])) {
                try {
                    sc.pending[i].finish();
                } catch (...) {
                    if (!err) err = std::current_exception();
                }
                sc.pending.erase(sc.pending.begin() + (ptrdiff_t)i);
            } else {
                i++;
            }
        }
    }
    std::vector<scope_pending_t> keep;
    for (scope_pending_t& p : sc.pending) {
        if (tag && p.tag != tag) {
            keep.push_back(std::move(p));
            continue;
        }
        try {
            HIP_TRY(hipEventSynchronize(p.done));
            p.finish();
        } catch (...) {
            if (!err) err = std::current_exception();
        }
    }
    sc.pending.swap(keep);
    if (sc.pending.empty()) {
        // nothing of an enqueued MSM is in flight any more: the staging areas and the events can be handed out again - on the MSM lanes (an MSM is
        // the only work they get) and on the scope's own lane (its events were "inputs ready" marks for those MSMs, or, without a free MSM lane, the
        // MSMs' own "done" marks; parked host values live in pin2, not in pin)
        sc.lane->pin_used = 0, sc.lane->scope_events_used = 0;
        for (int i = 0; i < sc.naux; i++) sc.aux[i]->pin_used = 0, sc.aux[i]->scope_events_used = 0;
    }
    if (err) std::rethrow_exception(err);
}
struct lane_guard {
    std::vector<lane_t*> lanes;
    int prev_device = -1;
    bool borrowed = false;        // every lane belongs to the thread's scope (nothing to give back)
    bool first_borrowed = false;  // lanes[0] is the scope's lane, the others are ours
    lane_guard() {}
    lane_guard(const lane_guard&) = delete;
    // one lane on logical device `dev`, or on the least busy device when dev < 0
    explicit lane_guard(int dev) {
        lane_t* sl = tl_scope().lane;
        if (sl && (dev < 0 || (dev < (int)g_rt.devs.size() && g_rt.devs[dev]->physical == sl->dev->physical))) {  // inside the thread's scope: its lane, its device is already current
            borrowed = true;
            lanes.push_back(sl);
            tu_kernel_attributes(sl->dev->logical);  // this unit's kernels may not have run on the device yet (the scope was opened by api.hip)
            return;
        }
        acquire(dev, 1);
    }
    void acquire(int dev, int want) {
        scope_flush();
        g_rt.configure();
        const int nd = (int)g_rt.devs.size();
        if (dev >= nd) throw hip_failure{hipErrorInvalidDevice, "logical device index out of range", __LINE__};
        if (lane_t* sl = tl_scope().lane) {
            if (dev < 0 || g_rt.devs[dev]->physical == sl->dev->physical) {
                // the scope's own lane (idle now) serves as the first lane of this call; more only if they are free right now
                lanes.push_back(sl);
                first_borrowed = true;
                tu_kernel_attributes(sl->dev->logical);
                lane_t* more[device_t::LANES];
                const int n = want > 1 ? sl->dev->try_take_n(more, want - 1) : 0;
                for (int i = 0; i < n; i++) lanes.push_back(more[i]);
                return;
            }
        }
        if (prev_device < 0 && hipGetDevice(&prev_device) != hipSuccess) prev_device = 0;
        device_t* d = nullptr;
        lane_t* got[device_t::LANES];
        int n = 0;
        if (dev >= 0) {
            d = g_rt.devs[dev].get();
        } else {
            const uint32_t s = g_rt.rr.fetch_add(1);
            for (int i = 0; i < nd && !d; i++) {  // first device (round-robin start) with a free lane
                device_t* cand = g_rt.devs[(s + i) % nd].get();
                if (want == 1) {
                    if (cand->try_take(&got[0])) {
                        d = cand;
                        n = 1;
                    }
                } else {
                    d = cand;
                }
            }
            if (!d) d = g_rt.devs[s % nd].get();
        }
        if (!n) n = d->take(got, want);
        try {  // a failure below must not leak the tokens: the destructor of a throwing constructor never runs
            HIP_TRY(hipSetDevice(d->physical));
            d->init();
            tu_kernel_attributes(d->logical);
        } catch (...) {
            for (int i = 0; i < n; i++) d->give(got[i]);
            if (prev_device >= 0) (void)hipSetDevice(prev_device);
            throw;
        }
        for (int i = 0; i < n; i++) lanes.push_back(got[i]);
    }
    lane_t& c() { return *lanes[0]; }
    ~lane_guard() {
        if (borrowed) return;
        for (size_t i = first_borrowed ? 1 : 0; i < lanes.size(); i++) lanes[i]->dev->give(lanes[i]);
        if (prev_device >= 0) (void)hipSetDevice(prev_device);
    }
};

inline ntt_ctx_t lane_t::ntt_ctx(hipStream_t st) { return ntt_ctx_t{st ? st : stream, &dev->tb, &dev->tw, &tw_leases}; }
inline void lane_t::begin_call() {
    phases.clear();
    events_used = 0;
}
inline void lane_t::phase_begin(const char* name) {
    if (!g_rt.profiling.load(std::memory_order_relaxed)) return;
    phase_rec r{name, new_event(), new_event(), 0.0};
    HIP_TRY(hipEventRecord(r.e0, stream));
    phases.push_back(r);
}
inline void lane_t::phase_end() {
    if (!g_rt.profiling.load(std::memory_order_relaxed) || phases.empty()) return;
    HIP_TRY(hipEventRecord(phases.back().e1, stream));
}
inline void lane_t::phase_host(const char* name, double ms) {  // a phase that ran on the calling thread (no events)
    if (!g_rt.profiling.load(std::memory_order_relaxed)) return;
    phases.push_back(phase_rec{name, nullptr, nullptr, ms});
}
static void ntt_tw_release(device_t* dev, std::vector<void*>& leases) { ntt_tw_release_entries(dev->tw, leases); }
inline void lane_t::flush_scope() {  // wait for everything the scope has enqueued, deliver parked host results, return the twiddle leases
    HIP_TRY(hipStreamSynchronize(stream));
    for (const deferred_out_t& d : deferred) memcpy(d.dst, pin2.as<uint8_t>() + d.off, d.bytes);
    deferred.clear();
    deferred_bytes = 0;
    if (!tw_leases.empty()) ntt_tw_release(dev, tw_leases);
    sv_drain_frees();
}
inline void lane_t::end_call() {
    if (in_scope) return;  // nothing is waited for; leases are held until the scope ends
    sv_drain_frees();
    if (!tw_leases.empty()) {
        HIP_TRY(hipStreamSynchronize(stream));
        ntt_tw_release(dev, tw_leases);
    }
    if (phases.empty()) return;
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<std::pair<std::string, double>> out;
    for (auto& r : phases) {
        if (r.e0) {
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, r.e0, r.e1));
            r.ms = ms;
        }
        out.emplace_back(r.name, r.ms);
    }
    std::lock_guard<std::mutex> lk(g_rt.prof_mu);
    g_rt.last_phases.swap(out);
}

// fn(logical device) on every listed device: inline for one device, one host thread per device otherwise (the reference's
// per-GPU `pool.spawn`, snarkvm.cu:262).  The first exception is rethrown on the caller's thread.
template <class Fn>
static void for_each_device(const std::vector<int>& devices, Fn fn) {
    if (devices.size() <= 1) {
        for (int d : devices) fn(d);
        return;
    }
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> errs(devices.size());
    for (size_t i = 0; i < devices.size(); i++)
        th.emplace_back([&, i] {
            try {
                fn(devices[i]);
            } catch (...) {
                errs[i] = std::current_exception();
            }
        });
    for (auto& t : th) t.join();
    for (auto& e : errs)
        if (e) std::rethrow_exception(e);
}

// registered base vectors: one replica per logical device (every device holds its own copy of the static SRS)
struct msm_ticket_t;
template <class F>
struct bases_handle_t {
    std::vector<aff_mem_t<F>*> d;  // [logical device]: tables * n entries: table j at d + j * n holds 2^(table_bits * j) * P_i
    size_t n = 0;
    int tables = 1;
    int table_bits = 256;  // table j = 2^(table_bits * j) * P
    // tickets of concurrent callers waiting to be fused (msm_coalesced)
    mutable std::mutex co_mu;
    mutable std::condition_variable co_cv;
    mutable std::deque<msm_ticket_t*> co_q;
    mutable int co_leaders = 0;
    void free_all() {
        int prev = 0;
        (void)hipGetDevice(&prev);
        for (size_t i = 0; i < d.size(); i++)
            if (d[i]) {
                (void)hipSetDevice(g_rt.devs[i]->physical);
                (void)hipFree(d[i]);
                d[i] = nullptr;
            }
        (void)hipSetDevice(prev);
    }
};
struct snarkvm_hip_bases : bases_handle_t<fq_t> {};

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

// ---- host-side finish of an MSM --------------------------------------------------------------------------------------
// The device leaves `nplanes` bit-plane sums (msm.hip.h 7b): the MSM result is sum_i 2^(pos[i]) * plane[i].  The remaining
// Horner chain (<= ~270 doublings of wave-uniform data) runs here, on the host, with the SAME field / curve code compiled
// for the host - the counterpart of the reference's host-side `dadd` collapse of its per-GPU results
// (algorithms/cuda/cuda/snarkvm.cu:290-295).  Several devices / chunks of one MSM simply add their planes into the same
// accumulator before the chain (msm_accum_t::add_planes), which is the whole multi-GPU combine.
static constexpr int MSM_MAX_POS = 320;
template <class F>
struct msm_accum_t {
    xyzz_t<F> at[MSM_MAX_POS];
    bool used[MSM_MAX_POS];
    int top = -1;
    msm_accum_t() {
        for (int i = 0; i < MSM_MAX_POS; i++) used[i] = false;
    }
    void add(int pos, const xyzz_t<F>& p) {
        if (p.is_inf()) return;
        if (pos < 0 || pos >= MSM_MAX_POS) throw std::runtime_error("msm: bit position out of range");
        if (!used[pos]) {
            at[pos] = p;
            used[pos] = true;
        } else {
            at[pos].add(p);
        }
        if (pos > top) top = pos;
    }
    // total = sum_p 2^p at[p], Jacobian memory image (the reference's Projective; infinity = (0, 1, 0))
    void finish(void* out) const {
        xyzz_t<F> t = xyzz_t<F>::inf();  // the chain stays in XYZZ (9 products per doubling, 14 per addition, no conversions)
        for (int p = top; p >= 0; p--) {
            t = t.dbl();
            if (used[p]) t.add(at[p]);
        }
        const jac_t<F> j = t.to_jacobian();
        uint32_t w[3 * F::MEM_WORDS];
        j.x.to_raw_words(w);
        j.y.to_raw_words(w + F::MEM_WORDS);
        j.z.to_raw_words(w + 2 * F::MEM_WORDS);
        memcpy(out, w, sizeof w);
    }
};
// what one device-side MSM run leaves for the host: planes (pinned host memory, valid after the lane's stream has been
// synchronised) and the bit position of each plane
struct msm_pending_t {
    const void* planes = nullptr;  // xyzz_mem_t<F>[nplanes]
    int nplanes = 0;
    int tail_windows = 0, nbits = 0;
    // position of plane (tw, j): DENSE (folded): tw = 2 w + sub -> c w + sub m + j; else tw = w -> c w + j
    int c = 0, m = 0;
    bool folded = false;
    int ninst = 0;  // fused multi-instance run: window w IS instance w (every instance has ONE bucket window at bit position 0)
    bool lazy = false;  // G1: the planes are raw lazy points (xyzz_mem_t<fqz_t>, 208 B each; tuning lazy_tail) - msm_collect converts them
    int pos(int idx) const {
        const int tw = idx / nbits, j = idx % nbits;
        if (ninst) return (tw & 1) * m + j;
        return folded ? c * (tw >> 1) + (tw & 1) * m + j : c * tw + j;
    }
};
// bytes of one bit plane / partial sum in the staging areas: the larger of the exact and the raw lazy image (G1: 192 / 208 B)
template <class F>
static constexpr size_t msm_point_bytes() {
    return sizeof(F) == sizeof(fq_t) && sizeof(xyzz_mem_t<fqz_t>) > sizeof(xyzz_mem_t<F>) ? sizeof(xyzz_mem_t<fqz_t>) : sizeof(xyzz_mem_t<F>);
}
// plane i of a pending run as an exact point (a raw lazy plane: four products by 2^377 on the host)
template <class F>
static xyzz_t<F> msm_plane(const msm_pending_t& pd, int i) {
    if constexpr (sizeof(F) == sizeof(fq_t)) {
        if (pd.lazy) {
            const xyzz_t<fqz_t> z = load_xyzz<fqz_t>(&((const xyzz_mem_t<fqz_t>*)pd.planes)[i]);
            if (z.is_inf()) return xyzz_t<F>::inf();
            return {z.x.to_exact(), z.y.to_exact(), z.zz.to_exact(), z.zzz.to_exact()};
        }
    }
    return load_xyzz<F>(&((const xyzz_mem_t<F>*)pd.planes)[i]);
}
template <class F>
static void msm_collect(msm_accum_t<F>& acc, const msm_pending_t& pd) {
    for (int i = 0; i < pd.nplanes; i++) acc.add(pd.pos(i), msm_plane<F>(pd, i));
}
// the planes of instance `inst` of a fused multi-instance run (2 * nbits consecutive planes)
template <class F>
static void msm_collect_inst(msm_accum_t<F>& acc, const msm_pending_t& pd, int inst) {
    const int per = 2 * pd.nbits;
    for (int i = inst * per; i < (inst + 1) * per; i++) acc.add(pd.pos(i), msm_plane<F>(pd, i));
}
// host description of a fused multi-instance run (msm_sort.hip.h: msm_inst_t)
struct msm_multi_t {
    const msm_inst_t* d_inst = nullptr;  // device table, K + 1 entries (sentinel: pstart = npad)
    uint32_t K = 0;
    size_t npad = 0;  // padded positions of all instances (multiple of SORT_TILE)
    size_t hn = 0;    // points of the registered vector: virtual index = table * hn + base index
    size_t plane_capacity = 0;  // planes the caller's staging area holds (checked before the copy is enqueued)
};

// Single-round accumulate grids are 256 workgroups of 4 waves for 256 CUs - one wave per SIMD when every CU gets exactly one
// workgroup.  The registers would let a CU take two, and the dispatcher does hand some CUs two while others stay idle; asking for
// more than half of a CU's 160 KB of LDS (unused) makes the second workgroup impossible.  tuning acc_lds overrides (0: off).
static size_t msm_acc_lds() {
    const long env = tuning().acc_lds;
    return env < 0 ? 0 : (size_t)env;
}
template <class F>
static bool msm_lazy_on();
// G1: the tail (reduce rounds, bucket merge, fold, bit planes) runs on the lazy arithmetic too (ffl.hip.h::fqz_t) and reads the accumulate
// kernel's raw partial sums as they are
template <class F>
static bool msm_lazy_tail_on() {
    return sizeof(F) == sizeof(fq_t) && msm_lazy_on<F>() && tuning().lazy_tail != 0;
}
// bytes of one partial sum / sink slot / plane on the device for the arithmetic the tail of an MSM over F runs on
template <class F>
static size_t msm_partial_bytes() {
    return msm_lazy_tail_on<F>() ? sizeof(xyzz_mem_t<fqz_t>) : sizeof(xyzz_mem_t<F>);
}
// Tail geometry of an MSM with `nwin` bucket windows of 2^(c - 1) buckets: windows of >= 2^11 buckets are first folded into two tail
// windows of 2^fold_m / 2^fold_hb - 1 entries; so are smaller windows when there are too few (window, bit) pairs to spread an
// unfolded tail over the chip (registered tables below 4 096 points: 2 windows x 8 bits would be 16 workgroups walking every
// partial sum; the fold gives 48).  Fills the pending record the host finish reads.
struct msm_tail_geom_t {
    int fold_m, fold_hb, tail_windows, nbits;
    bool fold;
};
static msm_tail_geom_t msm_tail_geometry(const msm_plan_t& pl, uint32_t nwin, msm_pending_t& pd, int ninst) {
    msm_tail_geom_t g;
    const int K = pl.c - 1;
    g.fold_m = (K + 1) / 2;
    g.fold_hb = K - g.fold_m;
    g.fold = K >= 11 || (K >= 4 && pl.c * pl.W < 128);
    g.tail_windows = g.fold ? 2 * (int)nwin : (int)nwin;
    g.nbits = g.fold ? g.fold_m + 1 : pl.c;  // weights run up to 2^fold_m (L sums) / 2^(c-1) (plain buckets)
    pd.tail_windows = g.tail_windows;
    pd.nbits = g.nbits;
    pd.nplanes = g.tail_windows * g.nbits;
    pd.c = pl.c;
    pd.m = g.fold_m;
    pd.folded = g.fold;
    pd.ninst = ninst;
    if ((!ninst && pd.nplanes > MSM_MAX_POS) || pl.c * (pl.W - 1) + (g.fold ? g.fold_m : 0) + g.nbits > MSM_MAX_POS)
        throw hip_failure{hipErrorInvalidValue, "msm: window geometry exceeds the tail's bit-position range", __LINE__};
    return g;
}
// 7.-9. of msm_run: per-bucket partial-sum lists (sums, start, cnt) -> fold -> bit-plane sums -> copy to `host_planes` (the host runs
// the Horner chain).  flat: flattened-list fold (any distribution of the partial sums over the buckets).
template <class F>
static void msm_tail_launch(lane_t& c, const msm_plan_t& pl, const msm_tail_geom_t& g, uint32_t nwin, uint32_t nbt, const xyzz_mem_t<F>* sums,
                            const uint32_t* start, const uint32_t* cnt, bool flat, const msm_pending_t& pd, void* host_planes) {
    hipStream_t st = c.stream;
    c.planes.ensure((size_t)pd.nplanes * sizeof(xyzz_mem_t<F>));
    const bool is_g2 = sizeof(F) == sizeof(fq2_t);
    const int hex = (is_g2 && tuning().hex2) ? 1 : 0;  // G2: the upper tree levels on sixteen lanes per addition (hex2.hip.h)
    // quad-strided accumulation in front of the trees (msm.hip.h), a bit mask: 1 = G2 bit planes, 2 = G2 fold, 4 = G1 bit planes, 8 = G1 fold.  Measured on the
    // 2^16 G2 tail (tools/g2_tail.sh, 17 x 15 geometry): bit planes 235 -> 202 us (186 with hex2 = 2), fold 375 -> 404 us; on one proof in transcript order (bench.py --workload proof1): 8.54 -> 8.29 - 8.37 ms with 13, 8.41 with 9, 8.37 with 15 - hence the default 13.
    // ... and only in the LATENCY regime (a small MSM's tail: a handful of entries per workgroup, the chip not full).  A big MSM's fold / bit planes are throughput-bound -
    // one wave per output walking thousands of entries - and four lanes repeating every addition there is four times the work: measured at 2^24 (12 x 22 geometry)
    // fold 1.31 -> 1.66 ms, bit planes 0.18 -> 0.20 ms (profiles/r06_summary.md), so the mask applies to folds that run 128 / 256 threads per output and planes of <= 256 entries.
    int quads_planes = (tuning().tail_quads >> (is_g2 ? 0 : 2)) & 1, quads_fold = (tuning().tail_quads >> (is_g2 ? 1 : 3)) & 1;
    if (g.fold_m > 8) quads_planes = 0;
    // Fq2: the kernels never compute P + P or P - P (msm.hip.h TAIL_FLAGGED): they flag the outputs whose additions met equal x coordinates, and a one-wave kernel per output kind
    // recomputes those with the plain law - an unflagged workgroup returns at once.  Flags: [fold slots | planes].
    uint32_t* fold_flags = nullptr;
    uint32_t* plane_flags = nullptr;
    if (is_g2) {
        const size_t nslots = g.fold ? ((size_t)nwin << (g.fold_m + 1)) : 0;
        c.tail_flags.ensure((nslots + (size_t)pd.nplanes) * 4);
        fold_flags = c.tail_flags.as<uint32_t>();
        plane_flags = fold_flags + nslots;
    }
    if (g.fold) {
        c.fold_sums.ensure(((size_t)nwin << (g.fold_m + 1)) * sizeof(xyzz_mem_t<F>));
        // 256 threads per output keep the serial part of a small fold short - as long as the whole grid is resident at once
        // (<= 512 workgroups at two waves per SIMD); many windows (table-less small MSMs: 20 windows x 128 outputs) or many
        // buckets are throughput-bound: one wave per output
        const unsigned fold_blocks = ((1u << g.fold_m) + (1u << g.fold_hb)) * (unsigned)nwin;
        // (G2 kernels hold one wave per SIMD: 256-thread workgroups sit one per CU, so 384 of them take two turns on 256 CUs; 128-thread
        // workgroups sit two per CU and lose one level of the tree besides - tuning fold_threads2)
        unsigned fold_threads = (nbt >= (1u << 18) || fold_blocks > 512u) ? 64u : 256u;
        // a fused group of three or four proof-sized G1 instances (768 / 1 024 workgroups; commitment rounds 4 and 5 of a proof): 128 threads per output still put the
        // whole grid on the chip at once (<= 2 048 waves at two per SIMD) and halve the serial walk of a lone wave - tuning fold_mid (64: round 5's one wave per output)
        if (sizeof(F) <= 64 && nbt < (1u << 18) && fold_blocks > 512u && fold_blocks <= 1024u && tuning().fold_mid == 128) fold_threads = 128u;
        // (measured, 17 x 15 geometry = 256 workgroups: 256 threads 0.38 ms, 128 threads 0.53 ms, 64 threads 0.83 ms - the halved workgroup only pays when the
        // grid would otherwise take two turns, tools/g2_tail.sh)
        if (sizeof(F) > 64 && fold_threads == 256u && fold_blocks > 256u && (tuning().fold_threads2 == 128 || tuning().fold_threads2 == 64)) fold_threads = (unsigned)tuning().fold_threads2;
        if (sizeof(F) > 64 && fold_threads == 256u && fold_blocks <= 256u && (tuning().fold_small2 == 128 || tuning().fold_small2 == 64)) fold_threads = (unsigned)tuning().fold_small2;
        if (fold_threads == 64u) quads_fold = 0;
        const dim3 fold_grid((1u << g.fold_m) + (1u << g.fold_hb), (unsigned)nwin);
        if (flat || fold_threads != 64u)
            hipLaunchKernelGGL((msm_fold_kernel<F, true>), fold_grid, dim3(fold_threads), 0, st, sums, start, cnt, c.fold_sums.as<xyzz_mem_t<F>>(), g.fold_m, g.fold_hb, hex,
                               quads_fold, fold_flags);
        else
            hipLaunchKernelGGL((msm_fold_kernel<F, false>), fold_grid, dim3(fold_threads), 0, st, sums, start, cnt, c.fold_sums.as<xyzz_mem_t<F>>(), g.fold_m, g.fold_hb, hex,
                               quads_fold, fold_flags);
        if constexpr (TAIL_FLAGGED<F>::value)
            hipLaunchKernelGGL((msm_fold_fix_kernel<F>), fold_grid, dim3(64), 0, st, sums, start, cnt, c.fold_sums.as<xyzz_mem_t<F>>(), g.fold_m, g.fold_hb,
                               (const uint32_t*)fold_flags);
        // one lane per entry of a plane (<= 2^fold_m); quad-strided: one QUAD per entry, up to 64 quads
        const unsigned plane_threads = quads_planes ? (g.fold_m <= 4 ? 64u : g.fold_m == 5 ? 128u : 256u) : (g.fold_m <= 6 ? 64u : g.fold_m == 7 ? 128u : 256u);
        const dim3 plane_grid((unsigned)g.nbits, (unsigned)g.tail_windows);
        hipLaunchKernelGGL((msm_bitplane_kernel<F, true>), plane_grid, dim3(plane_threads), 0, st, (const xyzz_mem_t<F>*)c.fold_sums.as<xyzz_mem_t<F>>(),
                           (const uint32_t*)nullptr, (const uint32_t*)nullptr, c.planes.as<xyzz_mem_t<F>>(), pl.nb, g.fold_m, g.fold_hb, hex, quads_planes, plane_flags);
        if constexpr (TAIL_FLAGGED<F>::value)
            hipLaunchKernelGGL((msm_bitplane_fix_kernel<F, true>), plane_grid, dim3(64), 0, st, (const xyzz_mem_t<F>*)c.fold_sums.as<xyzz_mem_t<F>>(), (const uint32_t*)nullptr,
                               (const uint32_t*)nullptr, c.planes.as<xyzz_mem_t<F>>(), pl.nb, g.fold_m, g.fold_hb, (const uint32_t*)plane_flags);
    } else {
        const dim3 plane_grid((unsigned)g.nbits, (unsigned)g.tail_windows);
        hipLaunchKernelGGL((msm_bitplane_kernel<F, false>), plane_grid, dim3(256), 0, st, sums, start, cnt, c.planes.as<xyzz_mem_t<F>>(), pl.nb, 0, 0, hex, 0, plane_flags);
        if constexpr (TAIL_FLAGGED<F>::value)
            hipLaunchKernelGGL((msm_bitplane_fix_kernel<F, false>), plane_grid, dim3(64), 0, st, sums, start, cnt, c.planes.as<xyzz_mem_t<F>>(), pl.nb, 0, 0,
                               (const uint32_t*)plane_flags);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host_planes, c.planes.p, (size_t)pd.nplanes * sizeof(xyzz_mem_t<F>), hipMemcpyDeviceToHost, st));
}
// A bucket sink: the chunks of ONE big MSM (snarkvm_msm over host bases: point-range chunks that arrive over PCIe one after the other)
// leave their per-bucket partial sums in a persistent accumulator instead of each running its own fold / bit-plane tail; the tail runs
// once, over the accumulator, after the last chunk.  L lanes work on the chunks concurrently: each owns one slot per bucket
// (bucket k, lane l -> acc[k * L + l]), so no two streams ever touch the same slot.
struct msm_bucket_sink_t {
    void* acc = nullptr;  // xyzz_mem_t<F>[nbt * L], zero-initialised (the point at infinity)
    uint32_t L = 1, slot = 0, nbt = 0;
    // L == 1 with several lanes (2^21 buckets of a wide-window geometry: one slot per bucket, not one per lane): the merges are
    // chained - a chunk's merge waits for `after` (the previous chunk's merge) and records `done`
    hipEvent_t after = nullptr, done = nullptr;
};
// Steps 6.-9. of msm_run on the tail arithmetic T (F itself, or fqz_t for a G1 MSM whose accumulate kernel left raw lazy partial sums):
// reduce rounds (cnt_a, start_a, part_a) -> (cnt_b, start_b, part_b) -> ..., then either the merge into a bucket sink (a chunk of a
// bigger MSM; pd.nplanes = 0) or fold -> bit planes -> copy to `host_planes`.
template <class T, class PB, class PE>
static void msm_reduce_and_tail(lane_t& c, const msm_plan_t& pl, const msm_tail_geom_t& tg, uint32_t nwin, uint32_t nbt, int rounds, size_t T0_max, size_t T1_max,
                                const msm_bucket_sink_t* sink, msm_pending_t& pd, void* host_planes, bool flat, PB&& phase_begin, PE&& phase_end) {
    hipStream_t st = c.stream;
    phase_begin("msm_reduce_partials");
    uint32_t *cnt_in = c.cnt_a.as<uint32_t>(), *cnt_out = c.cnt_b.as<uint32_t>();
    uint32_t *start_in = c.start_a.as<uint32_t>(), *start_out = c.start_b.as<uint32_t>();
    xyzz_mem_t<T> *pin = c.part_a.as<xyzz_mem_t<T>>(), *pout = c.part_b.as<xyzz_mem_t<T>>();
    size_t T_in_max = T0_max;
    for (int r = 0; r < rounds; r++) {
        size_t T_out_max = T_in_max / pl.S2 + nbt + 1;
        if (T_out_max > T1_max) T_out_max = T1_max;  // both ping-pong buffers hold >= T1_max partials
        hipLaunchKernelGGL(msm_alloc_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, cnt_in, cnt_out, nbt, pl.S2);
        exclusive_scan_u32(st, cnt_out, start_out, (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
        hipLaunchKernelGGL((msm_reduce_kernel<T>), dim3((unsigned)((T_out_max + 255) / 256)), dim3(256), 0, st, pin, start_in, cnt_in, start_out, pout, nbt,
                           pl.S2);
        std::swap(cnt_in, cnt_out);
        std::swap(start_in, start_out);
        std::swap(pin, pout);
        T_in_max = T_out_max;
    }
    phase_end();
    if (sink) {
        // a chunk of a bigger MSM: its per-bucket partial sums join the sink; the tail runs once, after the last chunk (msm_tail_from_sink)
        phase_begin("msm_bucket_merge");
        if (sink->after) HIP_TRY(hipStreamWaitEvent(st, sink->after, 0));
        hipLaunchKernelGGL((msm_bucket_merge_kernel<T>), dim3((nbt + 255) / 256), dim3(256), 0, st, (const xyzz_mem_t<T>*)pin, (const uint32_t*)start_in,
                           (const uint32_t*)cnt_in, (xyzz_mem_t<T>*)sink->acc, nbt, sink->L, sink->slot);
        if (sink->done) HIP_TRY(hipEventRecord(sink->done, st));
        phase_end();
        HIP_TRY(hipGetLastError());
        pd.nplanes = 0;
        return;
    }
    phase_begin("msm_bucket_reduce");
    msm_tail_launch<T>(c, pl, tg, nwin, nbt, pin, start_in, cnt_in, flat, pd, host_planes);
    phase_end();
}
// Device side of one MSM on lane `c`: d_bases = converted device bases; d_scalars = device scalars (32 B each).  Everything
// is enqueued on the lane's stream, ending with the copy of the bit-plane sums into `host_planes` (pinned, >=
// msm_plane_bytes<F>(...)); the caller synchronises the stream and runs msm_collect / msm_accum_t::finish.
template <class F>
static size_t msm_plane_bytes() {
    return (size_t)MSM_MAX_POS * msm_point_bytes<F>();  // upper bound on tail windows * bits
}
template <class F>
static msm_pending_t msm_run(lane_t& c, const aff_mem_t<F>* d_bases, const uint4* d_scalars, size_t n, void* host_planes, int window_bits,
                             const aff_mem_t<F>* d_bases1 = nullptr, size_t n0 = ~(size_t)0, int scalars_montgomery = 0, int tables = 1,
                             size_t table_stride = 0, bool profile = true, int table_bits = 0, const msm_multi_t* mu = nullptr,
                             const msm_bucket_sink_t* sink = nullptr, hipEvent_t scalars_read = nullptr) {
    // scalars_read: recorded on the lane's stream behind the last kernel that reads the scalar vectors (the digit kernel; wide windows:
    // the fused level-1 scatter) - from there on the caller may overwrite them while the MSM is still running
    // mu != nullptr: fused multi-instance run (msm_sort.hip.h): n = mu->npad padded positions, d_bases = the handle's table array,
    // d_scalars unused (the instance table carries the pointers), one bucket window per instance; host_planes holds
    // mu->K * 2 * (fold_m + 1) planes.
    // SNARKVM_HIP_TRACE=2 (diagnostics): wait for the stream after every phase and name it on stderr - locates a kernel that never returns
    static const int trace2 = getenv("SNARKVM_HIP_TRACE") ? atoi(getenv("SNARKVM_HIP_TRACE")) : 0;
    const char* cur_phase = "";
    auto phase_begin = [&](const char* name) {
        cur_phase = name;
        if (trace2 >= 2) fprintf(stderr, "[snarkvm_hip] msm n=%zu: %s ...\n", n, name);
        if (profile) c.phase_begin(name);
    };
    auto phase_end = [&]() {
        if (profile) c.phase_end();
        if (trace2 >= 2) {
            const hipError_t e = hipStreamSynchronize(c.stream);
            fprintf(stderr, "[snarkvm_hip] msm n=%zu: %s done (%s)\n", n, cur_phase, hipGetErrorString(e));
        }
    };
    msm_pending_t pd;
    pd.planes = host_planes;
    if (n0 > n) n0 = n;
    if (n == 0) return pd;  // no planes: the sum is the point at infinity
    if (n >= ((size_t)1 << 31)) throw hip_failure{hipErrorInvalidValue, "msm: npoints must be < 2^31", __LINE__};
    const msm_plan_t pl = msm_make_plan(n, mu ? table_bits : window_bits, tables, table_bits);
    const bool wide = pl.c > 16;  // u32 digits, three-level sort
    if (mu && (wide || pl.W != 1 || pl.c < 12 || (size_t)pl.J * mu->hn >= ((size_t)1 << 31) || n != mu->npad || n % SORT_TILE))
        throw hip_failure{hipErrorInvalidValue, "msm: geometry not eligible for a fused multi-instance run", __LINE__};
    if ((size_t)pl.Wd * n >= ((size_t)1 << 32)) throw hip_failure{hipErrorInvalidValue, "msm: windows * npoints must be < 2^32", __LINE__};
    if ((size_t)pl.J * n >= ((size_t)1 << 31)) throw hip_failure{hipErrorInvalidValue, "msm: tables * npoints must be < 2^31", __LINE__};
    const aff_mem_t<F>* vb1 = d_bases1 ? d_bases1 : d_bases;
    const aff_mem_t<F>* vbase = mu ? d_bases : (vb1 < d_bases ? vb1 : d_bases);
    if (!mu) {
        const size_t top0 = (size_t)(d_bases - vbase) + n0, top1 = (size_t)(vb1 - vbase) + (n - n0);
        if ((size_t)(pl.J - 1) * table_stride + (top0 > top1 ? top0 : top1) >= ((size_t)1 << 31))
            throw hip_failure{hipErrorInvalidValue, "msm: base slots must be addressable in 31 bits (tables * registered points < 2^31)", __LINE__};
    }
    hipStream_t st = c.stream;
    const size_t E_max = (size_t)pl.Wd * n;
    const uint32_t nwin = mu ? mu->K : (uint32_t)pl.W;  // bucket windows of the tail (multi: one per instance)
    const uint32_t nbt = nwin * pl.nb;

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
    const bool ltail = msm_lazy_tail_on<F>();
    pd.lazy = ltail;
    c.part_a.ensure(T0_max * msm_partial_bytes<F>());
    c.part_b.ensure(T1_max * msm_partial_bytes<F>());
    const int K = pl.c - 1;  // bucket-index bits
    const msm_tail_geom_t tg = msm_tail_geometry(pl, nwin, pd, mu ? (int)mu->K : 0);
    if (mu && (size_t)pd.nplanes > mu->plane_capacity) throw hip_failure{hipErrorInvalidValue, "msm: plane staging of the fused group too small", __LINE__};
    if (sink && (mu || sink->nbt != nbt)) throw hip_failure{hipErrorInvalidValue, "msm: bucket sink does not match the plan", __LINE__};
    c.planes.ensure((size_t)pd.nplanes * msm_partial_bytes<F>());

    // 1. scalar read.  Wide windows: fused with the level-1 partition below (the digits never exist in memory); otherwise the
    // stand-alone digit kernel writes the [rows][n] digit matrix.
    const int fused_env = tuning().fused;
    const bool fused = !mu && wide && fused_env && pl.c <= 22 && pl.Wd <= FUSED_MAX_ROWS;  // level-1 key of <= 7 bits: FUSED_G * 2^HB <= FUSED_THREADS
    msm_digit_params_t dp;
    memcpy(dp.bias, pl.bias, sizeof dp.bias);
    dp.c = pl.c;
    dp.W = pl.Wd;
    dp.n = n;
    dp.montgomery = scalars_montgomery;
    if (!fused) {
        phase_begin("msm_digits");
        c.digits.ensure(E_max * (wide ? sizeof(uint32_t) : sizeof(uint16_t)));
        size_t blocks = (n + 255) / 256;
        if (mu)
            hipLaunchKernelGGL(msm_digits_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, mu->d_inst, mu->K, c.digits.as<uint16_t>(), dp);
        if (blocks > 256 * 16) blocks = 256 * 16;
        if (mu)
            ;
        else if (wide)
            hipLaunchKernelGGL((msm_digits_kernel<uint32_t>), dim3((unsigned)blocks), dim3(256), 0, st, d_scalars, c.digits.as<uint32_t>(), dp);
        else
            hipLaunchKernelGGL((msm_digits_kernel<uint16_t>), dim3((unsigned)blocks), dim3(256), 0, st, d_scalars, c.digits.as<uint16_t>(), dp);
        phase_end();
        if (scalars_read) HIP_TRY(hipEventRecord(scalars_read, st));
    }
    int rounds = 0;
    // see step 5; a fused multi-instance run never reads back either: its instances are small (<= 2^18 points each), so the
    // flattened-list fold takes whatever partial sums the accumulate grid leaves
    const bool single_round = mu || (size_t)pl.Wd * n <= ((size_t)1 << 22);
    const bool prefetch_ok = (size_t)pl.Wd * n <= ((size_t)1 << 22);  // one wave per SIMD: nothing else hides the gather
    {
        // ---- 2.-4. LDS-staged radix partition (msm_sort.hip.h) -> bucket-major `sorted` + boff; two levels, three when wide
        msm_radix_params_t rp;
        rp.n = n;
        rp.c = pl.c;
        rp.W = pl.W;
        rp.J = pl.J;
        // virtual indices = slots relative to vbase (msm_radix_params_t): the lower of the two base ranges, or the handle's table array
        if (mu) {
            rp.inst = mu->d_inst;
            rp.ninst = mu->K;
            rp.vstride = (uint32_t)mu->hn;
        } else {
            rp.vn0 = (uint32_t)n0;
            rp.vr0 = (uint32_t)(d_bases - vbase);
            rp.vr1 = (uint32_t)(vb1 - vbase);
            rp.vstride = (uint32_t)table_stride;
        }
        const int LBL = K < 7 ? K : 7;  // key bits of the last level
        rp.LB = wide ? 14 : LBL;        // bits left below the level-1 key
        rp.HB = K - rp.LB;
        rp.nb = pl.nb;
        rp.xcd = (uint32_t)tuning().xcd;
        rp.tiles_per_row = fused ? (uint32_t)((n + FUSED_TILE - 1) / FUSED_TILE) : (uint32_t)((n + SORT_TILE - 1) / SORT_TILE);
        rp.TPW = (uint32_t)pl.J * rp.tiles_per_row;
        const uint32_t B1 = 1u << rp.HB;
        const uint32_t nbins = nwin * B1;
        // single: W windows x B1 bins x TPW tiles; multi: the windows (instances) partition the J * npad / TILE tiles among themselves
        const size_t ncounts1 = (size_t)(mu ? B1 : nbins) * rp.TPW;
        const size_t tiles1 = (size_t)(mu ? 1 : pl.W) * rp.TPW;
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
        if (fused) {
            // the scalar-read phase proper: a read-only pass over the scalars (32 B each) that leaves the level-1 histograms
            const uint32_t ntiles = rp.tiles_per_row, keys = (uint32_t)pl.Wd * B1;
            const uint32_t nchunks = (ntiles + FUSED_CHUNK - 1) / FUSED_CHUNK;
            const size_t ngroups = (size_t)keys * nchunks;
            c.counts.ensure((size_t)ntiles * keys * 4);
            c.offsets.ensure((size_t)ntiles * keys * 4);
            c.fchunk.ensure(2 * ngroups * 4);
            c.scan_tmp.ensure(scan_tmp_elems(ngroups > (size_t)nbt + 2 ? ngroups : (size_t)nbt + 2) * 4);
            counts1 = c.counts.as<uint32_t>();
            off1 = c.offsets.as<uint32_t>();
            uint32_t* csum = c.fchunk.as<uint32_t>();
            uint32_t* choff = csum + ngroups;
            phase_begin("msm_scalar_read");
            const size_t hist_lds = (size_t)keys * 4;
            // hist = 2: 1 024-thread workgroups with four private histogram copies (msm_sort.hip.h); 1: the round-3 kernel
            const int hist_variant = tuning().hist;
#define SV_FUSED_HIST(CB)                                                                                                                         \
    case CB:                                                                                                                                      \
        if (hist_variant == 2)                                                                                                                    \
            hipLaunchKernelGGL((radix_hist1_wide_kernel<CB>), dim3(ntiles), dim3(HISTW_THREADS), hist_lds * HISTW_COPIES, st, d_scalars, counts1, rp, dp); \
        else                                                                                                                                      \
            hipLaunchKernelGGL((radix_hist1_fused_kernel<CB>), dim3(ntiles), dim3(FUSED_THREADS), hist_lds, st, d_scalars, counts1, rp, dp);     \
        break;
            switch (pl.c) { SV_FUSED_HIST(17) SV_FUSED_HIST(18) SV_FUSED_HIST(19) SV_FUSED_HIST(20) SV_FUSED_HIST(21) SV_FUSED_HIST(22) }
#undef SV_FUSED_HIST
            phase_end();
            phase_begin("msm_sort_level1");
            const unsigned key_blocks = (keys + FUSED_THREADS - 1) / FUSED_THREADS;
            hipLaunchKernelGGL(fused_chunk_sums_kernel, dim3(nchunks, key_blocks), dim3(FUSED_THREADS), 0, st, (const uint32_t*)counts1, csum, ntiles, nchunks, keys, B1,
                               (uint32_t)pl.W, (uint32_t)pl.J);
            exclusive_scan_u32(st, csum, choff, ngroups, c.scan_tmp.as<uint32_t>());
            hipLaunchKernelGGL(fused_tile_offsets_kernel, dim3(nchunks, key_blocks), dim3(FUSED_THREADS), 0, st, (const uint32_t*)counts1, (const uint32_t*)choff,
                               (const uint32_t*)csum, off1, c.rbinstart.as<uint32_t>(), ntiles, nchunks, keys, B1, (uint32_t)pl.W, (uint32_t)pl.J);
#define SV_FUSED_SCATTER(CB)                                                                                                                      \
    case CB:                                                                                                                                      \
        hipLaunchKernelGGL((radix_scatter1_fused_kernel<CB>), dim3(ntiles), dim3(FUSED_THREADS), 0, st, d_scalars, (const uint32_t*)counts1,     \
                           (const uint32_t*)off1, c.rv1.as<uint32_t>(), c.rl1.as<uint16_t>(), rp, dp);                                            \
        break;
            switch (pl.c) { SV_FUSED_SCATTER(17) SV_FUSED_SCATTER(18) SV_FUSED_SCATTER(19) SV_FUSED_SCATTER(20) SV_FUSED_SCATTER(21) SV_FUSED_SCATTER(22) }
#undef SV_FUSED_SCATTER
            if (scalars_read) HIP_TRY(hipEventRecord(scalars_read, st));
        } else if (wide) {
            phase_begin("msm_sort_level1");
            hipLaunchKernelGGL((radix_hist1_kernel<uint32_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint32_t>(), counts1, rp);
            exclusive_scan_u32(st, counts1, off1, ncounts1, c.scan_tmp.as<uint32_t>());
            hipLaunchKernelGGL((radix_scatter1_kernel<uint32_t, uint16_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint32_t>(),
                               counts1, off1, c.rv1.as<uint32_t>(), c.rl1.as<uint16_t>(), rp);
        } else {
            phase_begin("msm_sort_level1");
            hipLaunchKernelGGL((radix_hist1_kernel<uint16_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint16_t>(), counts1, rp);
            exclusive_scan_u32(st, counts1, off1, ncounts1, c.scan_tmp.as<uint32_t>());
            hipLaunchKernelGGL((radix_scatter1_kernel<uint16_t, uint8_t>), dim3((unsigned)tiles1), dim3(SORT_THREADS), 0, st, c.digits.as<uint16_t>(),
                               counts1, off1, c.rv1.as<uint32_t>(), c.rl1.as<uint8_t>(), rp);
        }
        if (mu)
            hipLaunchKernelGGL(radix_bin_layout_multi_kernel, dim3((nbins + 1 + 255) / 256), dim3(256), 0, st, off1, counts1, ncounts1,
                               c.rbinstart.as<uint32_t>(), nbins, rp);
        else if (!fused)
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
                               c.rl2.as<uint8_t>(), nseg, 7, 7, (uint32_t)tuning().xcd);
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
                           (uint8_t*)nullptr, nseg, LBL, 0, (uint32_t)tuning().xcd);
        phase_end();
        // A single-round MSM (<= 2^22 digit entries: at most 2^16 accumulate threads) leaves at most 2^16 + nbt partial sums
        // whatever the scalars are, and the tail kernels walk them position by position (msm.hip.h 7a/7b): no reduce round.
        // Bigger MSMs run a FIXED number of reduce rounds (round 4: ONE round that shrinks a bucket's partial sums 16x; rounds 2-3: two of 8x) before the fold reads
        // them twice - what uniform scalars need anyway (the top digit row of a 253-bit scalar fills only 2^(253 mod c) buckets,
        // thousands of entries each) - and the flattened-list fold takes whatever is left of a heavier bucket (all scalars
        // equal at 2^24: 2 048 partial sums in one bucket, 32 additions per lane of its row and column).  Nothing is read back:
        // an MSM of any size is one uninterrupted enqueue (round 2 sized the rounds by the largest bucket: a 4-byte copy and
        // a stream synchronisation between sort and accumulate).
        // ---- 5. accumulate
        phase_begin("msm_accumulate");
        {
            // a bucket of s entries is touched by at most (s - 1) / S + 2 segment threads
            const int env_rounds = tuning().reduce_rounds;
            if (!single_round) rounds = env_rounds < 0 ? 0 : (env_rounds > 8 ? 8 : env_rounds);
            // fused groups: optional reduce rounds (tuning fuse_reduce).  They bound what one fold workgroup can meet when an instance's
            // scalars are all equal (a 2^18-pair instance then leaves ~70 000 partial sums in ONE bucket: 1 100 dependent additions per
            // lane of its row) at the price of one more pass over the partial sums of well-behaved instances.
            if (mu && tuning().fuse_reduce > 0) rounds = tuning().fuse_reduce > 4 ? 4 : tuning().fuse_reduce;
            if (mu && tuning().fuse_reduce < 0) rounds = mu->K >= 8 ? 1 : 0;
            hipLaunchKernelGGL(msm_alloc_seg_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, st, boffp, c.cnt_a.as<uint32_t>(), nbt, pl.S);
            exclusive_scan_u32(st, c.cnt_a.as<uint32_t>(), c.start_a.as<uint32_t>(), (size_t)nbt + 1, c.scan_tmp.as<uint32_t>());
            const size_t nthreads = (E_max + pl.S - 1) / pl.S;
#ifdef SV_BENCH  // profiling builds only (wrong results): restrict the gather to the first 2^k bases to separate ALU time from HBM gather time
            static const uint32_t dbg_mask = getenv("SNARKVM_HIP_DEBUG_IDX_MASK") ? (uint32_t)strtoul(getenv("SNARKVM_HIP_DEBUG_IDX_MASK"), nullptr, 0) : 0xffffffffu;
#else
            constexpr uint32_t dbg_mask = 0xffffffffu;
#endif
            // (a 3-waves-per-SIMD build of this kernel - 168 VGPRs - and a software-pipelined gather were measured: no gain)
            const int prefetch_env = tuning().prefetch;  // 0: never, 1: single-round launches only, 2: always (lazy kernel: -2 .. 3 %)
            if constexpr (sizeof(F) == sizeof(fq_t)) {
                if (msm_lazy_on<F>()) {
                    // raw partial sums (208 B each) go to their own buffer; the dense conversion pass fills part_a for the tail
                    const size_t tmax = nthreads + nbt + 1;  // every thread leaves >= 1 partial sum, one more per bucket boundary inside its segment
                    // lazy tail: the raw partial sums ARE the tail's input (part_a holds T0_max >= tmax of them); else they go to their own
                    // buffer and the dense conversion pass fills part_a
                    if (!ltail) c.part_raw.ensure(tmax * sizeof(g1_lazy_partial_t));
                    g1_lazy_partial_t* raw_out = ltail ? c.part_a.as<g1_lazy_partial_t>() : c.part_raw.as<g1_lazy_partial_t>();
                    // One workgroup per CU (a dynamic LDS request no second workgroup fits beside) = one accumulate wave per SIMD with half
                    // of the register file and ~64 KB of LDS left free: single-round grids always; multi-round grids when
                    // tuning acc_one_wg is set - the sort and tail kernels of the NEXT instance of a pipelined batch (another
                    // lane's stream) then find room beside the accumulate waves instead of waiting for gaps between its rounds.
                    const int one_wg_env = tuning().acc_one_wg;
                    if ((single_round && prefetch_ok && prefetch_env) || prefetch_env >= 2)
                        hipLaunchKernelGGL((msm_accumulate_lazy_kernel<true>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256),
                                           (single_round && prefetch_ok) || one_wg_env ? msm_acc_lds() : 0, st, vbase, c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(),
                                           raw_out, nbt, pl.S, dbg_mask);
                    else
                        hipLaunchKernelGGL((msm_accumulate_lazy_kernel<false>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, vbase,
                                           c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(), raw_out, nbt, pl.S, dbg_mask);
                    if (!ltail)
                        hipLaunchKernelGGL(g1_partials_to_exact_kernel, dim3((unsigned)((tmax + 255) / 256)), dim3(256), 0, st,
                                           (const g1_lazy_partial_t*)raw_out, c.part_a.as<g1_xyzz_mem_t>(), (const uint32_t*)c.start_a.as<uint32_t>(), nbt);
                    goto accumulated;
                }
            } else {
#ifndef SV_NO_G2
                if (msm_lazy_on<F>() && tuning().pair2) {
                    // G2 on a lane pair (ffl2p.hip.h): two lanes per segment, two waves per SIMD; raw 512-byte partial sums, then the dense conversion
                    const size_t tmax = nthreads + nbt + 1;
                    c.part_raw.ensure(tmax * sizeof(g2_pair_partial_t));
                    hipLaunchKernelGGL((msm_accumulate_pair2_kernel<false>), dim3((unsigned)((2 * nthreads + 255) / 256)), dim3(256), 0, st, vbase, c.sorted.as<uint32_t>(),
                                       boffp, c.start_a.as<uint32_t>(), c.part_raw.as<g2_pair_partial_t>(), nbt, pl.S, dbg_mask);
                    hipLaunchKernelGGL(g2_pair_partials_to_exact_kernel, dim3((unsigned)((8 * tmax + 255) / 256)), dim3(256), 0, st,
                                       (const g2_pair_partial_t*)c.part_raw.as<g2_pair_partial_t>(), c.part_a.as<xyzz_mem_t<fq2_t>>(),
                                       (const uint32_t*)c.start_a.as<uint32_t>(), nbt);
                    goto accumulated;
                }
                if (msm_lazy_on<F>()) {  // G2 on the lazy Fq2 arithmetic of ffl2.hip.h: raw 416-byte partial sums, then the dense conversion
                    const size_t tmax = nthreads + nbt + 1;
                    c.part_raw.ensure(tmax * sizeof(g2_lazy_partial_t));
                    hipLaunchKernelGGL((msm_accumulate_lazy2_kernel<false>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, vbase,
                                       c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(), c.part_raw.as<g2_lazy_partial_t>(), nbt, pl.S, dbg_mask);
                    hipLaunchKernelGGL(g2_partials_to_exact_kernel, dim3((unsigned)((tmax + 255) / 256)), dim3(256), 0, st,
                                       (const g2_lazy_partial_t*)c.part_raw.as<g2_lazy_partial_t>(), c.part_a.as<xyzz_mem_t<fq2_t>>(),
                                       (const uint32_t*)c.start_a.as<uint32_t>(), nbt);
                    goto accumulated;
                }
#endif
            }
            if (single_round && prefetch_ok && prefetch_env)
                hipLaunchKernelGGL((msm_accumulate_seg_kernel<F, 1, true>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, vbase,
                                   c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(), c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S, dbg_mask);
            else
                hipLaunchKernelGGL((msm_accumulate_seg_kernel<F, 1, false>), dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, st, vbase,
                                   c.sorted.as<uint32_t>(), boffp, c.start_a.as<uint32_t>(), c.part_a.as<xyzz_mem_t<F>>(), nbt, pl.S, dbg_mask);
        accumulated:;
        }
        phase_end();
    }
    // 6.-9. reduce rounds, then the bucket merge (a chunk of a bigger MSM) or fold -> bit-plane sums -> (host) Horner: on the lazy arithmetic
    // when the accumulate kernel left raw partial sums (G1, tuning lazy_tail), else on F's exact arithmetic
    const bool flat = single_round || tuning().fold_flat != 0;
    if constexpr (sizeof(F) == sizeof(fq_t)) {
        if (ltail) {
            msm_reduce_and_tail<fqz_t>(c, pl, tg, nwin, nbt, rounds, T0_max, T1_max, sink, pd, host_planes, flat, phase_begin, phase_end);
            return pd;
        }
    }
    msm_reduce_and_tail<F>(c, pl, tg, nwin, nbt, rounds, T0_max, T1_max, sink, pd, host_planes, flat, phase_begin, phase_end);
    return pd;
}
// The tail of a chunked MSM: fold + bit planes over the bucket sink (every bucket holds L partial sums, one per lane).
template <class F>
static msm_pending_t msm_tail_from_sink(lane_t& c, size_t chunk_n, int window_bits, const msm_bucket_sink_t& sink, void* host_planes, int tables = 1,
                                        int table_bits = 0) {
    msm_pending_t pd;
    pd.planes = host_planes;
    const msm_plan_t pl = msm_make_plan(chunk_n, window_bits, tables, table_bits);
    const uint32_t nwin = (uint32_t)pl.W, nbt = nwin * pl.nb;
    if (nbt != sink.nbt) throw hip_failure{hipErrorInvalidValue, "msm: bucket sink does not match the plan", __LINE__};
    const msm_tail_geom_t tg = msm_tail_geometry(pl, nwin, pd, 0);
    c.start_a.ensure(((size_t)nbt + 1) * 4);
    c.cnt_a.ensure(((size_t)nbt + 1) * 4);
    hipLaunchKernelGGL(msm_sink_lists_kernel, dim3((nbt + 1 + 255) / 256), dim3(256), 0, c.stream, c.start_a.as<uint32_t>(), c.cnt_a.as<uint32_t>(), nbt, sink.L);
    pd.lazy = msm_lazy_tail_on<F>();  // the sink holds what the chunks' merges left: raw lazy points then
    c.phase_begin("msm_bucket_reduce");
    if constexpr (sizeof(F) == sizeof(fq_t)) {
        if (pd.lazy) {
            msm_tail_launch<fqz_t>(c, pl, tg, nwin, nbt, (const xyzz_mem_t<fqz_t>*)sink.acc, c.start_a.as<uint32_t>(), c.cnt_a.as<uint32_t>(), true, pd, host_planes);
            c.phase_end();
            return pd;
        }
    }
    msm_tail_launch<F>(c, pl, tg, nwin, nbt, (const xyzz_mem_t<F>*)sink.acc, c.start_a.as<uint32_t>(), c.cnt_a.as<uint32_t>(), true, pd, host_planes);
    c.phase_end();
    return pd;
}
// synchronous single MSM: run, wait, finish on the host into `out` (Jacobian memory image)
template <class F>
static void msm_run_sync(lane_t& c, const aff_mem_t<F>* d_bases, const uint4* d_scalars, size_t n, void* out, int window_bits,
                         const aff_mem_t<F>* d_bases1 = nullptr, size_t n0 = ~(size_t)0, int scalars_montgomery = 0, int tables = 1,
                         size_t table_stride = 0, int table_bits = 0) {
    // A lane borrowed from the calling thread's scope: MSMs the scope enqueued on it (in-stream, or with no further lane free) keep their bit planes in
    // `pin` from offset 0 until the scope's flush has read them - this call stages at offset 0 too (and ensure() may move the block): deliver them first.
    if (c.in_scope && c.pin_used) scope_flush();
    c.pin.ensure(msm_plane_bytes<F>());
    const msm_pending_t pd = msm_run<F>(c, d_bases, d_scalars, n, c.pin.p, window_bits, d_bases1, n0, scalars_montgomery, tables, table_stride, true, table_bits);
    HIP_TRY(hipStreamSynchronize(c.stream));
    const double t0 = host_now_ms();
    msm_accum_t<F>* acc = new msm_accum_t<F>();
    std::unique_ptr<msm_accum_t<F>> hold(acc);
    msm_collect<F>(*acc, pd);
    acc->finish(out);
    c.phase_host("msm_host_finish", host_now_ms() - t0);  // the Horner chain over the bit planes, on the calling thread
}

// Accumulation runs on the lazily reduced arithmetic: G1 on ffl.hip.h (tuning lazy=0: the exact kernel), G2 on ffl2.hip.h (tuning
// lazy2=0).  Process wide: every base slot an MSM of that group reads - registered tables and the staging of table-less calls - then
// holds form406.
template <class F>
static bool msm_lazy_on() {
    return sizeof(F) == sizeof(fq_t) ? tuning().lazy != 0 : tuning().lazy2 != 0;
}
template <class F>
static void convert_bases(lane_t& c, const uint8_t* d_in, size_t stride, size_t n, aff_mem_t<F>* d_out, hipStream_t st = nullptr, bool for_msm = false) {
    if (!n) return;
    const int form406 = for_msm && msm_lazy_on<F>() ? 1 : 0;
    hipLaunchKernelGGL((convert_bases_kernel<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st ? st : c.stream, d_in, stride, n, d_out, form406);
    HIP_TRY(hipGetLastError());
}
// the last step of a G1 registration: every slot of every table, exact internal form -> form406
static void bases_to_lazy_form(lane_t& c, g1_aff_mem_t* d, size_t slots) {
    if (!slots || !msm_lazy_on<fq_t>()) return;
    hipLaunchKernelGGL(g1_bases_to_form406_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, c.stream, d, slots);
    HIP_TRY(hipGetLastError());
}
#ifndef SV_NO_G2
static void bases_to_lazy_form(lane_t& c, aff_mem_t<fq2_t>* d, size_t slots) {
    if (!slots || !msm_lazy_on<fq2_t>()) return;
    hipLaunchKernelGGL(g2_bases_to_form406_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, c.stream, d, slots);
    HIP_TRY(hipGetLastError());
}
#endif

// Precomputed base tables of a registered vector: table j = 2^(table_bits * j) * P_i, from table j - 1 (msm.hip.h).  Long
// vectors give every thread a run of points that share one inversion; a run of 1 keeps small vectors parallel.
template <class F>
static void precompute_tables_run(lane_t& c, aff_mem_t<F>* d, size_t n, int tables, int table_bits) {
    if (tables <= 1 || !n) return;
    int run = (int)(n >> 16);
    run = run < 1 ? 1 : (run > PRE_RUN ? PRE_RUN : run);
    const size_t slab = n < PRE_SLAB ? n : PRE_SLAB;
    c.gen_pts.ensure(4 * slab * sizeof(typename F::mem_t));
    for (int j = 1; j < tables; j++)
        for (size_t lo = 0; lo < n; lo += PRE_SLAB) {
            const size_t cnt = n - lo < PRE_SLAB ? n - lo : PRE_SLAB;
            const size_t threads = (cnt + run - 1) / run;
            hipLaunchKernelGGL((precompute_table_kernel<F>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream, d + (size_t)(j - 1) * n + lo,
                               d + (size_t)j * n + lo, cnt, table_bits, run, (typename F::mem_t*)c.gen_pts.p);
        }
    HIP_TRY(hipGetLastError());
}

// lanes a batch cycles through per device: more lanes hide more of the latency-bound tail of small MSMs, fewer keep the
// workspace footprint of big ones down (a 2^24 lane holds ~4 GB)
static int batch_lanes(size_t npoints) {
    const int env = tuning().lanes;
    int l = env > 0 ? env : (npoints >= ((size_t)1 << 20) ? 3 : 8);  // measured: 8 lanes +7 % below 2^20, no gain above
    return l < 1 ? 1 : (l > device_t::LANES ? device_t::LANES : l);
}
static constexpr size_t MSM_SPLIT_MIN = (size_t)1 << 18;  // pairs per device below which a point-range split costs more than it saves
static size_t msm_chunk_pairs() {  // pairs per upload / compute chunk of an MSM whose bases arrive from the host
    const int lg = tuning().msm_chunk_lg;
    return (size_t)1 << (lg < 16 ? 16 : (lg > 30 ? 30 : lg));
}
// pairs per scalar chunk of a host-scalar MSM over registered bases (tuning scalar_chunk_lg, default 2^22: the tail of
// a chunk costs < 1 ms, its upload 2.4 ms)
static size_t msm_scalar_chunk_pairs() {
    const int lg = tuning().scalar_chunk_lg;
    return (size_t)1 << (lg < 18 ? 18 : lg > 30 ? 30 : lg);
}

// `count` chunks of one call on the lanes of `lg` (a ring: chunk j uses lane j mod L).  A dedicated uploader thread runs
// upload(j, stream) - host-blocking copies of pageable caller memory - chunk after chunk, so PCIe stays busy back to back
// while the calling thread runs compute(j) (kernel launches plus the read-back that sizes the reduce rounds) for the chunks
// that have arrived.  up[j]: "chunk j is on the device" (event on the lane's second stream); used[j]: "the work of chunk j
// has consumed the lane's staging buffers" (event on the lane's stream).
template <class Upload, class Compute>
static void lane_ring_run(lane_guard& lg, size_t count, Upload&& upload, Compute&& compute, int trace, double t_begin) {
    const size_t L = lg.lanes.size();
    const int phys = lg.lanes[0]->dev->physical;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> uploaded(count, 0), enqueued(count, 0);
    std::vector<hipEvent_t> up(count), used(count);
    for (size_t j = 0; j < count; j++) {
        up[j] = lg.lanes[j % L]->new_event();
        used[j] = lg.lanes[j % L]->new_event();
    }
    std::exception_ptr up_err, cp_err;
    std::thread uploader([&] {
        try {
            HIP_TRY(hipSetDevice(phys));
            for (size_t j = 0; j < count; j++) {
                lane_t& c = *lg.lanes[j % L];
                if (j >= L) {  // the lane's previous chunk must have been consumed on the GPU
                    char state;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return enqueued[j - L] != 0; });
                        state = enqueued[j - L];
                    }
                    if (state == 2) break;  // the compute side failed
                    HIP_TRY(hipEventSynchronize(used[j - L]));
                }
                const double t0 = host_now_ms();
                upload(j, c.alt);
                HIP_TRY(hipEventRecord(up[j], c.alt));
                if (trace) fprintf(stderr, "[snarkvm_hip] chunk %zu dev %d lane %d: uploaded t+%.2f .. t+%.2f ms\n", j, c.dev->logical, c.index, t0 - t_begin, host_now_ms() - t_begin);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    uploaded[j] = 1;
                }
                cv.notify_all();
            }
        } catch (...) {
            up_err = std::current_exception();
            std::lock_guard<std::mutex> lk(mu);
            for (auto& u : uploaded) u = 2;
            cv.notify_all();
        }
    });
    try {
        for (size_t j = 0; j < count; j++) {
            lane_t& c = *lg.lanes[j % L];
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return uploaded[j] != 0; });
                if (uploaded[j] == 2) break;
            }
            HIP_TRY(hipStreamWaitEvent(c.stream, up[j], 0));
            const double t0 = host_now_ms();
            compute(j);
            HIP_TRY(hipEventRecord(used[j], c.stream));
            if (trace) fprintf(stderr, "[snarkvm_hip] chunk %zu: enqueued t+%.2f .. t+%.2f ms\n", j, t0 - t_begin, host_now_ms() - t_begin);
            {
                std::lock_guard<std::mutex> lk(mu);
                enqueued[j] = 1;
            }
            cv.notify_all();
        }
    } catch (...) {
        cp_err = std::current_exception();
        std::lock_guard<std::mutex> lk(mu);
        for (auto& e : enqueued) e = 2;
        cv.notify_all();
    }
    uploader.join();
    if (cp_err || up_err) {  // nothing of this call may still be in flight when the lanes go back to the pool
        for (lane_t* l : lg.lanes) {
            (void)hipStreamSynchronize(l->alt);
            (void)hipStreamSynchronize(l->stream);
        }
        std::rethrow_exception(cp_err ? cp_err : up_err);
    }
}

// The reference's FFI MSM (host bases, host scalars, no registration): G1: F = fq_t (stride >= 104), G2: F = fq2_t (>= 200).
// A big call is cut into point-range chunks that are dealt round-robin to the devices (the reference's per-GPU slices,
// snarkvm.cu:254-270) and, on each device, to a ring of up to three lanes: an uploader thread copies chunk after chunk into
// the lanes' staging buffers without a pause while the calling thread converts, sorts and accumulates the chunks that have
// arrived - the upload (PCIe, ~2.4 ns per pair) is the critical path and the compute (~2.4 ns per pair without precomputed
// tables) hides behind it.  Every chunk leaves only its bit-plane sums; they are added on the host before the one Horner chain.
template <class F>
static void msm_host_chunked(void* out, const void* points, size_t npoints, const void* scalars, size_t stride) {
    const size_t min_stride = 2 * sizeof(typename F::mem_t) + 8;
    if (stride < min_stride || (stride & 7)) throw hip_failure{hipErrorInvalidValue, "msm: bad ffi_affine_sz for this curve", __LINE__};
    scope_flush();  // (the per-device workers of a multi-GPU call are other threads)
    const int nd = g_rt.ndev();
    static const int trace = getenv("SNARKVM_HIP_TRACE") ? atoi(getenv("SNARKVM_HIP_TRACE")) : 0;
    const double t_begin = host_now_ms();
    size_t nchunks = npoints < 2 * MSM_SPLIT_MIN ? 1 : (npoints + msm_chunk_pairs() - 1) / msm_chunk_pairs();
    if (nchunks == 1 && npoints >= 2 * MSM_SPLIT_MIN && (nd > 1 || npoints >= ((size_t)1 << 20))) nchunks = 2;  // 2^20: 7.2 -> 7.0 ms, 2^21: 13.0 -> 12.4
    // chunk boundaries.  The upload is the critical path and nothing of the last chunk can start before its last byte has
    // arrived, so the LAST chunk is cut again into 1/2, 1/4, 1/4 (tuning taper): what is exposed after the final upload is the
    // computation of a quarter chunk plus the one tail (round 3: a whole 2^21-pair chunk, 8.4 ms of the 62 at 2^24).
    std::vector<size_t> bound;
    for (size_t i = 0; i <= nchunks; i++) bound.push_back(npoints * i / nchunks);
    const bool taper = tuning().taper != 0 && nchunks >= 3 && bound[nchunks] - bound[nchunks - 1] >= MSM_SPLIT_MIN;
    if (taper) {
        const size_t lo = bound[nchunks - 1], len = npoints - lo;
        bound.back() = lo + len / 2;
        bound.push_back(lo + len / 2 + len / 4);
        bound.push_back(npoints);
        nchunks += 2;
    }
    // ... and nothing can be computed before the FIRST chunk has arrived: at 2^24 the table-less arithmetic (16 digit rows per
    // point, ~47 ms of GPU time) outlasts the 41 ms of upload, so the 5 ms the GPU idles through the first 2^21-pair upload are 5 ms
    // of the call.  The first chunk is cut into 1/2^r, 1/2^r, 1/2^(r-1), ..., 1/2 (tuning ramp = r; pieces of >= 2^17 pairs).
    int ramp = tuning().taper != 0 && nchunks >= 2 ? tuning().ramp : 0;
    while (ramp > 0 && (bound[1] >> ramp) < ((size_t)1 << 17)) ramp--;
    if (ramp > 0) {
        const size_t len = bound[1];
        std::vector<size_t> front;
        size_t pos = len >> ramp;
        front.push_back(pos);
        for (int k = ramp; k >= 1; k--) {
            pos += len >> k;
            front.push_back(k == 1 ? len : pos);
        }
        bound.erase(bound.begin() + 1);
        bound.insert(bound.begin() + 1, front.begin(), front.end());
        nchunks += (size_t)ramp;
    }
    const int ndu = (int)(nchunks < (size_t)nd ? nchunks : (size_t)nd);
    std::unique_ptr<msm_accum_t<F>> acc(new msm_accum_t<F>());
    std::mutex acc_mu;
    std::vector<int> devs;
    if (ndu == 1)
        devs.push_back(-1);
    else
        for (int d = 0; d < ndu; d++) devs.push_back(d);
    const size_t slot = msm_plane_bytes<F>();
    for_each_device(devs, [&](int dev) {
        std::vector<size_t> mine;
        for (size_t i = (dev < 0 ? 0 : (size_t)dev); i < nchunks; i += (size_t)ndu) mine.push_back(i);
        lane_guard lg;
        const int ring = tuning().ring_lanes < 2 ? 2 : (tuning().ring_lanes > device_t::LANES ? device_t::LANES : tuning().ring_lanes);
        lg.acquire(dev, mine.size() > (size_t)ring ? ring : (int)mine.size());
        const int L = (int)lg.lanes.size();
        // Several chunks on this device: they share ONE set of buckets (16-bit windows whatever the chunk length) - every chunk
        // adds its per-bucket partial sums to a sink and the fold / bit-plane tail runs once, after the last chunk, instead of once
        // per chunk (~1.5 ms each at 2^21 pairs x 16 windows).  tuning taper=0: every chunk runs its own tail (round 3).
        const bool use_sink = tuning().taper != 0 && mine.size() >= 2;
        const int chunk_c = use_sink ? 16 : 0;
        std::vector<msm_pending_t> pend(use_sink ? 1 : mine.size());
        size_t max_cnt = 0;
        for (size_t j = 0; j < mine.size(); j++) {
            const size_t cnt = bound[mine[j] + 1] - bound[mine[j]];
            max_cnt = cnt > max_cnt ? cnt : max_cnt;
        }
        const size_t aff_bytes = (max_cnt * sizeof(aff_mem_t<F>) + 255) & ~(size_t)255;
        for (int l = 0; l < L; l++) {
            lane_t& c = *lg.lanes[l];
            c.begin_call();
            c.pin.ensure(slot * (use_sink ? 1 : (mine.size() + L - 1) / L));
            c.bases_tmp.ensure(aff_bytes + max_cnt * stride);
            c.scalars_tmp.ensure(max_cnt * 32);
        }
        msm_bucket_sink_t sink;
        hipEvent_t sink_ready = nullptr;
        if (use_sink) {
            lane_t& c0 = *lg.lanes[0];
            const msm_plan_t pl = msm_make_plan(max_cnt, chunk_c, 1, 0);
            sink.nbt = (uint32_t)pl.W * pl.nb;
            sink.L = (uint32_t)L;
            const size_t bytes = (size_t)sink.nbt * L * msm_partial_bytes<F>();
            c0.sink_acc.ensure(bytes);
            sink.acc = c0.sink_acc.p;
            HIP_TRY(hipMemsetAsync(sink.acc, 0, bytes, c0.stream));  // all-zero = the point at infinity
            sink_ready = c0.new_event();
            HIP_TRY(hipEventRecord(sink_ready, c0.stream));
            for (int l = 1; l < L; l++) HIP_TRY(hipStreamWaitEvent(lg.lanes[l]->stream, sink_ready, 0));
        }
        auto chunk_lo = [&](size_t j) { return bound[mine[j]]; };
        auto chunk_cnt = [&](size_t j) { return bound[mine[j] + 1] - bound[mine[j]]; };
        // upload of chunk j into its lane's staging buffers (host-blocking: the caller's memory is pageable)
        auto upload = [&](size_t j, hipStream_t st) {
            lane_t& c = *lg.lanes[j % L];
            uint8_t* raw = c.bases_tmp.template as<uint8_t>() + aff_bytes;
            HIP_TRY(hipMemcpyAsync(raw, (const uint8_t*)points + chunk_lo(j) * stride, chunk_cnt(j) * stride, hipMemcpyHostToDevice, st));
            HIP_TRY(hipMemcpyAsync(c.scalars_tmp.p, (const uint8_t*)scalars + chunk_lo(j) * 32, chunk_cnt(j) * 32, hipMemcpyHostToDevice, st));
        };
        auto compute = [&](size_t j, bool prof) {
            lane_t& c = *lg.lanes[j % L];
            uint8_t* raw = c.bases_tmp.template as<uint8_t>() + aff_bytes;
            if (prof) c.phase_begin("msm_convert_bases");
            convert_bases<F>(c, raw, stride, chunk_cnt(j), c.bases_tmp.template as<aff_mem_t<F>>(), nullptr, true);
            if (prof) c.phase_end();
            msm_bucket_sink_t mine_sink = sink;
            mine_sink.slot = (uint32_t)(j % L);
            const msm_pending_t pd = msm_run<F>(c, c.bases_tmp.template as<aff_mem_t<F>>(), c.scalars_tmp.template as<uint4>(), chunk_cnt(j),
                                                c.pin.template as<uint8_t>() + (use_sink ? 0 : slot * (j / L)), chunk_c, nullptr, ~(size_t)0, 0, 1, 0, prof, 0, nullptr,
                                                use_sink ? &mine_sink : nullptr);
            if (!use_sink) pend[j] = pd;
        };
        if (mine.size() == 1) {
            lane_t& c = *lg.lanes[0];
            c.phase_begin("msm_h2d");
            upload(0, c.stream);
            c.phase_end();
            compute(0, true);
        } else {
            lane_ring_run(lg, mine.size(), upload, [&](size_t j) { compute(j, false); }, trace, t_begin);
        }
        if (use_sink) {  // every lane's last merge, then the one tail on lane 0
            lane_t& c0 = *lg.lanes[0];
            for (int l = 1; l < L; l++) {
                hipEvent_t e = lg.lanes[l]->new_event();
                HIP_TRY(hipEventRecord(e, lg.lanes[l]->stream));
                HIP_TRY(hipStreamWaitEvent(c0.stream, e, 0));
            }
            pend[0] = msm_tail_from_sink<F>(c0, max_cnt, chunk_c, sink, c0.pin.p);
        }
        for (int l = 0; l < L; l++) {
            HIP_TRY(hipStreamSynchronize(lg.lanes[l]->alt));
            HIP_TRY(hipStreamSynchronize(lg.lanes[l]->stream));
        }
        const double t_sync = host_now_ms();
        {
            std::lock_guard<std::mutex> lk(acc_mu);
            for (auto& pd : pend) msm_collect<F>(*acc, pd);
        }
        if (trace) fprintf(stderr, "[snarkvm_hip] all chunks done at t+%.2f ms, planes collected in %.2f ms\n", t_sync - t_begin, host_now_ms() - t_sync);
        for (int l = 0; l < L; l++) lg.lanes[l]->end_call();
    });
    acc->finish(out);
}

// A batch of independent MSMs over one registered base vector, fanned out over devices x lanes (see
// snarkvm_hip_msm_registered_batch).  Every request names its own 144 / 288-byte output (Jacobian memory image).
//
// Instances of up to 2^18 pairs over windowed tables (one bucket window per table set: the geometries registered for proof-sized
// commitments, 17 x 15 / 16 x 16 bit) are FUSED: the instances a device received travel as groups through ONE launch sequence
// each (msm_sort.hip.h: instance id = top key of the radix partition, one accumulate grid, one fold and one bit-plane launch
// for the whole group, then one host finish per instance).  A prover round is such a batch (sonic_pc/mod.rs:186-245: the
// commitments of a round are independent MSMs over one committer key).  Per instance the fused run leaves fewer partial sums
// for the tail (the accumulate grid is sized for the group, not per instance) and ~25 launches are shared by the group.
static constexpr size_t MSM_FUSE_MAX_PAIRS = (size_t)1 << 18;   // per instance
static constexpr size_t MSM_FUSE_MAX_ENTRIES = (size_t)1 << 26;  // digit entries (tables x padded pairs) per fused group
static bool msm_fuse_enabled() {
    return tuning().fuse_batch != 0;  // A/B switch
}
static size_t msm_fuse_max_k() {
    const int k = tuning().fuse_max_k;
    return (size_t)(k < 2 ? 2 : (k > 256 ? 256 : k));
}
// one MSM of a batch: bases [off0, off0 + n0) followed by [off1, off1 + n1) (KZG10's hiding range; n1 = 0: none) against n0 + n1
// consecutive scalars; `out`: where its Jacobian memory image goes
struct msm_req_t {
    size_t off0 = 0, n0 = 0, off1 = 0, n1 = 0;
    const void* scalars = nullptr;
    void* out = nullptr;
};
// fn(i) for i < n on up to `max_threads` host threads (the calling thread is one of them).  Used for the Horner finishes of a fused
// group: 25 - 35 us each on one core, 64 of them per group.
template <class Fn>
static void host_parallel_for(size_t n, int max_threads, Fn fn) {
    size_t T = n / 4;
    if (T > (size_t)max_threads) T = (size_t)max_threads;
    if (T <= 1) {
        for (size_t i = 0; i < n; i++) fn(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::exception_ptr> errs(T);
    auto work = [&](size_t t) {
        try {
            for (size_t i; (i = next.fetch_add(1)) < n;) fn(i);
        } catch (...) {
            errs[t] = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    for (size_t t = 1; t < T; t++) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    for (auto& e : errs)
        if (e) std::rethrow_exception(e);
}
// the handle's geometry admits fused multi-instance groups: one bucket window of 12 .. 16 bits per table set, slots addressable in 31 bits
template <class F>
static bool msm_handle_fusable(const bases_handle_t<F>& h, int window_bits) {
    if (!msm_fuse_enabled() || h.tables <= 1 || h.table_bits < 12 || h.table_bits > 16 || (window_bits != 0 && window_bits != h.table_bits) ||
        (size_t)h.tables * h.n >= ((size_t)1 << 31) || h.n >= ((size_t)1 << 31))
        return false;
    const msm_plan_t pl = msm_make_plan(SORT_TILE, h.table_bits, h.tables, h.table_bits);  // what msm_run will ask of a fused group
    return pl.W == 1 && pl.c == h.table_bits;
}
// planes a fused instance leaves: two tail windows (row sums, column sums) of fold_m + 1 bits, fold_m = table_bits / 2 (msm_run)
template <class F>
static size_t msm_fuse_planes(const bases_handle_t<F>& h) {
    return 2 * ((size_t)h.table_bits / 2 + 1);
}
static size_t msm_padded(size_t n) { return (n + SORT_TILE - 1) / SORT_TILE * SORT_TILE; }
// the jobs the instances `mine` (indices into req) make on one device: fused groups of small instances (in order of appearance), single
// instances otherwise
template <class F>
static std::vector<std::vector<size_t>> msm_make_jobs(const bases_handle_t<F>& h, const msm_req_t* req, const std::vector<size_t>& mine, bool fusable_handle) {
    std::vector<std::vector<size_t>> jobs;
    std::vector<size_t> group;
    size_t group_entries = 0;
    auto flush = [&] {
        if (!group.empty()) jobs.push_back(group);  // a lone instance takes the single-MSM path (its own planner)
        group.clear();
        group_entries = 0;
    };
    for (size_t k : mine) {
        const size_t tot = req[k].n0 + req[k].n1;
        const bool small = fusable_handle && tot > 0 && tot <= MSM_FUSE_MAX_PAIRS;
        if (!small) {
            jobs.push_back({k});
            continue;
        }
        const size_t e = msm_padded(tot) * (size_t)h.tables;
        if (!group.empty() && (group.size() >= msm_fuse_max_k() || group_entries + e > MSM_FUSE_MAX_ENTRIES)) flush();
        group.push_back(k);
        group_entries += e;
    }
    flush();
    return jobs;
}
// pinned bytes job `job` needs: its bit planes, then (fused groups) its instance table
template <class F>
static void msm_job_staging(const bases_handle_t<F>& h, const std::vector<size_t>& job, size_t& plane_bytes, size_t& table_bytes) {
    const size_t K = job.size();
    plane_bytes = K > 1 ? K * msm_fuse_planes(h) * msm_point_bytes<F>() : msm_plane_bytes<F>();
    table_bytes = K > 1 ? ((K + 1) * sizeof(msm_inst_t) + 255) / 256 * 256 : 0;
}
// Enqueue job `job` on lane c (device `dev`): host_planes / tab = its pinned staging (valid until the planes have been collected).
template <class F>
static msm_pending_t msm_enqueue_job(lane_t& c, const bases_handle_t<F>& h, int dev, const msm_req_t* req, const std::vector<size_t>& job, uint8_t* host_planes,
                                     msm_inst_t* tab, int scalars_on_device, int scalars_montgomery, int window_bits, hipEvent_t scalars_read = nullptr) {
    if (job.size() == 1) {
        const msm_req_t& r = req[job[0]];
        const size_t n = r.n0 + r.n1;
        const uint4* d_sc = (const uint4*)r.scalars;
        if (!scalars_on_device && n) {
            // the lane's previous instance may still be reading its scalar buffer: stream order serialises the copy behind it
            c.scalars.ensure(n * 32);
            HIP_TRY(hipMemcpyAsync(c.scalars.p, r.scalars, n * 32, hipMemcpyHostToDevice, c.stream));
            d_sc = c.scalars.template as<uint4>();
        }
        return msm_run<F>(c, h.d[dev] + r.off0, d_sc, n, host_planes, window_bits, r.n1 ? h.d[dev] + r.off1 : nullptr, r.n1 ? r.n0 : ~(size_t)0, scalars_montgomery,
                          h.tables, h.n, false, h.table_bits, nullptr, nullptr, scalars_read);
    }
    // fused group: instance table (pinned -> device), scalars of host callers packed into the lane's scalar buffer
    const size_t K = job.size();
    size_t npad = 0, sc_bytes = 0;
    for (size_t q = 0; q < K; q++) sc_bytes += (req[job[q]].n0 + req[job[q]].n1) * 32;
    if (!scalars_on_device) c.scalars.ensure(sc_bytes);
    size_t sc_off = 0;
    for (size_t q = 0; q < K; q++) {
        const msm_req_t& r = req[job[q]];
        const size_t n = r.n0 + r.n1;
        msm_inst_t& in = tab[q];
        in.n = (uint32_t)n;
        in.n0 = r.n1 ? (uint32_t)r.n0 : in.n;
        in.off0 = (uint32_t)r.off0;
        in.off1 = r.n1 ? (uint32_t)r.off1 : 0u;
        in.pstart = (uint32_t)npad;
        in.ptiles = (uint32_t)(msm_padded(n) / SORT_TILE);
        npad += msm_padded(n);
        if (scalars_on_device) {
            in.scalars = (const uint4*)r.scalars;
        } else {
            uint8_t* dst = c.scalars.template as<uint8_t>() + sc_off;
            HIP_TRY(hipMemcpyAsync(dst, r.scalars, n * 32, hipMemcpyHostToDevice, c.stream));
            in.scalars = (const uint4*)dst;
            sc_off += n * 32;
        }
    }
    tab[K] = msm_inst_t{nullptr, 0, 0, 0, 0, (uint32_t)npad, 0};  // sentinel
    c.poly[4].ensure((K + 1) * sizeof(msm_inst_t));
    HIP_TRY(hipMemcpyAsync(c.poly[4].p, tab, (K + 1) * sizeof(msm_inst_t), hipMemcpyHostToDevice, c.stream));
    msm_multi_t mu;
    mu.d_inst = c.poly[4].template as<msm_inst_t>();
    mu.K = (uint32_t)K;
    mu.npad = npad;
    mu.hn = h.n;
    mu.plane_capacity = K * msm_fuse_planes(h);  // checked by msm_run BEFORE it enqueues the copy into the staging area
    return msm_run<F>(c, h.d[dev], nullptr, npad, host_planes, 0, nullptr, ~(size_t)0, scalars_montgomery, h.tables, h.n, false, h.table_bits, &mu, nullptr, scalars_read);
}
// the host finish of job `job` (its planes have arrived): one Horner chain per instance, the instances of a fused group on several threads
template <class F>
static void msm_finish_job(const msm_req_t* req, const std::vector<size_t>& job, const msm_pending_t& pd, int max_threads = 8) {
    if (job.size() == 1) {
        std::unique_ptr<msm_accum_t<F>> acc(new msm_accum_t<F>());
        msm_collect<F>(*acc, pd);
        acc->finish(req[job[0]].out);
        return;
    }
    host_parallel_for(job.size(), max_threads, [&](size_t q) {
        std::unique_ptr<msm_accum_t<F>> acc(new msm_accum_t<F>());
        msm_collect_inst<F>(*acc, pd, (int)q);
        acc->finish(req[job[q]].out);
    });
}
static void msm_check_requests(size_t hn, const msm_req_t* req, size_t count) {
    for (size_t k = 0; k < count; k++) {
        if (req[k].off0 + req[k].n0 > hn || (req[k].n1 && req[k].off1 + req[k].n1 > hn))
            throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: range exceeds the registered bases", __LINE__};
        if ((req[k].n0 + req[k].n1) && !req[k].scalars) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: null scalar vector", __LINE__};
        if (!req[k].out) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: null output", __LINE__};
    }
}
template <class F>
static void msm_batch_run(const bases_handle_t<F>& h, const msm_req_t* req, size_t count, int scalars_on_device, int scalars_montgomery, int window_bits) {
    auto total = [&](size_t k) { return req[k].n0 + req[k].n1; };
    scope_flush();  // the per-device workers below are other threads: what they read must be complete (and they cannot flush this thread's scope)
    const int nd = g_rt.ndev();
    std::vector<std::vector<size_t>> per_dev(nd);
    size_t largest = 0;
    msm_check_requests(h.n, req, count);
    for (size_t k = 0; k < count; k++) {
        largest = total(k) > largest ? total(k) : largest;
        int dev = (int)(k % (size_t)nd);
        if (scalars_on_device && total(k)) {
            dev = g_rt.device_of(req[k].scalars);
            if (dev < 0) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: scalars are not on a device in use", __LINE__};
        }
        per_dev[dev].push_back(k);
    }
    const int nlanes = batch_lanes(largest);
    std::vector<int> devs;
    for (int d = 0; d < nd; d++)
        if (!per_dev[d].empty()) devs.push_back(d);
    const bool fusable_handle = msm_handle_fusable(h, window_bits);
    for_each_device(devs, [&](int dev) {
        const std::vector<std::vector<size_t>> jobs = msm_make_jobs(h, req, per_dev[dev], fusable_handle);
        lane_guard lg;
        lg.acquire(dev, nlanes < (int)jobs.size() ? nlanes : (int)jobs.size());
        const int L = (int)lg.lanes.size();
        std::vector<msm_pending_t> pend(jobs.size());
        std::vector<hipEvent_t> done(jobs.size());
        // pinned staging per lane: the bit planes of its jobs, then the instance tables of its fused jobs
        std::vector<size_t> plane_off(jobs.size()), table_off(jobs.size()), lane_bytes(L, 0);
        for (size_t i = 0; i < jobs.size(); i++) {
            size_t pb, tb;
            msm_job_staging(h, jobs[i], pb, tb);
            plane_off[i] = lane_bytes[i % L];
            table_off[i] = plane_off[i] + pb;
            lane_bytes[i % L] += pb + tb;
        }
        for (int l = 0; l < L; l++) {
            lg.lanes[l]->begin_call();
            lg.lanes[l]->pin.ensure(lane_bytes[l] ? lane_bytes[l] : 256);
        }
        for (size_t i = 0; i < jobs.size(); i++) {
            lane_t& c = *lg.lanes[i % L];
            pend[i] = msm_enqueue_job<F>(c, h, dev, req, jobs[i], c.pin.template as<uint8_t>() + plane_off[i], (msm_inst_t*)(c.pin.template as<uint8_t>() + table_off[i]),
                                         scalars_on_device, scalars_montgomery, window_bits);
            done[i] = c.new_event();
            HIP_TRY(hipEventRecord(done[i], c.stream));
        }
        // the host finishes job i while the GPU works on the later ones; the instances of a fused group on several host threads
        for (size_t i = 0; i < jobs.size(); i++) {
            HIP_TRY(hipEventSynchronize(done[i]));
            msm_finish_job<F>(req, jobs[i], pend[i]);
        }
        for (int l = 0; l < L; l++) lg.lanes[l]->end_call();
    });
}
// An MSM call of a thread inside an SNARKVM_HIP_SCOPE_ASYNC_MSM scope, scalars in the scope device's memory: the instances are only ENQUEUED -
// on the next of the scope's MSM lanes, behind everything the scope's stream has been given so far - and the scope's stream in turn waits
// until the MSM has read its scalars (the caller may reuse those buffers in its next calls).  The outputs are written by the scope's flush
// (snarkvm_hip_scope_end, or any call that has to wait for the scope).  Returns false when the call does not qualify (the caller then
// takes the synchronous path).
template <class F>
static bool msm_scope_enqueue(const bases_handle_t<F>& h, const msm_req_t* req, size_t count, int scalars_on_device, int scalars_montgomery, int window_bits) {
    thread_scope_t& sc = tl_scope();
    if (!sc.lane || !(sc.flags & SNARKVM_HIP_SCOPE_ASYNC_MSM) || !scalars_on_device || !count || g_rt.profiling.load(std::memory_order_relaxed)) return false;
    msm_check_requests(h.n, req, count);
    device_t* d = sc.lane->dev;
    for (size_t k = 0; k < count; k++) {
        if (!(req[k].n0 + req[k].n1)) continue;
        const int dk = g_rt.device_of(req[k].scalars);
        if (dk < 0) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: scalars are not on a device in use", __LINE__};
        if (g_rt.devs[dk]->physical != d->physical) return false;  // another GPU: the synchronous path sorts that out
    }
    if ((size_t)d->logical >= h.d.size() || !h.d[d->logical]) return false;
    // the lane: the scope's MSM lanes in turn; a further one is added while fewer than SCOPE_AUX_MAX are held and one is free.
    // SNARKVM_HIP_SCOPE_MSM_IN_STREAM: the scope's own lane, in order with its transforms - no event hand-off between streams (what a caller
    // wants who collects this MSM before it issues anything else: nothing could run beside it anyway)
    const bool in_stream = (sc.flags & SNARKVM_HIP_SCOPE_MSM_IN_STREAM) != 0;
    if (!in_stream && sc.naux < SCOPE_AUX_MAX && !sc.aux_exhausted && (sc.naux == 0 || sc.aux_rr >= (unsigned)sc.naux)) {
        if (lane_t* l = d->take_for_scope(false)) {
            l->begin_call();
            l->pin_used = 0;
            l->scope_events_used = 0;
            sc.aux[sc.naux++] = l;
        } else {
            sc.aux_exhausted = true;
        }
    }
    lane_t& c = (sc.naux && !in_stream) ? *sc.aux[sc.aux_rr++ % (unsigned)sc.naux] : *sc.lane;
    std::vector<size_t> all(count);
    for (size_t k = 0; k < count; k++) all[k] = k;
    const std::vector<std::vector<size_t>> jobs = msm_make_jobs(h, req, all, msm_handle_fusable(h, window_bits));
    size_t need = 0;
    for (const auto& j : jobs) {
        size_t pb, tb;
        msm_job_staging(h, j, pb, tb);
        need += pb + tb;
    }
    // a scope's staging area is at least 1 MB from its first MSM on the lane: how far `pin_used` climbs inside a scope depends on when the pending MSMs happen to be
    // delivered (scope_collect hands over what has arrived) - an area sized by other paths (a few KB of planes) would be outgrown in SOME replay of a warmed shape
    // only, with a flush of the whole scope in front of the allocation
    if (c.pin_used == 0 && c.pin.cap < ((size_t)1 << 20)) c.pin.ensure((size_t)1 << 20);
    if (c.pin_used + need > c.pin.cap) {  // staging full: collect what is pending (its planes live there), then start over with a bigger area
        if (c.pin_used) scope_flush();    // (first use of a lane: nothing of the scope is in its area - no reason to deliver other MSMs early)
        c.pin.ensure(need > ((size_t)1 << 20) ? need : (size_t)1 << 20);
    }
    // shared by the finish closures: the requests (outputs) of this call
    std::shared_ptr<std::vector<msm_req_t>> rq(new std::vector<msm_req_t>(req, req + count));
    if (&c != sc.lane) {
        hipEvent_t ready = sc.lane->scope_event();
        HIP_TRY(hipEventRecord(ready, sc.lane->stream));
        HIP_TRY(hipStreamWaitEvent(c.stream, ready, 0));
    }
    // All or nothing: a failure on job k > 0 must not leave jobs 0 .. k-1 pending - the caller sees an error and may free or reuse the `out`
    // buffers their finishes would write at scope_end.  What this call added is taken back (after the lanes have drained: the kernels already
    // enqueued write into the staging that is being handed back).
    const size_t pending0 = sc.pending.size(), pin0 = c.pin_used, ev0 = c.scope_events_used;
    try {
        for (const auto& j : jobs) {
            size_t pb, tb;
            msm_job_staging(h, j, pb, tb);
            uint8_t* planes = c.pin.template as<uint8_t>() + c.pin_used;
            msm_inst_t* tab = (msm_inst_t*)(planes + pb);
            c.pin_used += pb + tb;
            // SNARKVM_HIP_SCOPE_STABLE_INPUTS: the caller leaves the scalar vectors alone until the scope ends - the scope's stream does not wait
            hipEvent_t read = (&c != sc.lane && !(sc.flags & SNARKVM_HIP_SCOPE_STABLE_INPUTS)) ? c.scope_event() : nullptr;
            const msm_pending_t pd = msm_enqueue_job<F>(c, h, d->logical, rq->data(), j, planes, tab, 1, scalars_montgomery, window_bits, read);
            if (read) HIP_TRY(hipStreamWaitEvent(sc.lane->stream, read, 0));
            hipEvent_t done = c.scope_event();
            HIP_TRY(hipEventRecord(done, c.stream));
            sc.pending.push_back(scope_pending_t{done, [rq, j, pd] { msm_finish_job<F>(rq->data(), j, pd, 4); }, req[0].out});
        }
    } catch (...) {
        (void)hipStreamSynchronize(c.stream);
        if (&c != sc.lane) (void)hipStreamSynchronize(sc.lane->stream);
        (void)hipGetLastError();
        sc.pending.erase(sc.pending.begin() + (ptrdiff_t)pending0, sc.pending.end());
        c.pin_used = pin0;
        c.scope_events_used = ev0;  // only this call's "read" / "done" marks were taken from c's pool since ev0, and c has drained
        throw;
    }
    return true;
}
// contiguous outputs (outs + k * sizeof(Jacobian)): the batch entry points of the C ABI
template <class F>
static std::vector<msm_req_t> msm_requests(void* outs, size_t count, const size_t* off0, const size_t* n0, const size_t* off1, const size_t* n1,
                                           const void* const* scalars) {
    std::vector<msm_req_t> req(count);
    for (size_t k = 0; k < count; k++) {
        req[k].off0 = off0[k];
        req[k].n0 = n0[k];
        req[k].off1 = (n1 && n1[k]) ? off1[k] : 0;
        req[k].n1 = n1 ? n1[k] : 0;
        req[k].scalars = scalars[k];
        req[k].out = (uint8_t*)outs + sizeof(jac_mem_t<F>) * k;
    }
    return req;
}

// ---- in-library coalescing of concurrent callers ---------------------------------------------------------------------------
// The reference prover issues one MSM per polynomial from rayon workers (sonic_pc/mod.rs:186-245, kzg10/mod.rs:117-119): many
// threads inside snarkvm_hip_msm_registered* at the same time, each with ONE proof-sized instance - the shape that runs at a
// third of the fused rate when every call travels alone.  Here such calls meet: a caller whose MSM is small enough for a fused
// group (msm_handle_fusable, <= 2^18 pairs) queues a ticket on the HANDLE; whoever finds a free dispatcher slot (two per handle:
// while one batch computes, the next is being enqueued) takes every compatible ticket that is waiting and runs them as ONE
// msm_batch_run - group commit.  While a batch is in flight new arrivals pile up, so the batch size adapts to the concurrency by
// itself; a dispatcher additionally waits `coalesce_us` for stragglers when another thread called within the last 300 us (a
// rayon fan-out arrives within tens of microseconds).  A lone caller (one thread, sequential calls) never waits and runs exactly
// the launch sequence of the direct path.  Results are bit-identical to the per-instance path: the same kernels on the same
// operands, only grouped (tests/test_gpu_proofs.py::test_coalesced_*).  tuning coalesce=0 switches it off.
// Errors stay with the caller that caused them: every ticket is validated before it is queued, and when a fused batch fails as a whole
// its tickets are run again one by one, so that one caller's bad request (or a failure only the group provokes) cannot make the other
// callers fall back to their CPU paths.
struct msm_ticket_t {
    msm_req_t req;
    int on_device = 0, montgomery = 0, window_bits = 0;
    int state = 0;  // 0 queued, 1 in flight, 2 done, 3 failed
    std::exception_ptr err;
};
// how the coalescer grouped its callers so far: {batches dispatched, tickets in them, largest batch, batches of one ticket}
extern std::atomic<uint64_t> g_co_stats[4];  // api.hip (process-wide: G1 and G2 callers)
static bool msm_other_caller_recently() {
    static std::atomic<uint64_t> last_ns{0}, last_tid{0};
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    const uint64_t now = (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
    const uint64_t tid = (uint64_t)std::hash<std::thread::id>()(std::this_thread::get_id()) | 1u;
    const uint64_t pt = last_ns.exchange(now), pid = last_tid.exchange(tid);
    return pid != 0 && pid != tid && now - pt < 300000ull;
}
template <class F>
static bool msm_coalescible(const bases_handle_t<F>& h, size_t n, int window_bits) {
    return tuning().coalesce && n > 0 && n <= MSM_FUSE_MAX_PAIRS && !g_rt.profiling.load(std::memory_order_relaxed) && msm_handle_fusable(h, window_bits);
}
template <class F>
static void msm_coalesced(const bases_handle_t<F>& h, msm_ticket_t* tix, size_t count) {
    if (!count) return;
    scope_flush();  // another thread may run these tickets: what they read must be complete
    for (size_t i = 0; i < count; i++) {  // a request that cannot run never reaches the queue (it would fail the group it lands in)
        msm_check_requests(h.n, &tix[i].req, 1);
        if (tix[i].on_device && (tix[i].req.n0 + tix[i].req.n1) && g_rt.device_of(tix[i].req.scalars) < 0)
            throw hip_failure{hipErrorInvalidValue, "msm_registered: scalars are not on a device in use", __LINE__};
    }
    const bool hint = msm_other_caller_recently();
    std::unique_lock<std::mutex> lk(h.co_mu);
    for (size_t i = 0; i < count; i++) h.co_q.push_back(&tix[i]);
    auto mine_done = [&] {
        for (size_t i = 0; i < count; i++)
            if (tix[i].state < 2) return false;
        return true;
    };
    bool waited = false;
    while (!mine_done()) {
        if (h.co_leaders < tuning().coalesce_slots && !h.co_q.empty()) {
            h.co_leaders++;
            if (!waited && (hint || h.co_leaders > 1) && tuning().coalesce_us > 0) {
                waited = true;  // once per call: stragglers of the same fan-out
                h.co_cv.wait_for(lk, std::chrono::microseconds(tuning().coalesce_us));
            }
            std::vector<msm_ticket_t*> batch;
            if (!h.co_q.empty()) {
                const msm_ticket_t key = *h.co_q.front();
                std::deque<msm_ticket_t*> rest;
                for (msm_ticket_t* t : h.co_q) {
                    if (batch.size() < 1024 && t->on_device == key.on_device && t->montgomery == key.montgomery && t->window_bits == key.window_bits) {
                        t->state = 1;
                        batch.push_back(t);
                    } else {
                        rest.push_back(t);
                    }
                }
                h.co_q.swap(rest);
            }
            lk.unlock();
            std::exception_ptr err;
            std::vector<std::exception_ptr> errs;  // per ticket, after a failed group was re-run singly
            if (!batch.empty()) {
                g_co_stats[0].fetch_add(1, std::memory_order_relaxed);
                g_co_stats[1].fetch_add(batch.size(), std::memory_order_relaxed);
                if (batch.size() == 1) g_co_stats[3].fetch_add(1, std::memory_order_relaxed);
                for (uint64_t cur = g_co_stats[2].load(); cur < batch.size() && !g_co_stats[2].compare_exchange_weak(cur, batch.size());) {
                }
                try {
                    std::vector<msm_req_t> req(batch.size());
                    for (size_t i = 0; i < batch.size(); i++) req[i] = batch[i]->req;
                    msm_batch_run<F>(h, req.data(), req.size(), batch[0]->on_device, batch[0]->montgomery, batch[0]->window_bits);
                } catch (...) {
                    err = std::current_exception();
                }
                if (err && batch.size() > 1) {
                    errs.assign(batch.size(), nullptr);
                    for (size_t i = 0; i < batch.size(); i++) {
                        try {
                            msm_batch_run<F>(h, &batch[i]->req, 1, batch[i]->on_device, batch[i]->montgomery, batch[i]->window_bits);
                        } catch (...) {
                            errs[i] = std::current_exception();
                        }
                    }
                }
            }
            lk.lock();
            for (size_t i = 0; i < batch.size(); i++) {
                msm_ticket_t* t = batch[i];
                t->err = errs.empty() ? err : errs[i];
                t->state = t->err ? 3 : 2;
            }
            h.co_leaders--;
            h.co_cv.notify_all();
        } else {
            h.co_cv.wait(lk);
        }
    }
    lk.unlock();
    for (size_t i = 0; i < count; i++)
        if (tix[i].state == 3 && tix[i].err) std::rethrow_exception(tix[i].err);
}
// a batch of requests through the coalescer when every one of them qualifies, else straight to msm_batch_run
template <class F>
static void msm_batch_dispatch(const bases_handle_t<F>& h, std::vector<msm_req_t>& req, int scalars_on_device, int scalars_montgomery, int window_bits) {
    if (msm_scope_enqueue<F>(h, req.data(), req.size(), scalars_on_device, scalars_montgomery, window_bits)) return;
    bool all_small = !req.empty();
    for (const msm_req_t& r : req) {
        if (r.off0 + r.n0 > h.n || (r.n1 && r.off1 + r.n1 > h.n)) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: range exceeds the registered bases", __LINE__};
        if ((r.n0 + r.n1) && !r.scalars) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: null scalar vector", __LINE__};
        if (scalars_on_device && (r.n0 + r.n1) && g_rt.device_of(r.scalars) < 0)
            throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: scalars are not on a device in use", __LINE__};
        all_small = all_small && msm_coalescible(h, r.n0 + r.n1, window_bits);
    }
    if (!all_small) {
        msm_batch_run<F>(h, req.data(), req.size(), scalars_on_device, scalars_montgomery, window_bits);
        return;
    }
    std::vector<msm_ticket_t> tix(req.size());
    for (size_t i = 0; i < req.size(); i++) {
        tix[i].req = req[i];
        tix[i].on_device = scalars_on_device ? 1 : 0;
        tix[i].montgomery = scalars_montgomery ? 1 : 0;
        tix[i].window_bits = window_bits;
    }
    msm_coalesced<F>(h, tix.data(), tix.size());
}

// ------------------------------------------------------------------------------------------------
// exported-function scaffolding
// ------------------------------------------------------------------------------------------------
// API_BEGIN: take one lane on any device;  API_BEGIN_DEV(d): on logical device d (d < 0: any).  `c` is the lane.
#define API_BEGIN_DEV(devsel)                      \
    try {                                          \
        lane_guard _lg(devsel);                    \
        lane_t& c = _lg.c();                       \
        c.begin_call();
#define API_BEGIN API_BEGIN_DEV(-1)
#define API_END                                    \
    c.end_call();                                  \
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
// the same handlers for functions that manage their lanes themselves
#define API_TRY try {
#define API_CATCH                                  \
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
// logical device for a call whose operands live in device memory (on_device != 0): the owner of `ptr`
static int device_for(const void* ptr, int on_device) {
    if (!on_device || !ptr) return -1;
    const int d = g_rt.device_of(ptr);
    if (d < 0) throw hip_failure{hipErrorInvalidValue, "device pointer does not belong to a device in use (snarkvm_hip_set_devices)", __LINE__};
    return d;
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

