// api.hip - C ABI (include/snarkvm_hip.h) of the gfx950 MSM / NTT backend: runtime configuration and the G1 MSM entry points.
//
// Host runtime (runtime.hip.h) = what algorithms/cuda/cuda/snarkvm.cu:73-312 (snarkvm_t) and snarkvm_api.cu:23-84 are in the
// reference: a lazily constructed per-process runtime over every selected GPU (device arenas, streams, twiddle tables), a
// pool of (device, stream) resource tokens handed to concurrent callers, staging of the caller's host buffers, the
// point-range split of one MSM over the GPUs with a host-side combine, error reporting as RustError.
// Fr entry points: api_fr.hip; point encoding + setup-time group operations: api_serde.hip; G2: api_g2.hip.
#define SV_TU_G1
#include "runtime.hip.h"
#include "ffl.hip.h"

runtime_t g_rt;
thread_local thread_scope_t g_tl_scope;
// A thread that ends inside a scope (a worker that died, a caller that forgot scope_end): its lanes return to the pool - they are a bounded resource
// (12 of a GPU's 16 may be held by scopes) and nobody else can end this scope.  The queued work is waited for; results that were still owed (MSM
// outputs, parked remainders) are NOT delivered: their buffers belong to a thread that is gone.
thread_scope_t::~thread_scope_t() {
    if (!lane) return;
    pending.clear();
    (void)hipStreamSynchronize(lane->stream);
    for (int i = 0; i < naux; i++) {
        (void)hipStreamSynchronize(aux[i]->stream);
        aux[i]->dev->give(aux[i], true);
    }
    lane->in_scope = false;
    lane->deferred.clear();
    lane->deferred_bytes = 0;
    if (!lane->tw_leases.empty()) ntt_tw_release(lane->dev, lane->tw_leases);
    lane->dev->give(lane, true);
    lane = nullptr;
}
std::atomic<uint64_t> g_co_stats[4];
static std::atomic<uint64_t> g_alloc_stats[5];  // {device allocations, device bytes, pinned allocations, pinned bytes, microseconds}
void sv_alloc_note(int slot, size_t bytes, double ms) {
    g_alloc_stats[slot].fetch_add(1, std::memory_order_relaxed);
    g_alloc_stats[slot + 1].fetch_add(bytes, std::memory_order_relaxed);
    g_alloc_stats[4].fetch_add((uint64_t)(ms * 1e3), std::memory_order_relaxed);
}

// Outgrown workspace blocks (runtime.hip.h: sv_defer_free) wait on a list of the THREAD that outgrew them and are released when that thread's call ends
// (lane_t::end_call) or its scope is flushed.  (Round 5 kept one process-wide list that every lane's end_call drained: hipFree waits for every stream
// of the device, so a short call of thread B inherited the wait for thread A's whole queued scope - the stall the deferral exists to avoid, moved to
// another caller.)  A thread that ends with blocks still parked releases them in the list's destructor.
struct tl_free_list_t {
    std::vector<void*> v;
    ~tl_free_list_t() {
        for (void* p : v) (void)hipFree(p);
    }
};
static thread_local tl_free_list_t g_tl_free;
void sv_defer_free(void* p) { g_tl_free.v.push_back(p); }
void sv_drain_frees() {
    if (g_tl_free.v.empty()) return;
    std::vector<void*> mine;
    mine.swap(g_tl_free.v);
    for (void* p : mine) (void)hipFree(p);  // hipFree takes a pointer of any device
}

extern "C" {

int snarkvm_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int snarkvm_hip_batch_lanes(size_t npoints) { return batch_lanes(npoints); }
// Select the devices this process uses (before the first compute call).  Duplicated ids make independent logical devices on
// one GPU (tests).  snarkvm_hip_set_device(d) == set_devices(&d, 1): the one-process-per-GPU deployment.
RustError snarkvm_hip_set_devices(const int32_t* ids, size_t n) {
    std::lock_guard<std::mutex> lk(g_rt.cfg_mu);
    if (n == 0 || !ids) return fail(1, "snarkvm_hip_set_devices: empty device list");
    std::vector<int> v(ids, ids + n);
    if (g_rt.configured) {
        bool same = v.size() == g_rt.devs.size();
        for (size_t i = 0; same && i < v.size(); i++) same = g_rt.devs[i]->physical == v[i];
        if (!same) return fail(1, "snarkvm_hip_set_devices: the runtime is already initialised on another device set");
        return ok();
    }
    g_rt.want = v;
    return ok();
}
RustError snarkvm_hip_set_device(int device) {
    const int32_t d = device;
    return snarkvm_hip_set_devices(&d, 1);
}
int snarkvm_hip_num_devices(void) {
    try {
        return g_rt.ndev();
    } catch (...) {
        return 0;
    }
}
void snarkvm_hip_set_profiling(int enabled) { g_rt.profiling.store(enabled != 0); }
int snarkvm_hip_get_phase_count(void) {
    std::lock_guard<std::mutex> lk(g_rt.prof_mu);
    return (int)g_rt.last_phases.size();
}
const char* snarkvm_hip_get_phase_name(int i) {
    std::lock_guard<std::mutex> lk(g_rt.prof_mu);
    return (i >= 0 && i < (int)g_rt.last_phases.size()) ? g_rt.last_phases[i].first.c_str() : "";
}
double snarkvm_hip_get_phase_ms(int i) {
    std::lock_guard<std::mutex> lk(g_rt.prof_mu);
    return (i >= 0 && i < (int)g_rt.last_phases.size()) ? g_rt.last_phases[i].second : 0.0;
}

// coalescer statistics (G1 and G2 callers): out[4] = {batches dispatched, tickets in them, largest
// batch, single-ticket batches}; reset != 0 clears them
void snarkvm_hip_coalescer_stats(uint64_t* out, int reset) {
    for (int i = 0; i < 4; i++) {
        if (out) out[i] = g_co_stats[i].load();
        if (reset) g_co_stats[i].store(0);
    }
}
void snarkvm_hip_alloc_stats(uint64_t* out, int reset) {
    for (int i = 0; i < 5; i++) {
        if (out) out[i] = g_alloc_stats[i].load();
        if (reset) g_alloc_stats[i].store(0);
    }
}
RustError snarkvm_hip_synchronize(void) {
    API_TRY
    g_rt.configure();
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (auto& d : g_rt.devs) {
        if (!d->ready) continue;
        HIP_TRY(hipSetDevice(d->physical));
        for (auto& l : d->lane) {
            HIP_TRY(hipStreamSynchronize(l.stream));
            HIP_TRY(hipStreamSynchronize(l.alt));
        }
    }
    (void)hipSetDevice(prev);
    API_CATCH
}

// ---- device memory for hosts without a HIP binding (a Rust prover) ---------------------------------------------
// Plain hipMalloc / hipFree / copies behind the C ABI, so that the device-resident entry points can be fed without torch or a HIP crate
// on the caller's side.  Copies run on a lane like every other call: inside the calling thread's scope that is the scope's own stream
// (ordered with its transforms), outside a scope any free lane of the device that owns the device pointer.
RustError snarkvm_hip_malloc(void** d_ptr, size_t bytes, int device) {
    API_TRY
    if (!d_ptr) throw hip_failure{hipErrorInvalidValue, "snarkvm_hip_malloc: null result pointer", __LINE__};
    *d_ptr = nullptr;
    g_rt.configure();
    lane_t* sl = tl_scope().lane;
    if (device < 0) device = sl ? sl->dev->logical : 0;
    if (device >= (int)g_rt.devs.size()) throw hip_failure{hipErrorInvalidDevice, "snarkvm_hip_malloc: logical device index out of range", __LINE__};
    if (bytes) {
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        const int phys = g_rt.devs[device]->physical;
        if (prev != phys) HIP_TRY(hipSetDevice(phys));
        const hipError_t e = hipMalloc(d_ptr, bytes);
        if (prev >= 0 && prev != phys) (void)hipSetDevice(prev);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            *d_ptr = nullptr;
            throw hip_failure{e, "snarkvm_hip_malloc: hipMalloc", __LINE__};
        }
    }
    API_CATCH
}
RustError snarkvm_hip_free(void* d_ptr) {
    API_TRY
    if (d_ptr) HIP_TRY(hipFree(d_ptr));  // waits for every stream of the owning device: nothing queued can still use the block
    API_CATCH
}
RustError snarkvm_hip_memcpy_h2d(void* d_dst, const void* src, size_t bytes) {
    if (!bytes) return ok();
    API_BEGIN_DEV(device_for(d_dst, 1))
    if (!d_dst || !src) throw hip_failure{hipErrorInvalidValue, "snarkvm_hip_memcpy_h2d: null pointer", __LINE__};
    HIP_TRY(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));  // host side: complete on return, also inside a scope (header)
    API_END
}
RustError snarkvm_hip_memcpy_d2h(void* dst, const void* d_src, size_t bytes) {
    if (!bytes) return ok();
    API_BEGIN_DEV(device_for(d_src, 1))
    if (!dst || !d_src) throw hip_failure{hipErrorInvalidValue, "snarkvm_hip_memcpy_d2h: null pointer", __LINE__};
    HIP_TRY(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    API_END
}
RustError snarkvm_hip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes) {
    if (!bytes) return ok();
    API_BEGIN_DEV(device_for(d_dst, 1))
    if (!d_dst || !d_src) throw hip_failure{hipErrorInvalidValue, "snarkvm_hip_memcpy_d2d: null pointer", __LINE__};
    const uint8_t *a = (const uint8_t*)d_dst, *b = (const uint8_t*)d_src;
    if (a < b + bytes && b < a + bytes) throw hip_failure{hipErrorInvalidValue, "snarkvm_hip_memcpy_d2d: the ranges overlap", __LINE__};
    if (g_rt.device_of(d_src) < 0) throw hip_failure{hipErrorInvalidValue, "snarkvm_hip_memcpy_d2d: source is not on a device in use", __LINE__};
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDefault, c.stream));
    c.sync_or_defer();
    API_END
}
RustError snarkvm_hip_memset(void* d_dst, int value, size_t bytes) {
    if (!bytes) return ok();
    API_BEGIN_DEV(device_for(d_dst, 1))
    HIP_TRY(hipMemsetAsync(d_dst, value, bytes, c.stream));
    c.sync_or_defer();
    API_END
}

// ---- deferred-synchronisation scope ---------------------------------------------------------------------
// One lane stays bound to the calling thread between _begin and _end; the thread's calls on device-resident operands
// (snarkvm_hip_ntt_device[_batch], snarkvm_hip_fr_* with on_device = 1) are enqueued on that lane's stream and return at once,
// small host results (the remainder of fr_divide_by_linear) are delivered by _end.  How a prover that keeps its polynomials in HBM
// issues a whole round - or the same round of many proofs in lock step - without a stream synchronisation per call (~40 us each
// on this stack: 45 transforms per proof).  Calls that leave the scope's lane (MSMs, host buffers) wait for the scope first.
static constexpr uint32_t SCOPE_FLAGS_ALL = SNARKVM_HIP_SCOPE_ASYNC_MSM | SNARKVM_HIP_SCOPE_STABLE_INPUTS | SNARKVM_HIP_SCOPE_MSM_IN_STREAM;
RustError snarkvm_hip_scope_begin_ex(const void* d_any, uint32_t flags) {
    API_TRY
    thread_scope_t& sc = tl_scope();
    if (sc.lane) throw hip_failure{hipErrorInvalidValue, "scope_begin: the calling thread already has an open scope", __LINE__};
    if (flags & ~SCOPE_FLAGS_ALL) throw hip_failure{hipErrorInvalidValue, "scope_begin: unknown flag", __LINE__};
    g_rt.configure();
    const int nd = (int)g_rt.devs.size();
    int dev = device_for(d_any, d_any ? 1 : 0);
    if (dev < 0) dev = (int)(g_rt.rr.fetch_add(1) % (uint32_t)nd);
    device_t* d = g_rt.devs[dev].get();
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = 0;
    lane_t* l = d->take_for_scope(true);  // waits while the GPU's scopes hold SCOPE_LANES_MAX lanes
    try {
        HIP_TRY(hipSetDevice(d->physical));
        d->init();
        tu_kernel_attributes(d->logical);
    } catch (...) {
        d->give(l, true);
        (void)hipSetDevice(prev);
        throw;
    }
    l->begin_call();
    l->in_scope = true;
    l->pin_used = 0;
    l->scope_events_used = 0;
    sc = thread_scope_t();
    sc.lane = l;
    sc.prev_device = prev;
    sc.flags = flags;
    API_CATCH
}
RustError snarkvm_hip_scope_begin(const void* d_any) { return snarkvm_hip_scope_begin_ex(d_any, 0); }
void* snarkvm_hip_scope_stream(void) {
    lane_t* l = tl_scope().lane;
    return l ? (void*)l->stream : nullptr;
}
RustError snarkvm_hip_scope_set_flags(uint32_t flags) {
    API_TRY
    thread_scope_t& sc = tl_scope();
    if (!sc.lane) throw hip_failure{hipErrorInvalidValue, "scope_set_flags: the calling thread has no open scope", __LINE__};
    if (flags & ~SCOPE_FLAGS_ALL) throw hip_failure{hipErrorInvalidValue, "scope_set_flags: unknown flag", __LINE__};
    sc.flags = flags;  // applies to the calls that follow; what is enqueued stays where it is
    API_CATCH
}
RustError snarkvm_hip_scope_collect(const void* out) {
    API_TRY
    scope_collect(out);
    API_CATCH
}
RustError snarkvm_hip_scope_end(void) {
    thread_scope_t& sc = tl_scope();
    lane_t* l = sc.lane;
    if (!l) return ok();
    RustError r = ok();
    try {
        scope_flush();
    } catch (const hip_failure& f) {
        r = from_failure(f);
    } catch (const std::exception& e) {
        r = fail(1, std::string("snarkvm_hip: scope_end: ") + e.what());
    } catch (...) {
        r = fail(1, "snarkvm_hip: scope_end failed");
    }
    l->in_scope = false;
    l->deferred.clear();
    l->deferred_bytes = 0;
    for (int i = 0; i < sc.naux; i++) {
        (void)hipStreamSynchronize(sc.aux[i]->stream);  // (already idle unless the flush failed half-way)
        sc.aux[i]->dev->give(sc.aux[i], true);
    }
    const int prev = sc.prev_device;
    sc = thread_scope_t();
    l->dev->give(l, true);
    if (prev >= 0) (void)hipSetDevice(prev);
    return r;
}

// ---- registered bases ---------------------------------------------------------------------------------
// tables 1 .. J-1 of one replica: table j = 2^table_bits * table j-1
static void precompute_tables(lane_t& c, snarkvm_hip_bases* h, g1_aff_mem_t* d) { precompute_tables_run<fq_t>(c, d, h->n, h->tables, h->table_bits); }
// One replica per logical device.  Host source: every device uploads and converts for itself (in parallel).  Device source:
// the owner converts and precomputes, the other devices receive the finished tables by peer copies over xGMI.
static void register_bases_impl(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device, int tables,
                                int table_bits) {
    if (!handle) throw hip_failure{hipErrorInvalidValue, "register_bases: null handle", __LINE__};
    if (npoints && !points) throw hip_failure{hipErrorInvalidValue, "register_bases: null points", __LINE__};
    if (ffi_affine_sz < 104 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "register_bases: bad stride", __LINE__};
    check_tables(tables, table_bits, "register_bases");
    scope_flush();  // device-resident points may come out of the caller's open scope; the per-device workers below are other threads
    const int nd = g_rt.ndev();
    std::unique_ptr<snarkvm_hip_bases> h(new snarkvm_hip_bases());
    h->n = npoints;
    h->tables = tables;
    h->table_bits = table_bits ? table_bits : 256 / tables;
    h->d.assign(nd, nullptr);
    const size_t bytes = (size_t)tables * npoints * sizeof(g1_aff_mem_t);
    try {
        if (npoints) {
            const int owner = on_device ? device_for(points, 1) : -1;
            auto build = [&](int dev) {
                lane_guard lg(dev);
                lane_t& c = lg.c();
                HIP_TRY(hipMalloc((void**)&h->d[dev], bytes));
                const uint8_t* src = (const uint8_t*)points;
                if (!on_device) {
                    c.bases_tmp.ensure(npoints * ffi_affine_sz);
                    HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, points, npoints * ffi_affine_sz, hipMemcpyHostToDevice, c.stream));
                    src = c.bases_tmp.as<uint8_t>();
                }
                convert_bases<fq_t>(c, src, ffi_affine_sz, npoints, h->d[dev]);
                precompute_tables(c, h.get(), h->d[dev]);
                bases_to_lazy_form(c, h->d[dev], (size_t)tables * npoints);  // last: the tables are derived from one another in the exact form
                HIP_TRY(hipStreamSynchronize(c.stream));
            };
            std::vector<int> all;
            for (int d = 0; d < nd; d++) all.push_back(d);
            if (!on_device) {
                for_each_device(all, build);
            } else {
                build(owner);
                for (int d = 0; d < nd; d++) {
                    if (d == owner) continue;
                    lane_guard lg(d);
                    HIP_TRY(hipMalloc((void**)&h->d[d], bytes));
                    HIP_TRY(hipMemcpyPeerAsync(h->d[d], g_rt.devs[d]->physical, h->d[owner], g_rt.devs[owner]->physical, bytes, lg.c().stream));
                    HIP_TRY(hipStreamSynchronize(lg.c().stream));
                }
            }
        }
    } catch (...) {
        h->free_all();
        throw;
    }
    *handle = h.release();
}
RustError snarkvm_hip_register_bases(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device) {
    API_TRY
    register_bases_impl(handle, points, npoints, ffi_affine_sz, on_device, 1, 0);
    API_CATCH
}
RustError snarkvm_hip_register_bases_tables(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device,
                                            int tables) {
    API_TRY
    register_bases_impl(handle, points, npoints, ffi_affine_sz, on_device, tables, 0);
    API_CATCH
}
RustError snarkvm_hip_register_bases_windowed(snarkvm_hip_bases_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int on_device,
                                              int tables, int window_bits) {
    API_TRY
    if (window_bits <= 0) throw hip_failure{hipErrorInvalidValue, "register_bases_windowed: window_bits must be positive", __LINE__};
    register_bases_impl(handle, points, npoints, ffi_affine_sz, on_device, tables, window_bits);
    API_CATCH
}
void snarkvm_hip_free_bases(snarkvm_hip_bases_t* h) {
    if (!h) return;
    h->free_all();
    delete h;
}
}  // extern "C"
// handle construction shared with api_serde.hip (bases decoded from their canonical bytes); C++ linkage
snarkvm_hip_bases* sv_new_bases_handle(size_t npoints, int tables, int table_bits) {
    snarkvm_hip_bases* h = new snarkvm_hip_bases();
    h->n = npoints;
    h->tables = tables;
    h->table_bits = table_bits;
    h->d.assign(g_rt.ndev(), nullptr);
    return h;
}
void sv_precompute_tables(lane_t& c, snarkvm_hip_bases* h, g1_aff_mem_t* d) { precompute_tables(c, h, d); }
extern "C" {

// ---- MSM over registered bases ----------------------------------------------------------------------------
static void check_window_bits(int window_bits, const char* who) {
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw std::runtime_error(std::string(who) + ": window_bits must be 0 or 2..23");
}
// One MSM over a registered range with HOST scalars, split by point range over the devices when it is big enough (the
// reference's multi-GPU MSM, snarkvm.cu:254-295): device d sums its n / ndev pairs over its own replica with the full
// window set; the per-device bit-plane sums are added on the host before the one Horner chain.  (off1, n1): KZG10's second
// base range (see snarkvm_hip_msm_registered_ex).
static void msm_registered_host_scalars(void* out, const snarkvm_hip_bases* h, size_t off0, size_t n0, size_t off1, size_t n1, const void* scalars,
                                        int scalars_montgomery, int window_bits) {
    const size_t n = n0 + n1;
    scope_flush();
    const int nd = g_rt.ndev();
    static const int trace = getenv("SNARKVM_HIP_TRACE") ? atoi(getenv("SNARKVM_HIP_TRACE")) : 0;
    const double t_begin = host_now_ms();
    // one part per device once every part keeps >= MSM_SPLIT_MIN pairs; big calls are cut further into scalar chunks whose
    // upload (32 B per pair over PCIe) hides behind the previous chunk's accumulation (lane_ring_run)
    size_t parts = n / MSM_SPLIT_MIN;
    if (parts > (size_t)nd) parts = (size_t)nd;
    if (parts < 1) parts = 1;
    const size_t chunk = msm_scalar_chunk_pairs();
    if (n >= 2 * chunk && (n + chunk - 1) / chunk > parts) parts = (n + chunk - 1) / chunk;
    // One device: GEOMETRIC chunks (tuning scalar_geo = g, default 4: sizes 1 : g [: g^2]).  Every chunk pays the bucket overhead of a
    // whole MSM (partial sums, their conversion, reduce rounds and the merge over 2^21 buckets at 22-bit windows: equal quarters of a 2^24 call cost 9.6 ms
    // each against 7.85 ms for a quarter of the unchunked MSM), so few chunks are better than many - and a chunk only has to keep
    // the GPU busy for as long as the NEXT chunk's scalars take to arrive (0.6 ms per 2^20 over PCIe against ~2 ms of arithmetic).
    std::vector<size_t> bound;
    const int geo = tuning().scalar_geo;
    if (nd == 1 && geo > 1 && tuning().taper != 0 && n >= ((size_t)1 << 22)) {
        parts = n >= ((size_t)1 << 23) ? 3 : 2;
        size_t wsum = 0, w = 1;
        for (size_t i = 0; i < parts; i++, w *= (size_t)geo) wsum += w;
        size_t acc_w = 0;
        w = 1;
        bound.push_back(0);
        for (size_t i = 0; i < parts; i++, w *= (size_t)geo) {
            acc_w += w;
            bound.push_back(i + 1 == parts ? n : (size_t)((unsigned __int128)n * acc_w / wsum));
        }
    } else {
        for (size_t i = 0; i <= parts; i++) bound.push_back(n * i / parts);
    }
    const int ndu = (int)(parts < (size_t)nd ? parts : (size_t)nd);
    std::unique_ptr<msm_accum_t<fq_t>> acc(new msm_accum_t<fq_t>());
    std::mutex acc_mu;
    std::vector<int> devs;
    if (ndu == 1) {
        devs.push_back(-1);  // any device with a free lane
    } else {
        for (int d = 0; d < ndu; d++) devs.push_back(d);
    }
    const size_t slot = msm_plane_bytes<fq_t>();
    for_each_device(devs, [&](int dev) {
        std::vector<size_t> mine;
        for (size_t i = (dev < 0 ? 0 : (size_t)dev); i < parts; i += (size_t)ndu) mine.push_back(i);
        lane_guard lg;
        lg.acquire(dev, mine.size() > 2 ? 3 : (int)mine.size());
        const size_t L = lg.lanes.size();
        size_t max_cnt = 0;
        for (size_t j = 0; j < mine.size(); j++) {
            const size_t cnt = bound[mine[j] + 1] - bound[mine[j]];
            max_cnt = cnt > max_cnt ? cnt : max_cnt;
        }
        // Several scalar chunks on this device share ONE set of buckets (the geometry of the largest chunk for all of them): every
        // chunk adds its per-bucket partial sums to a sink - one slot per bucket, the merges chained by events across the lanes - and
        // the reduce / fold / bit-plane tail runs once after the last chunk instead of once per chunk (2^24 pairs over 12 x 22-bit
        // tables: four tails of ~2.5 ms over 2^21 buckets each).  tuning taper=0: every chunk runs its own tail (round 3).
        const bool use_sink = tuning().taper != 0 && mine.size() >= 2;
        std::vector<msm_pending_t> pend(use_sink ? 1 : mine.size());
        int chunk_c = window_bits;
        msm_bucket_sink_t sink;
        std::vector<hipEvent_t> merged;
        for (size_t l = 0; l < L; l++) {
            lane_t& c = *lg.lanes[l];
            c.begin_call();
            c.scalars_tmp.ensure(max_cnt * 32 + 32);
            c.pin.ensure(slot * (use_sink ? 1 : (mine.size() + L - 1) / L));
        }
        if (use_sink) {
            lane_t& c0 = *lg.lanes[0];
            const msm_plan_t pl = msm_make_plan(max_cnt, window_bits, h->tables, h->table_bits);
            chunk_c = pl.c;
            sink.nbt = (uint32_t)pl.W * pl.nb;
            sink.L = 1;
            const size_t bytes = (size_t)sink.nbt * msm_partial_bytes<fq_t>();
            c0.sink_acc.ensure(bytes);
            sink.acc = c0.sink_acc.p;
            HIP_TRY(hipMemsetAsync(sink.acc, 0, bytes, c0.stream));  // all-zero = the point at infinity
            hipEvent_t ready = c0.new_event();
            HIP_TRY(hipEventRecord(ready, c0.stream));
            for (size_t l = 1; l < L; l++) HIP_TRY(hipStreamWaitEvent(lg.lanes[l]->stream, ready, 0));
            for (size_t j = 0; j < mine.size(); j++) merged.push_back(lg.lanes[j % L]->new_event());
        }
        const g1_aff_mem_t* base = h->d[lg.lanes[0]->dev->logical];
        auto upload = [&](size_t j, hipStream_t st) {
            lane_t& c = *lg.lanes[j % L];
            const size_t lo = bound[mine[j]], hi = bound[mine[j] + 1];
            if (hi > lo) HIP_TRY(hipMemcpyAsync(c.scalars_tmp.p, (const uint8_t*)scalars + lo * 32, (hi - lo) * 32, hipMemcpyHostToDevice, st));
        };
        auto compute = [&](size_t j, bool prof) {
            lane_t& c = *lg.lanes[j % L];
            // slice [lo, hi) of the concatenation (range 0 | range 1)
            const size_t lo = bound[mine[j]], hi = bound[mine[j] + 1];
            const size_t a0 = lo < n0 ? lo : n0, a1 = hi < n0 ? hi : n0;  // part inside range 0
            const size_t m0 = a1 - a0;
            const g1_aff_mem_t* b0 = base + off0 + a0;
            const g1_aff_mem_t* b1 = base + off1 + (lo > n0 ? lo - n0 : 0);
            msm_bucket_sink_t mine_sink = sink;
            if (use_sink) {
                mine_sink.after = j ? merged[j - 1] : nullptr;
                mine_sink.done = merged[j];
            }
            const msm_pending_t pd = msm_run<fq_t>(c, m0 ? b0 : b1, c.scalars_tmp.as<uint4>(), hi - lo, c.pin.as<uint8_t>() + (use_sink ? 0 : slot * (j / L)), chunk_c,
                                                   b1, m0 ? m0 : ~(size_t)0, scalars_montgomery, h->tables, h->n, prof, h->table_bits, nullptr,
                                                   use_sink ? &mine_sink : nullptr);
            if (!use_sink) pend[j] = pd;
        };
        if (mine.size() == 1) {
            lane_t& c = *lg.lanes[0];
            c.phase_begin("msm_h2d");
            upload(0, c.stream);
            c.phase_end();
            compute(0, parts == 1);
        } else {
            lane_ring_run(lg, mine.size(), upload, [&](size_t j) { compute(j, false); }, trace, t_begin);
        }
        if (use_sink) {  // the last merge (which waited for all earlier ones), then the one tail on lane 0
            lane_t& c0 = *lg.lanes[0];
            HIP_TRY(hipStreamWaitEvent(c0.stream, merged.back(), 0));
            pend[0] = msm_tail_from_sink<fq_t>(c0, max_cnt, chunk_c, sink, c0.pin.p, h->tables, h->table_bits);
        }
        for (size_t l = 0; l < L; l++) {
            HIP_TRY(hipStreamSynchronize(lg.lanes[l]->alt));
            HIP_TRY(hipStreamSynchronize(lg.lanes[l]->stream));
        }
        {
            std::lock_guard<std::mutex> lk(acc_mu);
            for (auto& pd : pend) msm_collect<fq_t>(*acc, pd);
        }
        if (trace) fprintf(stderr, "[snarkvm_hip] registered MSM, host scalars: %zu chunk(s) done at t+%.2f ms\n", mine.size(), host_now_ms() - t_begin);
        for (size_t l = 0; l < L; l++) lg.lanes[l]->end_call();
    });
    acc->finish(out);
}
// one proof-sized MSM of one caller: a ticket on the handle's coalescer (runtime.hip.h::msm_coalesced) - concurrent callers are fused
static void msm_single_coalesced(void* out, const snarkvm_hip_bases* h, size_t off0, size_t n0, size_t off1, size_t n1, const void* scalars,
                                 int scalars_on_device, int scalars_montgomery, int window_bits) {
    if (scalars_on_device && g_rt.device_of(scalars) < 0)
        throw hip_failure{hipErrorInvalidValue, "device pointer does not belong to a device in use (snarkvm_hip_set_devices)", __LINE__};
    msm_ticket_t t;
    t.req.off0 = off0, t.req.n0 = n0, t.req.off1 = n1 ? off1 : 0, t.req.n1 = n1, t.req.scalars = scalars, t.req.out = out;
    t.on_device = scalars_on_device ? 1 : 0;
    t.montgomery = scalars_montgomery ? 1 : 0;
    t.window_bits = window_bits;
    msm_coalesced<fq_t>(*h, &t, 1);
}
RustError snarkvm_hip_msm_registered(void* out, const snarkvm_hip_bases_t* h, size_t offset, size_t npoints, const void* scalars,
                                     int scalars_on_device, int window_bits) {
    API_TRY
    if (!h || offset + npoints > h->n) throw hip_failure{hipErrorInvalidValue, "msm_registered: range exceeds the registered bases", __LINE__};
    check_window_bits(window_bits, "msm_registered");
    if (!out || (npoints && !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_registered: null argument", __LINE__};
    msm_req_t one;
    one.off0 = offset, one.n0 = npoints, one.scalars = scalars, one.out = out;
    if (msm_scope_enqueue<fq_t>(*h, &one, 1, scalars_on_device, 0, window_bits)) {
    } else if (msm_coalescible(*h, npoints, window_bits)) {
        msm_single_coalesced(out, h, offset, npoints, 0, 0, scalars, scalars_on_device, 0, window_bits);
    } else if (!scalars_on_device) {
        msm_registered_host_scalars(out, h, offset, npoints, 0, 0, scalars, 0, window_bits);
    } else {
        lane_guard lg(device_for(scalars, npoints ? 1 : 0));
        lane_t& c = lg.c();
        c.begin_call();
        msm_run_sync<fq_t>(c, h->d[c.dev->logical] + offset, (const uint4*)scalars, npoints, out, window_bits, nullptr, ~(size_t)0, 0, h->tables, h->n,
                           h->table_bits);
        c.end_call();
    }
    API_CATCH
}
RustError snarkvm_hip_msm_registered_ex(void* out, const snarkvm_hip_bases_t* h, size_t off0, size_t n0, size_t off1, size_t n1,
                                        const void* scalars, int scalars_on_device, int scalars_montgomery, int window_bits) {
    API_TRY
    if (!h || off0 + n0 > h->n || off1 + n1 > h->n) throw hip_failure{hipErrorInvalidValue, "msm_registered_ex: range exceeds the registered bases", __LINE__};
    check_window_bits(window_bits, "msm_registered_ex");
    const size_t n = n0 + n1;
    if (!out || (n && !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_registered_ex: null argument", __LINE__};
    msm_req_t one;
    one.off0 = off0, one.n0 = n0, one.off1 = n1 ? off1 : 0, one.n1 = n1, one.scalars = scalars, one.out = out;
    if (msm_scope_enqueue<fq_t>(*h, &one, 1, scalars_on_device, scalars_montgomery, window_bits)) {
    } else if (msm_coalescible(*h, n, window_bits)) {
        msm_single_coalesced(out, h, off0, n0, off1, n1, scalars, scalars_on_device, scalars_montgomery, window_bits);
    } else if (!scalars_on_device) {
        msm_registered_host_scalars(out, h, off0, n0, off1, n1, scalars, scalars_montgomery, window_bits);
    } else {
        lane_guard lg(device_for(scalars, n ? 1 : 0));
        lane_t& c = lg.c();
        c.begin_call();
        const g1_aff_mem_t* base = h->d[c.dev->logical];
        msm_run_sync<fq_t>(c, base + off0, (const uint4*)scalars, n, out, window_bits, base + off1, n0, scalars_montgomery, h->tables, h->n, h->table_bits);
        c.end_call();
    }
    API_CATCH
}

// A batch of independent MSMs (the commitments of a batch of proofs) fanned out over devices x lanes.  Host scalars:
// instance k goes to device k mod ndev; device scalars: to the device that owns them.  On each device the instances cycle
// over several lanes, so the latency-bound tail of one overlaps the accumulation of the next; the host finishes instance k
// (Horner over its bit planes) as soon as its planes have arrived, while the GPU works on the later instances.
RustError snarkvm_hip_msm_registered_batch(void* outs, const snarkvm_hip_bases_t* h, size_t count, const size_t* offsets, const size_t* npoints,
                                           const void* const* scalars, int scalars_on_device, int scalars_montgomery, int window_bits) {
    API_TRY
    if (!h) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: null handle", __LINE__};
    check_window_bits(window_bits, "msm_registered_batch");
    if (count && (!outs || !offsets || !npoints || !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch: null argument", __LINE__};
    std::vector<msm_req_t> req = msm_requests<fq_t>(outs, count, offsets, npoints, nullptr, nullptr, scalars);
    msm_batch_dispatch<fq_t>(*h, req, scalars_on_device, scalars_montgomery, window_bits);
    API_CATCH
}

// The same with two base ranges per instance (KZG10::commit with hiding / SonicKZG10::commit's degree-bounded polynomials:
// instance k = bases [off0[k], + n0[k]) then [off1[k], + n1[k]), n0 + n1 consecutive scalars).
RustError snarkvm_hip_msm_registered_batch_ex(void* outs, const snarkvm_hip_bases_t* h, size_t count, const size_t* off0, const size_t* n0,
                                              const size_t* off1, const size_t* n1, const void* const* scalars, int scalars_on_device,
                                              int scalars_montgomery, int window_bits) {
    API_TRY
    if (!h) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch_ex: null handle", __LINE__};
    check_window_bits(window_bits, "msm_registered_batch_ex");
    if (count && (!outs || !off0 || !n0 || !off1 || !n1 || !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_registered_batch_ex: null argument", __LINE__};
    std::vector<msm_req_t> req = msm_requests<fq_t>(outs, count, off0, n0, off1, n1, scalars);
    msm_batch_dispatch<fq_t>(*h, req, scalars_on_device, scalars_montgomery, window_bits);
    API_CATCH
}

// ---- the reference's FFI MSM (host bases + host scalars) -----------------------------------------------------------------
// `snarkvm_msm` is STATELESS by default, like the reference's symbol (the callee must not retain the caller's
// pointers): bases and scalars are uploaded, converted and summed on every call and nothing is remembered after return.
//
// Opt-in base cache (SNARKVM_HIP_BASE_CACHE=1 / 2 / 4 / 8 / 16; an EXTENSION with its own contract): the
// reference's callers pass slices of ONE long-lived vector (`powers_of_beta_g[lz .. lz + len]`, kzg10/mod.rs:117-119).  With
// the cache on, a host range that is passed a SECOND time - the same address and length - is registered (converted, with
// precomputed tables, on every device) from the memory of THAT call, and later calls whose base range lies inside it skip
// the upload, the conversion and most of the Horner chain.  A hit is verified against raw copies of every CACHE_STEP-th
// point of the requested slice (a slice > 1024 points always contains at least 16 of them); a mismatch drops the entry.  The
// contract the caller accepts by setting the variable: the vector is immutable and outlives the process's use of it (true
// of an SRS; not checkable from here).  Host memory is only ever read inside the slice the current call passed.  A
// registration that fails (HBM exhausted) marks the entry do-not-retry and the call takes the stateless path.
// SNARKVM_HIP_BASE_CACHE_MB caps the device bytes per device (default 65536); least recently used ranges go first.
// Callers that can change their code should use snarkvm_hip_register_bases* + snarkvm_hip_msm_registered* instead.
struct base_cache_entry {
    const uint8_t* host = nullptr;
    size_t n = 0, stride = 0;
    std::shared_ptr<snarkvm_hip_bases> h;  // null: seen once, not registered yet (shared: a running call keeps a dropped entry's tables alive)
    std::vector<uint8_t> samples;     // 97 bytes (x, y, infinity) of points 0, CACHE_STEP, 2 * CACHE_STEP, ...
    uint64_t last_use = 0;
    bool failed = false;              // registration failed once (out of memory): never retried
    size_t bytes() const { return h ? (size_t)h->tables * h->n * sizeof(g1_aff_mem_t) : 0; }
};
static constexpr size_t CACHE_STEP = 64, CACHE_MAX = 8;
static std::mutex g_cache_mu;
static std::vector<base_cache_entry> g_base_cache;
static uint64_t g_cache_tick = 0;
static std::atomic<int> g_base_cache_override{-1};  // snarkvm_hip_set_base_cache: >= 0 replaces the environment's value
static int base_cache_tables() {
    static const int env = getenv("SNARKVM_HIP_BASE_CACHE") ? atoi(getenv("SNARKVM_HIP_BASE_CACHE")) : 0;  // off unless asked for
    const int o = g_base_cache_override.load(std::memory_order_relaxed);
    const int t = o >= 0 ? o : env;
    return (t == 1 || t == 2 || t == 4 || t == 8 || t == 16) ? t : 0;
}
static size_t base_cache_cap() {
    static const size_t mb = getenv("SNARKVM_HIP_BASE_CACHE_MB") ? (size_t)atoll(getenv("SNARKVM_HIP_BASE_CACHE_MB")) : 65536;
    return mb << 20;
}
static void base_cache_drop(size_t i) { g_base_cache.erase(g_base_cache.begin() + (long)i); }
// every sampled point of [off, off + npoints) still equals its raw copy
static bool samples_match(const base_cache_entry& e, size_t off, size_t npoints) {
    if (!npoints) return false;
    const size_t k0 = (off + CACHE_STEP - 1) / CACHE_STEP, k1 = (off + npoints - 1) / CACHE_STEP;  // sample indices inside the slice
    if (k0 > k1) return false;                                                                     // no sample inside: cannot vouch for it
    for (size_t k = k0; k <= k1; k++)
        if (memcmp(&e.samples[k * 97], e.host + k * CACHE_STEP * e.stride, 97) != 0) return false;
    return true;
}
// registered handle + offset covering [points, points + npoints * stride); nullptr: use the uncached path this time
static std::shared_ptr<snarkvm_hip_bases> base_cache_lookup(const void* points, size_t npoints, size_t stride, size_t& offset) {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    const uint8_t* p = (const uint8_t*)points;
    for (size_t i = 0; i < g_base_cache.size(); i++) {
        if (g_base_cache[i].stride != stride || p < g_base_cache[i].host || p + npoints * stride > g_base_cache[i].host + g_base_cache[i].n * stride ||
            (size_t)(p - g_base_cache[i].host) % stride)
            continue;
        const size_t off = (size_t)(p - g_base_cache[i].host) / stride;
        if (!samples_match(g_base_cache[i], off, npoints)) {  // the memory behind the range changed: forget it
            base_cache_drop(i);
            break;
        }
        g_base_cache[i].last_use = ++g_cache_tick;
        if (!g_base_cache[i].h) {
            // second sighting.  Only the memory of THIS call may be read: the range is registered when the call passes exactly
            // the remembered range (every sample of it was just verified above); a sub-slice of a range that is not registered
            // yet takes the stateless path.
            if (g_base_cache[i].failed || off != 0 || npoints != g_base_cache[i].n) return nullptr;
            // table geometry by size, as measured (profiles/r02_size_sweep.md, r04_geometry_23.md): 12 x 22-bit windows from 2^24 points, 13 x 20-bit from
            // 2^21, 17 x 15-bit below 2^18 (half the buckets of 16 x 16: the whole fold is one round of 256 workgroups), else the
            // configured count of 256 / tables-bit tables
            int tables = base_cache_tables(), bits = 0;
            if (tables == 16 && g_base_cache[i].n >= ((size_t)1 << 24)) tables = 12, bits = 22;  // 2^23: 13 x 20 is 4 % faster (profiles/r04_geometry_23.md)
            else if (tables == 16 && g_base_cache[i].n >= ((size_t)1 << 21)) tables = 13, bits = 20;
            else if (tables == 16 && g_base_cache[i].n < ((size_t)1 << 18)) tables = 17, bits = 15;
            const size_t need = (size_t)tables * g_base_cache[i].n * sizeof(g1_aff_mem_t);
            if (need > base_cache_cap()) return nullptr;
            for (;;) {  // make room: least recently used registered entries go first
                size_t used = 0, lru = (size_t)-1;
                for (size_t j = 0; j < g_base_cache.size(); j++) {
                    used += g_base_cache[j].bytes();
                    if (j != i && g_base_cache[j].h && (lru == (size_t)-1 || g_base_cache[j].last_use < g_base_cache[lru].last_use)) lru = j;
                }
                if (used + need <= base_cache_cap() || lru == (size_t)-1) break;
                base_cache_drop(lru);
                if (lru < i) i--;
            }
            snarkvm_hip_bases_t* nh = nullptr;
            try {
                register_bases_impl(&nh, points, npoints, stride, 0, tables, bits);
            } catch (...) {  // nothing leaks (register_bases_impl frees its replicas); this call and later ones go stateless
                (void)hipGetLastError();
                g_base_cache[i].failed = true;
                return nullptr;
            }
            g_base_cache[i].h = std::shared_ptr<snarkvm_hip_bases>(nh, [](snarkvm_hip_bases* b) { snarkvm_hip_free_bases(b); });
        }
        offset = off;
        return g_base_cache[i].h;
    }
    // first sighting: remember the range (a bigger range supersedes the ranges it contains)
    for (size_t i = g_base_cache.size(); i-- > 0;)
        if (g_base_cache[i].stride == stride && g_base_cache[i].host >= p && g_base_cache[i].host + g_base_cache[i].n * stride <= p + npoints * stride)
            base_cache_drop(i);
    while (g_base_cache.size() >= CACHE_MAX) {
        size_t lru = 0;
        for (size_t i = 1; i < g_base_cache.size(); i++)
            if (g_base_cache[i].last_use < g_base_cache[lru].last_use) lru = i;
        base_cache_drop(lru);
    }
    base_cache_entry e;
    e.host = p;
    e.n = npoints;
    e.stride = stride;
    for (size_t k = 0; k * CACHE_STEP < e.n; k++) e.samples.insert(e.samples.end(), e.host + k * CACHE_STEP * e.stride, e.host + k * CACHE_STEP * e.stride + 97);
    e.last_use = ++g_cache_tick;
    g_base_cache.push_back(std::move(e));
    return nullptr;
}

// The API form of SNARKVM_HIP_BASE_CACHE (a host that cannot set the environment before the library is loaded; A/B runs in one process).
// 0 also drops every cached range (calls in flight keep the tables they hold until they return).
RustError snarkvm_hip_set_base_cache(int tables) {
    if (!(tables == 0 || tables == 1 || tables == 2 || tables == 4 || tables == 8 || tables == 16)) return fail(1, "snarkvm_hip_set_base_cache: tables must be 0, 1, 2, 4, 8 or 16");
    g_base_cache_override.store(tables, std::memory_order_relaxed);
    if (tables == 0) {
        std::vector<base_cache_entry> drop;
        {
            std::lock_guard<std::mutex> lk(g_cache_mu);
            drop.swap(g_base_cache);
        }
        // the entries' tables are freed here, outside the lock (hipFree waits for the device)
    }
    return ok();
}

RustError snarkvm_msm(void* out, const void* points, size_t npoints, const void* scalars, size_t ffi_affine_sz) {
    API_TRY
    if (!out) throw hip_failure{hipErrorInvalidValue, "msm: null output", __LINE__};
    g_rt.configure();  // no device: an error, never a silent host result
    if (npoints == 0) {
        write_infinity<fq_t>(out);
    } else {
        if (!points || !scalars) throw hip_failure{hipErrorInvalidValue, "msm: null argument", __LINE__};
        std::shared_ptr<snarkvm_hip_bases> h;
        size_t offset = 0;
        if (base_cache_tables() && npoints > 1024 && ffi_affine_sz >= 104 && !(ffi_affine_sz & 7)) h = base_cache_lookup(points, npoints, ffi_affine_sz, offset);
        if (h && msm_coalescible(*h, npoints, 0))
            msm_single_coalesced(out, h.get(), offset, npoints, 0, 0, scalars, 0, 0, 0);
        else if (h)
            msm_registered_host_scalars(out, h.get(), offset, npoints, 0, 0, scalars, 0, 0);
        else
            msm_host_chunked<fq_t>(out, points, npoints, scalars, ffi_affine_sz);
    }
    API_CATCH
}

RustError snarkvm_hip_g1_sum(void* out, const void* in_projective, size_t n) {
    API_BEGIN
    if (!out || (n && !in_projective)) throw hip_failure{hipErrorInvalidValue, "g1_sum: null argument", __LINE__};
    if (n == 0) {
        write_infinity<fq_t>(out);
    } else {
        c.poly[0].ensure(n * 144 + 144);
        uint32_t* d_in = c.poly[0].as<uint32_t>();
        uint32_t* d_out = d_in + 36 * n;
        HIP_TRY(hipMemcpyAsync(d_in, in_projective, n * 144, hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(g1_sum_kernel, dim3(1), dim3(64), 0, c.stream, (const uint32_t*)d_in, n, d_out);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, d_out, 144, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    API_END
}
RustError snarkvm_hip_g1_to_affine(void* out_affine, const void* in_projective, size_t n) {
    API_BEGIN
    if (n) {
        if (!out_affine || !in_projective) throw hip_failure{hipErrorInvalidValue, "g1_to_affine: null argument", __LINE__};
        c.poly[0].ensure(n * 144);
        c.poly[1].ensure(n * 104);
        HIP_TRY(hipMemcpyAsync(c.poly[0].p, in_projective, n * 144, hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(g1_to_affine_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, c.poly[0].as<uint32_t>(), c.poly[1].as<uint32_t>(), n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_affine, c.poly[1].p, n * 104, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    API_END
}

// ---- synthetic bases ---------------------------------------------------------------------------
// G1 generator (g1.rs:219-253), memory Montgomery form, 64-bit limbs
static const uint64_t G1_GEN_X[6] = {1171681672315280277ull, 6528257384425852712ull,  7514971432460253787ull,
                                     2032708395764262463ull, 12876543207309632302ull, 107509843840671767ull};
static const uint64_t G1_GEN_Y[6] = {13572190014569192121ull, 15344828677741220784ull, 17067903700058808083ull,
                                     10342263224753415805ull, 1083990386877464092ull,  21335464879237822ull};
RustError snarkvm_hip_g1_generate_bases_device(void* d_out, uint64_t start, size_t npoints) {
    API_BEGIN_DEV(device_for(d_out, npoints ? 1 : 0))
    if (npoints) {
        // convert the generator on the host with the same arithmetic
        uint32_t xw[12], yw[12];
        memcpy(xw, G1_GEN_X, 48);
        memcpy(yw, G1_GEN_Y, 48);
        g1_aff_t g{fq_t::unpack(xw).from_mem_mont(), fq_t::unpack(yw).from_mem_mont()};
        g1_aff_mem_t gm;
        g.x.pack(gm.x.w);
        g.y.pack(gm.y.w);
        c.gen_pts.ensure(npoints * sizeof(g1_xyzz_mem_t));
        c.gen_prod.ensure(npoints * sizeof(fq_mem_t));
        const size_t threads = (npoints + GEN_RUN - 1) / GEN_RUN;
        hipLaunchKernelGGL(g1_generate_bases_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c.stream, gm, start, npoints,
                           (uint8_t*)d_out, (size_t)104, c.gen_pts.as<g1_xyzz_mem_t>(), c.gen_prod.as<fq_mem_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(c.stream));
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
    c.poly[0].ensure(bytes + 16);
    c.poly[1].ensure(bytes + 16);
    c.poly[2].ensure(bytes + 16);
    HIP_TRY(hipMemcpyAsync(c.poly[0].p, a, bytes, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemcpyAsync(c.poly[1].p, b ? b : a, bytes, hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(devtest_field_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c.stream, field, op, c.poly[0].as<uint32_t>(),
                       c.poly[1].as<uint32_t>(), c.poly[2].as<uint32_t>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c.poly[2].p, bytes, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    API_END
}
// The MSM planner on the host (no device needed): out = {c, W, J, Wd, nb, nbt, S, S2, L, wide}.  Returns 0.
int snarkvm_hip_selftest_msm_plan(size_t n, int window_bits, int tables, int table_bits, uint32_t* out) {
    const msm_plan_t p = msm_make_plan(n, window_bits, tables, table_bits);
    const uint32_t v[10] = {(uint32_t)p.c, (uint32_t)p.W, (uint32_t)p.J, (uint32_t)p.Wd, p.nb, p.nbt, p.S, p.S2, p.L, p.c > 16 ? 1u : 0u};
    for (int i = 0; i < 10; i++) out[i] = v[i];
    // bias must place one 2^(c-1) per digit row below MSM_BIAS_BITS
    uint32_t chk[10] = {0};
    for (int w = 0; w < p.Wd; w++) {
        const int bit = p.c - 1 + p.c * w;
        if (bit >= MSM_BIAS_BITS) return 1;
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
// The lazily reduced accumulate arithmetic (ffl.hip.h) against the exact one (ff.hip.h / ec.hip.h) on the host, no device needed:
// a chain of `iters` mixed additions of +-k G (k from a SplitMix64 stream over a pool of 64 multiples of the generator, so that
// P + P, P + (-P) and additions to infinity all occur), every coordinate of the accumulator compared after every step; also the
// field routines on values at the edges of their documented ranges.  Returns 0, or the (1-based) step of the first mismatch.
int snarkvm_hip_selftest_fq_lazy(uint64_t seed, int iters) {
    uint32_t xw[12], yw[12];
    memcpy(xw, G1_GEN_X, 48);
    memcpy(yw, G1_GEN_Y, 48);
    const g1_aff_t g{fq_t::unpack(xw).from_mem_mont(), fq_t::unpack(yw).from_mem_mont()};
    // pool[k] = (k + 1) G, affine, exact internal form
    std::vector<g1_aff_t> pool;
    g1_xyzz_t run = g1_xyzz_t::inf();
    for (int k = 0; k < 64; k++) {
        run.add_affine(g);
        const fq_t izzz = run.zzz.inverse();
        const fq_t izz = izzz.sqr() * run.zz.sqr();
        pool.push_back({run.x * izz, run.y * izzz});
    }
    const fq_t c406 = fq_t::from_table(FqLConv::C406), c348 = fq_t::from_table(FqLConv::C348);
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    g1_xyzz_t exact = g1_xyzz_t::inf();
    xyzz_lazy_t lazy = xyzz_lazy_t::infinity();
    int prev = -1;
    bool prev_neg = false;
    for (int it = 0; it < iters; it++) {
        const uint64_t r = next();
        int k = (int)(r % 64);
        bool neg = ((r >> 8) & 1) != 0;
        const int mode = (int)((r >> 16) % 16);
        if (mode == 0 && prev >= 0) k = prev, neg = prev_neg;    // the same point again
        if (mode == 1 && prev >= 0) k = prev, neg = !prev_neg;   // its negative
        if (mode == 2) {                                         // restart from infinity: acc = P, then P again -> doubling
            exact = g1_xyzz_t::inf();
            lazy = xyzz_lazy_t::infinity();
        }
        prev = k;
        prev_neg = neg;
        const g1_aff_t p = pool[k];
        exact.add_affine(p, neg);
        const fql_t px = fql_t::from_limbs(p.x * c406), py = fql_t::from_limbs(p.y * c406);
        if (!lazy.madd(px, py, neg)) {  // exceptional (or a false alarm of the low-limb filter): the exact arithmetic decides
            g1_xyzz_t e = lazy.to_exact();
            fq_t ex, ey;
            for (int i = 0; i < 13; i++) ex.v[i] = (uint32_t)px.v[i], ey.v[i] = (uint32_t)py.v[i];
            e.add_affine({ex * c348, ey * c348}, neg);
            lazy = xyzz_lazy_t::from_exact(e);
        }
        const g1_xyzz_t got = lazy.to_exact();
        if (got.is_inf() != exact.is_inf()) return it + 1;
        if (!exact.is_inf() && (got.x != exact.x || got.y != exact.y || got.zz != exact.zz || got.zzz != exact.zzz)) return it + 1;
    }
    // field routines on operands at the edges of their ranges: a = -q .. values near +-q with extreme limbs
    for (int t = 0; t < 200; t++) {
        fq_t a, b;
        for (int i = 0; i < 13; i++) a.v[i] = (uint32_t)(next() & LIMB_MASK), b.v[i] = (uint32_t)(next() & LIMB_MASK);
        a.v[12] &= 0x00ffffffu;
        b.v[12] &= 0x00ffffffu;
        if (t & 1) for (int i = 0; i < 12; i++) a.v[i] = LIMB_MASK;  // all-ones limbs
        if (t & 2) for (int i = 0; i < 12; i++) b.v[i] = (i & 1) ? LIMB_MASK : 0;
        const fql_t la = fql_t::from_exact(a), lb = fql_t::from_exact(b);
        if (fql_t::mul(la, lb).to_exact() != a * b) return -(4 * t + 1);
        if (fql_t::sqr(la - lb).to_exact() != (a - b).sqr()) return -(4 * t + 2);
        if (fql_t::diff_of_products(la - lb, lb - la, fql_t::from_exact(a * b).normalized(), la).to_exact() != fq_t::diff_of_products(a - b, b - a, a * b, a))
            return -(4 * t + 3);
        if (((la + lb) - (lb + lb)).normalized().to_exact() != (a - b)) return -(4 * t + 4);
        // raw memory image (how partial sums leave the accumulate kernel): a wide signed value through store_raw / load_raw
        alignas(16) uint32_t raw[12];
        const fql_t wide = (la - lb - lb - lb).normalized();
        wide.store_raw(raw);
        const fql_t back = fql_t::load_raw(raw);
        for (int i = 0; i < 13; i++)
            if (back.v[i] != wide.v[i]) return -(100000 + t);
    }
    return 0;
}
// The lazy arithmetic behind field-like operators (ffl.hip.h::fqz_t: the tail of a G1 MSM under tuning lazy_tail) against the exact
// one, on the host: two chains over the generic xyzz_t<F> - F = fq_t and F = fqz_t - fed the same GENERAL points (multiples of the
// generator with random zz / zzz scalings, so that the full addition law runs, not the mixed one), with repeated points (the doubling
// inside the addition), negatives (infinity), restarts and explicit doublings; every coordinate of the lazy accumulator is converted and
// compared after every step.  Returns 0, or the (1-based) step of the first mismatch.
int snarkvm_hip_selftest_g1_lazy_tail(uint64_t seed, int iters) {
    uint32_t xw[12], yw[12];
    memcpy(xw, G1_GEN_X, 48);
    memcpy(yw, G1_GEN_Y, 48);
    const g1_aff_t g{fq_t::unpack(xw).from_mem_mont(), fq_t::unpack(yw).from_mem_mont()};
    std::vector<g1_xyzz_t> pool;
    g1_xyzz_t run = g1_xyzz_t::inf();
    for (int k = 0; k < 48; k++) {
        run.add_affine(g);
        pool.push_back(run);
    }
    auto next = [&]() {
        seed += 0x9E3779B97F4A7C15ull;
        uint64_t z = seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    };
    auto to_lazy = [](const g1_xyzz_t& p) {
        xyzz_t<fqz_t> r = xyzz_t<fqz_t>::inf();
        if (!p.is_inf()) r = {{fql_t::from_exact(p.x)}, {fql_t::from_exact(p.y)}, {fql_t::from_exact(p.zz)}, {fql_t::from_exact(p.zzz)}};
        return r;
    };
    g1_xyzz_t exact = g1_xyzz_t::inf();
    xyzz_t<fqz_t> lazy = xyzz_t<fqz_t>::inf();
    g1_xyzz_t prev = g1_xyzz_t::inf();
    for (int it = 0; it < iters; it++) {
        const uint64_t r = next();
        g1_xyzz_t p = pool[r % 48];
        // another representative of the same point: (l^2 X, l^3 Y, l^2 ZZ, l^3 ZZZ)
        fq_t l = fq_t::zero();
        for (int i = 0; i < 12; i++) l.v[i] = (uint32_t)(next() & LIMB_MASK);
        l.v[0] |= 1;
        const fq_t l2 = l.sqr(), l3 = l2 * l;
        p = {p.x * l2, p.y * l3, p.zz * l2, p.zzz * l3};
        if ((r >> 8) & 1) p.y = p.y.neg();
        const int mode = (int)((r >> 16) % 16);
        if (mode == 0) p = prev;                                  // the same representative again
        if (mode == 1) p = {prev.x, prev.y.neg(), prev.zz, prev.zzz};  // its negative
        if (mode == 2) {                                         // restart: acc = P, then (next step, mode 0 or 3) P again -> doubling
            exact = g1_xyzz_t::inf();
            lazy = xyzz_t<fqz_t>::inf();
        }
        if (mode == 3) p = exact;                                // acc + acc: the doubling inside the addition
        if (mode == 4) {                                         // explicit doubling
            exact = exact.dbl();
            lazy = lazy.dbl();
        } else {
            prev = p;
            exact.add(p);
            lazy.add(to_lazy(p));
        }
        if (lazy.is_inf() != exact.is_inf()) return it + 1;
        if (!exact.is_inf() && (lazy.x.to_exact() != exact.x || lazy.y.to_exact() != exact.y || lazy.zz.to_exact() != exact.zz || lazy.zzz.to_exact() != exact.zzz))
            return it + 1;
        // the memory image the tail kernels exchange
        xyzz_mem_t<fqz_t> m;
        store_xyzz<fqz_t>(&m, lazy);
        lazy = load_xyzz<fqz_t>(&m);
    }
    // a representative of zero other than 0 is still the point at infinity / an equal coordinate: q and -q as limbs
    fqz_t zq;
    for (int i = 0; i < 13; i++) zq.a.v[i] = FqL::MOD[i];
    if (!zq.is_zero() || !zq.neg().is_zero() || !(zq + zq).is_zero() || (zq + fqz_t::one()).is_zero()) return -1;
    return 0;
}
// The host-side finish of an MSM (runtime.hip.h msm_accum_t) on its own, no device needed: out (144 B) = sum_i 2^pos[i] *
// planes[i] for `n` G1 points given as Jacobian memory images.  Lets the CPU test-suite pin the Horner code on the oracle.
int snarkvm_hip_selftest_g1_finish(const void* planes_jacobian, const int32_t* pos, size_t n, void* out) {
    try {
        std::unique_ptr<msm_accum_t<fq_t>> acc(new msm_accum_t<fq_t>());
        const uint32_t* src = (const uint32_t*)planes_jacobian;
        for (size_t i = 0; i < n; i++) {
            const g1_jac_t j = {fq_t::from_raw_words(src + 36 * i), fq_t::from_raw_words(src + 36 * i + 12), fq_t::from_raw_words(src + 36 * i + 24)};
            acc->add(pos[i], g1_xyzz_t::from_jacobian(j));
        }
        acc->finish(out);
        return 0;
    } catch (...) {
        return 1;
    }
}

}  // extern "C"
