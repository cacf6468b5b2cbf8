// api.hip - C ABI (include/snarkvm_hip.h) of the gfx950 MSM / NTT backend: the G1 / Fr entry points.
//
// Host runtime (runtime.hip.h) = what algorithms/cuda/cuda/snarkvm.cu:73-312 (snarkvm_t) and snarkvm_api.cu:23-84 are in the
// reference: a lazily constructed per-process context (device arenas, streams, twiddle tables), staging of the caller's host
// buffers, error reporting as RustError, serialisation of concurrent callers.  The G2 entry points live in api_g2.hip.
#define SV_TU_MSM_G1
#include "runtime.hip.h"

context_t g_ctx;

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

// tables 1 .. J-1 of a registered base vector: table j = 2^(256 / J) * table j-1
static void precompute_tables(snarkvm_hip_bases* h) {
    for (int j = 1; j < h->tables; j++)
        hipLaunchKernelGGL((precompute_table_kernel<fq_t>), dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, g_ctx.stream,
                           h->d + (size_t)(j - 1) * h->n, h->d + (size_t)j * h->n, h->n, h->table_bits);
    HIP_TRY(hipGetLastError());
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

// ---- canonical (de)serialisation of G1 points (serde.hip.h) ---------------------------------------
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
