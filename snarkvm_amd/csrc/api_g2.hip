// api_g2.hip - the G2 (Fq2) entry points of the C ABI: the MSM engine, table precomputation and point encoding instantiated
// over fq2_t.  A separate translation unit only because these instantiations are half of the compile time: build.py compiles
// both units in parallel.
#include "runtime.hip.h"

extern "C" {

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

}  // extern "C"
