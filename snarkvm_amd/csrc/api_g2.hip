// api_g2.hip - the G2 (Fq2) entry points of the C ABI: the MSM engine, table precomputation and point encoding instantiated
// over fq2_t.  A separate translation unit only because these instantiations are half of the compile time: build.py compiles
// the units in parallel.
#include "runtime.hip.h"

struct snarkvm_hip_bases_g2 : bases_handle_t<fq2_t> {};

extern "C" {

RustError snarkvm_hip_msm_g2(void* out, const void* points, size_t npoints, const void* scalars, size_t ffi_affine_sz) {
    API_TRY
#ifdef SV_NO_G2  // development builds only (python -m snarkvm_amd.build --fast): skips the Fq2 kernel instantiations
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!out) throw hip_failure{hipErrorInvalidValue, "msm_g2: null output", __LINE__};
    g_rt.configure();
    if (npoints == 0) {
        write_infinity<fq2_t>(out);
    } else {
        if (!points || !scalars) throw hip_failure{hipErrorInvalidValue, "msm_g2: null argument", __LINE__};
        msm_host_chunked<fq2_t>(out, points, npoints, scalars, ffi_affine_sz);
    }
#endif
    API_CATCH
}

// ---- registered G2 bases (extension): same engine over fq2_t, precomputed tables remove most of the Horner chain of a
// one-shot G2 MSM
RustError snarkvm_hip_register_bases_g2(snarkvm_hip_bases_g2_t** handle, const void* points, size_t npoints, size_t ffi_affine_sz, int tables,
                                        int window_bits) {
    API_TRY
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!handle || (npoints && !points)) throw hip_failure{hipErrorInvalidValue, "register_bases_g2: null argument", __LINE__};
    if (ffi_affine_sz < 200 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "register_bases_g2: bad stride", __LINE__};
    check_tables(tables, window_bits, "register_bases_g2");
    const int nd = g_rt.ndev();
    std::unique_ptr<snarkvm_hip_bases_g2> h(new snarkvm_hip_bases_g2());
    h->n = npoints;
    h->tables = tables;
    h->table_bits = window_bits ? window_bits : 256 / tables;
    h->d.assign(nd, nullptr);
    if (npoints) {
        try {
            std::vector<int> all;
            for (int d = 0; d < nd; d++) all.push_back(d);
            for_each_device(all, [&](int dev) {
                lane_guard lg(dev);
                lane_t& c = lg.c();
                HIP_TRY(hipMalloc((void**)&h->d[dev], (size_t)tables * npoints * sizeof(aff_mem_t<fq2_t>)));
                c.bases_tmp.ensure(npoints * ffi_affine_sz);
                HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, points, npoints * ffi_affine_sz, hipMemcpyHostToDevice, c.stream));
                convert_bases<fq2_t>(c, c.bases_tmp.as<uint8_t>(), ffi_affine_sz, npoints, h->d[dev]);
                precompute_tables_run<fq2_t>(c, h->d[dev], npoints, tables, h->table_bits);
                HIP_TRY(hipStreamSynchronize(c.stream));
            });
        } catch (...) {
            h->free_all();
            throw;
        }
    }
    *handle = h.release();
#endif
    API_CATCH
}
void snarkvm_hip_free_bases_g2(snarkvm_hip_bases_g2_t* h) {
    if (!h) return;
    h->free_all();
    delete h;
}
#ifndef SV_NO_G2
// a proof-sized G2 MSM of one caller: concurrent callers over the same handle are fused (runtime.hip.h::msm_coalesced)
static RustError msm_g2_single_coalesced(void* out, const snarkvm_hip_bases_g2_t* h, size_t offset, size_t npoints, const void* scalars, int scalars_on_device,
                                         int window_bits) {
    API_TRY
    if (scalars_on_device && g_rt.device_of(scalars) < 0)
        throw hip_failure{hipErrorInvalidValue, "device pointer does not belong to a device in use (snarkvm_hip_set_devices)", __LINE__};
    msm_ticket_t t;
    t.req.off0 = offset, t.req.n0 = npoints, t.req.scalars = scalars, t.req.out = out;
    t.on_device = scalars_on_device ? 1 : 0;
    t.window_bits = window_bits;
    msm_coalesced<fq2_t>(*h, &t, 1);
    API_CATCH
}
#endif
RustError snarkvm_hip_msm_g2_registered(void* out, const snarkvm_hip_bases_g2_t* h, size_t offset, size_t npoints, const void* scalars,
                                        int scalars_on_device, int window_bits) {
#ifndef SV_NO_G2
    if (h && out && scalars && offset + npoints <= h->n && (!window_bits || (window_bits >= 2 && window_bits <= MSM_C_MAX)) && g_rt.configured &&
        msm_coalescible(*h, npoints, window_bits))
        return msm_g2_single_coalesced(out, h, offset, npoints, scalars, scalars_on_device, window_bits);
#endif
    API_BEGIN_DEV(device_for(scalars, (scalars_on_device && npoints) ? 1 : 0))
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!h || offset + npoints > h->n) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: range exceeds the registered bases", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: window_bits must be 0 or 2..23", __LINE__};
    if (!out || (npoints && !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered: null argument", __LINE__};
    const uint4* d_sc = (const uint4*)scalars;
    if (!scalars_on_device && npoints) {
        c.scalars_tmp.ensure(npoints * 32);
        c.phase_begin("msm_h2d");
        HIP_TRY(hipMemcpyAsync(c.scalars_tmp.p, scalars, npoints * 32, hipMemcpyHostToDevice, c.stream));
        c.phase_end();
        d_sc = c.scalars_tmp.as<uint4>();
    }
    msm_run_sync<fq2_t>(c, h->d[c.dev->logical] + offset, d_sc, npoints, out, window_bits, nullptr, ~(size_t)0, 0, h->tables, h->n, h->table_bits);
#endif
    API_END
}
// A batch of independent G2 MSMs over one registered vector (BASELINE configs[4]: one per proof), fanned out like the G1 batch.
RustError snarkvm_hip_msm_g2_registered_batch(void* outs, const snarkvm_hip_bases_g2_t* h, size_t count, const size_t* offsets, const size_t* npoints,
                                              const void* const* scalars, int scalars_on_device, int window_bits) {
    API_TRY
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (!h) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered_batch: null handle", __LINE__};
    if (window_bits && (window_bits < 2 || window_bits > MSM_C_MAX)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered_batch: window_bits must be 0 or 2..23", __LINE__};
    if (count && (!outs || !offsets || !npoints || !scalars)) throw hip_failure{hipErrorInvalidValue, "msm_g2_registered_batch: null argument", __LINE__};
    std::vector<msm_req_t> req = msm_requests<fq2_t>(outs, count, offsets, npoints, nullptr, nullptr, scalars);
    msm_batch_dispatch<fq2_t>(*h, req, scalars_on_device, 0, window_bits);
#endif
    API_CATCH
}

static RustError g2_deserialize_impl(void* out_affine, const void* bytes, size_t n, int compressed, int validate) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (n) {
        if (!out_affine || !bytes) throw hip_failure{hipErrorInvalidValue, "g2_deserialize: null argument", __LINE__};
        const size_t psz = compressed ? 96 : 192;
        c.bases_tmp.ensure(n * psz);
        c.poly[0].ensure(n * 200);
        c.serde_status.ensure(4);
        HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, bytes, n * psz, hipMemcpyHostToDevice, c.stream));
        HIP_TRY(hipMemsetAsync(c.serde_status.p, 0, 4, c.stream));
        hipLaunchKernelGGL(g2_deserialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, c.bases_tmp.as<uint8_t>(), n, compressed, validate,
                           c.poly[0].as<uint8_t>(), c.serde_status.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        uint32_t st = 0;
        HIP_TRY(hipMemcpyAsync(&st, c.serde_status.p, 4, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipMemcpyAsync(out_affine, c.poly[0].p, n * 200, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
        serde_throw_on_status(st, "g2_deserialize");
    }
#endif
    API_END
}
static RustError g2_serialize_impl(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz, int compressed) {
    API_BEGIN
#ifdef SV_NO_G2
    throw hip_failure{hipErrorNotSupported, "this development build was compiled without G2 (SV_NO_G2)", __LINE__};
#else
    if (n) {
        if (!out_bytes || !affine) throw hip_failure{hipErrorInvalidValue, "g2_serialize: null argument", __LINE__};
        if (ffi_affine_sz < 200 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "g2_serialize: bad stride", __LINE__};
        c.bases_tmp.ensure(n * ffi_affine_sz);
        const size_t psz = compressed ? 96 : 192;
        c.poly[0].ensure(n * psz);
        HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, affine, n * ffi_affine_sz, hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(g2_serialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, c.bases_tmp.as<uint8_t>(), ffi_affine_sz, n,
                           compressed, c.poly[0].as<uint8_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_bytes, c.poly[0].p, n * psz, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
#endif
    API_END
}

RustError snarkvm_hip_g2_deserialize(void* out_affine, const void* bytes, size_t n, int validate) { return g2_deserialize_impl(out_affine, bytes, n, 0, validate); }
RustError snarkvm_hip_g2_deserialize_compressed(void* out_affine, const void* bytes, size_t n, int validate) {
    return g2_deserialize_impl(out_affine, bytes, n, 1, validate);
}
RustError snarkvm_hip_g2_serialize(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz) { return g2_serialize_impl(out_bytes, affine, n, ffi_affine_sz, 0); }
RustError snarkvm_hip_g2_serialize_compressed(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz) {
    return g2_serialize_impl(out_bytes, affine, n, ffi_affine_sz, 1);
}

}  // extern "C"
