// api_serde.hip - canonical (de)serialisation of G1 points on the device (serde.hip.h): the reference's point encoding
// (curves/src/templates/macros.rs:66-140, utilities/src/serialize/flags.rs:72-99) and `.usrs` bodies straight into registered
// bases.  A separate translation unit so that build.py compiles the square-root / subgroup-check kernels beside the MSM units.
#include "runtime.hip.h"

snarkvm_hip_bases* sv_new_bases_handle(size_t npoints, int tables, int table_bits);  // api.hip
void sv_precompute_tables(lane_t& c, snarkvm_hip_bases* h, g1_aff_mem_t* d);        // api.hip

// bytes (host) -> native base slots and / or Rust-layout records (both device); returns the SERDE_* status bits
static uint32_t g1_deserialize_run(lane_t& c, const void* bytes, size_t n, int compressed, int validate, g1_aff_mem_t* d_native, uint8_t* d_rust) {
    const size_t psz = compressed ? 48 : 96;
    c.bases_tmp.ensure(n * psz);
    c.serde_status.ensure(4);
    HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, bytes, n * psz, hipMemcpyHostToDevice, c.stream));
    HIP_TRY(hipMemsetAsync(c.serde_status.p, 0, 4, c.stream));
    hipLaunchKernelGGL(g1_deserialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, c.bases_tmp.as<uint8_t>(), n, compressed, validate,
                       d_native, d_rust, c.serde_status.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    uint32_t st = 0;
    HIP_TRY(hipMemcpyAsync(&st, c.serde_status.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIP_TRY(hipStreamSynchronize(c.stream));
    return st;
}

extern "C" {

RustError snarkvm_hip_register_bases_serialized(snarkvm_hip_bases_t** handle, const void* bytes, size_t npoints, int compressed, int validate,
                                                int tables) {
    API_TRY
    if (!handle || (npoints && !bytes)) throw hip_failure{hipErrorInvalidValue, "register_bases_serialized: null argument", __LINE__};
    check_tables(tables, 0, "register_bases_serialized");
    std::unique_ptr<snarkvm_hip_bases> h(sv_new_bases_handle(npoints, tables, 256 / tables));
    if (npoints) {
        try {
            std::vector<int> all;
            for (int d = 0; d < g_rt.ndev(); d++) all.push_back(d);
            for_each_device(all, [&](int dev) {  // every device decodes its own replica from the host bytes
                lane_guard lg(dev);
                lane_t& c = lg.c();
                HIP_TRY(hipMalloc((void**)&h->d[dev], (size_t)tables * npoints * sizeof(g1_aff_mem_t)));
                serde_throw_on_status(g1_deserialize_run(c, bytes, npoints, compressed, validate, h->d[dev], nullptr), "register_bases_serialized");
                sv_precompute_tables(c, h.get(), h->d[dev]);
                bases_to_lazy_form(c, h->d[dev], (size_t)tables * npoints);
                HIP_TRY(hipStreamSynchronize(c.stream));
            });
        } catch (...) {
            h->free_all();
            throw;
        }
    }
    *handle = h.release();
    API_CATCH
}
RustError snarkvm_hip_g1_deserialize(void* out_affine, const void* bytes, size_t n, int compressed, int validate) {
    API_BEGIN
    if (n) {
        if (!out_affine || !bytes) throw hip_failure{hipErrorInvalidValue, "g1_deserialize: null argument", __LINE__};
        c.poly[0].ensure(n * 104);
        const uint32_t st = g1_deserialize_run(c, bytes, n, compressed, validate, nullptr, c.poly[0].as<uint8_t>());
        serde_throw_on_status(st, "g1_deserialize");
        HIP_TRY(hipMemcpyAsync(out_affine, c.poly[0].p, n * 104, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    API_END
}
RustError snarkvm_hip_g1_serialize(void* out_bytes, const void* affine, size_t n, size_t ffi_affine_sz, int compressed) {
    API_BEGIN
    if (n) {
        if (!out_bytes || !affine) throw hip_failure{hipErrorInvalidValue, "g1_serialize: null argument", __LINE__};
        if (ffi_affine_sz < 104 || (ffi_affine_sz & 7)) throw hip_failure{hipErrorInvalidValue, "g1_serialize: bad stride", __LINE__};
        const size_t psz = compressed ? 48 : 96;
        c.bases_tmp.ensure(n * ffi_affine_sz);
        c.poly[0].ensure(n * psz);
        HIP_TRY(hipMemcpyAsync(c.bases_tmp.p, affine, n * ffi_affine_sz, hipMemcpyHostToDevice, c.stream));
        hipLaunchKernelGGL(g1_serialize_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c.stream, c.bases_tmp.as<uint8_t>(), ffi_affine_sz, n,
                           compressed, c.poly[0].as<uint8_t>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out_bytes, c.poly[0].p, n * psz, hipMemcpyDeviceToHost, c.stream));
        HIP_TRY(hipStreamSynchronize(c.stream));
    }
    API_END
}

}  // extern "C"
