// snarkvm_hip.hpp - C++ host mirror of the reference's accelerator crate `snarkvm_algorithms_cuda`
// (algorithms/cuda/src/lib.rs:77-168): the same three functions NTT / polymul / msm with the same argument
// meaning and error behaviour, over the C ABI of snarkvm_hip.h.  Header-only; link with -lsnarkvm_hip.
//
// `Err(cuda::Error)` becomes a thrown snarkvm_hip::Error (code + message); the two argument checks the Rust
// wrapper performs before the FFI call (power-of-two domain, lib.rs:84-86,107-109; npoints <= points.len(),
// lib.rs:150-152) throw std::invalid_argument, mirroring Rust's panic.
#pragma once
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "snarkvm_hip.h"

namespace snarkvm_hip {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(RustError e) {
    if (e.code != 0) {
        std::string m = e.message ? e.message : "";
        std::free(e.message);  // the callee allocated it (TAKE_RESPONSIBILITY_FOR_ERROR_MESSAGE, build.rs:79)
        throw Error(e.code, m);
    }
}

// lib.rs:77-97
template <class T>
inline void NTT(size_t domain_size, T* inout, NTTInputOutputOrder order, NTTDirection direction, NTTType type) {
    static_assert(sizeof(T) == 32, "the NTT operates on 32-byte Fr elements (fft/domain.rs:377)");
    if (domain_size == 0 || (domain_size & (domain_size - 1)) != 0) throw std::invalid_argument("domain_size is not power of 2");
    uint32_t lg = 0;
    while (((size_t)1 << lg) < domain_size) lg++;
    check(snarkvm_ntt(inout, lg, order, direction, type));
}

// lib.rs:100-145: returns a vector of `domain` elements pre-filled with `zero`
template <class T>
inline std::vector<T> polymul(size_t domain, const std::vector<std::vector<T>>& polynomials,
                              const std::vector<std::vector<T>>& evaluations, const T& zero) {
    static_assert(sizeof(T) == 32, "Fr elements are 32 bytes");
    if (domain == 0 || (domain & (domain - 1)) != 0) throw std::invalid_argument("domain_size is not power of 2");
    uint32_t lg = 0;
    while (((size_t)1 << lg) < domain) lg++;
    std::vector<const void*> pptrs, eptrs;
    std::vector<size_t> plens, elens;
    for (auto& p : polynomials) {
        pptrs.push_back(p.data());
        plens.push_back(p.size());
    }
    for (auto& e : evaluations) {
        eptrs.push_back(e.data());
        elens.push_back(e.size());
    }
    std::vector<T> out(domain, zero);
    check(snarkvm_polymul(out.data(), pptrs.size(), pptrs.data(), plens.data(), eptrs.size(), eptrs.data(), elens.data(), lg));
    return out;
}

// lib.rs:148-168
template <class Affine, class Projective, class Scalar>
inline Projective msm(const Affine* points, size_t npoints_available, const Scalar* scalars, size_t nscalars) {
    static_assert(sizeof(Scalar) == 32, "scalars are BigInteger256");
    if (nscalars > npoints_available) throw std::invalid_argument("length mismatch: fewer points than scalars");
    Projective ret;
    check(snarkvm_msm(&ret, points, nscalars, scalars, sizeof(Affine)));
    return ret;
}

// ---- extensions (part 2 of rust/snarkvm-algorithms-hip/src/lib.rs: `resident`) -------------------------------------------
// Deferred-synchronisation scope (snarkvm_hip.h: snarkvm_hip_scope_begin / _end): device-resident calls of this thread between
// construction and destruction are enqueued on one stream and waited for once.  Not copyable; end() reports errors, the
// destructor swallows them (like the Rust guard's Drop).
class Scope {
public:
    // flags: 0, or SNARKVM_HIP_SCOPE_ASYNC_MSM (registered-bases MSMs over device scalars are enqueued too; their outputs are written by end())
    // [| SNARKVM_HIP_SCOPE_STABLE_INPUTS: the scalar vectors of those MSMs are not touched before end()]
    explicit Scope(const void* d_any = nullptr, uint32_t flags = 0) { check(snarkvm_hip_scope_begin_ex(d_any, flags)); }
    void* stream() const { return snarkvm_hip_scope_stream(); }  // the hipStream_t of the scope's calls
    void set_flags(uint32_t flags) { check(snarkvm_hip_scope_set_flags(flags)); }  // for the calls that follow (e.g. | SNARKVM_HIP_SCOPE_MSM_IN_STREAM after the independent MSM is out)
    void collect(const void* out = nullptr) { check(snarkvm_hip_scope_collect(out)); }  // the outputs of that MSM call (null: of all enqueued so far); the scope stays open
    Scope(const Scope&) = delete;
    Scope& operator=(const Scope&) = delete;
    void end() {
        if (open_) {
            open_ = false;
            check(snarkvm_hip_scope_end());
        }
    }
    ~Scope() {
        if (open_) {
            RustError e = snarkvm_hip_scope_end();
            if (e.message) std::free(e.message);
        }
    }

private:
    bool open_ = true;
};

// Device memory owned through the library (snarkvm_hip_malloc / _free / _memcpy_*; rust: resident::DeviceBuffer): the operands of every `d_*`
// argument of the extension ABI without a HIP header on the caller's side.  Host-side copies are complete on return; copy_from() / fill() inside a
// Scope are only enqueued on the scope's stream.  Not copyable; movable.
class DeviceBuffer {
public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes, int device = -1) : bytes_(bytes) { check(snarkvm_hip_malloc(&p_, bytes, device)); }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr, o.bytes_ = 0; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept {
        if (this != &o) {
            release();
            p_ = o.p_, bytes_ = o.bytes_;
            o.p_ = nullptr, o.bytes_ = 0;
        }
        return *this;
    }
    ~DeviceBuffer() { release(); }
    void* data() const { return p_; }
    size_t size() const { return bytes_; }
    void* at(size_t byte_offset) const {
        if (byte_offset > bytes_) throw std::out_of_range("DeviceBuffer::at");
        return static_cast<char*>(p_) + byte_offset;
    }
    void upload(const void* src, size_t bytes, size_t byte_offset = 0) const {
        if (byte_offset + bytes > bytes_) throw std::out_of_range("DeviceBuffer::upload");
        check(snarkvm_hip_memcpy_h2d(at(byte_offset), src, bytes));
    }
    void download(void* dst, size_t bytes, size_t byte_offset = 0) const {
        if (byte_offset + bytes > bytes_) throw std::out_of_range("DeviceBuffer::download");
        check(snarkvm_hip_memcpy_d2h(dst, at(byte_offset), bytes));
    }
    void copy_from(size_t byte_offset, const void* d_src, size_t bytes) const {
        if (byte_offset + bytes > bytes_) throw std::out_of_range("DeviceBuffer::copy_from");
        check(snarkvm_hip_memcpy_d2d(at(byte_offset), d_src, bytes));
    }
    void fill(size_t byte_offset, int value, size_t bytes) const {
        if (byte_offset + bytes > bytes_) throw std::out_of_range("DeviceBuffer::fill");
        check(snarkvm_hip_memset(at(byte_offset), value, bytes));
    }

private:
    void release() {
        if (p_) {
            RustError e = snarkvm_hip_free(p_);
            if (e.message) std::free(e.message);
            p_ = nullptr;
        }
    }
    void* p_ = nullptr;
    size_t bytes_ = 0;
};

// Registered bases (an SRS resident in HBM with precomputed window tables): register once, commit per call.  `tables` x
// `window_bits` must cover 254 bits (17 x 15 for proof-sized MSMs, 12 x 22 at 2^24).  Concurrent commit() calls of proof size
// are fused inside the library (runtime.hip.h::msm_coalesced).
template <class Affine, class Projective>
class RegisteredBases {
public:
    RegisteredBases(const Affine* bases, size_t n, bool on_device, int tables, int window_bits) : n_(n) {
        check(snarkvm_hip_register_bases_windowed(&h_, bases, n, sizeof(Affine), on_device ? 1 : 0, tables, window_bits));
    }
    RegisteredBases(const RegisteredBases&) = delete;
    RegisteredBases& operator=(const RegisteredBases&) = delete;
    ~RegisteredBases() {
        if (h_) snarkvm_hip_free_bases(h_);
    }
    size_t size() const { return n_; }
    const snarkvm_hip_bases_t* handle() const { return h_; }  // for the raw snarkvm_hip_msm_registered* calls (device-resident scalars, batches)
    // sum_i scalars[i] * bases[offset + i]  (+ sum_j scalars[n + j] * bases[offset1 + j] when n1 > 0: KZG10's hiding term)
    Projective commit(size_t offset, size_t n, const void* scalars, bool scalars_on_device, bool scalars_montgomery = false, size_t offset1 = 0,
                      size_t n1 = 0) const {
        Projective out;
        check(snarkvm_hip_msm_registered_ex(&out, h_, offset, n, offset1, n1, scalars, scalars_on_device ? 1 : 0, scalars_montgomery ? 1 : 0, 0));
        return out;
    }

private:
    snarkvm_hip_bases_t* h_ = nullptr;
    size_t n_;
};

}  // namespace snarkvm_hip
