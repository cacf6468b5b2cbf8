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

}  // namespace snarkvm_hip
