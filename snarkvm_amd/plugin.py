"""Mirror of the reference's accelerator plugin crate `snarkvm_algorithms_cuda` (algorithms/cuda/src/lib.rs:77-168):
the three functions `NTT`, `polymul`, `msm` with the same names, argument meaning and error behaviour
(`Err(cuda::Error)` becomes `HipError`; a length mismatch panics / raises before the FFI call), bound to the
C ABI of libsnarkvm_hip.so.  Data crosses the boundary as numpy views of the Rust memory layouts
(snarkvm_amd.layout)."""
import ctypes

import numpy as np

from . import _lib
from .layout import G1_AFFINE, G1_PROJECTIVE, NTTDirection, NTTInputOutputOrder, NTTType

__all__ = ["NTT", "polymul", "msm", "NTTInputOutputOrder", "NTTDirection", "NTTType"]


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def NTT(domain_size, inout, ntt_order, ntt_direction, ntt_type):
    """lib.rs:77-97.  In-place NTT of `inout` ((domain_size, 4) u64 Montgomery limbs, C-contiguous)."""
    if domain_size & (domain_size - 1):
        raise ValueError("domain_size is not power of 2")  # lib.rs:84-86 panics
    if not (isinstance(inout, np.ndarray) and inout.dtype == np.uint64 and inout.flags.c_contiguous and inout.size == 4 * domain_size):
        raise ValueError("inout must be a C-contiguous uint64 array of domain_size x 4 limbs")
    lg = domain_size.bit_length() - 1
    err = _lib.lib().snarkvm_ntt(_ptr(inout), ctypes.c_uint32(lg), ctypes.c_int(ntt_order), ctypes.c_int(ntt_direction),
                                 ctypes.c_int(ntt_type))
    _lib.check(err)


def NTT_device_batch(lg, device_ptrs, directions=None, types=None, ntt_order=0):
    """Extension (no reference counterpart): `len(device_ptrs)` independent in-place transforms of 2^lg elements over device
    vectors, one enqueue and one synchronisation (`snarkvm_hip_ntt_device_batch`).  directions / types: per vector or None."""
    k = len(device_ptrs)
    if k == 0:
        return
    ptrs = (ctypes.c_void_p * k)(*[int(p) for p in device_ptrs])
    dirs = (ctypes.c_int * k)(*[int(d) for d in directions]) if directions is not None else None
    tys = (ctypes.c_int * k)(*[int(t) for t in types]) if types is not None else None
    _lib.check(_lib.lib().snarkvm_hip_ntt_device_batch(ptrs, ctypes.c_size_t(k), ctypes.c_uint32(lg), ctypes.c_int(ntt_order), dirs, tys))


def polymul(domain, polynomials, evaluations, zero=None):
    """lib.rs:100-145.  Returns the product as a (domain, 4) array (full domain length; the reference trims
    trailing zeros afterwards in DensePolynomial::from_coefficients_vec, multiplier.rs:93)."""
    if domain & (domain - 1):
        raise ValueError("domain_size is not power of 2")
    lg = domain.bit_length() - 1
    polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polynomials]
    evals = [np.ascontiguousarray(e, dtype=np.uint64).reshape(-1, 4) for e in evaluations]
    out = np.zeros((domain, 4), dtype=np.uint64)  # lib.rs:126-127: pre-filled with `zero`
    pp = (ctypes.c_void_p * max(1, len(polys)))(*[p.ctypes.data for p in polys])
    pl = (ctypes.c_size_t * max(1, len(polys)))(*[p.shape[0] for p in polys])
    ep = (ctypes.c_void_p * max(1, len(evals)))(*[e.ctypes.data for e in evals])
    el = (ctypes.c_size_t * max(1, len(evals)))(*[e.shape[0] for e in evals])
    err = _lib.lib().snarkvm_polymul(_ptr(out), ctypes.c_size_t(len(polys)), pp, pl, ctypes.c_size_t(len(evals)), ep, el,
                                     ctypes.c_uint32(lg))
    _lib.check(err)
    return out


def msm(points, scalars):
    """lib.rs:148-168.  points: G1_AFFINE array (Rust layout, 104 B stride); scalars: (n, 4) u64 canonical
    integers.  npoints = len(scalars); fewer points than scalars is the caller's bug (lib.rs:150-152 panics)."""
    points = np.ascontiguousarray(points, dtype=G1_AFFINE).reshape(-1)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    npoints = scalars.shape[0]
    if npoints > points.shape[0]:
        raise ValueError(f"length mismatch {points.shape[0]} points < {npoints} scalars")
    ret = np.zeros(1, dtype=G1_PROJECTIVE)
    err = _lib.lib().snarkvm_msm(_ptr(ret), _ptr(points), ctypes.c_size_t(npoints), _ptr(scalars),
                                 ctypes.c_size_t(G1_AFFINE.itemsize))
    _lib.check(err)
    return ret
