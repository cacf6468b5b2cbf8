"""Setup-time group operations on the gfx950 backend.

    FixedBase::{get_mul_window_size, get_window_table, msm}     algorithms/src/msm/fixed_base.rs:26-97
    UniversalParams::lagrange_basis (group-element iFFT)         algorithms/src/polycommit/kzg10/data_structures.rs:68-72

Results are group elements; like the reference's callers (`batch_normalization_into_affine`) compare / use them after
affine normalisation (`kzg10.to_affine`)."""
import ctypes

import numpy as np

from . import _lib
from .kzg10 import to_affine
from .layout import G1_AFFINE, G1_PROJECTIVE


class FixedBase:
    @staticmethod
    def get_mul_window_size(num_scalars):
        """fixed_base.rs:26-31 (`ln_without_floats`, msm/mod.rs:29-32)."""
        if num_scalars < 32:
            return 3
        lg = 0
        while (1 << lg) < num_scalars:
            lg += 1
        return lg * 69 // 100 + 2

    @staticmethod
    def get_window_table(scalar_size, window, g):
        """fixed_base.rs:33-68.  The device builds its own 8-bit table of multiples of `g` inside `msm`; the returned
        object only carries the base (the window geometry does not change any result)."""
        return {"scalar_size": scalar_size, "window": window, "g": np.ascontiguousarray(g, dtype=G1_AFFINE).reshape(1)}

    @staticmethod
    def msm(scalar_size, window, table, v):
        """fixed_base.rs:87-97: [v_i * g] as G1_PROJECTIVE records; `v` are Fr elements ((n, 4) Montgomery limbs)."""
        v = np.ascontiguousarray(v, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(v.shape[0], dtype=G1_PROJECTIVE)
        _lib.check(_lib.lib().snarkvm_hip_g1_fixed_base_msm(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(table["g"].ctypes.data),
                                                           ctypes.c_void_p(v.ctypes.data), ctypes.c_size_t(v.shape[0])))
        return out


def group_ntt(points_projective, inverse=False):
    """Radix-2 (i)FFT of 2^k G1Projective records with Fr twiddles (`DomainCoeff` for group elements, fft/domain.rs)."""
    pts = np.array(points_projective, dtype=G1_PROJECTIVE, copy=True).reshape(-1)
    n = pts.shape[0]
    lg = n.bit_length() - 1
    if n == 0 or 1 << lg != n:
        raise ValueError("domain_size is not power of 2")
    _lib.check(_lib.lib().snarkvm_hip_g1_group_ntt(ctypes.c_void_p(pts.ctypes.data), ctypes.c_uint32(lg), ctypes.c_int(1 if inverse else 0)))
    return pts


def lagrange_basis(powers_of_beta_g):
    """data_structures.rs:68-72: `batch_normalization_into_affine(domain.ifft(powers as projective))`.
    `powers_of_beta_g`: G1_AFFINE records, a power-of-two count."""
    aff = np.ascontiguousarray(powers_of_beta_g, dtype=G1_AFFINE).reshape(-1)
    proj = np.zeros(aff.shape[0], dtype=G1_PROJECTIVE)
    proj["x"] = aff["x"]
    proj["y"] = aff["y"]
    one = np.array([202099033278250856, 5854854902718660529, 11492539364873682930, 8885205928937022213, 5545221690922665192, 39800542322357402],
                   dtype=np.uint64)  # Fq R (fq.rs:134-141): Z = 1
    proj["z"] = np.where(aff["infinity"][:, None] != 0, 0, one[None, :])
    inf = aff["infinity"] != 0
    proj["x"][inf] = 0
    proj["y"][inf] = one  # Projective::zero() = (0, 1, 0)
    return to_affine(group_ntt(proj, inverse=True))
