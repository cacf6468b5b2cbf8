"""`VariableBase::msm` for G1 (algorithms/src/msm/variable_base/mod.rs:28-49), bound to the gfx950 backend.

The reference dispatches G1 MSMs with more than 1024 scalars to the accelerator and silently falls back to
`batched::msm` otherwise / on error.  This package *is* the accelerator side: every size goes to the device
(the reference's own GPU test drives the FFI from 2^2, mod.rs:109-119) and errors are raised, never masked by
a CPU path.  `RegisteredBases` is the extension for SRS vectors that stay resident in HBM."""
import ctypes

import numpy as np

from . import _lib, plugin
from .layout import G1_AFFINE, G1_PROJECTIVE


class VariableBase:
    @staticmethod
    def msm(bases, scalars):
        """bases: G1_AFFINE array; scalars: (n,4) u64 canonical integers (`Fr::to_bigint`).  Returns a
        G1_PROJECTIVE record (Jacobian; compare after to_affine, like mod.rs:116-117)."""
        return plugin.msm(bases, scalars)


class RegisteredBases:
    """Device-resident base vector (snarkvm_hip_register_bases)."""

    def __init__(self, bases=None, device_ptr=None, npoints=None, tables=1, window_bits=0):
        """tables > 1 precomputes 2^(256/tables * j) * P_i (j < tables) once, shrinking the serial tail of every MSM.
        window_bits > 0 selects the general form: table j = 2^(window_bits * j) * P_i with tables * window_bits >= 254;
        window_bits up to 23 (e.g. tables=12, window_bits=22 for n ~ 2^24) trades bucket additions for bucket count."""
        L = _lib.lib()
        self._h = ctypes.c_void_p()
        self.tables = int(tables)
        self.window_bits = int(window_bits)
        if device_ptr is not None:
            self.n = int(npoints)
            src, on_device = ctypes.c_void_p(device_ptr), 1
        else:
            bases = np.ascontiguousarray(bases, dtype=G1_AFFINE).reshape(-1)
            self.n = bases.shape[0]
            src, on_device = ctypes.c_void_p(bases.ctypes.data), 0
        if self.window_bits:
            err = L.snarkvm_hip_register_bases_windowed(ctypes.byref(self._h), src, ctypes.c_size_t(self.n), ctypes.c_size_t(G1_AFFINE.itemsize),
                                                        ctypes.c_int(on_device), ctypes.c_int(self.tables), ctypes.c_int(self.window_bits))
        else:
            err = L.snarkvm_hip_register_bases_tables(ctypes.byref(self._h), src, ctypes.c_size_t(self.n), ctypes.c_size_t(G1_AFFINE.itemsize),
                                                      ctypes.c_int(on_device), ctypes.c_int(self.tables))
        _lib.check(err)

    @classmethod
    def from_serialized(cls, data, npoints, compressed=False, validate=False, tables=1):
        """Register bases straight from their canonical encoding (e.g. the body of a `.usrs` SRS file): the bytes are
        decoded on the device into the engine's base slots (snarkvm_hip_register_bases_serialized)."""
        from . import serialize

        self = cls.__new__(cls)
        self.tables = int(tables)
        self.window_bits = 0
        self.n = int(npoints)
        self._h = serialize.register_bases_serialized(data, npoints, compressed, validate, tables)
        return self

    def msm(self, scalars=None, offset=0, device_ptr=None, npoints=None, window_bits=0):
        """sum_i scalars[i] * bases[offset + i]; scalars either a host (n,4) u64 array or a device pointer."""
        L = _lib.lib()
        out = np.zeros(1, dtype=G1_PROJECTIVE)
        if device_ptr is not None:
            n = int(npoints)
            err = L.snarkvm_hip_msm_registered(ctypes.c_void_p(out.ctypes.data), self._h, ctypes.c_size_t(offset), ctypes.c_size_t(n),
                                               ctypes.c_void_p(device_ptr), ctypes.c_int(1), ctypes.c_int(window_bits))
        else:
            scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
            n = scalars.shape[0]
            err = L.snarkvm_hip_msm_registered(ctypes.c_void_p(out.ctypes.data), self._h, ctypes.c_size_t(offset), ctypes.c_size_t(n),
                                               ctypes.c_void_p(scalars.ctypes.data), ctypes.c_int(0), ctypes.c_int(window_bits))
        _lib.check(err)
        return out

    def msm_batch(self, scalars_list=None, offsets=None, device_ptrs=None, npoints=None, window_bits=0, montgomery=False):
        """Independent MSMs over this base vector, pipelined on several HIP streams.  Either `scalars_list` (host
        (n_k, 4) arrays) or `device_ptrs` + `npoints` (device-resident scalar vectors).  Returns a G1_PROJECTIVE array."""
        L = _lib.lib()
        if device_ptrs is not None:
            count = len(device_ptrs)
            ns = [int(v) for v in npoints]
            ptrs = [int(p) for p in device_ptrs]
            on_dev = 1
            keep = None
        else:
            keep = [np.ascontiguousarray(s, dtype=np.uint64).reshape(-1, 4) for s in scalars_list]
            count = len(keep)
            ns = [k.shape[0] for k in keep]
            ptrs = [k.ctypes.data for k in keep]
            on_dev = 0
        offs = [0] * count if offsets is None else [int(o) for o in offsets]
        out = np.zeros(count, dtype=G1_PROJECTIVE)
        c_offs = (ctypes.c_size_t * max(1, count))(*offs)
        c_ns = (ctypes.c_size_t * max(1, count))(*ns)
        c_ptrs = (ctypes.c_void_p * max(1, count))(*ptrs)
        _lib.check(L.snarkvm_hip_msm_registered_batch(ctypes.c_void_p(out.ctypes.data), self._h, ctypes.c_size_t(count), c_offs, c_ns, c_ptrs,
                                                      ctypes.c_int(on_dev), ctypes.c_int(1 if montgomery else 0), ctypes.c_int(window_bits)))
        return out

    def close(self):
        if self._h:
            _lib.lib().snarkvm_hip_free_bases(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def msm_g2(bases, scalars):
    """G2 variable-base MSM on the device (extension: the reference sends G2 to its CPU `standard::msm`,
    variable_base/mod.rs:45-47).  bases: G2_AFFINE array (200 B stride); returns a G2_PROJECTIVE record."""
    from .layout import G2_AFFINE, G2_PROJECTIVE

    bases = np.ascontiguousarray(bases, dtype=G2_AFFINE).reshape(-1)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = scalars.shape[0]
    if n > bases.shape[0]:
        raise ValueError(f"length mismatch {bases.shape[0]} points < {n} scalars")
    out = np.zeros(1, dtype=G2_PROJECTIVE)
    err = _lib.lib().snarkvm_hip_msm_g2(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(bases.ctypes.data), ctypes.c_size_t(n),
                                        ctypes.c_void_p(scalars.ctypes.data), ctypes.c_size_t(G2_AFFINE.itemsize))
    _lib.check(err)
    return out


class RegisteredBasesG2:
    """Device-resident G2 base vector with precomputed tables (snarkvm_hip_register_bases_g2)."""

    def __init__(self, bases, tables=16, window_bits=0):
        from .layout import G2_AFFINE

        bases = np.ascontiguousarray(bases, dtype=G2_AFFINE).reshape(-1)
        self.n = bases.shape[0]
        self._h = ctypes.c_void_p()
        _lib.check(_lib.lib().snarkvm_hip_register_bases_g2(ctypes.byref(self._h), ctypes.c_void_p(bases.ctypes.data), ctypes.c_size_t(self.n),
                                                           ctypes.c_size_t(G2_AFFINE.itemsize), ctypes.c_int(int(tables)), ctypes.c_int(int(window_bits))))

    def msm(self, scalars=None, offset=0, window_bits=0, device_ptr=None, npoints=None):
        """sum_i scalars[i] * bases[offset + i]; scalars either a host (n,4) u64 array or a device pointer + npoints."""
        from .layout import G2_PROJECTIVE

        out = np.zeros(1, dtype=G2_PROJECTIVE)
        if device_ptr is not None:
            _lib.check(_lib.lib().snarkvm_hip_msm_g2_registered(ctypes.c_void_p(out.ctypes.data), self._h, ctypes.c_size_t(offset),
                                                               ctypes.c_size_t(int(npoints)), ctypes.c_void_p(device_ptr), ctypes.c_int(1),
                                                               ctypes.c_int(window_bits)))
            return out
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        _lib.check(_lib.lib().snarkvm_hip_msm_g2_registered(ctypes.c_void_p(out.ctypes.data), self._h, ctypes.c_size_t(offset),
                                                           ctypes.c_size_t(scalars.shape[0]), ctypes.c_void_p(scalars.ctypes.data), ctypes.c_int(0),
                                                           ctypes.c_int(window_bits)))
        return out

    def msm_batch(self, scalars_list=None, offsets=None, device_ptrs=None, npoints=None, window_bits=0):
        """Independent G2 MSMs over this base vector fanned out over devices and lanes (snarkvm_hip_msm_g2_registered_batch)."""
        from .layout import G2_PROJECTIVE

        if device_ptrs is not None:
            count, ns, ptrs, on_dev, keep = len(device_ptrs), [int(v) for v in npoints], [int(p) for p in device_ptrs], 1, None
        else:
            keep = [np.ascontiguousarray(s, dtype=np.uint64).reshape(-1, 4) for s in scalars_list]
            count, ns, ptrs, on_dev = len(keep), [k.shape[0] for k in keep], [k.ctypes.data for k in keep], 0
        offs = [0] * count if offsets is None else [int(o) for o in offsets]
        out = np.zeros(count, dtype=G2_PROJECTIVE)
        c_offs = (ctypes.c_size_t * max(1, count))(*offs)
        c_ns = (ctypes.c_size_t * max(1, count))(*ns)
        c_ptrs = (ctypes.c_void_p * max(1, count))(*ptrs)
        _lib.check(_lib.lib().snarkvm_hip_msm_g2_registered_batch(ctypes.c_void_p(out.ctypes.data), self._h, ctypes.c_size_t(count), c_offs, c_ns, c_ptrs,
                                                                 ctypes.c_int(on_dev), ctypes.c_int(window_bits)))
        return out

    def close(self):
        if self._h:
            _lib.lib().snarkvm_hip_free_bases_g2(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_devices(ids):
    """Devices this process uses (snarkvm_hip_set_devices): before the first compute call.  A repeated id makes an independent
    logical device on the same GPU."""
    arr = (ctypes.c_int32 * len(ids))(*[int(i) for i in ids])
    _lib.check(_lib.lib().snarkvm_hip_set_devices(arr, ctypes.c_size_t(len(ids))))


def num_devices():
    return _lib.lib().snarkvm_hip_num_devices()


def g1_sum(points):
    """Sum of G1Projective records on the device (snarkvm_hip_g1_sum): the combine step of a point-range-split MSM."""
    pts = np.ascontiguousarray(points)
    raw = np.frombuffer(pts.tobytes(), dtype=np.uint8)
    if raw.size % G1_PROJECTIVE.itemsize:
        raise ValueError("g1_sum: input is not a whole number of G1Projective records")
    n = raw.size // G1_PROJECTIVE.itemsize
    out = np.zeros(1, dtype=G1_PROJECTIVE)
    _lib.check(_lib.lib().snarkvm_hip_g1_sum(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(raw.ctypes.data), ctypes.c_size_t(n)))
    return out
