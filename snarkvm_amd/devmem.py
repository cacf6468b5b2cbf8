"""Device memory owned through the C ABI (snarkvm_hip_malloc / _free / _memcpy_h2d / _d2h / _d2d / _memset, include/snarkvm_hip.h): what a host
without a HIP binding of its own - the Rust prover of north_star - uses to keep its vectors in HBM.  The ctypes twin of
rust/snarkvm-algorithms-hip/src/lib.rs::resident::DeviceBuffer; no torch anywhere in this module.

The reference's plugin owns every device byte itself (algorithms/cuda/cuda/snarkvm.cu:51,123-151: per-GPU arenas behind snarkvm_t); a caller of
the device-resident extension ABI needs the same ownership on its side of the boundary, and these six calls are it.
"""
import ctypes

import numpy as np

from . import _lib


class HipMem:
    """One block of device memory.  `device`: index into the devices in use, -1 = the device of the calling thread's open scope (logical
    device 0 outside a scope).  Freed by free() or when the object dies (snarkvm_hip_free waits for the device's queued work)."""

    def __init__(self, nbytes, device=-1):
        self.nbytes = int(nbytes)
        p = ctypes.c_void_p()
        _lib.check(_lib.lib().snarkvm_hip_malloc(ctypes.byref(p), self.nbytes, device))
        self.ptr = p.value or 0

    @classmethod
    def from_numpy(cls, arr, device=-1):
        arr = np.ascontiguousarray(arr)
        m = cls(arr.nbytes, device)
        m.upload(arr)
        return m

    def data_ptr(self):  # the name torch tensors use: either kind can be handed to the replay code
        return self.ptr

    def at(self, byte_offset):
        assert 0 <= byte_offset <= self.nbytes
        return self.ptr + byte_offset

    def upload(self, arr, byte_offset=0):
        """host -> device; complete on return"""
        arr = np.ascontiguousarray(arr)
        assert byte_offset + arr.nbytes <= self.nbytes
        _lib.check(_lib.lib().snarkvm_hip_memcpy_h2d(self.at(byte_offset), arr.ctypes.data, arr.nbytes))

    def download(self, nbytes=None, byte_offset=0, dtype=np.uint8):
        """device -> host as a new numpy array of `dtype`; complete on return"""
        nbytes = self.nbytes - byte_offset if nbytes is None else nbytes
        assert byte_offset + nbytes <= self.nbytes
        out = np.empty(nbytes, dtype=np.uint8)
        if nbytes:
            _lib.check(_lib.lib().snarkvm_hip_memcpy_d2h(out.ctypes.data, self.at(byte_offset), nbytes))
        return out.view(dtype)

    def copy_from(self, byte_offset, src_ptr, nbytes):
        """device -> device from the raw pointer `src_ptr` (ranges must not overlap); inside a scope only enqueued"""
        assert byte_offset + nbytes <= self.nbytes
        _lib.check(_lib.lib().snarkvm_hip_memcpy_d2d(self.at(byte_offset), src_ptr, nbytes))

    def fill(self, byte_offset, value, nbytes):
        """memset; inside a scope only enqueued"""
        assert byte_offset + nbytes <= self.nbytes
        _lib.check(_lib.lib().snarkvm_hip_memset(self.at(byte_offset), value, nbytes))

    def free(self):
        if self.ptr:
            err = _lib.lib().snarkvm_hip_free(self.ptr)
            self.ptr = 0
            _lib.check(err)

    def __del__(self):
        try:
            self.free()
        except Exception:  # interpreter shutdown / a dead device: nothing left to do
            pass
