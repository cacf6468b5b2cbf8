"""ctypes loader of snarkvm_amd/lib/libsnarkvm_hip.so (the C ABI of include/snarkvm_hip.h).

There is deliberately no CPU fallback: if the HIP library is missing or a call fails, an exception is
raised (the reference's *caller* owns the CPU fallback, msm/variable_base/mod.rs:39-43).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SNARKVM_HIP_LIB") or os.path.join(_HERE, "lib", "libsnarkvm_hip.so")  # override: A/B experiments only

# every symbol include/snarkvm_hip.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "snarkvm_ntt", "snarkvm_polymul", "snarkvm_msm", "snarkvm_hip_set_base_cache",
    "snarkvm_hip_device_count", "snarkvm_hip_batch_lanes", "snarkvm_hip_set_device", "snarkvm_hip_set_devices", "snarkvm_hip_num_devices",
    "snarkvm_hip_malloc", "snarkvm_hip_free", "snarkvm_hip_memcpy_h2d", "snarkvm_hip_memcpy_d2h", "snarkvm_hip_memcpy_d2d", "snarkvm_hip_memset", "snarkvm_hip_ntt_device", "snarkvm_hip_ntt_device_batch",
    "snarkvm_hip_scope_begin", "snarkvm_hip_scope_end", "snarkvm_hip_scope_begin_ex", "snarkvm_hip_scope_collect", "snarkvm_hip_scope_set_flags", "snarkvm_hip_scope_stream", "snarkvm_hip_alloc_stats",
    "snarkvm_hip_register_bases", "snarkvm_hip_register_bases_tables", "snarkvm_hip_register_bases_windowed", "snarkvm_hip_free_bases", "snarkvm_hip_msm_registered", "snarkvm_hip_msm_g2",
    "snarkvm_hip_msm_registered_ex", "snarkvm_hip_msm_registered_batch", "snarkvm_hip_msm_registered_batch_ex", "snarkvm_hip_g1_to_affine",
    "snarkvm_hip_fr_mul_device", "snarkvm_hip_fr_convert_device", "snarkvm_hip_g1_generate_bases_device",
    "snarkvm_hip_fr_vec_op", "snarkvm_hip_fr_divide_by_linear", "snarkvm_hip_fr_batch_inversion_and_mul",
    "snarkvm_hip_fr_distribute_powers", "snarkvm_hip_fr_lagrange_coefficients", "snarkvm_hip_fr_divide_by_vanishing",
    "snarkvm_hip_fr_mul_by_vanishing",
    "snarkvm_hip_fr_vec_op_strided", "snarkvm_hip_fr_divide_by_linear_strided", "snarkvm_hip_fr_divide_by_vanishing_strided",
    "snarkvm_hip_register_bases_serialized", "snarkvm_hip_g1_deserialize", "snarkvm_hip_g1_serialize", "snarkvm_hip_g1_sum", "snarkvm_hip_g2_deserialize", "snarkvm_hip_g2_serialize", "snarkvm_hip_g2_deserialize_compressed", "snarkvm_hip_g2_serialize_compressed",
    "snarkvm_hip_register_bases_g2", "snarkvm_hip_free_bases_g2", "snarkvm_hip_msm_g2_registered", "snarkvm_hip_msm_g2_registered_batch",
    "snarkvm_hip_g1_fixed_base_msm", "snarkvm_hip_g1_group_ntt",
    "snarkvm_hip_set_profiling", "snarkvm_hip_get_phase_count", "snarkvm_hip_get_phase_name",
    "snarkvm_hip_get_phase_ms", "snarkvm_hip_synchronize", "snarkvm_hip_coalescer_stats",
    "snarkvm_hip_selftest_field", "snarkvm_hip_selftest_g1_msm_naive", "snarkvm_hip_selftest_msm_plan", "snarkvm_hip_selftest_g1_finish", "snarkvm_hip_selftest_fq_lazy", "snarkvm_hip_selftest_g1_lazy_tail", "snarkvm_hip_selftest_fr_signed", "snarkvm_hip_selftest_fq2_lazy", "snarkvm_hip_selftest_fq2_pair", "snarkvm_hip_selftest_g2_hex", "snarkvm_hip_devtest_field", "snarkvm_hip_devtest_g2_tail_repeat",
]


class RustError(ctypes.Structure):
    """sppark `cuda::Error` (algorithms/cuda/src/lib.rs:20): code 0 == success."""
    _fields_ = [("code", ctypes.c_int32), ("message", ctypes.c_void_p)]


class HipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"snarkvm_hip error {code}: {message}")
        self.code = code
        self.message = message


_lib = None
_libc = None


def lib():
    global _lib, _libc
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing - build it with `python -m snarkvm_amd.build` (hipcc, gfx950). "
                "snarkvm_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64.
        # If torch is installed, load it first so that this library binds to the runtime that is already in the
        # process (same SONAME); two runtimes in one process leave the second one without visible GPUs.
        # SNARKVM_HIP_NO_TORCH=1: a torch-free host (tests/test_gpu_devmem.py proves a whole proof's call list that way): the library then binds to the
        # HIP runtime of /opt/rocm like any C++ / Rust caller's process does.
        if not os.environ.get("SNARKVM_HIP_NO_TORCH"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = ctypes.CDLL(LIB_PATH)
        err_fns = ["snarkvm_ntt", "snarkvm_polymul", "snarkvm_msm", "snarkvm_hip_set_base_cache", "snarkvm_hip_set_device", "snarkvm_hip_set_devices",
                   "snarkvm_hip_malloc", "snarkvm_hip_free", "snarkvm_hip_memcpy_h2d", "snarkvm_hip_memcpy_d2h", "snarkvm_hip_memcpy_d2d", "snarkvm_hip_memset", "snarkvm_hip_ntt_device", "snarkvm_hip_ntt_device_batch",
                   "snarkvm_hip_scope_begin", "snarkvm_hip_scope_end", "snarkvm_hip_scope_begin_ex", "snarkvm_hip_scope_collect", "snarkvm_hip_scope_set_flags",
                   "snarkvm_hip_register_bases", "snarkvm_hip_register_bases_tables", "snarkvm_hip_register_bases_windowed", "snarkvm_hip_msm_registered", "snarkvm_hip_msm_g2", "snarkvm_hip_msm_registered_ex", "snarkvm_hip_msm_registered_batch", "snarkvm_hip_msm_registered_batch_ex", "snarkvm_hip_g1_to_affine", "snarkvm_hip_fr_mul_device",
                   "snarkvm_hip_fr_convert_device", "snarkvm_hip_g1_generate_bases_device", "snarkvm_hip_synchronize",
                   "snarkvm_hip_fr_vec_op", "snarkvm_hip_fr_divide_by_linear", "snarkvm_hip_fr_batch_inversion_and_mul", "snarkvm_hip_fr_distribute_powers", "snarkvm_hip_fr_lagrange_coefficients", "snarkvm_hip_fr_divide_by_vanishing", "snarkvm_hip_fr_mul_by_vanishing",
                   "snarkvm_hip_fr_vec_op_strided", "snarkvm_hip_fr_divide_by_linear_strided", "snarkvm_hip_fr_divide_by_vanishing_strided",
                   "snarkvm_hip_register_bases_serialized", "snarkvm_hip_g1_deserialize", "snarkvm_hip_g1_serialize", "snarkvm_hip_g1_sum", "snarkvm_hip_g2_deserialize", "snarkvm_hip_g2_serialize", "snarkvm_hip_g2_deserialize_compressed", "snarkvm_hip_g2_serialize_compressed",
                   "snarkvm_hip_register_bases_g2", "snarkvm_hip_msm_g2_registered", "snarkvm_hip_msm_g2_registered_batch",
                   "snarkvm_hip_g1_fixed_base_msm", "snarkvm_hip_g1_group_ntt",
                   "snarkvm_hip_devtest_field", "snarkvm_hip_devtest_g2_tail_repeat"]
        for name in err_fns:
            getattr(L, name).restype = RustError
        L.snarkvm_hip_device_count.restype = ctypes.c_int
        L.snarkvm_hip_batch_lanes.restype = ctypes.c_int
        L.snarkvm_hip_num_devices.restype = ctypes.c_int
        L.snarkvm_hip_selftest_g1_finish.restype = ctypes.c_int
        L.snarkvm_hip_selftest_fq_lazy.restype = ctypes.c_int
        L.snarkvm_hip_selftest_g1_lazy_tail.restype = ctypes.c_int
        L.snarkvm_hip_selftest_fr_signed.restype = ctypes.c_int
        L.snarkvm_hip_selftest_fq2_lazy.restype = ctypes.c_int
        L.snarkvm_hip_selftest_fq2_pair.restype = ctypes.c_int
        L.snarkvm_hip_selftest_g2_hex.restype = ctypes.c_int
        L.snarkvm_hip_get_phase_count.restype = ctypes.c_int
        L.snarkvm_hip_get_phase_name.restype = ctypes.c_char_p
        L.snarkvm_hip_get_phase_ms.restype = ctypes.c_double
        L.snarkvm_hip_selftest_field.restype = ctypes.c_int
        L.snarkvm_hip_selftest_g1_msm_naive.restype = ctypes.c_int
        L.snarkvm_hip_selftest_msm_plan.restype = ctypes.c_int
        L.snarkvm_hip_free_bases.restype = None
        L.snarkvm_hip_free_bases_g2.restype = None
        L.snarkvm_hip_set_profiling.restype = None
        L.snarkvm_hip_coalescer_stats.restype = None
        L.snarkvm_hip_alloc_stats.restype = None
        L.snarkvm_hip_scope_stream.restype = ctypes.c_void_p
        L.snarkvm_hip_scope_begin_ex.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.snarkvm_hip_scope_set_flags.argtypes = [ctypes.c_uint32]
        L.snarkvm_hip_alloc_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]  # (without argtypes a bare Python int address would travel as a 32-bit C int)
        L.snarkvm_hip_coalescer_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.snarkvm_hip_malloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_int]
        L.snarkvm_hip_free.argtypes = [ctypes.c_void_p]
        L.snarkvm_hip_memcpy_h2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.snarkvm_hip_memcpy_d2h.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.snarkvm_hip_memcpy_d2d.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
        L.snarkvm_hip_memset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _libc = ctypes.CDLL(None)
        _libc.free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def check(err):
    """Raise HipError for a non-zero RustError (and free its message, as Rust's Drop would)."""
    if err.code != 0:
        msg = ctypes.string_at(err.message).decode(errors="replace") if err.message else ""
        if err.message:
            _libc.free(err.message)
        raise HipError(err.code, msg)


def device_count():
    return lib().snarkvm_hip_device_count()
