"""ctypes binding of oracle/liboracle.so (CPU restatement of the reference's MSM/NTT path).

TEST INFRASTRUCTURE ONLY - see the header of cpu_oracle.cpp.  Importable from tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never from snarkvm_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

# Rust in-memory layouts
FR = np.dtype(("<u8", 4))
G1_AFFINE = np.dtype([("x", "<u8", 6), ("y", "<u8", 6), ("infinity", "u1"), ("pad", "u1", 7)])
G1_PROJECTIVE = np.dtype([("x", "<u8", 6), ("y", "<u8", 6), ("z", "<u8", 6)])
G2_AFFINE = np.dtype([("x", "<u8", 12), ("y", "<u8", 12), ("infinity", "u1"), ("pad", "u1", 7)])
G2_PROJECTIVE = np.dtype([("x", "<u8", 12), ("y", "<u8", 12), ("z", "<u8", 12)])
assert G1_AFFINE.itemsize == 104 and G1_PROJECTIVE.itemsize == 144
assert G2_AFFINE.itemsize == 200 and G2_PROJECTIVE.itemsize == 288

ORDER_NN, ORDER_NR, ORDER_RN, ORDER_RR = 0, 1, 2, 3
FORWARD, INVERSE = 0, 1
STANDARD, COSET = 0, 1

MSM_BATCHED, MSM_STANDARD, MSM_NAIVE = 0, 1, 2


def build(force=False):
    """Compile liboracle.so with the Makefile next to this file (g++ only)."""
    src = os.path.join(_HERE, "cpu_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
BUILD_FLAGS = "-O3 -march=x86-64-v3 (portable: compiled in the build container, shipped prebuilt)"


def use_native():
    """bench.py's cpu_baseline leg times the restatement on the GPU box's host cores: compile it THERE with -march=native (BASELINE.md 3 promises that for the
    CPU column; the shipped liboracle.so is -march=x86-64-v3 because it is built in another container) into a temporary directory - never in-tree, a native
    build must not travel to another machine - and make it the library this process uses.  Must be called before the first oracle call; falls back to the
    shipped build (and says so in BUILD_FLAGS) when no compiler is at hand.  Returns the flags in effect."""
    global _lib, _LIB_PATH, BUILD_FLAGS
    if _lib is not None:
        return BUILD_FLAGS
    import tempfile

    out_dir = os.path.join(tempfile.gettempdir(), f"oracle_native_{os.getuid()}")
    out = os.path.join(out_dir, "liboracle.so")
    src = os.path.join(_HERE, "cpu_oracle.cpp")
    try:
        os.makedirs(out_dir, mode=0o700, exist_ok=True)
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp", "-Wno-unused-function", "-shared", "-o", out + ".tmp", src],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
            os.replace(out + ".tmp", out)
        _LIB_PATH = out
        BUILD_FLAGS = "-O3 -march=native (compiled on this host by oracle.use_native())"
    except Exception as e:  # noqa: BLE001
        BUILD_FLAGS += f"; native rebuild failed: {type(e).__name__}"
    return BUILD_FLAGS


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_max_threads.restype = ctypes.c_int
        _lib.oracle_domain.restype = ctypes.c_int
        _lib.oracle_ntt.restype = ctypes.c_int
        _lib.oracle_polymul.restype = ctypes.c_int
        _lib.oracle_g1_is_on_curve.restype = ctypes.c_int
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def set_threads(n):
    lib().oracle_set_threads(ctypes.c_int(int(n)))


def max_threads():
    return lib().oracle_max_threads()


_OPS = {"add": 0, "sub": 1, "mul": 2, "inverse": 3, "from_bigint": 4, "to_bigint": 5, "neg": 6, "sqr": 7}


def _field_op(fn, limbs, op, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, limbs)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, limbs)
    out = np.empty_like(a)
    fn(ctypes.c_int(_OPS[op]), _p(a), _p(b), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def fr_op(op, a, b=None):
    return _field_op(lib().oracle_fr_op, 4, op, a, b)


def fq_op(op, a, b=None):
    return _field_op(lib().oracle_fq_op, 6, op, a, b)


def domain(lg):
    """(group_gen, group_gen_inv, size_inv, generator_inv, size_as_field_element), Montgomery limbs."""
    out = np.empty((5, 4), dtype=np.uint64)
    rc = lib().oracle_domain(ctypes.c_uint32(lg), _p(out))
    if rc:
        raise ValueError("domain too large")
    return out


def ntt(x, order=ORDER_NN, direction=FORWARD, kind=STANDARD):
    """Out-of-place wrapper of oracle_ntt (same argument meaning as snarkvm_ntt)."""
    x = np.array(x, dtype=np.uint64, copy=True).reshape(-1, 4)
    n = x.shape[0]
    lg = n.bit_length() - 1
    assert 1 << lg == n
    rc = lib().oracle_ntt(_p(x), ctypes.c_uint32(lg), ctypes.c_int(order), ctypes.c_int(direction), ctypes.c_int(kind))
    if rc:
        raise RuntimeError(f"oracle_ntt rc={rc}")
    return x


def polymul(lg, polys, evals=()):
    n = 1 << lg
    out = np.zeros((n, 4), dtype=np.uint64)
    polys = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 4) for p in polys]
    evals = [np.ascontiguousarray(e, dtype=np.uint64).reshape(-1, 4) for e in evals]
    pp = (ctypes.c_void_p * max(1, len(polys)))(*[p.ctypes.data for p in polys])
    pl = (ctypes.c_size_t * max(1, len(polys)))(*[p.shape[0] for p in polys])
    ep = (ctypes.c_void_p * max(1, len(evals)))(*[e.ctypes.data for e in evals])
    el = (ctypes.c_size_t * max(1, len(evals)))(*[e.shape[0] for e in evals])
    rc = lib().oracle_polymul(_p(out), ctypes.c_size_t(len(polys)), pp, pl, ctypes.c_size_t(len(evals)), ep, el,
                              ctypes.c_uint32(lg))
    if rc:
        raise RuntimeError(f"oracle_polymul rc={rc}")
    return out


def _msm(fn, aff_dt, proj_dt, kind, bases, scalars):
    bases = np.ascontiguousarray(bases, dtype=aff_dt)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(1, dtype=proj_dt)
    fn(ctypes.c_int(kind), _p(bases), ctypes.c_size_t(bases.shape[0]), _p(scalars),
       ctypes.c_size_t(scalars.shape[0]), _p(out))
    return out


def g1_msm(bases, scalars, kind=MSM_BATCHED):
    return _msm(lib().oracle_g1_msm, G1_AFFINE, G1_PROJECTIVE, kind, bases, scalars)


def g2_msm(bases, scalars, kind=MSM_STANDARD):
    return _msm(lib().oracle_g2_msm, G2_AFFINE, G2_PROJECTIVE, kind, bases, scalars)


def g1_to_affine(proj):
    proj = np.ascontiguousarray(proj, dtype=G1_PROJECTIVE).reshape(-1)
    out = np.zeros(proj.shape[0], dtype=G1_AFFINE)
    lib().oracle_g1_to_affine(_p(proj), _p(out), ctypes.c_size_t(proj.shape[0]))
    return out


def g2_to_affine(proj):
    proj = np.ascontiguousarray(proj, dtype=G2_PROJECTIVE).reshape(-1)
    out = np.zeros(proj.shape[0], dtype=G2_AFFINE)
    lib().oracle_g2_to_affine(_p(proj), _p(out), ctypes.c_size_t(proj.shape[0]))
    return out


def g1_is_on_curve(aff):
    aff = np.ascontiguousarray(aff, dtype=G1_AFFINE).reshape(-1)
    return all(lib().oracle_g1_is_on_curve(ctypes.c_void_p(aff[i : i + 1].ctypes.data)) for i in range(aff.shape[0]))


def g1_mul(base, scalar_limbs):
    base = np.ascontiguousarray(base, dtype=G1_AFFINE).reshape(1)
    s = np.ascontiguousarray(scalar_limbs, dtype=np.uint64).reshape(4)
    out = np.zeros(1, dtype=G1_PROJECTIVE)
    lib().oracle_g1_mul(_p(base), _p(s), _p(out))
    return out


def g2_mul(base, scalar_limbs):
    base = np.ascontiguousarray(base, dtype=G2_AFFINE).reshape(1)
    s = np.ascontiguousarray(scalar_limbs, dtype=np.uint64).reshape(4)
    out = np.zeros(1, dtype=G2_PROJECTIVE)
    lib().oracle_g2_mul(_p(base), _p(s), _p(out))
    return out


def g1_add(p1, p2):
    p1 = np.ascontiguousarray(p1, dtype=G1_PROJECTIVE).reshape(1)
    p2 = np.ascontiguousarray(p2, dtype=G1_PROJECTIVE).reshape(1)
    out = np.zeros(1, dtype=G1_PROJECTIVE)
    lib().oracle_g1_add(_p(p1), _p(p2), _p(out))
    return out


def g1_gen_bases(gen_affine, start, n):
    """bases[i] = (start + i) * G in the Rust affine layout."""
    gen_affine = np.ascontiguousarray(gen_affine, dtype=G1_AFFINE).reshape(1)
    out = np.zeros(n, dtype=G1_AFFINE)
    lib().oracle_g1_gen_bases(_p(gen_affine), ctypes.c_uint64(start), ctypes.c_size_t(n), _p(out))
    return out


# ---- prover-round polynomial helpers (SURVEY.md §8 N2, a15 `open`) ----
VEC_OPS = {"add": 0, "sub": 1, "mul": 2, "mul_sub": 3, "scale": 4, "sub_scalar": 5, "axpy": 6}


def _fr(x):
    return np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)


def fr_vec_op(op, a, b, c=None):
    a, b = _fr(a), _fr(b)
    c = b if c is None else _fr(c)
    out = np.empty_like(a)
    lib().oracle_fr_vec_op(ctypes.c_int(VEC_OPS[op]), _p(a), _p(b), _p(c), _p(out), ctypes.c_size_t(a.shape[0]))
    return out


def poly_divide(poly, divisor_terms):
    """Polynomial::divide_with_q_and_r; divisor_terms = [(degree, coeff limbs)], sorted.  Returns (quotient, remainder), trimmed."""
    poly = _fr(poly)
    n = poly.shape[0]
    deg = (ctypes.c_size_t * len(divisor_terms))(*[d for d, _ in divisor_terms])
    cf = _fr(np.stack([np.asarray(c, dtype=np.uint64).reshape(4) for _, c in divisor_terms]))
    q = np.zeros((max(n, 1), 4), dtype=np.uint64)
    r = np.zeros((max(n, 1), 4), dtype=np.uint64)
    ql, rl = ctypes.c_size_t(), ctypes.c_size_t()
    rc = lib().oracle_fr_poly_divide(_p(poly), ctypes.c_size_t(n), deg, _p(cf), ctypes.c_size_t(len(divisor_terms)), _p(q),
                                     ctypes.byref(ql), _p(r), ctypes.byref(rl))
    if rc:
        raise ZeroDivisionError("Dividing by zero polynomial is undefined")
    return q[: ql.value].copy(), r[: rl.value].copy()


def poly_evaluate(poly, point):
    poly = _fr(poly)
    point = _fr(point)
    out = np.zeros((1, 4), dtype=np.uint64)
    lib().oracle_fr_evaluate(_p(poly), ctypes.c_size_t(poly.shape[0]), _p(point), _p(out))
    return out


def batch_inversion_and_mul(v, coeff):
    v = np.array(v, dtype=np.uint64, copy=True).reshape(-1, 4)
    lib().oracle_fr_batch_inversion_and_mul(_p(v), ctypes.c_size_t(v.shape[0]), _p(_fr(coeff)))
    return v


def distribute_powers(v, g, c):
    v = np.array(v, dtype=np.uint64, copy=True).reshape(-1, 4)
    lib().oracle_fr_distribute_powers(_p(v), ctypes.c_size_t(v.shape[0]), _p(_fr(g)), _p(_fr(c)))
    return v


def lagrange_coefficients(lg, tau):
    out = np.zeros((1 << lg, 4), dtype=np.uint64)
    rc = lib().oracle_fr_lagrange_coefficients(ctypes.c_uint32(lg), _p(_fr(tau)), _p(out))
    if rc:
        raise ValueError("domain too large")
    return out


def mul_by_vanishing(poly, domain_size):
    poly = _fr(poly)
    out = np.zeros((poly.shape[0] + domain_size, 4), dtype=np.uint64)
    lib().oracle_fr_mul_by_vanishing(_p(poly), ctypes.c_size_t(poly.shape[0]), ctypes.c_size_t(domain_size), _p(out))
    return out


# ---- setup-time group operations ----
def g1_fixed_base_msm(g_affine, scalars_mont, scalar_size=253, window=None):
    """FixedBase::msm with the reference's own window rule when `window` is None."""
    g = np.ascontiguousarray(g_affine, dtype=G1_AFFINE).reshape(1)
    v = _fr(scalars_mont)
    if window is None:
        n = v.shape[0]
        lg = 0
        while (1 << lg) < n:
            lg += 1
        window = 3 if n < 32 else lg * 69 // 100 + 2
    out = np.zeros(v.shape[0], dtype=G1_PROJECTIVE)
    lib().oracle_g1_fixed_base_msm(_p(g), ctypes.c_size_t(scalar_size), ctypes.c_size_t(window), _p(v), ctypes.c_size_t(v.shape[0]), _p(out))
    return out


def g1_group_ntt(points_proj, inverse=False):
    pts = np.array(points_proj, dtype=G1_PROJECTIVE, copy=True).reshape(-1)
    lg = pts.shape[0].bit_length() - 1
    assert 1 << lg == pts.shape[0]
    lib().oracle_g1_group_ntt(_p(pts), ctypes.c_uint32(lg), ctypes.c_int(1 if inverse else 0))
    return pts
