"""Shared helpers for the test-suite (conversions between Python ints and limb arrays, fixtures)."""
import os

import numpy as np

from oracle import pyref
from oracle import cpu as oracle
from snarkvm_amd.layout import G1_AFFINE, G1_PROJECTIVE, G2_AFFINE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def limbs(v, n):
    return np.array(pyref.to_limbs(v, n), dtype=np.uint64)


def ints_to_fr(vals):
    """canonical ints -> (n,4) canonical limb array"""
    return np.array([pyref.to_limbs(v % pyref.R_MOD, 4) for v in vals], dtype=np.uint64).reshape(-1, 4)


def ints_to_fr_mont(vals):
    return np.array([pyref.to_limbs(pyref.fr_to_mont(v % pyref.R_MOD), 4) for v in vals], dtype=np.uint64).reshape(-1, 4)


def fr_mont_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [pyref.fr_from_mont(pyref.from_limbs(row)) for row in arr]


def fr_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [pyref.from_limbs(row) for row in arr]


def g1_affine_from_ints(points):
    """[(x, y) | None] canonical ints -> Rust-layout affine array (Montgomery coordinates)."""
    out = np.zeros(len(points), dtype=G1_AFFINE)
    for i, p in enumerate(points):
        if p is None:
            out[i]["x"] = 0
            out[i]["y"] = pyref.to_limbs(pyref.fq_to_mont(1), 6)
            out[i]["infinity"] = 1
        else:
            out[i]["x"] = pyref.to_limbs(pyref.fq_to_mont(p[0]), 6)
            out[i]["y"] = pyref.to_limbs(pyref.fq_to_mont(p[1]), 6)
    return out


def g1_affine_to_ints(arr):
    arr = np.asarray(arr, dtype=G1_AFFINE).reshape(-1)
    out = []
    for a in arr:
        if a["infinity"]:
            out.append(None)
        else:
            out.append((pyref.fq_from_mont(pyref.from_limbs(a["x"])), pyref.fq_from_mont(pyref.from_limbs(a["y"]))))
    return out


def g2_affine_from_ints(points):
    out = np.zeros(len(points), dtype=G2_AFFINE)
    for i, p in enumerate(points):
        if p is None:
            out[i]["y"][:6] = pyref.to_limbs(pyref.fq_to_mont(1), 6)
            out[i]["infinity"] = 1
        else:
            (x0, x1), (y0, y1) = p
            out[i]["x"] = pyref.to_limbs(pyref.fq_to_mont(x0), 6) + pyref.to_limbs(pyref.fq_to_mont(x1), 6)
            out[i]["y"] = pyref.to_limbs(pyref.fq_to_mont(y0), 6) + pyref.to_limbs(pyref.fq_to_mont(y1), 6)
    return out


def g2_affine_to_ints(arr):
    out = []
    for a in np.asarray(arr, dtype=G2_AFFINE).reshape(-1):
        if a["infinity"]:
            out.append(None)
        else:
            f = lambda l: pyref.fq_from_mont(pyref.from_limbs(l))
            out.append(((f(a["x"][:6]), f(a["x"][6:])), (f(a["y"][:6]), f(a["y"][6:]))))
    return out


def srs_points_ints(raw, n=None):
    """Decode the uncompressed SRS fixture: 96 B per point, canonical LE x | y (macros.rs:86-95)."""
    cnt = len(raw) // 96
    n = cnt if n is None else min(n, cnt)
    pts = []
    for i in range(n):
        x = int.from_bytes(raw[96 * i : 96 * i + 48], "little")
        y = int.from_bytes(raw[96 * i + 48 : 96 * i + 96], "little")
        pts.append((x, y))
    return pts


def srs_config1_bases(raw):
    """every point of powers-of-beta-15.usrs followed by its negation (x, q - y) - 2 * 32768 = 2^16 bases,
    Rust-layout affine records (benches/msm/variable_base.rs:29-32 tiles real SRS points the same way)."""
    pts = srs_points_ints(raw)
    return g1_affine_from_ints(pts + [(x, (pyref.Q_MOD - y) % pyref.Q_MOD) for x, y in pts])


def affine_equal(a, b):
    """Rust `Affine == Affine` (derive(PartialEq): x, y, infinity all equal)."""
    a = np.asarray(a, dtype=G1_AFFINE).reshape(-1)
    b = np.asarray(b, dtype=G1_AFFINE).reshape(-1)
    return (
        a.shape == b.shape
        and np.array_equal(a["x"], b["x"])
        and np.array_equal(a["y"], b["y"])
        and np.array_equal(a["infinity"], b["infinity"])
    )


def g1_generator_affine():
    return g1_affine_from_ints([pyref.G1_GEN])


def weighted_sum_mod_r(scalars, start=1):
    """sum_i (start + i) * scalars[i] mod r for an (n,4) u64 array, vectorised (16-bit pieces, chunked)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    n = s.shape[0]
    total = 0
    CH = 1 << 18
    for lo in range(0, n, CH):
        hi = min(n, lo + CH)
        wgt = np.arange(start + lo, start + hi, dtype=np.uint64)
        for limb in range(4):
            col = s[lo:hi, limb]
            for piece in range(4):
                part = (col >> np.uint64(16 * piece)) & np.uint64(0xFFFF)
                # weights < 2^26, pieces < 2^16, chunk 2^18 -> sums < 2^60
                total += int(np.sum(part * wgt, dtype=np.uint64)) << (64 * limb + 16 * piece)
    return total % pyref.R_MOD
