"""Canonical (de)serialisation of G1 points on the device.

Mirror of `CanonicalSerialize / CanonicalDeserialize for Affine<P>` (curves/src/templates/macros.rs:66-140) and of the
`.usrs` universal-SRS files (parameters/src/mainnet/resources: a u64 point count, then uncompressed points; loaded by
`PowersOfBetaG::load`, parameters/src/mainnet/powers.rs).  Points decoded here never pass through a CPU field
implementation: the bytes are copied to HBM and converted by one kernel (snarkvm_amd/csrc/serde.hip.h)."""
import ctypes
import struct

import numpy as np

from . import _lib
from .layout import G1_AFFINE

UNCOMPRESSED_SIZE = 96
COMPRESSED_SIZE = 48


class SerializationError(ValueError):
    pass


def _check(err):
    try:
        _lib.check(err)
    except _lib.HipError as e:
        if e.code == 1:  # SerializationError raised by the decoder (flags / non-canonical / InvalidData)
            raise SerializationError(e.message) from None
        raise


def g1_deserialize(data, compressed=False, validate=False):
    """bytes -> G1_AFFINE record array (`deserialize_{un,}compressed[_unchecked]`)."""
    psz = COMPRESSED_SIZE if compressed else UNCOMPRESSED_SIZE
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    if buf.shape[0] % psz:
        raise SerializationError("truncated input")
    n = buf.shape[0] // psz
    out = np.zeros(n, dtype=G1_AFFINE)
    _check(_lib.lib().snarkvm_hip_g1_deserialize(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n),
                                                ctypes.c_int(int(compressed)), ctypes.c_int(int(validate))))
    return out


def g1_serialize(points, compressed=False):
    """G1_AFFINE records -> bytes (`serialize_{un,}compressed`)."""
    points = np.ascontiguousarray(points, dtype=G1_AFFINE).reshape(-1)
    psz = COMPRESSED_SIZE if compressed else UNCOMPRESSED_SIZE
    out = np.zeros(points.shape[0] * psz, dtype=np.uint8)
    _check(_lib.lib().snarkvm_hip_g1_serialize(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(points.ctypes.data), ctypes.c_size_t(points.shape[0]),
                                              ctypes.c_size_t(G1_AFFINE.itemsize), ctypes.c_int(int(compressed))))
    return out.tobytes()


def split_usrs(data):
    """`.usrs` = `Vec<G1Affine>::serialize_uncompressed`: u64 little-endian count, then count * 96 bytes."""
    (count,) = struct.unpack_from("<Q", data, 0)
    body = memoryview(data)[8 : 8 + count * UNCOMPRESSED_SIZE]
    if len(body) != count * UNCOMPRESSED_SIZE:
        raise SerializationError("truncated .usrs")
    return count, body


def register_bases_serialized(data, npoints, compressed=False, validate=False, tables=1):
    """bytes -> opaque registered-bases handle (snarkvm_hip_register_bases_serialized)."""
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    h = ctypes.c_void_p()
    _check(_lib.lib().snarkvm_hip_register_bases_serialized(ctypes.byref(h), ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(npoints),
                                                           ctypes.c_int(int(compressed)), ctypes.c_int(int(validate)), ctypes.c_int(int(tables))))
    return h


G2_UNCOMPRESSED_SIZE = 192
G2_COMPRESSED_SIZE = 96


def g2_deserialize(data, validate=False, compressed=False):
    """bytes -> G2_AFFINE record array: uncompressed (192 B, e.g. `beta-h.usrs`) or compressed (96 B: x only; y is the Fq2
    square root of x^3 + b' selected by the sign flag, recovered on the device)."""
    from .layout import G2_AFFINE

    psz = G2_COMPRESSED_SIZE if compressed else G2_UNCOMPRESSED_SIZE
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    if buf.shape[0] % psz:
        raise SerializationError("truncated input")
    n = buf.shape[0] // psz
    out = np.zeros(n, dtype=G2_AFFINE)
    fn = _lib.lib().snarkvm_hip_g2_deserialize_compressed if compressed else _lib.lib().snarkvm_hip_g2_deserialize
    _check(fn(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n), ctypes.c_int(int(validate))))
    return out


def g2_serialize(points, compressed=False):
    from .layout import G2_AFFINE

    points = np.ascontiguousarray(points, dtype=G2_AFFINE).reshape(-1)
    out = np.zeros(points.shape[0] * (G2_COMPRESSED_SIZE if compressed else G2_UNCOMPRESSED_SIZE), dtype=np.uint8)
    fn = _lib.lib().snarkvm_hip_g2_serialize_compressed if compressed else _lib.lib().snarkvm_hip_g2_serialize
    _check(fn(ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(points.ctypes.data), ctypes.c_size_t(points.shape[0]), ctypes.c_size_t(G2_AFFINE.itemsize)))
    return out.tobytes()
