"""Builds snarkvm_amd/lib/libsnarkvm_hip.so (hand-written HIP for gfx950) with hipcc.

hipcc cross-compiles without a GPU; the .so is kept in-tree (git-ignored) so that it travels to the
GPU box with the repository snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsnarkvm_hip.so")
SOURCES = ["api.hip"]
HEADERS = ["ff.cuh", "ec.cuh", "ntt.cuh", "msm.cuh", os.path.join("..", "..", "include", "snarkvm_hip.h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False, fast=False):
    """fast=True (development only) compiles without the G2 / Fq2 instantiations (about half the compile time)."""
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-result",
           "-o", LIB] + (["-DSV_NO_G2"] if fast else []) + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, fast="--fast" in sys.argv))
