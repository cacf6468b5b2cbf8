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
SOURCES = ["api.hip", "api_fr.hip", "api_g2.hip", "api_serde.hip"]  # compiled in parallel (the Fq2 instantiations are half of the compile time), then linked


def _inputs():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "snarkvm_hip.h"))
    return files


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _inputs())


OBJDIR = os.environ.get("SNARKVM_HIP_OBJDIR", "/tmp/snarkvm_hip_obj")  # objects stay out of the tree (they would travel to the GPU box)


def build(force=False, verbose=False, fast=False, only=None, ool=False):
    """fast=True (development only) compiles without the G2 / Fq2 instantiations.  ool=True (A/B switch): every exceptional
    path out of line (-DSV_COLD_OOL): kernels a few percent slower, see ff.hip.h.  only=[...] (development only): recompile just
    the listed translation units and link them with the objects kept from the last build, whatever their age - for experiments
    on a kernel that one unit instantiates (each unit takes 2 - 5 minutes); the driver's build() always compiles everything."""
    if not force and not only and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"] + (["-DSV_NO_G2"] if fast else []) + (["-DSV_COLD_OOL"] if ool else [])
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if only and src not in only:
            if not os.path.exists(obj):
                raise FileNotFoundError(f"{obj}: no kept object for {src}; run a full build first")
            continue
        cmd = [hipcc] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return LIB


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a.endswith(".hip")]
    print(build(force="--force" in sys.argv, verbose=True, fast="--fast" in sys.argv, only=only or None, ool="--ool" in sys.argv))
