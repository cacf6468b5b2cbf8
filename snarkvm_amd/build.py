"""Builds snarkvm_amd/lib/libsnarkvm_hip.so (hand-written HIP for gfx950) with hipcc.

hipcc cross-compiles without a GPU; the .so is kept in-tree (git-ignored) so that it travels to the
GPU box with the repository snapshot.
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libsnarkvm_hip.so")
SOURCES = ["api.hip", "api_fr.hip", "api_g2.hip", "api_serde.hip", "tail_g1.hip", "tail_g2.hip", "tail_g2_planes.hip", "tail_g2_fix.hip"]  # compiled in parallel, then linked (the tail_* units: the fold / bit-plane kernels, msm.hip.h)


def _inputs():
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(HERE, "..", "include", "snarkvm_hip.h"))
    return files


def _headers():
    return [f for f in _inputs() if f.endswith(".h")]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _inputs())


def _objdir(flags):
    """Objects stay out of the tree (they would travel to the GPU box).  One private directory per (checkout, flag set): two
    checkouts, or a --fast / --ool build beside the product build, never see each other's objects.  SNARKVM_HIP_OBJDIR overrides."""
    env = os.environ.get("SNARKVM_HIP_OBJDIR")
    if env:
        return env
    key = hashlib.sha256((os.path.realpath(HERE) + "\0" + " ".join(flags)).encode()).hexdigest()[:16]
    base = os.path.join(os.environ.get("XDG_CACHE_HOME") or os.path.join(tempfile.gettempdir(), f"snarkvm_hip_{os.getuid()}"), "obj")
    return os.path.join(base, key)


def _stamp(obj):
    return obj + ".json"


def _kept_object_is_current(obj, src, flags):
    """A kept object may be linked only if it was compiled with the same flags, from this source, after every header's last change."""
    try:
        with open(_stamp(obj)) as f:
            st = json.load(f)
    except (OSError, ValueError):
        return False
    if st.get("flags") != flags or st.get("src") != os.path.realpath(src):
        return False
    t = os.path.getmtime(obj)
    return all(os.path.getmtime(h) <= t for h in _headers() + [src])


TSAN_LIB = os.path.join(LIBDIR, "libsnarkvm_hip_tsan.so")


def build(force=False, verbose=False, fast=False, only=None, ool=False, tsan=False):
    """fast=True (development only) compiles without the G2 / Fq2 instantiations.  ool=True (A/B switch): every exceptional
    path out of line (-DSV_COLD_OOL): kernels a few percent slower, see ff.hip.h.  only=[...] (development only): recompile just
    the listed translation units and link them with the objects kept from the last build of THIS checkout with THESE flags -
    refused when a kept object is older than any header (a mixed library must never reach the GPU box); the driver's build()
    always compiles everything.
    tsan=True (diagnostics: tools/soak.py --tsan): the HOST side of every unit compiled with -fsanitize=thread (device code untouched) into a SEPARATE library,
    lib/libsnarkvm_hip_tsan.so - never the product library; a process that loads it needs the ThreadSanitizer runtime (an instrumented executable, or
    LD_PRELOAD of clang's libclang_rt.tsan-x86_64.so)."""
    out_lib = TSAN_LIB if tsan else LIB
    if not tsan and not force and not only and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # Device-function calls (round 6, ROCm 7.2 hipcc): the ~260 000-instruction Fq2 fold / bit-plane kernels used to keep their exceptional additions out of line;
    # a call inside them came back with live registers of the caller overwritten - with the backend's interprocedural register allocation on (its default) in one
    # source revision, with `-mllvm -enable-ipra=false` in other kernels - so those kernels now hold no call at all (msm.hip.h TAIL_FLAGGED + the fix kernels) and the
    # flags stay the toolchain's defaults, the configuration every other kernel's parity tests ran under.
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-pass-failed"] + (["-DSV_NO_G2"] if fast else []) + (["-DSV_COLD_OOL"] if ool else [])
    if tsan:
        flags += ["-Xarch_host", "-fsanitize=thread", "-Xarch_host", "-g"]
    objdir = _objdir(flags)
    os.makedirs(objdir, mode=0o700, exist_ok=True)
    if os.path.islink(objdir) or os.stat(objdir).st_uid != os.getuid():
        raise PermissionError(f"{objdir}: object directory is a symlink or belongs to another user")
    objs, procs = [], []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if only and src not in only:
            if not os.path.exists(obj) or not _kept_object_is_current(obj, path, flags):
                raise RuntimeError(f"{src}: the kept object is missing, was built with other flags, or is older than a header it includes - "
                                   f"add it to the list or run a full build")
            continue
        if os.path.exists(_stamp(obj)):
            os.remove(_stamp(obj))
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, obj, path, subprocess.Popen(cmd)))
    for cmd, obj, path, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
        with open(_stamp(obj), "w") as f:
            json.dump({"flags": flags, "src": os.path.realpath(path)}, f)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_lib] + objs
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    return out_lib


if __name__ == "__main__":
    only = [a for a in sys.argv[1:] if a.endswith(".hip")]
    print(build(force="--force" in sys.argv, verbose=True, fast="--fast" in sys.argv, only=only or None, ool="--ool" in sys.argv, tsan="--tsan" in sys.argv))
