"""CPU-only checks of the drop-in boundary: the HIP library builds/loads and exports every symbol declared in
include/snarkvm_hip.h; argument validation that needs no device; the oracle is not reachable from the product."""
import ctypes
import os
import re

import numpy as np
import pytest

from snarkvm_amd import _lib
from tests import util


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(util.ROOT, "include", "snarkvm_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(snarkvm_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name


def _rust_width(t):
    """Width class of a Rust FFI type as written in sys.rs (the twin of tools/gen_bindings.py::width_class on the C side)."""
    t = t.strip()
    depth = 0
    while t.startswith("*"):
        t = re.sub(r"^\*(const|mut)\s+", "", t)
        depth += 1
    if depth:
        return ("ptr", depth)
    if t == "Error":
        return ("err",)
    if t in ("NTTInputOutputOrder", "NTTDirection", "NTTType"):
        return ("int", 4)  # #[repr(C)] enums
    return {"usize": ("int", 8), "i32": ("int", 4), "u32": ("int", 4), "u64": ("int", 8), "f64": ("f64",)}[t]


def test_rust_sys_matches_the_header():
    """rust/snarkvm-algorithms-hip/src/sys.rs against include/snarkvm_hip.h: every declared function present (and nothing else), the same number
    of arguments, the same width class per argument and for the result, pointer depth included; the three #[repr(C)] enums of lib.rs carry the
    header's discriminants; and the committed file is exactly what tools/gen_bindings.py yields today."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_bindings", os.path.join(util.ROOT, "tools", "gen_bindings.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    funcs, consts = gen.parse_header()
    c_side = {f["name"]: f for f in funcs}
    assert set(c_side) == set(_lib.SYMBOLS)  # the header parser sees what the ctypes table lists
    rust_dir = os.path.join(util.ROOT, "rust", "snarkvm-algorithms-hip", "src")
    sys_rs = open(os.path.join(rust_dir, "sys.rs")).read()
    block = sys_rs[sys_rs.index('extern "C" {') :]
    rust_side = {}
    for m in re.finditer(r"pub fn (\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
        args = [a.split(":", 1) for a in m.group(2).split(",") if a.strip()]
        rust_side[m.group(1)] = ([(n.strip(), _rust_width(t)) for n, t in args], _rust_width(m.group(3)) if m.group(3) else ("void",))
    assert set(rust_side) == set(c_side), set(rust_side) ^ set(c_side)
    for name, f in c_side.items():
        r_args, r_ret = rust_side[name]
        assert len(r_args) == len(f["args"]), name
        for (rn, rw), (cn, ct) in zip(r_args, f["args"]):
            assert rw == gen.width_class(ct), (name, cn, rw, ct)
        assert r_ret == gen.width_class(f["ret"]), name
    lib_rs = open(os.path.join(rust_dir, "lib.rs")).read()
    assert "pub mod sys;" in lib_rs and 'extern "C"' not in lib_rs  # no second, hand-written declaration block
    for enum in ("NTTInputOutputOrder", "NTTDirection", "NTTType"):
        body = re.search(r"#\[repr\(C\)\][^{]*pub enum " + enum + r"\s*\{([^}]*)\}", lib_rs).group(1)
        for k, v in re.findall(r"(\w+)\s*=\s*(\d+)", body):
            assert consts[k] == int(v), (enum, k)
    for k in ("SNARKVM_HIP_SCOPE_ASYNC_MSM", "SNARKVM_HIP_SCOPE_STABLE_INPUTS", "SNARKVM_HIP_SCOPE_MSM_IN_STREAM"):
        assert re.search(rf"pub const {k}: u32 = {consts[k]};", sys_rs), k
    # every sys:: item the hand-written wrappers (and INTEGRATION.md's sketches) name is a declared one
    used = set(re.findall(r"sys::(snarkvm_\w+)", lib_rs + open(os.path.join(util.ROOT, "INTEGRATION.md")).read())) - set(gen.OPAQUE)
    assert used and used <= set(c_side), used - set(c_side)
    assert sys_rs == gen.generate(), "sys.rs is stale: run python tools/gen_bindings.py"


def test_rust_error_layout():
    # sppark cuda::Error {code: i32, message: *mut c_char}: 16 bytes on x86-64
    assert ctypes.sizeof(_lib.RustError) == 16


def test_polymul_host_only_corner_cases():
    """snarkvm.cu:196-201: no inputs -> no-op success; exactly one polynomial -> plain copy (no device involved)."""
    from snarkvm_amd import plugin

    out = plugin.polymul(8, [], [])
    assert not out.any()
    p = np.arange(12, dtype=np.uint64).reshape(3, 4)
    out = plugin.polymul(8, [p], [])
    assert np.array_equal(out[:3], p) and not out[3:].any()


def test_argument_validation_raises_before_ffi():
    from snarkvm_amd import plugin
    from snarkvm_amd.layout import G1_AFFINE

    with pytest.raises(ValueError):
        plugin.NTT(12, np.zeros((12, 4), dtype=np.uint64), 0, 0, 0)  # not a power of two (lib.rs:84-86)
    with pytest.raises(ValueError):
        plugin.msm(np.zeros(3, dtype=G1_AFFINE), np.zeros((4, 4), dtype=np.uint64))  # lib.rs:150-152


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a device every compute entry point must fail loudly (the reference's caller owns the CPU fallback)."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    from snarkvm_amd import plugin

    with pytest.raises(_lib.HipError):
        plugin.NTT(8, np.zeros((8, 4), dtype=np.uint64), 0, 0, 0)
    from snarkvm_amd.layout import G1_AFFINE

    with pytest.raises(_lib.HipError):
        plugin.msm(np.zeros(4, dtype=G1_AFFINE), np.zeros((4, 4), dtype=np.uint64))


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under snarkvm_amd/ (Python or C++/HIP) or include/ may import, include,
    link or dlopen it; bench.py only inside its cpu_baseline leg; __graft_entry__ only in build()/smoke()."""
    pat_py = re.compile(r"^\s*(from\s+oracle\b|import\s+oracle\b)", re.M)
    pat_c = re.compile(r"#\s*include\s*[\"<][^\">]*oracle|liboracle|cpu_oracle")
    for top in ("snarkvm_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(util.ROOT, top)):
            for f in files:
                path = os.path.join(dirpath, f)
                if f.endswith(".py"):
                    assert not pat_py.search(open(path).read()), path
                elif f.endswith((".hip", ".h", ".hpp", ".cpp")):
                    assert not pat_c.search(open(path).read()), path
    bench = open(os.path.join(util.ROOT, "bench.py")).read()
    # every oracle import of bench.py sits inside a checking leg that runs after the timed region of its function
    markers = ("CPU baseline + oracle checks (rank 0, N = 1 only)", "# ---- checks (outside the timed regions)")
    for m in re.finditer(r"from oracle import", bench):
        func_start = bench.rfind("\ndef ", 0, m.start())
        body = bench[func_start : m.start()]
        if body.startswith("\ndef oracle_check_proof("):
            continue  # the checker of the proof-shaped legs: its CALL SITES are held to the same rule below
        assert any(k in body for k in markers), bench[m.start() - 200 : m.start() + 40]
        assert "time.perf_counter() - t0" in body  # the timed region of that function has ended before the import
    calls = [m for m in re.finditer(r"(?<!def )oracle_check_proof\(", bench)]
    assert calls
    for m in calls:
        func_start = bench.rfind("\ndef ", 0, m.start())
        body = bench[func_start : m.start()]
        assert "# ---- checks (outside the timed regions)" in body, bench[m.start() - 200 : m.start() + 40]
        assert "time.perf_counter() - t0" in body
    assert len(re.findall(r"from oracle import", bench)) >= 2
