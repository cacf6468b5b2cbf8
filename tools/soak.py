#!/usr/bin/env python3
"""Builds and runs tools/soak.cpp (the randomized soak of the host runtime through the C ABI) and keeps its report.

    python tools/soak.py --seconds 300 --threads 16 --out gpurun_out/r06/soak.json            # the product library
    python tools/soak.py --seconds 120 --tsan --out gpurun_out/r06/soak_tsan.json              # host side of library + soak under ThreadSanitizer
    python tools/soak.py --seconds 120 --devices 2                                             # two logical devices on one GPU

--tsan: the library is rebuilt with the HOST side of every unit under -fsanitize=thread (snarkvm_amd/build.py tsan=True -> lib/libsnarkvm_hip_tsan.so,
device code untouched; ~7 min the first time - do it where the build cache lives, the .so travels), soak.cpp is compiled by clang++ with the same flag, and
the ThreadSanitizer reports (stderr) are counted and kept beside the JSON report.  The HIP / HSA runtimes are not instrumented: races INSIDE them are
invisible, calls into them are synchronisation-free as far as the tool knows - reports that only name runtime-internal frames are listed separately.
"""
import argparse
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build_soak(tsan, lib):
    exe = os.path.join(ROOT, "tools", "soak_tsan" if tsan else "soak")
    libdir = os.path.dirname(lib)
    libname = os.path.basename(lib)[3:-3]
    cxx = [CLANG, "-fsanitize=thread", "-g", "-O1"] if tsan else ["g++", "-O2"]
    cmd = cxx + ["-std=c++17", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "soak.cpp"), "-o", exe, "-L", libdir, f"-l{libname}",
                 f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--devices", type=int, default=1, help="logical devices (the visible GPU listed that many times)")
    ap.add_argument("--tsan", action="store_true")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from snarkvm_amd import build as hip_build

    lib = hip_build.TSAN_LIB if args.tsan else hip_build.LIB
    if args.tsan and args.build_only and (not os.path.exists(lib) or any(os.path.getmtime(f) > os.path.getmtime(lib) for f in hip_build._inputs())):
        hip_build.build(tsan=True, verbose=True)  # (only with --build-only, i.e. where the build cache lives: never a 7-minute compile on the GPU box)
    if not os.path.exists(lib):
        raise SystemExit(f"{lib} is missing: build it first (python -m snarkvm_amd.build{' --tsan' if args.tsan else ''})")
    exe = build_soak(args.tsan, lib)
    if args.build_only:
        print(exe)
        return
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    if args.tsan:
        # The HIP / HSA runtimes are not instrumented: their internal synchronisation (own atomics, futexes, doorbells) is invisible to the tool, so every
        # heap word they recycle between their threads looks like a race.  Reports whose racing access sits INSIDE those libraries are suppressed; what stays
        # are races whose accesses are in this library, in the soak, or in instrumented code they call.
        supp = os.path.join(ROOT, "tools", "tsan_suppressions.txt")
        env["TSAN_OPTIONS"] = env.get("TSAN_OPTIONS", f"halt_on_error=0 report_signal_unsafe=0 history_size=4 second_deadlock_stack=1 exitcode=0 suppressions={supp}")
    r = subprocess.run([exe, str(args.seconds), str(args.threads), str(args.seed), str(args.devices)], capture_output=True, text=True, env=env,
                       timeout=args.seconds + 900)
    line = next((ln for ln in r.stdout.splitlines() if ln.startswith("{")), None)
    rep = json.loads(line) if line else {"ok": False, "error": "no report line", "stdout_tail": r.stdout[-2000:]}
    rep["exit_code"] = r.returncode
    rep["library"] = os.path.basename(lib)
    if args.tsan:
        reports = re.split(r"(?m)^==================\n", r.stderr)
        reports = [x for x in reports if "WARNING: ThreadSanitizer" in x]

        def where(x):  # the module / source location of the SUMMARY line = the racing access the tool blames
            m = re.search(r"SUMMARY: ThreadSanitizer: [^/(]*\(?([^\s)]+)", x)
            return m.group(1) if m else "?"

        runtime_internal = [x for x in reports if re.search(r"libamdhip64|libhsa-runtime64|libhsakmt|librocprofiler|libdrm", where(x))]
        ours = [x for x in reports if x not in runtime_internal]
        by_site = {}
        for x in ours:
            by_site[where(x)] = by_site.get(where(x), 0) + 1
        rep["tsan"] = {"reports_after_suppressions": len(reports), "inside_the_uninstrumented_hip_hsa_runtimes": len(runtime_internal), "in_this_library_or_the_soak": len(ours),
                       "sites": by_site, "kinds": sorted({m.group(1).strip() for x in reports for m in [re.search(r"WARNING: ThreadSanitizer: ([^(\n]+)", x)] if m}),
                       "suppressed_by_tool": (re.findall(r"ThreadSanitizer: Matched (\d+) suppressions", r.stderr) or ["0"])[-1]}
        rep["ok"] = bool(rep.get("ok")) and not ours
        if args.out:
            with open(os.path.splitext(args.out)[0] + ".tsan.txt", "w") as f:
                f.write("\n==================\n".join(ours[:40]) if ours else ("no ThreadSanitizer report names this library or the soak\n" + "\n==================\n".join(runtime_internal[:3])))
    else:
        rep["stderr_tail"] = r.stderr[-1500:]
    print(json.dumps(rep))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rep, f, indent=1)
    sys.exit(0 if rep.get("ok") and (r.returncode == 0 or args.tsan) else 1)


if __name__ == "__main__":
    main()
