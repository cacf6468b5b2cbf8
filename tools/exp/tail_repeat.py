"""diag: G2 fold / bit-plane kernels over fixed inputs, repeated: bit-identical outputs?"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("ROOT", "/root/repo"))
from snarkvm_amd import _lib, synthetic
L = _lib.lib()
pts = synthetic.g2_points(512)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for (m, hb) in ((7, 7), (7, 6), (8, 7)):
    for threads in (256, 128, 64):
        for hex_ in (0, 1):
            for quads in (0, 3):
                if threads == 64 and quads: continue
                rep = np.zeros(10, dtype=np.uint32)
                _lib.check(L.snarkvm_hip_devtest_g2_tail_repeat(ctypes.c_void_p(pts.ctypes.data), ctypes.c_size_t(512), m, hb, threads, 256 if quads else 128, hex_, quads, iters,
                                                                ctypes.c_void_p(rep.ctypes.data)))
                print(f"m={m} hb={hb} threads={threads} hex={hex_} quads={quads}: fold launches differing {rep[0]}/{iters - 1} (slots {rep[1]}, first {list(rep[4:8])}), plane launches differing {rep[2]} (planes {rep[3]}); flagged fold {rep[8]} planes {rep[9]}", flush=True)
