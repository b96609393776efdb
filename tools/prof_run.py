"""Small driver for rocprofv3: python3 tools/prof_run.py msm <log_n> [reps] | ntt <log_n> [reps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref
import support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
what, log_n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
gl.init()
rng = np.random.default_rng(1)
n = 1 << log_n
if what == "msm":
    C = pyref.CURVES["mnt4753_g1"]
    prng = pyref.Rng(3)
    pts = []
    H = C.mul(prng.next_u64() | 1, C.G); P = C.mul(prng.next_u64(), C.G)
    for _ in range(1024):
        pts.append(P); P = C.add(P, H)
    b = np.zeros((1024, 24), dtype=np.uint64)
    for i, Q in enumerate(pts):
        b[i, :12] = pyref.fe_to_abi(C.F, Q[0][0]); b[i, 12:] = pyref.fe_to_abi(C.F, Q[1][0])
    bases = np.tile(b, (max(1, n // 1024), 1))[:n]
    s = S.random_scalars_np(n, seed=11, below=C.order)        # uniform below r, the reference's sampling shape (fields/macros.rs:11-28)
    rb = gl.ResidentBases("mnt4753_g1", bases)
    if len(sys.argv) > 4 and sys.argv[4] == "table":
        print("shift table window", rb.precompute(0), flush=True)
    ds = gl.DeviceBuffer(n * 96).upload(s)
    for r in range(reps):
        t0 = time.time(); rb.msm_dev(ds, n)
        print("msm log_n", log_n, "wall %.2f ms" % ((time.time() - t0) * 1e3), gl.msm_last_timing(), flush=True)
else:
    a = S.random_scalars_np(n, seed=12, below=pyref.P6.p)     # field elements of MNT4-753 Fr
    buf = gl.DeviceBuffer(n * 96).upload(a)
    dom = gl.EvaluationDomain("mnt4753_fr", n)
    for r in range(reps):
        dom.fft_dev(buf, 0)
        print("fft log_n", log_n, "kernel ms", gl.fft_last_kernel_ms(), flush=True)
