"""Latency of tiny MSMs (1..64 pairs) through gh_msm and of one host-side gh_proj_mul: what the prover's
2-pair "inputs" MSMs would cost if they were issued on their own (ginger-lib_amd/groth16.py folds them into
the large queries instead)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref, support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg(); gl.init()
for curve in ("mnt4753_g1", "mnt4753_g2"):
    C = pyref.CURVES[curve]
    rng = pyref.Rng(3)
    pts = S.chain_points(C, 64, rng)
    b, inf = S.bases_array(C, pts)
    for n in (1, 2, 3, 8, 64):
        s = S.scalar_array([rng.field_elem(C.order) for _ in range(n)])
        gl.VariableBaseMSM.multi_scalar_mul(curve, b[:n], s)
        t0 = time.perf_counter()
        for _ in range(3):
            gl.VariableBaseMSM.multi_scalar_mul(curve, b[:n], s)
        dt = (time.perf_counter() - t0) / 3 * 1e3
        print(curve, n, "%.2f ms" % dt, gl.msm_last_timing(), flush=True)
    xyz = S.proj_array(C, pts[0])
    k = S.scalar_array([rng.field_elem(C.order)])[0]
    t0 = time.perf_counter()
    for _ in range(5):
        gl.proj_mul(curve, xyz, k)
    print(curve, "proj_mul %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
