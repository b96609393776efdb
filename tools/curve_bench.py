"""MSM timing for any of the four curves: python3 tools/curve_bench.py <curve> <log_n> [reps] [window]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref, support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
curve, log_n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
gl.init()
if len(sys.argv) > 4:
    gl.msm_set_window(int(sys.argv[4]))
C = pyref.CURVES[curve]
n = 1 << log_n
pool_n = min(n, 256)
pool = S.chain_points(C, pool_n, pyref.Rng(1))
pb, _ = S.bases_array(C, pool)
bases = np.tile(pb, (n // pool_n, 1))
s = S.random_scalars_np(n, seed=5, below=C.order)
rb = gl.ResidentBases(curve, bases)
ds = gl.DeviceBuffer(n * 96).upload(s)
for r in range(reps):
    t0 = time.perf_counter(); out = rb.msm_dev(ds, n); dt = time.perf_counter() - t0
    tm = gl.msm_last_timing()
    print(curve, "log_n", log_n, "c", tm["window_bits"], "wall %.1f ms" % (dt * 1e3), "%.2f M pairs/s" % (n / dt / 1e6),
          {k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")}, flush=True)
if log_n <= 14:
    exp = S.oracle_msm(curve, bases, None, s, 16)
    g_xy, g_inf = gl.proj_to_affine(curve, out); e_xy, e_inf = S.oracle_affine(curve, exp)
    print("matches oracle:", g_inf == e_inf and bool((g_xy == e_xy).all()))
