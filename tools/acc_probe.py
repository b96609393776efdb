"""One-line timings of the MSM stages for a given shape: python3 tools/acc_probe.py <curve> <log_n> [table=1] [batch=6] [reps=3]
Knobs come from the environment (GH_AFFINE, GH_AFF_ROUNDS, GH_AFF_BMIN, ...), one process per setting."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref
import support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
curve, log_n = sys.argv[1], int(sys.argv[2])
table = int(sys.argv[3]) if len(sys.argv) > 3 else 1
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 6
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
nocheck = len(sys.argv) > 6 and sys.argv[6] == "nocheck"
c_table = int(os.environ.get("PROBE_C", "0"))                     # explicit window for the shift table (0 = the library's choice)       # profiling runs: the last MSM of the process is the measured one
gl.init()
C = pyref.CURVES[curve]
n = 1 << log_n
pool_n = min(n, 4096 if C.deg == 1 else 512)
pb, _ = S.bases_array(C, S.chain_points(C, pool_n, pyref.Rng(1)))
bases = np.tile(pb, (n // pool_n, 1))
s = S.random_scalars_np(n, seed=1000, below=C.order)
rb = gl.ResidentBases(curve, bases)
ds = gl.DeviceBuffer(n * 96).upload(s)
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("GH_"))
if table:
    t0 = time.perf_counter()
    c = rb.precompute(c_table)
    print("precompute c=%d %.2f s" % (c, time.perf_counter() - t0), flush=True)
ref = None
for r in range(reps):
    t0 = time.perf_counter()
    out = rb.msm_dev(ds, n)
    wall = (time.perf_counter() - t0) * 1e3
    tm = gl.msm_last_timing()
    a = gl.proj_to_affine(curve, out)
    if ref is None:
        ref = a
    same = a[1] == ref[1] and bool((a[0] == ref[0]).all())
    print("[%s] %s 2^%d table=%d single: wall %.2f ms  %.2f M/s | c=%d W=%d sort %.2f acc %.2f heavy %.2f reduce %.2f fold %.2f | same %s" % (
        tag, curve, log_n, table, wall, n / wall / 1e3, tm["window_bits"], tm["num_windows"], tm["sort_ms"], tm["accumulate_ms"],
        tm["heavy_ms"], tm["reduce_ms"], tm["fold_ms"], same), flush=True)
if batch > 1:
    gl.msm_batch_dev([(rb, ds, n)] * 2)
    t0 = time.perf_counter()
    outs = gl.msm_batch_dev([(rb, ds, n)] * batch)
    wall = (time.perf_counter() - t0) * 1e3 / batch
    accs = " ".join("%.1f" % gl.msm_batch_timing(i)["accumulate_ms"] for i in range(batch))
    a = gl.proj_to_affine(curve, outs[-1])
    same = a[1] == ref[1] and bool((a[0] == ref[0]).all())
    print("[%s] %s 2^%d table=%d batch of %d: %.2f ms per MSM  %.2f M/s | acc %s | same %s" % (tag, curve, log_n, table, batch, wall, n / wall / 1e3, accs, same), flush=True)
if nocheck:
    sys.exit(0)
gl.msm_set_affine(0)
a = gl.proj_to_affine(curve, rb.msm_dev(ds, n))
print("projective kernel gives the same point:", a[1] == ref[1] and bool((a[0] == ref[0]).all()), " acc %.2f ms" % gl.msm_last_timing()["accumulate_ms"], flush=True)
