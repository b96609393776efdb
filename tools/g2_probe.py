"""MSMs over a chain key (distinct bases, as bench.py's; any of the four curves) for profiling and shard-size sweeps:
python3 tools/g2_probe.py <curve> <log_n> [reps=2] [window=0] [batch=0]   (batch > 0: also a pipelined batch of that many MSMs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyref, support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
curve, log_n = sys.argv[1], int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
gl.init()
n = 1 << log_n
C = pyref.CURVES[curve]
prng = pyref.Rng(5)
p0, step = C.mul(prng.next_u64() | 1, C.G), C.mul(prng.next_u64() | 1, C.G)
xy, _ = S.bases_array(C, [p0, step])
rb = gl.ResidentBases.chain(curve, xy[0], xy[1], n)
c = rb.precompute(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
batch = int(sys.argv[5]) if len(sys.argv) > 5 else 0
sc = S.random_scalars_np(n, seed=9, below=C.order)
ds = gl.DeviceBuffer(n * 96).upload(sc)
for r in range(reps):
    t0 = time.perf_counter(); out = rb.msm_dev(ds, n); dt = time.perf_counter() - t0
    tm = gl.msm_last_timing()
    print(curve, "log_n", log_n, "c", tm["window_bits"], "wall %.1f ms" % (dt * 1e3), {k: round(v, 2) for k, v in tm.items() if k.endswith("_ms")}, flush=True)
if batch:
    gl.msm_batch_dev([(rb, ds, n)] * 2)
    t0 = time.perf_counter(); gl.msm_batch_dev([(rb, ds, n)] * batch); dt = (time.perf_counter() - t0) / batch
    print(curve, "log_n", log_n, "c", c, "pipelined batch of %d: %.2f ms per MSM  %.2f M scalar-muls/s" % (batch, dt * 1e3, n / dt / 1e6), flush=True)
