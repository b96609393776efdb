#!/usr/bin/env python3
"""Window sweep of the MSM on a precomputed shift table (gh_bases_precompute) against the plain
per-window path.  Usage: python tools/sweep_precompute.py <curve> <log_n> <c0> <c1> [reps]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref
import support as S
from __graft_entry__ import _load_pkg

curve, log_n, c0, c1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
gl = _load_pkg()
gl.init()
C = pyref.CURVES[curve]
n = 1 << log_n
pool_n = min(n, 1 << (12 if C.deg == 1 else 8))
pool = S.chain_points(C, pool_n, pyref.Rng(1))
pb, _ = S.bases_array(C, pool)
bases = np.tile(pb, (n // pool_n, 1))
scalars = S.random_scalars_np(n, seed=1000, below=C.order)
rb = gl.ResidentBases(curve, bases)
ds = gl.DeviceBuffer(n * 96).upload(scalars)


def run(tag):
    rb.msm_dev(ds, n)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = rb.msm_dev(ds, n)
        dt = (time.perf_counter() - t0) * 1e3
        tm = gl.msm_last_timing()
        if best is None or dt < best[0]:
            best = (dt, tm)
    dt, tm = best
    print("%-10s c=%2d W=%2d  wall %8.2f ms  %6.2f M/s | sort %6.2f acc %7.2f heavy %5.2f (%d) reduce %6.2f fold %5.2f" % (
        tag, tm["window_bits"], tm["num_windows"], dt, n / dt / 1e3, tm["sort_ms"], tm["accumulate_ms"], tm["heavy_ms"],
        tm["heavy_buckets"], tm["reduce_ms"], tm["fold_ms"]), flush=True)
    k = 6
    gl.msm_batch_dev([(rb, ds, n)] * 2)
    t0 = time.perf_counter()
    outs = gl.msm_batch_dev([(rb, ds, n)] * k)
    dtb = (time.perf_counter() - t0) * 1e3 / k
    tms = [gl.msm_batch_timing(i) for i in range(k)]
    ra = gl.proj_to_affine(curve, out)
    same = all((lambda a: a[1] == ra[1] and bool((a[0] == ra[0]).all()))(gl.proj_to_affine(curve, o)) for o in outs)
    print("           batch of %d pipelined: %8.2f ms per MSM  %6.2f M/s | acc %s | reduce %s | same result: %s" % (
        k, dtb, n / dtb / 1e3, " ".join("%.1f" % t["accumulate_ms"] for t in tms), " ".join("%.1f" % t["reduce_ms"] for t in tms), same), flush=True)
    return gl.proj_to_affine(curve, out)


ref = run("plain")
for c in range(c0, c1 + 1):
    t0 = time.perf_counter()
    try:
        rb.precompute(c)
    except Exception as e:
        print("c=%d: %s" % (c, e), flush=True)
        continue
    pre_s = time.perf_counter() - t0
    got = run("table")
    ok = got[1] == ref[1] and bool((got[0] == ref[0]).all())
    print("           precompute %.2f s   same affine result as plain: %s" % (pre_s, ok), flush=True)
    if not ok:
        sys.exit(1)
