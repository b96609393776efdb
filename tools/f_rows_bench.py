"""Timings of the SURVEY 8f rows at BASELINE config 5's size (device-resident entry points, one MI355X):
python3 tools/f_rows_bench.py [log_n=20]   ->  one line per entry point (wall ms around the _dev call + gh_dev_sync, mean of 5)"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref
import support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << log_n
gl.init()
lib = gl.load_library()
F = "mnt4753_fr"
fid = gl.FIELDS[F]
p = pyref.P6.p
rows = [S.random_scalars_np(n, seed=s, below=p) for s in (1, 2, 3)]
d = S.random_scalars_np(3, seed=4, below=p)
bufs = [gl.DeviceBuffer(n * 96 + 192) for _ in range(4)]


def timed(name, fn, reps=5, work=None):
    fn(); lib.gh_dev_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    lib.gh_dev_sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print("%-34s 2^%d  %8.3f ms%s" % (name, log_n, ms, "   %.1f M elements/s" % (work / ms / 1e3) if work else ""), flush=True)


def reload():
    for b, r in zip(bufs, rows):
        b.upload(r)


reload()
timed("gh_witness_map_dev (R1CS->QAP)", lambda: gl._check(lib.gh_witness_map_dev(fid, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, log_n, gl._ptr(d[0]), gl._ptr(d[1]), gl._ptr(d[2]), bufs[3].ptr)), work=n)
reload()
timed("gh_sap_witness_map_dev (R1CS->SAP)", lambda: gl._check(lib.gh_sap_witness_map_dev(fid, bufs[0].ptr, bufs[2].ptr, log_n, gl._ptr(d[0]), gl._ptr(d[1]), bufs[3].ptr)), work=n)
reload()
timed("gh_batch_inverse_dev", lambda: gl._check(lib.gh_batch_inverse_dev(fid, bufs[0].ptr, n)), work=n)
tau = S.random_scalars_np(1, seed=9, below=p)[0]
timed("gh_lagrange_coefficients_dev", lambda: gl._check(lib.gh_lagrange_coefficients_dev(fid, log_n, gl._ptr(tau), bufs[3].ptr)), work=n)
timed("gh_fft_dev (one transform)", lambda: gl._check(lib.gh_fft_dev(fid, bufs[0].ptr, log_n, 0)), work=n)
# FixedBaseMSM at generator scale: one 2^log_n-scalar call on G1 and on G2 (window by the reference's rule), affine output
for curve in ("mnt4753_g1", "mnt4753_g2"):
    C = pyref.CURVES[curve]
    w = gl.FixedBaseMSM.get_mul_window_size(n)
    t0 = time.perf_counter()
    tab = gl.FixedBaseMSM(curve, S.proj_array(C, C.G), 753, w)
    t_tab = time.perf_counter() - t0
    s = S.random_scalars_np(n, seed=5, below=C.order)
    tab.multi_scalar_mul_affine(s[:1024])
    t0 = time.perf_counter()
    tab.multi_scalar_mul_affine(s)
    dt = time.perf_counter() - t0
    tab.free()
    print("gh_fixed_base_msm_affine %-10s 2^%d  window %d: table %.2f s, %8.1f ms incl. PCIe both ways   %.2f M scalar-muls/s" % (curve, log_n, w, t_tab, dt * 1e3, n / dt / 1e6), flush=True)
for b in bufs:
    b.free()
