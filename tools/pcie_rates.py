"""PCIe-inclusive rates of the host-buffer entry points next to the resident ones:
python3 tools/pcie_rates.py [log_n]      MNT4-753 G1, one MI355X"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref
import support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
gl.init()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
curve = "mnt4753_g1"
C = pyref.CURVES[curve]
n = 1 << log_n
prng = pyref.Rng(5)
xy, _ = S.bases_array(C, [C.mul(prng.next_u64() | 1, C.G), C.mul(prng.next_u64() | 1, C.G)])
rb = gl.ResidentBases.chain(curve, xy[0], xy[1], n)
bases = rb.download(0, n)                             # the same key as a host array (ABI layout)
scalars = S.random_scalars_np(n, seed=9, below=C.order)
ds = gl.DeviceBuffer(n * 96).upload(scalars)

def timed(f, reps=3):
    f()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = f()
    return (time.perf_counter() - t0) / reps * 1e3, out

ms_host, o0 = timed(lambda: gl.VariableBaseMSM.multi_scalar_mul(curve, bases, scalars))            # gh_msm: bases + scalars cross PCIe every call
ms_res, o1 = timed(lambda: rb.msm(scalars))                                         # gh_msm_resident: scalars cross PCIe
ms_dev, o2 = timed(lambda: rb.msm_dev(ds, n))                                       # gh_msm_resident_dev: nothing crosses
rb.precompute(0)
ms_res_t, o3 = timed(lambda: rb.msm(scalars))
ms_dev_t, o4 = timed(lambda: rb.msm_dev(ds, n))
aff = [gl.proj_to_affine(curve, o) for o in (o0, o1, o2, o3, o4)]
same = all(a[1] == aff[0][1] and (a[0] == aff[0][0]).all() for a in aff)
print("MNT4-753 G1 2^%d pairs, ms per MSM (M scalar-muls/s); same result: %s" % (log_n, same))
for name, ms in (("gh_msm (host bases 192 B + host scalars 96 B per pair)", ms_host), ("gh_msm_resident (host scalars), per-window path", ms_res),
                 ("gh_msm_resident_dev, per-window path", ms_dev), ("gh_msm_resident (host scalars), shift table", ms_res_t),
                 ("gh_msm_resident_dev, shift table", ms_dev_t)):
    print("  %-66s %8.2f ms  %6.2f M/s" % (name, ms, n / ms / 1e3))
# witness map: host rows in / host coefficients out against device-resident
F = "mnt4753_fr"
N = 1 << log_n
rows = [S.random_scalars_np(N, seed=20 + i, below=pyref.P6.p) for i in range(3)]
z = np.zeros(12, dtype=np.uint64)
ms_wm_host, _ = timed(lambda: gl.witness_map(F, rows[0], rows[1], rows[2], z, z, z), reps=2)
print("  %-66s %8.2f ms" % ("gh_witness_map 2^%d (three host rows in, h out over PCIe)" % log_n, ms_wm_host))
