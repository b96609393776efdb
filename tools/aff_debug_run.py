"""one G2 MSM with GH_AFF_DEBUG=1: which affine rounds fall back to the C++ kernel and why (development aid)
python3 tools/aff_debug_run.py <curve> <log_n> [table window]"""
import os, sys
os.environ["GH_AFF_DEBUG"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyref, support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
curve, log_n = sys.argv[1], int(sys.argv[2])
gl.init()
n = 1 << log_n
C = pyref.CURVES[curve]
prng = pyref.Rng(5)
p0, step = C.mul(prng.next_u64() | 1, C.G), C.mul(prng.next_u64() | 1, C.G)
xy, _ = S.bases_array(C, [p0, step])
rb = gl.ResidentBases.chain(curve, xy[0], xy[1], n)
if len(sys.argv) > 3:
    rb.precompute(int(sys.argv[3]))
sc = S.random_scalars_np(n, seed=9, below=C.order)
ds = gl.DeviceBuffer(n * 96).upload(sc)
gl.msm_set_affine(1)
out = rb.msm_dev(ds, n)
e_xy, e_inf = S.affine_abi_of_point(C, S.chain_msm_closed_form(C, p0, step, sc))
g_xy, g_inf = gl.proj_to_affine(curve, out)
print("closed form ok:", bool(g_inf == e_inf and (np.asarray(g_xy).reshape(-1) == e_xy).all()), gl.msm_last_timing())
