"""Soak: MSMs of changing sizes / curves / modes in one process; the device memory the library holds must settle
(pool buffers grow to the largest request and stay), results must stay equal run to run.
python3 tools/soak.py [rounds]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pyref, support as S
from __graft_entry__ import _load_pkg
import torch
gl = _load_pkg(); gl.init()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
specs = [("mnt4753_g1", 17), ("mnt4753_g2", 16), ("mnt6753_g1", 18), ("mnt6753_g2", 15), ("mnt4753_g1", 14)]
keys = []
for curve, lg in specs:
    C = pyref.CURVES[curve]; n = 1 << lg
    pb, _ = S.bases_array(C, S.chain_points(C, 256, pyref.Rng(lg)))
    rb = gl.ResidentBases(curve, np.tile(pb, (n // 256, 1)))
    ds = gl.DeviceBuffer(n * 96).upload(S.random_scalars_np(n, seed=lg, below=C.order))
    keys.append((curve, n, rb, ds))
ref = {}
free_hist = []
for r in range(rounds):
    for i, (curve, n, rb, ds) in enumerate(keys):
        if r == 2 and i % 2 == 0:
            rb.precompute(0)
        m = n - (r * 977) % (n // 2)
        outs = [rb.msm_dev(ds, m)] + gl.msm_batch_dev([(rb, ds, m)] * 3)
        aff = [gl.proj_to_affine(curve, o) for o in outs]
        assert all(a[1] == aff[0][1] and (a[0] == aff[0][0]).all() for a in aff), (curve, r)
        key = (i, m)
        if key in ref:
            assert ref[key][1] == aff[0][1] and (ref[key][0] == aff[0][0]).all()
        ref[key] = aff[0]
        if os.environ.get("SOAK_VERBOSE"):
            print("   round %d %s m=%d free %.3f GB" % (r, curve, m, torch.cuda.mem_get_info()[0] / 2**30), flush=True)
    gl.load_library().gh_dev_sync()
    free_b, total_b = torch.cuda.mem_get_info()
    free_hist.append(free_b)
    print("round %d: free %.2f GB" % (r, free_b / 2**30), flush=True)
# one-time steps are expected (pool buffers, shift tables in round 2, the HIP runtime's scratch arena the first time a kernel
# with a large private segment runs -- e.g. msm_heavy_combine_kernel<Mnt6G2>: 16.5 KB per lane, about 6 GB); a leak would keep growing
grow = (free_hist[-3] - free_hist[-1]) / 2**20 if rounds >= 4 else 0.0
print("device memory taken over the last two rounds: %.1f MB" % grow)
sys.exit(1 if grow > 64 else 0)
