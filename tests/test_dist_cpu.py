"""The N > 1 path of the MSM (shard, local sum, all-gather of partials, fold) on two CPU ranks
over gloo.  The local per-shard MSM is the CPU oracle here (no GPU in this container); the
exchange and the fold (gh_proj_add from the product library, host side) are the real code."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, curve, n, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import pyref
    import support as S
    from __graft_entry__ import _load_pkg
    gl = _load_pkg()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    C = pyref.CURVES[curve]
    rng = pyref.Rng(1234)                      # same inputs on every rank
    pts = S.chain_points(C, n, rng)
    scal = [rng.field_elem(C.order) for _ in range(n)]
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    sys.path.insert(0, os.path.join(ROOT, "ginger-lib_amd"))
    import importlib
    distmod = importlib.import_module("ginger_lib_amd.dist")
    lo, hi = distmod.shard_bounds(n, rank, world)
    sharded = distmod.ShardedMSM(curve, lambda sc: S.oracle_msm(curve, b[lo:hi], inf[lo:hi], sc, 2),
                                 lambda acc, p: gl.proj_add(curve, acc, p))
    total = sharded.multi_scalar_mul(s[lo:hi])
    xy, is_inf = gl.proj_to_affine(curve, total)
    # the same exchange through the C entry point (include/ginger_hip_dist.h) with gloo as the transport
    cd = distmod.CDist(gl, rank, world, transport="callback", allgather=distmod.gloo_allgather_bytes(dist))
    partial = S.oracle_msm(curve, b[lo:hi], inf[lo:hi], s[lo:hi], 2)
    total_c = cd.allgather_fold(curve, partial)
    xy_c, inf_c = gl.proj_to_affine(curve, total_c)
    c_ok = cd.world_seen == world and inf_c == is_inf and bool((xy_c == xy).all())
    # the callback transport says so: no RCCL rank took part
    c_ok = c_ok and cd.rccl_ranks == 0
    # a pipelined batch's results in ONE exchange: three partial sums (the shard, the shard's first half, infinity) -> three totals
    mid = lo + (hi - lo) // 2
    p_half = S.oracle_msm(curve, b[lo:mid], inf[lo:mid], s[lo:mid], 2)
    zero = np.zeros_like(partial)
    zero[12 * C.deg] = 1
    tot3 = cd.allgather_fold_batch(curve, [partial, p_half, zero])
    xy0, inf0 = gl.proj_to_affine(curve, tot3[0])
    c_ok = c_ok and inf0 == is_inf and bool((xy0 == xy).all())
    halves = [distmod.shard_bounds(n, r, world) for r in range(world)]
    idx = np.concatenate([np.arange(l, l + (h - l) // 2) for l, h in halves])
    exp_h = S.oracle_affine(curve, S.oracle_msm(curve, b[idx], inf[idx], s[idx], 2))
    xy1, inf1 = gl.proj_to_affine(curve, tot3[1])
    c_ok = c_ok and inf1 == exp_h[1] and bool((xy1 == exp_h[0]).all())
    c_ok = c_ok and gl.proj_to_affine(curve, tot3[2])[1]           # infinity + infinity
    cd.shutdown()
    # gh_shutdown tears a communicator down as well (and a second shutdown is harmless)
    cd2 = distmod.CDist(gl, rank, world, transport="callback", allgather=distmod.gloo_allgather_bytes(dist))
    gl.shutdown()
    import ctypes
    rr = ctypes.c_int(-1)
    gl.load_library().gh_dist_transport(ctypes.byref(rr), None, None, 0)
    c_ok = c_ok and rr.value == 0 and gl.load_library().gh_dist_info(None, None) != 0
    full = S.oracle_msm(curve, b, inf, s, 2)
    exy, einf = S.oracle_affine(curve, full)
    ok = c_ok and (is_inf == einf) and bool((xy == exy).all())
    ret[rank] = ok
    dist.destroy_process_group()


@pytest.mark.parametrize("curve,n,world", [("mnt4753_g1", 37, 2), ("mnt6753_g2", 9, 2), ("mnt6753_g1", 41, 8)])
def test_two_rank_sharded_msm(curve, n, world):
    """world 2 on two curves; world 8 = the rank count of BASELINE config 4 (shards of 5 / 6 pairs, the all-gather and the fold in
    rank order over eight partial sums) rehearsed on CPU ranks over gloo, as the GPU node itself is not the builder's to use"""
    port = 29500 + (os.getpid() % 2000) + world
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, curve, n, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world))


def test_shard_bounds_cover():
    sys.path.insert(0, ROOT)
    from __graft_entry__ import _load_pkg
    _load_pkg()
    import importlib
    distmod = importlib.import_module("ginger_lib_amd.dist")
    for n in (0, 1, 7, 8, 1000):
        for world in (1, 2, 3, 8):
            cuts = [distmod.shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
