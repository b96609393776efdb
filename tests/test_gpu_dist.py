"""The RCCL transport of include/ginger_hip_dist.h on the one GPU this box has: librccl is loaded at run time, a one-rank
communicator is created from a unique id, and gh_partials_allgather_fold returns the rank's own partial sum (the fold over
one rank).  The multi-rank fold itself is covered on CPU ranks over gloo (tests/test_dist_cpu.py: same C entry point, custom
transport); real multi-GPU runs are the driver's."""
import ctypes
import importlib

import numpy as np
import pytest

import pyref
import support as S

pytestmark = pytest.mark.gpu


def test_rccl_single_rank_exchange(gpu):
    lib = gpu.load_library()
    uid = (ctypes.c_char * 128)()
    gpu._check(lib.gh_dist_unique_id(uid))
    assert any(b != b"\x00" for b in uid)
    gpu._check(lib.gh_dist_init_rccl(uid, 0, 1))
    try:
        r, w = ctypes.c_int(), ctypes.c_int()
        gpu._check(lib.gh_dist_info(ctypes.byref(r), ctypes.byref(w)))
        assert (r.value, w.value) == (0, 1)
        assert lib.gh_dist_init_rccl(uid, 0, 1) != 0               # one communicator per process
        for curve in ("mnt4753_g1", "mnt6753_g2"):
            C = pyref.CURVES[curve]
            P = S.chain_points(C, 1, pyref.Rng(3))[0]
            part = S.proj_array(C, P)
            out = np.zeros_like(part)
            us = ctypes.c_double()
            gpu._check(lib.gh_partials_allgather_fold(gpu.CURVES[curve], gpu._ptr(part), gpu._ptr(out), ctypes.byref(us)))
            assert S.affine_of_xyz(C, out) == P and us.value > 0
            # three partial sums in one exchange
            Q = S.chain_points(C, 3, pyref.Rng(4))
            parts = np.stack([S.proj_array(C, q) for q in Q])
            outs = np.zeros_like(parts)
            gpu._check(lib.gh_partials_allgather_fold_batch(gpu.CURVES[curve], gpu._ptr(parts), 3, gpu._ptr(outs), ctypes.byref(us)))
            assert [S.affine_of_xyz(C, o) for o in outs] == Q
        # the transport reports itself: one RCCL rank, a version, and the file the symbols came from
        rr, rv = ctypes.c_int(), ctypes.c_int()
        path = ctypes.create_string_buffer(512)
        gpu._check(lib.gh_dist_transport(ctypes.byref(rr), ctypes.byref(rv), path, 512))
        assert rr.value == 1 and rv.value > 20000 and b"librccl" in path.value
        print("RCCL in use:", rv.value, path.value.decode())
    finally:
        lib.gh_dist_shutdown()
    assert lib.gh_dist_info(None, None) != 0


def test_sharded_msm_through_c_exchange_single_process(gpu):
    """two shards of one MSM on this GPU, their partial sums folded through gh_partials_allgather_fold with a custom
    (in-process) transport standing in for the second rank: the affine result equals the unsharded MSM and the oracle"""
    distmod = importlib.import_module("ginger_lib_amd.dist")
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    rng = pyref.Rng(21)
    n = 900
    pts = S.chain_points(C, n, rng)
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array([rng.field_elem(C.order) for _ in range(n)])
    lo, hi = distmod.shard_bounds(n, 0, 2)
    p0 = gpu.VariableBaseMSM.multi_scalar_mul(curve, b[lo:hi], s[lo:hi])
    p1 = gpu.VariableBaseMSM.multi_scalar_mul(curve, b[hi:], s[hi:])
    cd = distmod.CDist(gpu, 0, 2, transport="callback", allgather=lambda data: data + p1.tobytes())
    try:
        total = cd.allgather_fold(curve, p0)
    finally:
        cd.shutdown()
    exp = S.oracle_affine(curve, S.oracle_msm(curve, b, inf, s, 8))
    got = gpu.proj_to_affine(curve, total)
    assert got[1] == exp[1] and (got[0] == exp[0]).all()
