"""CPU emulation of the affine-round bucket accumulation: the same __host__ __device__ lane bodies the GPU
kernels run (ginger-lib_amd/csrc/aff_kernels.h), driven lane by lane by tests/host_shim/aff_shim.hip, against
textbook affine sums (pyref).  Covers what the reference's bucket loop covers through add_assign_mixed
(algebra/src/curves/models/short_weierstrass_projective.rs:481-519): P + Q, P + P (doubling branch :492-495),
P + (-P), sums passing through infinity, plus the list shapes of the rounds (odd sizes, empty buckets, one huge
bucket, tails shorter than a wave, more lanes than work)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import pyref
import support as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "build", "libaff_shim.so")
SRC = os.path.join(ROOT, "tests", "host_shim", "aff_shim.hip")


@pytest.fixture(scope="module")
def shim():
    deps = [SRC] + [os.path.join(ROOT, "ginger-lib_amd", "csrc", f) for f in ("aff_kernels.h", "ec29.h", "fp29.h")]
    if not os.path.exists(SHIM) or any(os.path.getmtime(d) > os.path.getmtime(SHIM) for d in deps):
        os.makedirs(os.path.dirname(SHIM), exist_ok=True)
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "--offload-host-only", "-shared", "-fPIC", "-Wno-unused-result",
                               "-o", SHIM, SRC])
    lib = ctypes.CDLL(SHIM)
    vp, sz, ci, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
    lib.aff_tree_host.argtypes = [ci, vp, sz, vp, sz, vp, vp, u32, ci, u32, u32, vp, vp]
    return lib


def run_case(shim, curve, buckets, pts, R, waves, bmin):
    """buckets: list of lists of (point index, negate); returns per-bucket affine sums from the emulation"""
    C = pyref.CURVES[curve]
    bases, _ = S.bases_array(C, pts)
    counts = np.array([len(b) for b in buckets], dtype=np.uint32)
    starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
    entries = np.array([i | (0x80000000 if neg else 0) for b in buckets for (i, neg) in b] or [0], dtype=np.uint32)
    n_entries = int(counts.sum())
    out = np.zeros((len(buckets), 36), dtype=np.uint64)
    marks = ctypes.c_uint32()
    rc = shim.aff_tree_host(S.CURVE_ID[curve], S.ptr(bases), len(pts), S.ptr(entries), n_entries, S.ptr(starts), S.ptr(counts),
                            len(buckets), R, waves, bmin, S.ptr(out), ctypes.byref(marks))
    assert rc == 0
    got = [S.affine_of_xyz(C, out[b]) for b in range(len(buckets))]
    exp = []
    for b in buckets:
        acc = None
        for (i, neg) in b:
            acc = C.add(acc, C.neg(pts[i]) if neg else pts[i])
        exp.append(acc)
    return got, exp, marks.value


@pytest.mark.parametrize("curve", ["mnt4753_g1", "mnt6753_g1"])
def test_affine_rounds_group_law_cases(shim, curve):
    C = pyref.CURVES[curve]
    rng = pyref.Rng(11)
    pts = S.chain_points(C, 12, rng)
    P, Q, T = 0, 1, 2
    buckets = [
        [],                                             # empty
        [(P, 0)],                                       # single
        [(P, 0), (Q, 0)],                               # generic
        [(P, 0), (P, 0)],                               # doubling
        [(P, 0), (P, 1)],                               # cancellation -> infinity
        [(P, 0), (P, 1), (Q, 0)],                       # marker + single
        [(P, 0), (P, 1), (Q, 0), (T, 1)],               # marker + sum
        [(P, 0), (P, 1), (Q, 0), (Q, 1)],               # marker + marker
        [(P, 0), (P, 0), (P, 0), (P, 0)],               # 2P + 2P -> doubling in round 1
        [(P, 1), (P, 1), (P, 0), (P, 0)],               # -2P + 2P -> cancellation in round 1
        [(P, 0), (Q, 0), (Q, 0), (P, 0)],               # (P+Q) + (Q+P): equal sums met in round 1
        [(i % 12, i % 3 == 0) for i in range(7)],       # odd size
        [(i % 12, 0) for i in range(12)] * 3 + [(3, 1)],
        [],
        [(5, 1)],
    ]
    for R in (1, 2, 3, 6):
        for waves, bmin in ((1, 1), (1, 4), (2, 1)):
            got, exp, marks = run_case(shim, curve, buckets, pts, R, waves, bmin)
            assert got == exp, (R, waves, bmin)
            assert marks >= 3


def test_affine_rounds_random_shapes(shim):
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    rng = pyref.Rng(5)
    pts = S.chain_points(C, 40, rng)
    nrng = np.random.default_rng(3)
    sizes = list(nrng.integers(0, 9, size=150)) + [300, 1, 0, 65, 64, 63]
    buckets = [[(int(nrng.integers(0, 40)), bool(nrng.integers(0, 2))) for _ in range(int(s))] for s in sizes]
    for R, waves, bmin in ((3, 1, 2), (5, 1, 1), (9, 2, 3)):
        got, exp, _ = run_case(shim, curve, buckets, pts, R, waves, bmin)
        assert got == exp, (R, waves, bmin)
