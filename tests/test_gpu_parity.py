"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs, against the committed golden fixtures, and -- at BASELINE.json's full sizes -- through
size-independent properties (round trips, linearity, window-size invariance).
Bit-exact: FFT outputs limb for limb; MSM after into_affine()."""
import json
import os

import numpy as np
import pytest

import pyref
import support as S

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
MSM_G = json.load(open(os.path.join(G, "msm_golden.json")))
NTT_G = json.load(open(os.path.join(G, "ntt_golden.json")))
FLAGS = (("fft", 0), ("ifft", 1), ("coset_fft", 2), ("coset_ifft", 3))


def affine_eq(gl, curve, got_xyz, exp_xyz):
    g_xy, g_inf = gl.proj_to_affine(curve, got_xyz)
    e_xy, e_inf = S.oracle_affine(curve, exp_xyz)
    return g_inf == e_inf and bool((g_xy == e_xy).all())


# ------------------------------------------------------------------------------ NTT
@pytest.mark.parametrize("name", list(NTT_G))
def test_ntt_golden(gpu, name):
    case = NTT_G[name]
    F = S.FIELD_OF[case["field"]]
    a = S.fe_array(F, [int(x, 16) for x in case["input"]])
    dom = gpu.EvaluationDomain(case["field"], 1 << case["log_n"])
    for nm, _ in FLAGS:
        assert S.fe_list(F, getattr(dom, nm)(a)) == [int(x, 16) for x in case[nm]], (name, nm)


@pytest.mark.parametrize("field,log_n", [("mnt4753_fr", l) for l in (0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 13, 16)] +
                         [("mnt6753_fr", l) for l in (1, 6, 12, 14)])
def test_ntt_vs_oracle(gpu, field, log_n):
    n = 1 << log_n
    F = S.FIELD_OF[field]
    for n_in in sorted({n, max(1, n - 3), n + 5}):          # exact, zero-padded, truncated (domain.rs:121)
        a = S.random_scalars_np(n_in, seed=log_n * 7 + n_in % 5, below=F.p)
        dom = gpu.EvaluationDomain(field, n)
        assert dom.size == n
        for nm, flags in FLAGS:
            got = getattr(dom, nm)(a).reshape(-1, 12)
            exp = S.oracle_fft(field, a, log_n, flags, 16)
            assert (got == exp).all(), (field, log_n, n_in, nm)


def test_domain_limits(gpu):
    assert gpu.EvaluationDomain.new("mnt4753_fr", 1 << 30) is None          # 2-adicity 30 (domain.rs:69-71)
    assert gpu.EvaluationDomain.new("mnt6753_fr", (1 << 14) + 1) is None    # 2-adicity 15
    assert gpu.EvaluationDomain.new("mnt6753_fr", 1 << 14).size == 1 << 14
    assert gpu.EvaluationDomain.new("mnt4753_fr", 0).size == 1


def test_ntt_full_size_roundtrips(gpu):
    """BASELINE config 2 size (2^20): ifft(fft(x)) = x and coset_ifft(coset_fft(x)) = x, device resident."""
    log_n = 20
    n = 1 << log_n
    a = S.random_scalars_np(n, seed=3, below=pyref.P6.p)
    dom = gpu.EvaluationDomain("mnt4753_fr", n)
    buf = gpu.DeviceBuffer(n * 96).upload(a)
    for fwd, inv in ((0, 1), (2, 3)):
        dom.fft_dev(buf, fwd)
        mid = buf.download().reshape(n, 12)
        assert not (mid == a).all()
        dom.fft_dev(buf, inv)
        assert (buf.download().reshape(n, 12) == a).all()
    buf.free()
    # spot-check the forward transform itself against the oracle at 2^18
    log_n = 18
    n = 1 << log_n
    a = S.random_scalars_np(n, seed=4, below=pyref.P6.p)
    got = gpu.EvaluationDomain("mnt4753_fr", n).coset_fft(a).reshape(n, 12)
    assert (got == S.oracle_fft("mnt4753_fr", a, log_n, 2, 16)).all()



def test_assembly_pass_and_cpp_pass_give_the_same_vectors(gpu):
    """The generated NTT pass (asmgen/ntt_pass.py) and the hipcc pass (ntt_kernels.h) are interchangeable: a child process with
    GH_NTT_ASM=0 runs the same four transforms on the same seeded inputs; the vectors must be equal bit for bit (2^13: passes of 7 + 6
    stages; 2^17: 6 + 6 + 5, a pass of each kind inside ONE transform; 2^20: the prover's domain)."""
    import subprocess
    import sys
    import tempfile
    import zlib
    script = r"""
import os, sys, zlib, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, pyref, support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg(); gl.init()
out = {}
for log_n in (13, 17, 20):
    n = 1 << log_n
    a = S.random_scalars_np(n, seed=40 + log_n, below=pyref.P6.p)
    dom = gl.EvaluationDomain("mnt4753_fr", n)
    buf = gl.DeviceBuffer(n * 96).upload(a)
    for fl in (0, 2, 3, 1):
        dom.fft_dev(buf, fl)
        out["%%d_%%d" %% (log_n, fl)] = zlib.crc32(buf.download().tobytes())
    buf.free()
print("CRC " + json.dumps(out, sort_keys=True))
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GH_NTT_ASM="0")
    p = subprocess.run([sys.executable, "-c", script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("CRC ")][-1]
    ref = json.loads(line[4:])
    for log_n in (13, 17, 20):
        n = 1 << log_n
        a = S.random_scalars_np(n, seed=40 + log_n, below=pyref.P6.p)
        dom = gpu.EvaluationDomain("mnt4753_fr", n)
        buf = gpu.DeviceBuffer(n * 96).upload(a)
        for fl in (0, 2, 3, 1):
            dom.fft_dev(buf, fl)
            assert zlib.crc32(buf.download().tobytes()) == ref["%d_%d" % (log_n, fl)], (log_n, fl)
        buf.free()


def test_vec_ops(gpu):
    F = pyref.P6
    n = 1000
    a = S.random_scalars_np(n, seed=8, below=F.p)
    b = S.random_scalars_np(n, seed=9, below=F.p)
    dom = gpu.EvaluationDomain("mnt4753_fr", 1024)
    got = dom.mul_polynomials_in_evaluation_domain(a, b).reshape(n, 12)
    exp = a.copy()
    S.oracle().oracle_vec_mul(0, S.ptr(exp), S.ptr(b), n)
    assert (got == exp).all()
    # divide_by_vanishing_poly_on_coset: multiply by (g^N - 1)^-1  (domain.rs:245-256)
    inv = np.zeros(12, dtype=np.uint64)
    S.oracle().oracle_vanishing_inv_on_coset(0, 10, S.ptr(inv))
    got = gpu.vec_scale("mnt4753_fr", a, inv).reshape(n, 12)
    bb = np.tile(inv, (n, 1))
    exp = a.copy()
    S.oracle().oracle_vec_mul(0, S.ptr(exp), S.ptr(bb), n)
    assert (got == exp).all()


@pytest.mark.parametrize("field,log_n", [("mnt4753_fr", 0), ("mnt4753_fr", 1), ("mnt4753_fr", 10), ("mnt4753_fr", 15), ("mnt6753_fr", 9)])
def test_witness_map_vs_oracle(gpu, field, log_n):
    """R1CStoQAP::witness_map transform pipeline (r1cs_to_qap.rs:121-166), arbitrary rows and d1 d2 d3"""
    F = S.FIELD_OF[field]
    n = 1 << log_n
    a, b, c = (S.random_scalars_np(n, seed=s0 + log_n, below=F.p) for s0 in (100, 200, 300))
    for ds in (([0] * 3), None):
        if ds is None:
            d = S.random_scalars_np(3, seed=9, below=F.p)
        else:
            d = np.zeros((3, 12), dtype=np.uint64)
        got = gpu.witness_map(field, a, b, c, d[0], d[1], d[2]).reshape(n + 1, 12)
        exp = S.oracle_witness_map(field, a, b, c, d[0], d[1], d[2], 16)
        assert (got == exp).all()


@pytest.mark.parametrize("field,log_n", [("mnt4753_fr", 0), ("mnt4753_fr", 1), ("mnt4753_fr", 11), ("mnt4753_fr", 16), ("mnt6753_fr", 8)])
def test_sap_witness_map_vs_oracle(gpu, field, log_n):
    """R1CStoSAP::witness_map transform pipeline (gm17/r1cs_to_sap.rs:191-240), arbitrary rows and d1 d2"""
    F = S.FIELD_OF[field]
    n = 1 << log_n
    a, c = (S.random_scalars_np(n, seed=s0 + log_n, below=F.p) for s0 in (400, 500))
    for zero_d in (True, False):
        d = np.zeros((2, 12), dtype=np.uint64) if zero_d else S.random_scalars_np(2, seed=19, below=F.p)
        got = gpu.sap_witness_map(field, a, c, d[0], d[1]).reshape(n + 1, 12)
        exp = S.oracle_sap_witness_map(field, a, c, d[0], d[1], 16)
        assert (got == exp).all()


@pytest.mark.parametrize("field", ["mnt4753_fr", "mnt6753_fr"])
def test_batch_inversion_and_lagrange_vs_oracle(gpu, field):
    """batch_inversion (fields/mod.rs:412-442: zeros skipped; sizes around the per-thread run of 32) and
    evaluate_all_lagrange_coefficients (domain.rs:183-219: tau outside the domain, tau = a domain element, sizes 1 .. 2^12)"""
    F = S.FIELD_OF[field]
    for n in (1, 2, 31, 32, 33, 1000, 5000):
        a = S.random_scalars_np(n, seed=600 + n, below=F.p)
        a[::7] = 0
        assert (gpu.batch_inversion(field, a).reshape(n, 12) == S.oracle_batch_inversion(field, a)).all(), n
    a = S.fe_array(F, [1, F.p - 1, 2])
    assert S.fe_list(F, gpu.batch_inversion(field, a).reshape(3, 12)) == [1, F.p - 1, pow(2, -1, F.p)]
    tau = S.random_scalars_np(1, seed=77, below=F.p)[0]
    for log_n in (0, 1, 5, 12):
        assert (gpu.evaluate_all_lagrange_coefficients(field, log_n, tau) == S.oracle_lagrange(field, log_n, tau)).all(), log_n
    w = pyref.domain_params(F, 6)
    inside = S.fe_array(F, [pow(w, 37, F.p)])[0]
    got = gpu.evaluate_all_lagrange_coefficients(field, 6, inside)
    assert (got == S.oracle_lagrange(field, 6, inside)).all() and S.fe_list(F, got) == [1 if i == 37 else 0 for i in range(64)]


# ------------------------------------------------------------------------------ MSM
def load_msm_case(name):
    case = MSM_G[name]
    curve = name.replace("_zero_sum", "")
    C = pyref.CURVES[curve]
    pts = [None if b is None else (tuple(int(c, 16) for c in b[0]), tuple(int(c, 16) for c in b[1])) for b in case["bases"]]
    scal = [int(s, 16) for s in case["scalars"]]
    e = case["expected_affine"]
    exp = None if e is None else (tuple(int(c, 16) for c in e[0]), tuple(int(c, 16) for c in e[1]))
    return curve, C, pts, scal, exp


@pytest.mark.parametrize("name", list(MSM_G))
def test_msm_golden(gpu, name):
    curve, C, pts, scal, exp = load_msm_case(name)
    b, inf = S.bases_array(C, pts)
    out = gpu.VariableBaseMSM.multi_scalar_mul(curve, b, S.scalar_array(scal), inf)
    assert S.affine_of_xyz(C, out) == exp
    xy, is_inf = gpu.proj_to_affine(curve, out)
    assert is_inf == (exp is None)


@pytest.mark.parametrize("curve,sizes", [("mnt4753_g1", (0, 1, 2, 31, 32, 33, 200, 3000)), ("mnt6753_g1", (1, 40, 1000)),
                                         ("mnt4753_g2", (1, 35, 300)), ("mnt6753_g2", (2, 33, 200))])
def test_msm_vs_oracle(gpu, curve, sizes):
    C = pyref.CURVES[curve]
    r = C.order
    rng = pyref.Rng(sum(map(ord, curve)))
    base_pool = S.chain_points(C, min(max(sizes), 512), rng)
    for n in sizes:
        pts = [base_pool[i % len(base_pool)] for i in range(n)]
        scal = [rng.field_elem(r) for _ in range(n)]
        if n >= 31:
            scal[0], scal[1], scal[2], scal[3] = 0, 1, r - 1, 1 << 300
            scal[4] = scal[5]
            pts[6] = None
            pts[8] = pts[7]; scal[8] = scal[7]
            pts[10] = C.neg(pts[9]); scal[10] = scal[9]
        b, inf = S.bases_array(C, pts)
        s = S.scalar_array(scal)
        got = gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s, inf)
        exp = S.oracle_msm(curve, b, inf, s, 16)
        assert affine_eq(gpu, curve, got, exp), (curve, n)
        if n == 0:
            k = C.deg      # empty input -> (0, 1, 0)  (variable_base.rs + swp.rs:372-378)
            assert pyref.ext_from_abi(C.F, [int(v) for v in got[12 * k:24 * k]], k) == C.E.one()
            assert not got[24 * k:].any()


@pytest.mark.parametrize("curve", ["mnt4753_g1", "mnt6753_g1"])
def test_msm_window_sizes_vs_oracle(gpu, curve):
    """every digit-extraction regime: c | 752 (unsigned two-region top window: 4, 8, 16), leftover top
    windows of different fill (5, 13, 15, 18), scalars around r/2 (sign folding) and the extremes"""
    C = pyref.CURVES[curve]
    r = C.order
    rng = pyref.Rng(31)
    n = 96
    pts = S.chain_points(C, n, rng)
    scal = [rng.field_elem(r) for _ in range(n)]
    scal[:10] = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, (r + 3) // 2, 1 << 751, (1 << 752) + 12345]
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    exp = S.oracle_msm(curve, b, inf, s, 16)
    try:
        for c in (4, 5, 8, 13, 15, 16, 18):
            gpu.msm_set_window(c)
            got = gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s, inf)
            assert gpu.msm_last_timing()["window_bits"] == c
            assert affine_eq(gpu, curve, got, exp), (curve, c)
    finally:
        gpu.msm_set_window(0)


def test_msm_unequal_lengths_and_resident(gpu):
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    rng = pyref.Rng(77)
    pts = S.chain_points(C, 120, rng)
    scal = [rng.field_elem(C.order) for _ in range(150)]
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    exp = S.oracle_msm(curve, b, inf, s, 8)                     # zip truncates to 120 (variable_base.rs:36)
    assert affine_eq(gpu, curve, gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s, inf), exp)
    exp2 = S.oracle_msm(curve, b, inf, s[:70], 8)
    assert affine_eq(gpu, curve, gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s[:70], inf), exp2)
    rb = gpu.ResidentBases(curve, b, inf)
    assert affine_eq(gpu, curve, rb.msm(s), exp)
    assert affine_eq(gpu, curve, rb.msm(s[:70]), exp2)
    ds = gpu.DeviceBuffer(s.nbytes).upload(s)
    assert affine_eq(gpu, curve, rb.msm_dev(ds, 150), exp)
    ds.free()
    rb.free()


def test_msm_skewed_scalars(gpu):
    """Witness-like scalars: mostly 0 / 1 / small / repeated values -> a few very long buckets
    (wave-cooperative heavy-bucket path) and the scalar == 1 shortcut of variable_base.rs:37-41."""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    rng = pyref.Rng(5)
    pool = S.chain_points(C, 256, rng)
    n = 6000
    pts = [pool[(i * 7) % 256] for i in range(n)]
    big = rng.field_elem(C.order)
    scal = []
    for i in range(n):
        m = i % 10
        scal.append(1 if m < 5 else 0 if m == 5 else 2 if m == 6 else big if m == 7 else (i * 12345) % 65536 if m == 8 else rng.field_elem(C.order))
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    got = gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s, inf)
    assert affine_eq(gpu, curve, got, S.oracle_msm(curve, b, inf, s, 16))


def test_msm_full_size_properties(gpu, no_dedup):
    """BASELINE config 3 size (2^20 pairs): linearity msm(s) + msm(t) == msm(s + t mod r) and
    invariance of the affine result under the window size; spot check vs the oracle at 2^14."""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    r = C.order
    rng = pyref.Rng(2020)
    pool = S.chain_points(C, 4096, rng)
    pb, _ = S.bases_array(C, pool)
    n = 1 << 20
    bases = np.tile(pb, (n // 4096, 1))
    s = S.random_scalars_np(n, seed=11, below=r)
    t = S.random_scalars_np(n, seed=12, below=r)
    # (s + t) mod r, vectorised on Python ints per row would be slow: do limb arithmetic with object ints in chunks
    def add_mod(a, b):
        out = np.empty_like(a)
        av = a.astype(object)
        bv = b.astype(object)
        for i in range(0, len(a), 1 << 16):
            xs = [(pyref.limbs_to_int(x) + pyref.limbs_to_int(y)) % r for x, y in zip(av[i:i + (1 << 16)], bv[i:i + (1 << 16)])]
            out[i:i + (1 << 16)] = np.array([pyref.int_to_limbs(x) for x in xs], dtype=np.uint64)
        return out
    m = 1 << 17                                                    # the ints loop is the slow part: use a 2^17 slice for s+t
    st = add_mod(s[:m], t[:m])
    rb = gpu.ResidentBases(curve, bases)
    full_s = rb.msm(s)                                             # full 2^20 run (also timing sanity)
    tm = gpu.msm_last_timing()
    assert tm["accumulate_madds"] > 0.9 * n * tm["num_windows"] * 0.9
    ms, mt, mst = rb.msm(s[:m]), rb.msm(t[:m]), rb.msm(st)
    lhs = gpu.proj_add(curve, ms, mt)
    a1, i1 = gpu.proj_to_affine(curve, lhs)
    a2, i2 = gpu.proj_to_affine(curve, mst)
    assert i1 == i2 and (a1 == a2).all()
    # window-size invariance on the full input
    ref_xy, ref_inf = gpu.proj_to_affine(curve, full_s)
    for c in (13, 18):
        gpu.msm_set_window(c)
        xy, inf = gpu.proj_to_affine(curve, rb.msm(s))
        assert gpu.msm_last_timing()["window_bits"] == c
        assert inf == ref_inf and (xy == ref_xy).all()
    gpu.msm_set_window(0)
    # oracle spot check at 2^14 pairs
    k = 1 << 14
    exp = S.oracle_msm(curve, bases[:k], None, s[:k], 16)
    assert affine_eq(gpu, curve, rb.msm(s[:k]), exp)
    # the same key with its shift table (one bucket set, c = 21) and as a pipelined batch: same affine results
    assert rb.precompute(0) == 21
    xy, inf = gpu.proj_to_affine(curve, rb.msm(s))
    assert gpu.msm_last_timing()["num_windows"] == 36
    assert inf == ref_inf and (xy == ref_xy).all()
    ds, dt = gpu.DeviceBuffer(s.nbytes).upload(s), gpu.DeviceBuffer(t.nbytes).upload(t)
    outs = gpu.msm_batch_dev([(rb, ds, n), (rb, dt, m), (rb, ds, m), (rb, ds, n)])
    for o in (outs[0], outs[3]):
        xy, inf = gpu.proj_to_affine(curve, o)
        assert inf == ref_inf and (xy == ref_xy).all()
    for o, r0 in ((outs[1], mt), (outs[2], ms)):
        a1, i1 = gpu.proj_to_affine(curve, o)
        a2, i2 = gpu.proj_to_affine(curve, r0)
        assert i1 == i2 and (a1 == a2).all()
    assert affine_eq(gpu, curve, rb.msm(s[:k]), exp)
    ds.free(); dt.free()
    rb.free()


# ------------------------------------------------------------------------------ MSM on a precomputed shift table
@pytest.mark.parametrize("curve,n,windows", [("mnt4753_g1", 300, (0, 4, 7, 13, 16, 17, 19)), ("mnt6753_g1", 200, (0, 11, 18)),
                                             ("mnt4753_g2", 90, (0, 9, 17)), ("mnt6753_g2", 60, (0, 12))])
def test_msm_precomputed_vs_oracle(gpu, no_dedup, curve, n, windows):
    """gh_bases_precompute: every window files into one bucket set (table row w = 2^(c w) P).  Same
    affine result as the oracle for: c | 752 (carry-only top window), one and several pseudo-windows
    of the reduction (c <= 16 / c >= 17), shorter scalar vectors, infinity / duplicate / opposite bases,
    scalars 0, 1, r - 1, around r/2."""
    C = pyref.CURVES[curve]
    r = C.order
    rng = pyref.Rng(4242 + n)
    pool = S.chain_points(C, min(n, 64), rng)
    pts = [pool[i % len(pool)] for i in range(n)]
    scal = [rng.field_elem(r) for _ in range(n)]
    scal[:10] = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, (r + 3) // 2, 1 << 751, (1 << 752) + 12345]
    pts[12] = None
    pts[14] = pts[13]; scal[14] = scal[13]
    pts[16] = C.neg(pts[15]); scal[16] = scal[15]
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    exp = S.oracle_msm(curve, b, inf, s, 16)
    exp_short = S.oracle_msm(curve, b, inf, s[:n // 3], 16)
    rb = gpu.ResidentBases(curve, b, inf)
    plain = rb.msm(s)
    assert affine_eq(gpu, curve, plain, exp)
    for c in windows:
        used = rb.precompute(c)
        assert used == (c if c else used) and used >= 2
        got = rb.msm(s)
        tm = gpu.msm_last_timing()
        assert tm["window_bits"] == used and tm["num_windows"] == 752 // used + 1
        assert affine_eq(gpu, curve, got, exp), (curve, c)
        assert affine_eq(gpu, curve, rb.msm(s[:n // 3]), exp_short), (curve, c, "short")
    rb.free()


@pytest.mark.parametrize("curve,n", [("mnt4753_g1", 400), ("mnt6753_g1", 150), ("mnt4753_g2", 80), ("mnt6753_g2", 50)])
def test_msm_partial_table_vs_oracle(gpu, no_dedup, curve, n):
    """gh_bases_precompute_rows: at most max_rows rows (row j = 2^(c G j) P, G = ceil(windows / max_rows) bucket sets; window
    w = j G + g reads row j and files into set g; the sets are folded with c doublings each).  Every (c, max_rows) below --
    one row (= the bases themselves: one bucket set per window, like the plain path), two rows, a row count that does not
    divide the windows, more rows than windows (= the full table) -- gives the oracle's affine sum, also on a shorter
    scalar vector and in a pipelined batch; the degenerate inputs are those of the full-table test."""
    C = pyref.CURVES[curve]
    r = C.order
    rng = pyref.Rng(777 + n)
    pool = S.chain_points(C, min(n, 48), rng)
    pts = [pool[i % len(pool)] for i in range(n)]
    scal = [rng.field_elem(r) for _ in range(n)]
    scal[:10] = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, (r + 1) // 2, (r + 3) // 2, 1 << 751, (1 << 752) + 12345]
    pts[12] = None
    pts[14] = pts[13]; scal[14] = scal[13]
    pts[16] = C.neg(pts[15]); scal[16] = scal[15]
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    exp = S.oracle_msm(curve, b, inf, s, 16)
    exp_short = S.oracle_msm(curve, b, inf, s[:n // 3], 16)
    rb = gpu.ResidentBases(curve, b, inf)
    ds = gpu.DeviceBuffer(s.nbytes).upload(s)
    try:
        for c, max_rows in ((13, 1), (13, 2), (13, 5), (16, 7), (17, 3), (9, 1000), (0, 4)):
            used = rb.precompute(c, max_rows)
            windows = 752 // used + 1
            sets = 1 if max_rows >= windows else -(-windows // max_rows)
            assert rb.table_rows() == -(-windows // sets), (c, max_rows)
            assert affine_eq(gpu, curve, rb.msm(s), exp), (curve, c, max_rows)
            assert gpu.msm_last_timing()["window_bits"] == used
            assert affine_eq(gpu, curve, rb.msm(s[:n // 3]), exp_short), (curve, c, max_rows, "short")
            outs = gpu.msm_batch_dev([(rb, ds, n), (rb, ds, n // 3), (rb, ds, n)])
            assert affine_eq(gpu, curve, outs[0], exp) and affine_eq(gpu, curve, outs[1], exp_short) and affine_eq(gpu, curve, outs[2], exp)
    finally:
        ds.free()
        rb.free()


def test_msm_precomputed_skewed_and_large(gpu, no_dedup):
    """witness-like scalars (long buckets -> chunk path) on the merged bucket set, and a 2^16-pair run
    against the oracle (BASELINE config 1 size) with the automatic table window"""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    rng = pyref.Rng(6)
    pool = S.chain_points(C, 256, rng)
    n = 6000
    pts = [pool[(i * 7) % 256] for i in range(n)]
    big = rng.field_elem(C.order)
    scal = []
    for i in range(n):
        m = i % 10
        scal.append(1 if m < 5 else 0 if m == 5 else 2 if m == 6 else big if m == 7 else (i * 12345) % 65536 if m == 8 else rng.field_elem(C.order))
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    rb = gpu.ResidentBases(curve, b, inf)
    rb.precompute(0)
    assert affine_eq(gpu, curve, rb.msm(s), S.oracle_msm(curve, b, inf, s, 16))
    rb.free()
    n = 1 << 16
    pb, _ = S.bases_array(C, pool)
    bases = np.tile(pb, (n // 256, 1))
    s = S.random_scalars_np(n, seed=21, below=C.order)
    rb = gpu.ResidentBases(curve, bases)
    c = rb.precompute(0)
    got = rb.msm(s)
    assert gpu.msm_last_timing()["window_bits"] == c
    assert affine_eq(gpu, curve, got, S.oracle_msm(curve, bases, None, s, 16))
    # the override of the window size switches a table-carrying key back to the per-window path
    gpu.msm_set_window(13)
    assert affine_eq(gpu, curve, rb.msm(s), S.oracle_msm(curve, bases, None, s, 16))
    assert gpu.msm_last_timing()["num_windows"] == 58
    gpu.msm_set_window(0)
    rb.free()


def test_msm_batch_pipelined(gpu):
    """gh_msm_resident_dev_batch: several MSMs pipelined over streams (two buffer slots) give exactly
    the results of the same MSMs issued one by one -- different keys, lengths (incl. empty), with and
    without a shift table, more jobs than slots."""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    rng = pyref.Rng(99)
    pool = S.chain_points(C, 128, rng)
    keys, scal = [], []
    for i, n in enumerate((500, 128, 1000, 64, 777, 300, 2000)):
        pts = [pool[(j * (i + 3)) % 128] for j in range(n)]
        b, inf = S.bases_array(C, pts)
        rb = gpu.ResidentBases(curve, b, inf)
        if i % 3 != 1:
            rb.precompute(0 if i % 2 else 13)
        s = S.scalar_array([rng.field_elem(C.order) for _ in range(n)])
        keys.append((rb, b, inf))
        scal.append(s)
    jobs, expected = [], []
    bufs = []
    for i, ((rb, b, inf), s) in enumerate(zip(keys, scal)):
        m = 0 if i == 3 else (len(s) if i % 2 == 0 else len(s) // 2)
        d = gpu.DeviceBuffer(max(96, s.nbytes)).upload(s)
        bufs.append(d)
        jobs.append((rb, d, m))
        expected.append(S.oracle_msm(curve, b, inf, s[:m], 8))
    jobs.append(jobs[0]); expected.append(expected[0])          # a key may repeat
    got = gpu.msm_batch_dev(jobs)
    assert len(got) == len(jobs)
    assert gpu.msm_batch_timing(len(jobs) - 1)["window_bits"] > 0 and gpu.msm_batch_timing(3)["window_bits"] == 0   # job 3 is empty
    for i, (g_xyz, e_xyz) in enumerate(zip(got, expected)):
        assert affine_eq(gpu, curve, g_xyz, e_xyz), i
    one_by_one = [rb.msm_dev(d, m) for rb, d, m in jobs]         # (projective triples differ run to run: list order
    for a, b2 in zip(got, one_by_one):                           #  inside a bucket comes from atomics)
        xa, ia = gpu.proj_to_affine(curve, a)
        xb, ib = gpu.proj_to_affine(curve, b2)
        assert ia == ib and (xa == xb).all()
    for d in bufs:
        d.free()
    for rb, _, _ in keys:
        rb.free()


def test_msm_batch_with_growing_jobs_in_a_reused_slot(gpu):
    """A pipelined batch whose jobs GROW in the position that re-uses a buffer slot (job k + 2 after job k: the GM17 prover's
    batch mixes key sizes): the pool replaces the slot's buffers while the job before may still be reducing out of them --
    only after the device has drained (msm_impl.h issue_sort(k + 1) before finish(k - 1)).  Every sum of the batch
    must equal the same MSM run alone."""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    sizes = [1 << 12, 1 << 12, 1 << 15, (1 << 15) + 777, 1 << 17, 1 << 13, (1 << 17) + 5]
    nmax = max(sizes)
    rb = gpu.ResidentBases.chain(curve, *S.bases_array(C, S.chain_points(C, 2, pyref.Rng(77)))[0], nmax)
    keys = []
    for i, n in enumerate(sizes):
        pts = rb.download(0, n)
        k = gpu.ResidentBases(curve, pts)
        if i % 2 == 0:
            k.precompute(0)
        s = S.random_scalars_np(n, seed=500 + i, below=C.order)
        keys.append((k, gpu.DeviceBuffer(n * 96).upload(s), n))
    try:
        gpu.dev_trim()                                               # empty pools: every larger job has to grow its slot
        batch = gpu.msm_batch_dev(keys)
        for (k, d, n), got in zip(keys, batch):
            a, b = gpu.proj_to_affine(curve, got), gpu.proj_to_affine(curve, k.msm_dev(d, n))
            assert a[1] == b[1] and (a[0] == b[0]).all(), n
    finally:
        for k, d, _ in keys:
            k.free()
            d.free()
        rb.free()


@pytest.mark.parametrize("curve,n,windows", [("mnt4753_g1", 3000, (0, 9, 13)), ("mnt6753_g1", 700, (0, 12)),
                                             ("mnt4753_g2", 640, (0, 9)), ("mnt6753_g2", 620, (0, 8))])
def test_msm_affine_bucket_sums_vs_oracle(gpu, no_dedup, curve, n, windows):
    """gh_msm_set_affine(1): bucket sums by affine rounds (aff_kernels.h: pairwise rounds over the flat list, batched
    safegcd inversion).  Duplicate and opposite bases in one bucket (doubling / cancellation handled in the round),
    sums that pass through infinity, buckets of every length (c = 9: hundreds of entries per bucket; many equal small
    scalars: thousands -> extra rounds), zero / one / r - 1 scalars, an infinity base; on a shift table and on the
    per-window path; same affine result as the oracle and as the projective kernel."""
    C = pyref.CURVES[curve]
    r = C.order
    rng = pyref.Rng(777 + n)
    pool = S.chain_points(C, 48, rng)
    pts = [pool[(i * 5) % 48] for i in range(n)]
    scal = [rng.field_elem(r) for _ in range(n)]
    scal[:6] = [0, 1, r - 1, 2, (r - 1) // 2, (r + 1) // 2]
    for i in range(10, 200, 7):            # equal pairs (P + P inside a bucket) and opposite pairs (P - P)
        pts[i + 1] = pts[i]; scal[i + 1] = scal[i]
        pts[i + 3] = C.neg(pts[i + 2]); scal[i + 3] = scal[i + 2]
    for i in range(300, 600):              # many equal small scalars: long buckets of window 0
        scal[i] = 3
    pts[7] = None
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    exp = S.oracle_msm(curve, b, inf, s, 16)
    exp_short = S.oracle_msm(curve, b, inf, s[:n // 2], 16)
    rb = gpu.ResidentBases(curve, b, inf)
    try:
        for mode in (1, 0):
            gpu.msm_set_affine(mode)
            for c in (0, 5, 13):          # per-window path (no table yet), automatic and forced window sizes
                gpu.msm_set_window(c)
                assert affine_eq(gpu, curve, rb.msm(s), exp), (curve, "per-window", c, mode)
                assert affine_eq(gpu, curve, gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s[:n // 2], inf), exp_short), (curve, c, mode)
            gpu.msm_set_window(0)
        for c in windows:
            rb.precompute(c)
            for mode in (1, 0):
                gpu.msm_set_affine(mode)
                assert affine_eq(gpu, curve, rb.msm(s), exp), (curve, c, "affine mode", mode)
                assert affine_eq(gpu, curve, rb.msm(s[:n // 2]), exp_short), (curve, c, "short", mode)
    finally:
        gpu.msm_set_affine(2)
        gpu.msm_set_window(0)
        rb.free()


@pytest.mark.parametrize("curve", ["mnt4753_g1", "mnt6753_g1", "mnt4753_g2", "mnt6753_g2"])
def test_msm_affine_degenerate_inputs(gpu, no_dedup, curve):
    """affine rounds on inputs made of the group law's special cases only: all bases equal (every addition of every
    round is a doubling), bases in opposite pairs with equal scalars (every bucket cancels to infinity), a single pair,
    all scalars zero"""
    C = pyref.CURVES[curve]
    r = C.order
    rng = pyref.Rng(91)
    P, Q = S.chain_points(C, 2, rng)
    k = rng.field_elem(r)
    cases = [([P] * 257, [k] * 257), ([P, C.neg(P)] * 100, [k] * 200), ([P, C.neg(P)] * 64 + [Q], [k] * 128 + [5]),
             ([P], [k]), ([P, Q, P], [0, 0, 0]), ([P] * 70 + [C.neg(P)] * 70, list(range(1, 71)) * 2)]
    gpu.msm_set_affine(1)
    try:
        for pts, scal in cases:
            b, inf = S.bases_array(C, pts)
            s = S.scalar_array(scal)
            exp = S.oracle_msm(curve, b, inf, s, 8)
            assert affine_eq(gpu, curve, gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s, inf), exp), len(pts)
            rb = gpu.ResidentBases(curve, b, inf)
            rb.precompute(7)
            assert affine_eq(gpu, curve, rb.msm(s), exp), (len(pts), "table")
            rb.free()
    finally:
        gpu.msm_set_affine(2)


@pytest.mark.parametrize("curve", ["mnt4753_g1", "mnt4753_g2", "mnt6753_g2"])
def test_msm_adds_up_the_scalars_of_equal_bases(gpu, curve):
    """A key with a shift table knows its equal bases (msm_impl.h dedup_bases: hashed on the device, grouped on the host,
    verified limb for limb) and adds their scalars up before the MSM: sum s_i P = (sum s_i) P, opposite bases with the
    opposite sign.  One large group (every third base the same point, as a proving key's b_query has for the variables of a
    closing constraint), pairs of equal and of opposite bases, an infinity base, scalars 0 / 1 / r - 1 inside groups, fewer
    scalars than bases -- against the oracle's plain multi_scalar_mul (variable_base.rs:10-83), table path and batch."""
    C = pyref.CURVES[curve]
    r = C.order
    n = 3000
    rng = pyref.Rng(88)
    pool = S.chain_points(C, n, rng)
    pts = list(pool)
    for i in range(0, n, 3):
        pts[i] = pool[0]                                   # one big group
    for i in range(1, 600, 6):
        pts[i + 3] = pts[i]                                # pairs P, P
        pts[i + 4] = C.neg(pts[i + 1])                     # pairs Q, -Q (i + 1 is not a multiple of 3 here: i = 1 mod 6)
    pts[7] = None
    scal = [rng.field_elem(r) for _ in range(n)]
    scal[0], scal[3], scal[6], scal[9] = 0, 1, r - 1, (r + 1) // 2
    scal[1], scal[4] = r - 5, 5                            # a pair whose scalars cancel
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    rb = gpu.ResidentBases(curve, b, inf)
    try:
        rb.precompute(7)
        for m in (n, n - 1000, 5):
            exp = S.oracle_msm(curve, b[:m], inf[:m], s[:m], 8)
            assert affine_eq(gpu, curve, rb.msm(s[:m]), exp), m
        d = gpu.DeviceBuffer(n * 96).upload(s)
        outs = gpu.msm_batch_dev([(rb, d, n), (rb, d, n - 1000), (rb, d, n)])
        assert affine_eq(gpu, curve, outs[0], S.oracle_msm(curve, b, inf, s, 8)) and affine_eq(gpu, curve, outs[2], S.oracle_msm(curve, b, inf, s, 8))
        assert affine_eq(gpu, curve, outs[1], S.oracle_msm(curve, b[:n - 1000], inf[:n - 1000], s[:n - 1000], 8))
        d.free()
    finally:
        rb.free()


@pytest.mark.parametrize("curve", ["mnt4753_g2", "mnt6753_g2"])
def test_g2_key_of_duplicate_bases_leaves_the_assembly_rounds(gpu, curve):
    """A proving key's b_g2_query holds equal points wherever two variables have the same polynomial, and an assignment with
    equal values sends both to the same bucket in every window: every such pair is a doubling (the reference's P == Q branch,
    swp.rs:492).  The assembly rounds list such elements for aff_fix_kernel; when a round lists more than the list holds the
    round is redone by the C++ kernel (which doubles inline) and the key stays on it for later MSMs (BasesBase::aff_asm_off).
    Both calls must give sum s_i P_i, checked as twice the MSM over one copy of the bases (itself on the tested paths), and
    an interleaved key with mostly distinct bases must still agree with the oracle-tested per-window path."""
    C = pyref.CURVES[curve]
    half = 1 << 15
    rb_half = gpu.ResidentBases.chain(curve, *S.bases_array(C, S.chain_points(C, 2, pyref.Rng(61)))[0], half)
    pts = rb_half.download(0, half)
    s_half = S.random_scalars_np(half, seed=303, below=C.order)
    dup = np.repeat(pts, 2, axis=0)                                   # P0 P0 P1 P1 ...: neighbours in every bucket list
    s_dup = np.repeat(s_half, 2, axis=0)
    rb = gpu.ResidentBases(curve, dup)
    gpu.msm_set_affine(1)
    try:
        # (no shift table for the doubled key: building one would find the equal bases and add their scalars up front --
        #  test_msm_adds_up_the_scalars_of_equal_bases -- and no doubling would be left for the rounds)
        rb_half.precompute(0)
        ref = gpu.proj_add(curve, rb_half.msm(s_half), rb_half.msm(s_half))
        e_xy, e_inf = gpu.proj_to_affine(curve, ref)
        for call in range(3):                                         # overflow + redo, then the sticky C++ path twice
            g_xy, g_inf = gpu.proj_to_affine(curve, rb.msm(s_dup))
            assert g_inf == e_inf and (g_xy == e_xy).all(), call
        # the same key with other scalars (few coincidences): still correct on the path it was moved to
        s2 = S.random_scalars_np(2 * half, seed=304, below=C.order)
        a = gpu.proj_to_affine(curve, rb.msm(s2))
        gpu.msm_set_affine(0)
        b = gpu.proj_to_affine(curve, rb.msm(s2))
        assert a[1] == b[1] and (a[0] == b[0]).all()
    finally:
        gpu.msm_set_affine(2)
        rb.free()
        rb_half.free()


# ------------------------------------------------------------------------------ proving-key wire format
def _wire(C, pts):
    """GroupAffine::write (short_weierstrass_projective.rs:185-192): x || y || infinity, each base-field
    coefficient as 96 LE bytes of its canonical integer; zero() = (0, 1, true)."""
    out = bytearray()
    for P in pts:
        x, y = (tuple([0] * C.deg), tuple([1] + [0] * (C.deg - 1))) if P is None else P
        for coord in (x, y):
            for c in coord:
                out += int(c).to_bytes(96, "little")
        out.append(1 if P is None else 0)
    return bytes(out)


@pytest.mark.parametrize("curve,n", [("mnt4753_g1", 200), ("mnt6753_g1", 100), ("mnt4753_g2", 60), ("mnt6753_g2", 40)])
def test_bases_upload_wire_vs_oracle(gpu, curve, n):
    """gh_bases_upload_wire: bases from the reference's serialised form (canonical little-endian coefficients,
    infinity flag byte) give the same MSM as the same points uploaded as Montgomery limbs, and as the oracle."""
    C = pyref.CURVES[curve]
    rng = pyref.Rng(55 + n)
    pts = S.chain_points(C, n, rng)
    pts[3] = None
    scal = [rng.field_elem(C.order) for _ in range(n)]
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    exp = S.oracle_msm(curve, b, inf, s, 8)
    rb = gpu.ResidentBases.from_wire(curve, _wire(C, pts))
    assert rb.n == n
    assert affine_eq(gpu, curve, rb.msm(s), exp)
    rb.precompute(0)
    assert affine_eq(gpu, curve, rb.msm(s), exp)
    rb.free()
    # what FromBytes rejects: a coefficient >= p (fp_768.rs:791-805), a flag byte that is no bool
    bad = bytearray(_wire(C, pts[:2]))
    bad[0:96] = int(C.F.p).to_bytes(96, "little")
    with pytest.raises(gpu.GingerHipError):
        gpu.ResidentBases.from_wire(curve, bytes(bad))
    bad = bytearray(_wire(C, pts[:2]))
    bad[192 * C.deg] = 2
    with pytest.raises(gpu.GingerHipError):
        gpu.ResidentBases.from_wire(curve, bytes(bad))


def test_wire_byte_order_matches_reference_fixture():
    """the reference's own 96-byte fixtures (fields/mnt{4,6}753/test_vec/*_tobyte) are canonical little-endian
    integers below p -- the convention _wire() and gh_bases_upload_wire use"""
    kats = json.load(open(os.path.join(G, "ref_kats.json")))
    for tag, F in (("mnt4753", pyref.P4), ("mnt6753", pyref.P6)):
        raw = bytes.fromhex(kats["test_vec/%s_tobyte" % tag])
        assert len(raw) == 96 and int.from_bytes(raw, "little") < F.p


def test_msm_batch_argument_errors(gpu):
    """a batch stays on one curve; bad handles and null arguments are reported, nothing is left in flight"""
    import ctypes
    C1, C2 = pyref.CURVES["mnt4753_g1"], pyref.CURVES["mnt6753_g1"]
    rng = pyref.Rng(1)
    b1, _ = S.bases_array(C1, S.chain_points(C1, 8, rng))
    b2, _ = S.bases_array(C2, S.chain_points(C2, 8, rng))
    r1, r2 = gpu.ResidentBases("mnt4753_g1", b1), gpu.ResidentBases("mnt6753_g1", b2)
    s = S.scalar_array([rng.field_elem(C1.order) for _ in range(8)])
    d = gpu.DeviceBuffer(s.nbytes).upload(s)
    with pytest.raises(gpu.GingerHipError):
        gpu.msm_batch_dev([(r1, d, 8), (r2, d, 8)])
    lib = gpu.load_library()
    out = np.zeros(36, dtype=np.uint64)
    bad = (ctypes.c_void_p * 1)(ctypes.c_void_p(0))
    ptrs = (ctypes.c_void_p * 1)(d.ptr)
    ns = (ctypes.c_size_t * 1)(8)
    assert lib.gh_msm_resident_dev_batch(bad, ptrs, ns, 1, out.ctypes.data_as(ctypes.c_void_p)) == -6      # GH_E_BAD_HANDLE
    assert lib.gh_msm_resident_dev_batch(None, ptrs, ns, 1, out.ctypes.data_as(ctypes.c_void_p)) == -1     # GH_E_BAD_ARG
    assert lib.gh_msm_resident_dev_batch(None, None, None, 0, None) == 0
    # the library is still usable afterwards
    exp = S.oracle_msm("mnt4753_g1", b1, None, s, 4)
    assert affine_eq(gpu, "mnt4753_g1", gpu.msm_batch_dev([(r1, d, 8)])[0], exp)
    d.free(); r1.free(); r2.free()


def test_msm_cached_is_a_pure_function_of_its_arguments(gpu):
    """gh_msm_cached: two different base sets through ONE host buffer (same address, length, first and last base -- what the
    round-2 Rust shim keyed on) give two different sums, each equal to the oracle's; repeats are hits (scalars only), the
    shift table appears at the second sighting, and a small budget evicts the least recently used key."""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    n = 1 << 13
    pts = S.chain_points(C, n, pyref.Rng(21))
    buf, _ = S.bases_array(C, pts)
    other, _ = S.bases_array(C, S.chain_points(C, n, pyref.Rng(22)))
    s1 = S.random_scalars_np(n, seed=71, below=C.order)
    s2 = S.random_scalars_np(n, seed=72, below=C.order)
    gpu.key_cache_clear()
    gpu.key_cache_config(64 << 30, 2)
    base = gpu.key_cache_stats()

    def aff(x):
        xy, inf = gpu.proj_to_affine(curve, x)
        return inf, xy.tobytes()

    def oaff(b, s):
        xy, inf = S.oracle_affine(curve, S.oracle_msm(curve, b, None, s, 16))
        return inf, xy.tobytes()

    exp_a1, exp_a2 = oaff(buf, s1), oaff(buf, s2)
    assert aff(gpu.msm_cached(curve, buf, s1)) == exp_a1                       # miss: upload
    assert gpu.msm_last_timing()["num_windows"] == 752 // gpu.msm_last_timing()["window_bits"] + 1
    assert aff(gpu.msm_cached(curve, buf, s2)) == exp_a2                       # hit: second sighting builds the table
    st = gpu.key_cache_stats()
    assert (st["misses"] - base["misses"], st["hits"] - base["hits"], st["tables_built"] - base["tables_built"], st["entries"]) == (1, 1, 1, 1)
    addr = buf.ctypes.data
    buf[1:n - 1] = other[1:n - 1]                                             # same buffer, same ends, other interior
    assert buf.ctypes.data == addr
    exp_b1 = oaff(buf, s1)
    assert exp_b1 != exp_a1
    assert aff(gpu.msm_cached(curve, buf, s1)) == exp_b1                       # NOT the stale key's sum
    st = gpu.key_cache_stats()
    assert st["misses"] - base["misses"] == 2 and st["entries"] == 2
    # ragged lengths: the key is the first min(n_bases, n_scalars) bases
    m = n - 777
    assert aff(gpu.msm_cached(curve, buf, s1[:m])) == oaff(buf[:m], s1[:m])
    # a budget below two keys: the least recently used one goes
    before = gpu.key_cache_stats()
    gpu.key_cache_config(int(n * 208 * 1.5), 0)
    st = gpu.key_cache_stats()
    assert st["entries"] == 1 and st["evictions"] > before["evictions"]
    assert aff(gpu.msm_cached(curve, buf, s2)) == oaff(buf, s2)
    gpu.key_cache_config()
    gpu.key_cache_clear()
    assert gpu.key_cache_stats()["entries"] == 0


def test_msm_cached_survives_forced_hash_collisions_and_a_failed_table_build(gpu):
    """Two round-3 defects of gh_msm_cached, exercised by fault injection on every GPU run instead of by filling the card
    (the test that did that aborted the suite once and was deleted):
    (a) a shift-table build that runs out of memory drops every pooled scratch buffer -- GH_TEST_TABLE_NOMEM=1 makes every
        build take exactly that path (msm_impl.h precompute_bases).  The scalars' PCIe copy is in flight at that moment; round 3
        copied into a pooled buffer that the builder freed under the copy (GPU fault at 2^22 MNT6 G2 pairs).  The call must
        succeed on the per-window path with the oracle's sum, `tables_built` unchanged;
    (b) a hit used to be decided by 128 hash bits alone.  gh_test_hooks(1) forces the selection lanes of every key to one
        value: different bases of one size must still be different keys (verification lanes), each with its own sum."""
    gpu.key_cache_clear()
    gpu.key_cache_config(None, 2)

    def aff(curve, x):
        xy, inf = gpu.proj_to_affine(curve, x)
        return inf, xy.tobytes()

    def oaff(curve, b, s):
        xy, inf = S.oracle_affine(curve, S.oracle_msm(curve, b, None, s, 16))
        return inf, xy.tobytes()

    os.environ["GH_TEST_TABLE_NOMEM"] = "1"
    try:
        for curve, n in (("mnt4753_g1", 1 << 13), ("mnt6753_g2", 1 << 12)):
            C = pyref.CURVES[curve]
            buf, _ = S.bases_array(C, S.chain_points(C, n, pyref.Rng(31)))
            base = gpu.key_cache_stats()
            for k in range(3):                                        # sighting 2 and 3 both try the table and fall back
                s = S.random_scalars_np(n, seed=80 + k, below=C.order)
                assert aff(curve, gpu.msm_cached(curve, buf, s)) == oaff(curve, buf, s)
                tm = gpu.msm_last_timing()
                assert tm["num_windows"] == 752 // tm["window_bits"] + 1          # the per-window path
            st = gpu.key_cache_stats()
            assert (st["misses"] - base["misses"], st["hits"] - base["hits"], st["tables_built"] - base["tables_built"]) == (1, 2, 0)
    finally:
        del os.environ["GH_TEST_TABLE_NOMEM"]
    # without the injected failure the same bases, seen afresh, get their table at the second sighting
    gpu.key_cache_clear()
    base = gpu.key_cache_stats()
    for k in range(2):
        s = S.random_scalars_np(n, seed=90 + k, below=C.order)
        assert aff(curve, gpu.msm_cached(curve, buf, s)) == oaff(curve, buf, s)
    assert gpu.key_cache_stats()["tables_built"] - base["tables_built"] == 1
    gpu.key_cache_clear()

    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    n = 1 << 12
    a, _ = S.bases_array(C, S.chain_points(C, n, pyref.Rng(41)))
    b, _ = S.bases_array(C, S.chain_points(C, n, pyref.Rng(42)))
    s = S.random_scalars_np(n, seed=95, below=C.order)
    ea, eb = oaff(curve, a, s), oaff(curve, b, s)
    assert ea != eb
    gpu.set_test_hooks(1)
    try:
        assert gpu.bases_key_id(curve, a)[:2] == gpu.bases_key_id(curve, b)[:2]
        base = gpu.key_cache_stats()
        assert aff(curve, gpu.msm_cached(curve, a, s)) == ea           # miss
        assert aff(curve, gpu.msm_cached(curve, b, s)) == eb           # collides with a's entry: verified, rejected, miss
        assert aff(curve, gpu.msm_cached(curve, a, s)) == ea           # hit (table)
        assert aff(curve, gpu.msm_cached(curve, b, s)) == eb           # hit behind a colliding entry
        st = gpu.key_cache_stats()
        assert (st["misses"] - base["misses"], st["hits"] - base["hits"], st["entries"]) == (2, 2, 2)
        assert st["collisions"] - base["collisions"] >= 2
    finally:
        gpu.set_test_hooks(0)
        gpu.key_cache_clear()
        gpu.key_cache_config()


def test_msm_parity_again_with_the_lean_reduction_forced(gpu):
    """Inside a pipelined batch of large MSMs the bucket reduction runs in its lane-level form (msm_impl.h: `lean`; level 1 hands
    every lane's two sums to level 2).  Sizes the oracle can referee never reach it by themselves, so the MSM parity tests of
    this file run once more in a child process with GH_REDUCE_LEAN=1 (the switch is read once per process): every curve's G1
    path, all window regimes, skewed scalars, heavy buckets, resident keys and batches -- against the oracle as before."""
    import os, subprocess, sys
    if os.environ.get("GH_REDUCE_LEAN"):
        pytest.skip("already the forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GH_REDUCE_LEAN="1")
    sel = "test_msm_golden or test_msm_vs_oracle or test_msm_window_sizes_vs_oracle or test_msm_unequal_lengths_and_resident or test_msm_skewed_scalars"
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k", sel],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout


def test_affine_rounds_parity_again_with_every_round_split(gpu):
    """Large affine rounds go out as two halves on two streams (msm_impl.h launch_tree); sizes the oracle can referee stay below the
    threshold, so the affine-round parity tests of this file run once more in a child process with GH_AFF_SPLIT_B=1 (every round with
    more than one tile of outputs is split: ragged second halves, exception lists in both control blocks, duplicate bases that overflow
    them) -- against the oracle as before."""
    import os, subprocess, sys
    if os.environ.get("GH_AFF_SPLIT_B"):
        pytest.skip("already the forced run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GH_AFF_SPLIT_B="1")
    sel = ("test_msm_affine_bucket_sums_vs_oracle or test_msm_affine_degenerate_inputs or test_msm_adds_up_the_scalars_of_equal_bases or "
           "test_g2_key_of_duplicate_bases_leaves_the_assembly_rounds or test_msm_precomputed_skewed_and_large")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q", "-k", sel],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout
