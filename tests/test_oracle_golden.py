"""CPU oracle vs the first-principles golden vectors (tests/golden/{msm,ntt}_golden.json) and the
algebraic identities the reference itself tests: naive == Pippenger incl. unequal lengths
(variable_base.rs:102-151), ifft(fft) = id and coset round trip (fft/test.rs:9-43),
parallel_fft == serial_fft (fft/test.rs:45-72)."""
import json
import os

import numpy as np
import pytest

import pyref
import support as S

G = os.path.join(os.path.dirname(__file__), "golden")
MSM = json.load(open(os.path.join(G, "msm_golden.json")))
NTT = json.load(open(os.path.join(G, "ntt_golden.json")))


def load_msm_case(name):
    case = MSM[name]
    curve = name.replace("_zero_sum", "")
    C = pyref.CURVES[curve]
    pts = [None if b is None else (tuple(int(c, 16) for c in b[0]), tuple(int(c, 16) for c in b[1])) for b in case["bases"]]
    scal = [int(s, 16) for s in case["scalars"]]
    e = case["expected_affine"]
    exp = None if e is None else (tuple(int(c, 16) for c in e[0]), tuple(int(c, 16) for c in e[1]))
    return curve, C, pts, scal, exp


@pytest.mark.parametrize("name", list(MSM))
def test_oracle_msm_golden(name):
    curve, C, pts, scal, exp = load_msm_case(name)
    b, inf = S.bases_array(C, pts)
    for threads in (1, 4):
        out = S.oracle_msm(curve, b, inf, S.scalar_array(scal), threads)
        assert S.affine_of_xyz(C, out) == exp


@pytest.mark.parametrize("name", list(NTT))
def test_oracle_ntt_golden(name):
    case = NTT[name]
    F = S.FIELD_OF[case["field"]]
    a = S.fe_array(F, [int(x, 16) for x in case["input"]])
    for nm, flags in (("fft", 0), ("ifft", 1), ("coset_fft", 2), ("coset_ifft", 3)):
        for threads in (1, 2, 8):
            got = S.fe_list(F, S.oracle_fft(case["field"], a, case["log_n"], flags, threads))
            assert got == [int(x, 16) for x in case[nm]], (name, nm, threads)


def test_oracle_msm_empty_and_small():
    for curve, C in pyref.CURVES.items():
        out = S.oracle_msm(curve, np.zeros((0, 24 * C.deg), np.uint64), None, np.zeros((0, 12), np.uint64))
        assert S.affine_of_xyz(C, out) is None
        # (0, 1, 0): swp.rs:372-378
        k = C.deg
        assert pyref.ext_from_abi(C.F, [int(v) for v in out[12 * k:24 * k]], k) == C.E.one()


def test_oracle_msm_vs_naive_window_regimes():
    """len < 32 -> c = 3 (251 windows); len >= 32 -> c from the log2 formula (variable_base.rs:14-18)"""
    rng = pyref.Rng(99)
    C = pyref.CURVES["mnt4753_g1"]
    for n in (5, 31, 32, 100):
        pts = S.chain_points(C, n, rng)
        scal = [rng.field_elem(C.order) for _ in range(n)]
        b, inf = S.bases_array(C, pts)
        out = S.oracle_msm("mnt4753_g1", b, inf, S.scalar_array(scal), 8)
        assert S.affine_of_xyz(C, out) == C.msm(pts, scal)
    # unequal lengths: zip truncates (variable_base.rs:134-151)
    out = S.oracle_msm("mnt4753_g1", b[:60], inf[:60], S.scalar_array(scal), 8)
    assert S.affine_of_xyz(C, out) == C.msm(pts[:60], scal)


@pytest.mark.parametrize("field", ["mnt4753_fr", "mnt6753_fr"])
def test_oracle_fft_roundtrip_and_variants(field):
    F = S.FIELD_OF[field]
    rng = pyref.Rng(5)
    for log_n in (0, 1, 4, 9):
        n = 1 << log_n
        vals = [rng.field_elem(F.p) for _ in range(n)]
        a = S.fe_array(F, vals)
        f = S.oracle_fft(field, a, log_n, 0, 8)
        assert S.fe_list(F, f) == pyref.ntt_fast(F, vals, log_n)
        assert S.fe_list(F, S.oracle_fft(field, f, log_n, 1, 8)) == vals
        c = S.oracle_fft(field, a, log_n, 2, 8)
        assert S.fe_list(F, S.oracle_fft(field, c, log_n, 3, 8)) == vals
        if log_n >= 4:   # parallel_fft == serial_fft for every log_cpus (fft/test.rs:45-72)
            ser = a.copy()
            S.oracle().oracle_fft_variant(S.FIELD_ID[field], S.ptr(ser), log_n, -1)
            for log_cpus in (1, 2, 3):
                par = a.copy()
                S.oracle().oracle_fft_variant(S.FIELD_ID[field], S.ptr(par), log_n, log_cpus)
                assert (par == ser).all()


def test_oracle_domain_limits():
    import ctypes
    out = np.zeros(48, dtype=np.uint64)
    lg = ctypes.c_uint32()
    o = S.oracle()
    assert o.oracle_domain(0, 1 << 29, S.ptr(out), ctypes.byref(lg)) == 1 and lg.value == 29
    assert o.oracle_domain(0, (1 << 29) + 1, S.ptr(out), ctypes.byref(lg)) == 0      # log >= two-adicity 30 -> None (domain.rs:69-71)
    assert o.oracle_domain(1, 1 << 14, S.ptr(out), ctypes.byref(lg)) == 1
    assert o.oracle_domain(1, (1 << 14) + 1, S.ptr(out), ctypes.byref(lg)) == 0      # MNT6 Fr: two-adicity 15
    assert o.oracle_domain(0, 0, S.ptr(out), ctypes.byref(lg)) == 1 and lg.value == 0  # new(0) -> size 1


def test_oracle_witness_map_first_principles():
    """witness_map (r1cs_to_qap.rs:121-166): for rows satisfying a_i b_i = c_i on the domain, h must be
    the exact quotient (A B - C) / (X^N - 1); with d1 d2 d3 the reference's h[0] / h[N] corrections."""
    F = pyref.P6
    p = F.p
    log_n, n = 3, 8
    rng = pyref.Rng(4)
    a = [rng.field_elem(p) for _ in range(n)]
    b = [rng.field_elem(p) for _ in range(n)]
    c = [(x * y) % p for x, y in zip(a, b)]
    pa, pb, pc = (pyref.ntt_fast(F, v, log_n, inverse=True) for v in (a, b, c))
    prod = [0] * (2 * n - 1)
    for i, x in enumerate(pa):
        for j, y in enumerate(pb):
            prod[i + j] = (prod[i + j] + x * y) % p
    for i, x in enumerate(pc):
        prod[i] = (prod[i] - x) % p
    q = [0] * (n - 1)
    for i in range(2 * n - 2, n - 1, -1):
        q[i - n] = prod[i]
        prod[i - n] = (prod[i - n] + prod[i]) % p
        prod[i] = 0
    assert not any(prod)
    A, B, C = S.fe_array(F, a), S.fe_array(F, b), S.fe_array(F, c)
    z = S.fe_array(F, [0])
    assert S.fe_list(F, S.oracle_witness_map("mnt4753_fr", A, B, C, z, z, z)) == q + [0, 0]
    d1, d2, d3 = (rng.field_elem(p) for _ in range(3))
    h = S.fe_list(F, S.oracle_witness_map("mnt4753_fr", A, B, C, S.fe_array(F, [d1]), S.fe_array(F, [d2]), S.fe_array(F, [d3])))
    assert h == [(q[0] - d3 - d1 * d2) % p] + q[1:] + [0, (d1 * d2) % p]


def test_batch_inversion_lagrange_and_sap_map_vs_python():
    """round-2 oracle additions against first principles: batch_inversion (fields/mod.rs:412-442) with zeros in the batch,
    evaluate_all_lagrange_coefficients (domain.rs:183-219) outside and inside the domain (L_i(tau) = (tau^N - 1) w^i / (N (tau - w^i))),
    and the SAP witness map (gm17/r1cs_to_sap.rs:191-240) against the same steps on Python integers"""
    for field in ("mnt4753_fr", "mnt6753_fr"):
        F = S.FIELD_OF[field]
        p = F.p
        rng = pyref.Rng(17)
        vals = [rng.field_elem(p) for _ in range(41)]
        vals[0] = vals[7] = vals[40] = 0
        assert S.fe_list(F, S.oracle_batch_inversion(field, S.fe_array(F, vals))) == [0 if v == 0 else pow(v, -1, p) for v in vals]
        log_n = 5
        n = 1 << log_n
        w = pyref.domain_params(F, log_n)
        tau = rng.field_elem(p)
        tn = pow(tau, n, p)
        exp = [(tn - 1) * pow(n, -1, p) % p * pow(w, i, p) % p * pow((tau - pow(w, i, p)) % p, -1, p) % p for i in range(n)]
        assert S.fe_list(F, S.oracle_lagrange(field, log_n, S.fe_array(F, [tau])[0])) == exp
        assert sum(exp) % p == 1                                                    # the Lagrange basis sums to one
        assert S.fe_list(F, S.oracle_lagrange(field, log_n, S.fe_array(F, [pow(w, 9, p)])[0])) == [1 if i == 9 else 0 for i in range(n)]
        av, cv = [rng.field_elem(p) for _ in range(n)], [rng.field_elem(p) for _ in range(n)]
        d1, d2 = rng.field_elem(p), rng.field_elem(p)
        ac = pyref.ntt_fast(F, av, log_n, inverse=True)
        hh = [2 * d1 * x % p for x in ac]
        hh[0] = (hh[0] - d2 - d1 * d1) % p
        hh.append(d1 * d1 % p)
        ae = pyref.ntt_fast(F, ac, log_n, coset=True)
        ce = pyref.ntt_fast(F, pyref.ntt_fast(F, cv, log_n, inverse=True), log_n, coset=True)
        vi = pow((pow(F.generator, n, p) - 1) % p, -1, p)
        q = pyref.ntt_fast(F, [(x * x - y) * vi % p for x, y in zip(ae, ce)], log_n, inverse=True, coset=True)
        for i in range(n - 1):
            hh[i] = (hh[i] + q[i]) % p
        got = S.oracle_sap_witness_map(field, S.fe_array(F, av), S.fe_array(F, cv), S.fe_array(F, [d1])[0], S.fe_array(F, [d2])[0], 2)
        assert S.fe_list(F, got) == hh
