"""FixedBaseMSM on the device (gh_fixed_base_*) against the oracle's restatement of algebra/src/msm/fixed_base.rs:7-79
(window table + windowed_mul) and against textbook k * g: every output point, compared after into_affine()."""
import ctypes

import numpy as np
import pytest

import pyref
import support as S

pytestmark = pytest.mark.gpu


def oracle_fixed(curve, g_xyz, window, scal_mont, n, threads=8, scalar_size=753):
    C = pyref.CURVES[curve]
    O = S.oracle()
    O.oracle_fixed_base_msm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p, ctypes.c_int]
    out = np.zeros((n, 36 * C.deg), dtype=np.uint64)
    w = O.oracle_fixed_base_msm(S.CURVE_ID[curve], S.ptr(g_xyz), scalar_size, window, S.ptr(scal_mont), n, S.ptr(out), threads)
    assert w > 0
    return out, w


@pytest.mark.parametrize("curve,n", [("mnt4753_g1", 700), ("mnt6753_g1", 300), ("mnt4753_g2", 200), ("mnt6753_g2", 120)])
def test_fixed_base_msm_vs_oracle(gpu, curve, n):
    C = pyref.CURVES[curve]
    F = S.FIELD_OF["mnt4753_fr" if curve.startswith("mnt4") else "mnt6753_fr"]
    r = C.order
    rng = pyref.Rng(40 + n)
    ks = [rng.field_elem(r) for _ in range(n)]
    ks[:8] = [0, 1, 2, r - 1, r - 2, (r - 1) // 2, 1 << 752, (1 << 700) - 1]
    canon = S.scalar_array(ks)
    mont = S.fe_array(F, ks)
    g = S.chain_points(C, 1, rng)[0]
    g_xyz = S.proj_array(C, g)                                   # a projective representative with Z != 1
    assert gpu.FixedBaseMSM.get_mul_window_size(n) == (3 if n < 32 else int(np.ceil(np.log(n))))
    exp, w_ref = oracle_fixed(curve, g_xyz, 0, mont, n)
    assert w_ref == gpu.FixedBaseMSM.get_mul_window_size(n)
    for window in (w_ref, 3, 11):
        tab = gpu.FixedBaseMSM(curve, g_xyz, 753, window)
        try:
            got = tab.multi_scalar_mul(canon)
        finally:
            tab.free()
        for i in range(n):
            assert gpu.proj_to_affine(curve, got[i])[1] == S.oracle_affine(curve, exp[i])[1], (curve, window, i)
            a, b = gpu.proj_to_affine(curve, got[i])[0], S.oracle_affine(curve, exp[i])[0]
            assert (a == b).all(), (curve, window, i)
    for i in (0, 1, 3, 9):                                        # and the textbook multiple
        assert S.affine_of_xyz(C, got[i]) == (None if ks[i] % r == 0 else C.mul(ks[i] % r, g))
    # infinity as the base: every multiple is infinity
    tab = gpu.FixedBaseMSM(curve, S.proj_array(C, None), 753, 5)
    out = tab.multi_scalar_mul(canon[:10])
    tab.free()
    assert all(gpu.proj_to_affine(curve, out[i])[1] for i in range(10))


@pytest.mark.parametrize("curve", ["mnt4753_g1", "mnt6753_g2"])
def test_fixed_base_short_scalar_size_and_affine_output(gpu, curve):
    """(a) scalar_size < MODULUS_BITS with scalars that have higher bits set: windowed_mul (fixed_base.rs:45-66) still reads
    `window` bits per row below bit 753, and a digit of the last row at or beyond last_in_window meets the zero the table was
    initialised with (:22-33) -- the device follows that, digit by digit, against the oracle's literal restatement;
    (b) gh_fixed_base_msm_affine = multi_scalar_mul + batch_normalization + into_affine (generator.rs:247-335): Montgomery rows
    equal gh_proj_to_affine of every projective output, canonical rows are the same integers out of Montgomery form, and
    infinity is GroupAffine::zero() = (0, 1, true)."""
    C = pyref.CURVES[curve]
    F = S.FIELD_OF["mnt4753_fr" if curve.startswith("mnt4") else "mnt6753_fr"]
    r = C.order
    n = 150
    rng = pyref.Rng(91)
    ks = [rng.field_elem(r) for _ in range(n)]
    ks[:6] = [0, 1, 3, 4, (1 << 250) - 1, 1 << 250]
    canon, mont = S.scalar_array(ks), S.fe_array(F, ks)
    g_xyz = S.proj_array(C, S.chain_points(C, 1, rng)[0])
    for scalar_size, window in ((250, 8), (753, 7), (100, 3)):
        exp, _ = oracle_fixed(curve, g_xyz, window, mont, n, scalar_size=scalar_size)
        tab = gpu.FixedBaseMSM(curve, g_xyz, scalar_size, window)
        try:
            got = tab.multi_scalar_mul(canon)
            xy_m, inf_m = tab.multi_scalar_mul_affine(canon)
            xy_c, inf_c = tab.multi_scalar_mul_affine(canon, canonical=True)
        finally:
            tab.free()
        k = C.deg
        for i in range(n):
            e_xy, e_inf = S.oracle_affine(curve, exp[i])
            a_xy, a_inf = gpu.proj_to_affine(curve, got[i])
            assert a_inf == e_inf and (a_xy == e_xy).all(), (scalar_size, window, i)
            assert bool(inf_m[i]) == e_inf and bool(inf_c[i]) == e_inf
            assert (xy_m[i] == e_xy).all(), (scalar_size, window, i)          # infinity: (0, 1) in Montgomery form on both sides
            want = [C.F.from_mont(pyref.limbs_to_int([int(v) for v in e_xy[12 * c:12 * c + 12]])) for c in range(2 * k)]
            have = [pyref.limbs_to_int([int(v) for v in xy_c[i][12 * c:12 * c + 12]]) for c in range(2 * k)]
            assert have == want
        assert inf_m[0] == 1 and inf_m[1] == 0


def test_fixed_base_list_kernels_equal_the_per_scalar_kernel(gpu):
    """The sums run on the MSM's accumulation kernels (fixed_base.hip: one list per scalar); GH_FIXED_NAIVE=1 keeps the
    one-thread-per-scalar kernel of round 2.  Both must give the same affine rows -- the naive one in a child process, because
    the switch is read once per process."""
    import hashlib, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for curve in ("mnt4753_g1", "mnt4753_g2", "mnt6753_g2"):
        C = pyref.CURVES[curve]
        n = 500
        s = S.random_scalars_np(n, seed=123, below=C.order)
        s[0] = 0
        s[1, 1:] = 0                                              # a scalar with one non-zero row
        tab = gpu.FixedBaseMSM(curve, S.proj_array(C, C.G), 753, 9)
        xy, inf = tab.multi_scalar_mul_affine(s, canonical=True)
        tab.free()
        digests[curve] = hashlib.sha256(xy.tobytes() + inf.tobytes()).hexdigest()
        assert inf[0] == 1 and not inf[1:].any()
    child = (
        "import sys, hashlib, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import pyref, support as S\n"
        "from __graft_entry__ import _load_pkg\n"
        "gl = _load_pkg(); gl.init()\n"
        "for curve in ('mnt4753_g1', 'mnt4753_g2', 'mnt6753_g2'):\n"
        "    C = pyref.CURVES[curve]\n"
        "    s = S.random_scalars_np(500, seed=123, below=C.order); s[0] = 0; s[1, 1:] = 0\n"
        "    tab = gl.FixedBaseMSM(curve, S.proj_array(C, C.G), 753, 9)\n"
        "    xy, inf = tab.multi_scalar_mul_affine(s, canonical=True); tab.free()\n"
        "    print(curve, hashlib.sha256(xy.tobytes() + inf.tobytes()).hexdigest())\n"
    ) % (root, os.path.join(root, "tests"))
    env = dict(os.environ, GH_FIXED_NAIVE="1")
    out = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = dict(l.split() for l in out.stdout.strip().splitlines() if l.startswith("mnt"))
    assert got == digests


def test_fixed_base_generator_scale(gpu):
    """generator-sized call (generator.rs:243: one multi_scalar_mul per query): 2^16 scalars on G1, spot-checked"""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    n = 1 << 16
    s = S.random_scalars_np(n, seed=77, below=C.order)
    g_xyz = S.proj_array(C, C.G)
    w = gpu.FixedBaseMSM.get_mul_window_size(n)
    tab = gpu.FixedBaseMSM(curve, g_xyz, 753, w)
    out = tab.multi_scalar_mul(s)
    tab.free()
    for i in (0, 1, 12345, n - 1):
        k = pyref.limbs_to_int([int(v) for v in s[i]])
        assert S.affine_of_xyz(C, out[i]) == C.mul(k, C.G)


@pytest.mark.parametrize("curve,n", [("mnt4753_g1", 1000), ("mnt6753_g1", 333), ("mnt4753_g2", 150), ("mnt6753_g2", 97)])
def test_bases_generate_chain(gpu, curve, n):
    """gh_bases_generate_chain: the resident synthetic key P_0 + i H equals the textbook chain point for point, and an MSM
    over it equals the oracle's over the downloaded rows"""
    C = pyref.CURVES[curve]
    rng = pyref.Rng(900 + n)
    H = C.mul(rng.next_u64() | 1, C.G)
    P0 = C.mul(rng.next_u64() | 1, C.G)
    xy, _ = S.bases_array(C, [P0, H])
    rb = gpu.ResidentBases.chain(curve, xy[0], xy[1], n)
    try:
        assert rb.n == n
        rows = rb.download(0, n)
        exp = []
        P = P0
        for _ in range(n):
            exp.append(P)
            P = C.add(P, H)
        eb, _ = S.bases_array(C, exp)
        assert (rows == eb).all()
        assert (rb.download(n - 5, 5) == eb[n - 5:]).all()
        s = S.scalar_array([rng.field_elem(C.order) for _ in range(n)])
        got = gpu.proj_to_affine(curve, rb.msm(s))
        e_xy, e_inf = S.oracle_affine(curve, S.oracle_msm(curve, rows, None, s, 8))
        assert got[1] == e_inf and (got[0] == e_xy).all()
    finally:
        rb.free()


def test_bases_generate_chain_large(gpu):
    """2^20 distinct G1 bases in one call: rows far apart are the chain points (i H by scalar multiplication)"""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    n = 1 << 20
    H = C.mul(0x1234567, C.G)
    P0 = C.mul(0x7654321, C.G)
    xy, _ = S.bases_array(C, [P0, H])
    rb = gpu.ResidentBases.chain(curve, xy[0], xy[1], n)
    try:
        for i in (0, 1, 31, 32, 33, 65535, 1 << 19, n - 1):
            row = rb.download(i, 1)
            exp, _ = S.bases_array(C, [C.add(P0, C.mul(i, H)) if i else P0])
            assert (row == exp).all(), i
    finally:
        rb.free()
