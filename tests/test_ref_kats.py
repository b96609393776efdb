"""Pin the CPU oracle (and the big-int model) with the reference's own known-answer tests.
Vectors: tests/golden/ref_kats.json (extracted by tests/golden/extract_ref_kats.py from
algebra/src/fields/mnt{4,6}753/tests.rs and algebra/src/curves/mnt{4,6}753/tests.rs).
"ctor" = "new": raw Montgomery limbs (Fq::new); "from_repr": canonical integer."""
import json
import os

import numpy as np
import pytest

import pyref
import support as S

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_kats.json")))
FIELD_FILES = {"fields/mnt4753/tests.rs": (4, pyref.P4), "fields/mnt6753/tests.rs": (6, pyref.P6)}


def vecs(path, name):
    return KATS[path][name]["vectors"]


def raw(v):
    return int(v["v"], 16)


def mont_raw(F, v):
    """Montgomery limbs of the value the literal denotes"""
    x = int(v["v"], 16)
    return x if v["ctor"] == "new" else F.to_mont(x)


def fp(fid, op, a, b=0):
    out = np.zeros(12, dtype=np.uint64)
    rc = S.oracle().oracle_fp_op(fid, op, S.ptr(S.u64(a)), S.ptr(S.u64(b)), S.ptr(out))
    assert rc == 1
    return S.to_int(out)


@pytest.mark.parametrize("path", list(FIELD_FILES))
def test_fq_add_sub_mul_square_neg(path):
    fid, F = FIELD_FILES[path]
    p = F.p
    # test_fq_add_assign (mnt4753 tests.rs:270-448 / mnt6753 :475-656): tmp, tmp+0, tmp+1, b, tmp+1+b, q-1, c, d, c+d (= q-1)
    v = [mont_raw(F, x) for x in vecs(path, "test_fq_add_assign")]
    assert fp(fid, 2, v[0], 0) == v[1] == v[0]
    assert fp(fid, 2, v[0], 1) == v[2]
    assert fp(fid, 2, v[2], v[3]) == v[4]
    assert v[5] == p - 1 and fp(fid, 2, v[5], 1) == 0
    assert fp(fid, 2, v[6], v[7]) == v[8] == p - 1
    # test_fq_sub_assign (:449-603 / :657-811): (a, b, a-b) twice, then (a, a-0)
    v = [mont_raw(F, x) for x in vecs(path, "test_fq_sub_assign")]
    assert fp(fid, 3, v[0], v[1]) == v[2]
    assert fp(fid, 3, v[3], v[4]) == v[5]
    assert fp(fid, 3, v[6], 0) == v[7]
    # test_fq_mul_assign (:604-646 / :812-903): raw Montgomery a * b = c
    v = [mont_raw(F, x) for x in vecs(path, "test_fq_mul_assign")]
    assert fp(fid, 0, v[0], v[1]) == v[2]
    assert (v[0] * v[1] * F.Rinv) % p == v[2]                       # big-int model agrees
    # test_fq_squaring (:696-747 / :904-956)
    v = [mont_raw(F, x) for x in vecs(path, "test_fq_squaring")]
    assert fp(fid, 1, v[0]) == v[1]
    # test_neg_one (:202-219 / :406-423): literal == -one
    v = [mont_raw(F, x) for x in vecs(path, "test_neg_one")]
    assert fp(fid, 4, F.R) == v[0]
    # round trips and inverse
    x = v[0]
    assert fp(fid, 7, fp(fid, 8, x)) == x
    assert fp(fid, 0, x, fp(fid, 6, x)) == F.R


def test_fq_bytes_fixture():
    # test_fq_bytes (mnt4753 tests.rs:896-917, mnt6753 :1195-1214): 96-byte LE canonical serialisation
    for tag, F in (("mnt4753", pyref.P4), ("mnt6753", pyref.P6)):
        b = bytes.fromhex(KATS["test_vec/%s_tobyte" % tag])
        assert len(b) == 96
        assert int.from_bytes(b, "little") < F.p


def ext_op(tower, op, a, b=None):
    k = tower
    out = np.zeros(12 * k, dtype=np.uint64)
    aa = np.array(a, dtype=np.uint64)
    bb = np.array(b, dtype=np.uint64) if b is not None else None
    rc = S.oracle().oracle_ext_op(tower, op, S.ptr(aa), S.ptr(bb) if bb is not None else None, S.ptr(out))
    assert rc == 1
    return [int(x) for x in out]


@pytest.mark.parametrize("path,tower,tag", [("fields/mnt4753/tests.rs", 2, "fq2"), ("fields/mnt6753/tests.rs", 3, "fq3")])
def test_tower_kats(path, tower, tag):
    fid, F = FIELD_FILES[path]
    k = tower
    E = pyref.Ext(F, k, 13 if k == 2 else 11)

    def elems(name):
        v = [F.from_mont(mont_raw(F, x)) for x in vecs(path, name)]   # canonical ints
        assert len(v) % k == 0
        return [tuple(v[i:i + k]) for i in range(0, len(v), k)]

    def abi(e):
        return pyref.ext_to_abi(F, e)

    def back(l):
        return pyref.ext_from_abi(F, l, k)

    a, r = elems("test_%s_squaring" % tag)                 # fq2 :1071-1153, fq3 :1278-1382
    assert back(ext_op(tower, 1, abi(a))) == r == E.mul(a, a)
    a, b, r = elems("test_%s_mul" % tag)                   # fq2 :1154-1250, fq3 :1383-1522
    assert back(ext_op(tower, 0, abi(a), abi(b))) == r == E.mul(a, b)
    a, r = elems("test_%s_inverse" % tag)                  # fq2 :1251-1320, fq3 :1523-1620
    assert back(ext_op(tower, 6, abi(a))) == r == E.inv(a)
    a, b, r = elems("test_%s_addition" % tag)
    assert back(ext_op(tower, 2, abi(a), abi(b))) == r
    a, b, r = elems("test_%s_subtraction" % tag)
    assert back(ext_op(tower, 3, abi(a), abi(b))) == r
    a, r = elems("test_%s_negation" % tag)
    assert back(ext_op(tower, 4, abi(a))) == r
    a, r = elems("test_%s_doubling" % tag)
    assert back(ext_op(tower, 5, abi(a))) == r


def ec(curve, op, p, q=None, flag=0, outn=None):
    C = pyref.CURVES[curve]
    out = np.zeros(outn or 36 * C.deg, dtype=np.uint64)
    pp = np.array(p, dtype=np.uint64)
    qq = np.array(q, dtype=np.uint64) if q is not None else None
    rc = S.oracle().oracle_ec_op(S.CURVE_ID[curve], op, S.ptr(pp), S.ptr(qq) if qq is not None else None, flag, S.ptr(out))
    return rc, out


@pytest.mark.parametrize("path,g1,g2", [("curves/mnt4753/tests.rs", "mnt4753_g1", "mnt4753_g2"),
                                        ("curves/mnt6753/tests.rs", "mnt6753_g1", "mnt6753_g2")])
def test_curve_kats(path, g1, g2):
    for curve, pref in ((g1, "g1"), (g2, "g2")):
        C = pyref.CURVES[curve]
        F, k = C.F, C.deg

        def elems(name):
            v = [int(x["v"], 16) for x in vecs(path, name)]
            assert all(x["ctor"] == "from_repr" for x in vecs(path, name))
            return [tuple(v[i:i + k]) for i in range(0, len(v), k)]

        def proj(X, Y, Z):
            return pyref.ext_to_abi(F, X) + pyref.ext_to_abi(F, Y) + pyref.ext_to_abi(F, Z)

        def affine(xyz):
            rc, out = ec(curve, 4, xyz, outn=24 * k)
            if rc:
                return None
            o = [int(x) for x in out]
            return (pyref.ext_from_abi(F, o[:12 * k], k), pyref.ext_from_abi(F, o[12 * k:], k))

        # test_gX_addition_correctness (mnt4753 :623-752 / :1010-1267): P, Q projective; expected affine of P+Q
        e = elems("test_%s_addition_correctness" % pref)
        P, Q, R = e[0:3], e[3:6], e[6:8]
        _, s = ec(curve, 0, proj(*P), proj(*Q))
        assert affine(s) == (R[0], R[1])
        assert C.add(C.proj_to_affine(*P), C.proj_to_affine(*Q)) == (R[0], R[1])     # big-int model agrees
        # mixed addition against the same expectation
        Qa = C.proj_to_affine(*Q)
        _, s = ec(curve, 2, proj(*P), pyref.ext_to_abi(F, Qa[0]) + pyref.ext_to_abi(F, Qa[1]), 0)
        assert affine(s) == (R[0], R[1])
        # test_gX_doubling_correctness (:753-839 / :1268-1434)
        e = elems("test_%s_doubling_correctness" % pref)
        P, R = e[0:3], e[3:5]
        _, s = ec(curve, 1, proj(*P))
        assert affine(s) == (R[0], R[1])
        assert C.add(C.proj_to_affine(*P), C.proj_to_affine(*P)) == (R[0], R[1])
        # test_gX_affine_projective_conversion (:926-1009 / :1435-1596)
        e = elems("test_%s_affine_projective_conversion" % pref)
        P, R = e[0:3], e[3:5]
        assert affine(proj(*P)) == (R[0], R[1])
        if pref == "g1":
            # test_g1_scalar_multiplication (:840-925): affine a, scalar (Fr::from_repr), expected affine
            v = vecs(path, "test_g1_scalar_multiplication")
            x, y, sc, rx, ry = [int(t["v"], 16) for t in v]
            _, s = ec(curve, 3, proj((x,), (y,), (1,)), pyref.int_to_limbs(sc))
            assert affine(s) == ((rx,), (ry,))
            assert C.mul(sc, ((x,), (y,))) == ((rx,), (ry,))
