"""The device arithmetic headers (ginger-lib_amd/csrc/fp29.h, ec29.h) compiled for the host with
g++ (tests/host_shim/fp29_shim.cpp) and checked against Python integers: the exact code the HIP
kernels run (29-bit reduced radix, Montgomery 2^754), including the ABI conversions."""
import ctypes
import os
import subprocess

import pytest

import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "build", "libfp29_shim.so")
U = ctypes.c_uint32


@pytest.fixture(scope="module")
def shim():
    src = os.path.join(ROOT, "tests", "host_shim", "fp29_shim.cpp")
    os.makedirs(os.path.dirname(SHIM), exist_ok=True)
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(ROOT, "ginger-lib_amd", "csrc", "fp29.h")), os.path.getmtime(os.path.join(ROOT, "ginger-lib_amd", "csrc", "ec29.h"))):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SHIM, src])
    return ctypes.CDLL(SHIM)


def words(x, n=24):
    return (U * n)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)])


def toint(w):
    return sum(int(v) << (32 * i) for i, v in enumerate(w))


@pytest.mark.parametrize("fid,F", [(4, pyref.P4), (6, pyref.P6)])
def test_field_ops(shim, fid, F):
    p = F.p
    RI = pow(2, 754, p)
    RIinv = pow(RI, -1, p)
    rng = pyref.Rng(7 + fid)
    for it in range(200):
        a, b = rng.field_elem(p), rng.field_elem(p)
        if it == 0:
            a = 0
        if it == 1:
            a = b = p - 1
        if it == 2:
            b = a
        if it == 3:
            a, b = 1, p - 1
        exp = {0: (a * b * RIinv) % p, 1: (a * a * RIinv) % p, 2: (a + b) % p, 3: (a - b) % p, 4: (-a) % p, 5: (2 * a) % p,
               6: (a * pow(2, 740, p) * RIinv) % p, 7: (a * pow(2, 768, p) * RIinv) % p,
               8: 11 * a % p, 9: 13 * a % p, 10: 26 * a % p, 11: 121 * a % p}
        out = (U * 24)()
        for op, e in exp.items():
            shim.t_fp_op(fid, op, words(a), words(b), out)
            assert toint(out) == e, (op, it)
        # dual product with one reduction: (a b + c d) 2^-754, extremes included
        c, d = (p - 1, p - 1) if it < 4 else (rng.field_elem(p), rng.field_elem(p))
        shim.t_fp_mul2(fid, words(a), words(b), words(c), words(d), out)
        assert toint(out) == ((a * b + c * d) * RIinv) % p, it
        # triple product with one reduction; extremes (all p - 1: the value passes 2^754) included
        e, f = (p - 1, p - 1) if it < 6 else (rng.field_elem(p), rng.field_elem(p))
        if it in (4, 5):
            a = b = c = d = p - 1
        shim.t_fp_mul3(fid, words(a), words(b), words(c), words(d), words(e), words(f), out)
        assert toint(out) == ((a * b + c * d + e * f) * RIinv) % p, it


@pytest.mark.parametrize("fid,F", [(4, pyref.P4), (6, pyref.P6)])
def test_mul_small_quotient_edges(shim, fid, F):
    """fp_mul_small's one-pass form estimates q = floor(K x / p) from the top limb: inputs on either side of every multiple of p / K,
    smallest and largest top limbs"""
    p = F.p
    out = (U * 24)()
    for op, K in ((8, 11), (9, 13)):
        xs = [0, 1, p - 1, p - 2, (1 << 725) - 1, 1 << 725, (p >> 725) << 725, ((p >> 725) << 725) - 1]
        for j in range(1, K):
            xs += [(j * p) // K + d for d in (-2, -1, 0, 1, 2)]
        for x in xs:
            x %= p
            shim.t_fp_op(fid, op, words(x), words(0), out)
            assert toint(out) == K * x % p, (K, hex(x))
    # the register-multiplier form (fp_mul_small_rt): every k it can be given, k = 1 must return x itself
    rng = pyref.Rng(3)
    for k in range(1, 16):
        for x in [0, 1, p - 1, (p >> 725) << 725] + [(j * p) // k + d for j in range(1, k) for d in (-1, 0, 1)] + [rng.field_elem(p) for _ in range(4)]:
            x %= p
            kb = (U * 24)(*([k] + [0] * 23))
            shim.t_fp_op(fid, 14, words(x), kb, out)
            assert toint(out) == k * x % p, (k, hex(x))


@pytest.mark.parametrize("fid,F", [(4, pyref.P4), (6, pyref.P6)])
def test_field_inverse_safegcd(shim, fid, F):
    """fp_inv / fp_inv_plain (Bernstein-Yang divsteps, 61 batches of 29) against Python's pow(x, -1, p):
    random values, small and large ones, powers of two, and 0 -> 0 (fp_768.rs:551-554 returns None there)."""
    p = F.p
    RI = pow(2, 754, p)
    rng = pyref.Rng(1234 + fid)
    vals = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 752, (1 << 752) - 1, 17, RI, pow(RI, -1, p)]
    vals += [1 << k for k in range(0, 752, 37)] + [p - (1 << k) for k in range(1, 752, 41)]
    vals += [rng.field_elem(p) for _ in range(1500)] + [rng.next_u64() + 1 for _ in range(50)]
    out = (U * 24)()
    for x in vals:
        shim.t_fp_op(fid, 13, words(x), words(0), out)
        assert toint(out) == (pow(x, -1, p) if x else 0), hex(x)
        shim.t_fp_op(fid, 12, words(x), words(0), out)          # Montgomery in / out: (a R) -> a^-1 R
        a = x * pow(RI, -1, p) % p
        assert toint(out) == (pow(a, -1, p) * RI % p if x else 0), hex(x)


@pytest.mark.parametrize("cid,name", list(enumerate(("mnt4753_g1", "mnt4753_g2", "mnt6753_g1", "mnt6753_g2"))))
def test_curve_ops(shim, cid, name):
    C = pyref.CURVES[name]
    F, E, k = C.F, C.E, C.deg
    rng = pyref.Rng(100 + cid)

    def abi_ext(e):
        x = 0
        for i, c in enumerate(e):
            x |= F.to_mont(c) << (768 * i)
        return words(x, 24 * k)

    def pack(parts):
        buf = (U * (24 * k * len(parts)))()
        for j, e in enumerate(parts):
            w = abi_ext(e)
            for i in range(24 * k):
                buf[j * 24 * k + i] = w[i]
        return buf

    def proj(P):
        if P is None:
            return pack((E.zero(), E.one(), E.zero()))
        z = tuple(rng.field_elem(F.p) for _ in range(k))
        return pack((E.mul(P[0], z), E.mul(P[1], z), z))

    def result(out):
        vals = []
        for j in range(3):
            vals.append(tuple(F.from_mont(toint(out[(j * k + c) * 24:(j * k + c + 1) * 24])) for c in range(k)))
        return C.proj_to_affine(*vals)

    P, Q = C.mul(12345, C.G), C.mul(987654321, C.G)
    assert C.on_curve(P) and C.on_curve(Q)
    out = (U * (72 * k))()
    for A, B in ((P, Q), (P, P), (P, C.neg(P)), (None, Q)):
        shim.t_ec_op(cid, 0, proj(A), pack(B), out)          # mixed add, incl. doubling and inverse branches
        assert result(out) == C.add(A, B)
        shim.t_ec_op(cid, 1, proj(A), proj(B), out)          # full add
        assert result(out) == C.add(A, B)
    shim.t_ec_op(cid, 1, proj(P), proj(None), out)
    assert result(out) == P
    shim.t_ec_op(cid, 2, proj(P), proj(None), out)
    assert result(out) == C.add(P, P)
    shim.t_ec_op(cid, 2, proj(None), proj(None), out)
    assert result(out) is None


@pytest.mark.parametrize("fid,F", [(4, pyref.P4), (6, pyref.P6)])
def test_dual_product_single_accumulator(shim, fid, F):
    """fp_mul2s: (a b + c d) 2^-754 on ONE 64-bit accumulator chain.  Its columns stay below 2^64 only because all operands are
    fully reduced; the operands with every limb at its maximum (2^29 - 1, top limb p_25 - 1: still below p) come closest."""
    p = F.p
    RIinv = pow(pow(2, 754, p), -1, p)
    top = p >> 725
    ones = sum(((1 << 29) - 1) << (29 * i) for i in range(25)) | ((top - 1) << 725)
    assert ones < p
    rng = pyref.Rng(90 + fid)
    out = (U * 24)()
    cases = [(ones, ones, ones, ones), (p - 1, p - 1, p - 1, p - 1), (ones, p - 1, p - 1, ones), (0, 0, 0, 0), (1, p - 1, p - 1, 1), (ones, 0, 0, ones)]
    cases += [tuple(rng.field_elem(p) for _ in range(4)) for _ in range(300)]
    # operands that drive the m digits high as well: a b + c d = -1 * 2^(29 k) patterns are not constructible directly; random
    # values with saturated low limbs cover the carry paths
    for _ in range(100):
        a, b, c, d = (rng.field_elem(p) | ((1 << 290) - 1) for _ in range(4))
        cases.append((a % p, b % p, c % p, d % p))
    for a, b, c, d in cases:
        shim.t_fp_mul2s(fid, words(a), words(b), words(c), words(d), out)
        assert toint(out) == ((a * b + c * d) * RIinv) % p


@pytest.mark.parametrize("cid,name", list(enumerate(("mnt4753_g1", "mnt4753_g2", "mnt6753_g1", "mnt6753_g2"))))
def test_xyzz_mixed_addition(shim, cid, name):
    """xyzz_madd / xyzz_to_proj (ec29.h: madd-2008-s on (X, Y, ZZ, ZZZ)) against the affine group law in Python: a chain of
    additions, the empty accumulator, P + (-P) -> infinity, and P == Q reported through `same`."""
    C = pyref.CURVES[name]
    F, E, k = C.F, C.E, C.deg
    rng = pyref.Rng(400 + cid)

    def pack(parts):
        buf = (U * (24 * k * len(parts)))()
        for j, e in enumerate(parts):
            for c in range(k):
                w = words(F.to_mont(e[c]))
                for i in range(24):
                    buf[(j * k + c) * 24 + i] = w[i]
        return buf

    def xyzz(Pt):
        if Pt is None:
            return pack((E.zero(), E.one(), E.zero(), E.zero()))
        z = tuple(rng.field_elem(F.p) for _ in range(k))
        zz = E.mul(z, z)
        zzz = E.mul(zz, z)
        return pack((E.mul(Pt[0], zz), E.mul(Pt[1], zzz), zz, zzz))

    def ext(out, j):
        return tuple(F.from_mont(toint(out[(j * k + c) * 24:(j * k + c + 1) * 24])) for c in range(k))

    def results(out):
        X, Y, ZZ, ZZZ = (ext(out, j) for j in range(4))
        if all(v == 0 for v in ZZ):
            a1 = None
        else:
            a1 = (E.mul(X, E.inv(ZZ)), E.mul(Y, E.inv(ZZZ)))
            assert E.mul(E.mul(ZZ, ZZ), ZZ) == E.mul(ZZZ, ZZZ)          # the invariant ZZ^3 = ZZZ^2
        a2 = C.proj_to_affine(ext(out, 4), ext(out, 5), ext(out, 6))
        assert a1 == a2
        return a1

    out = (U * (24 * k * 7))()
    P, Q = C.mul(424242, C.G), C.mul(777777777, C.G)
    for A, B in ((P, Q), (Q, P), (None, Q), (P, C.neg(P)), (C.add(P, Q), C.neg(Q))):
        assert shim.t_xyzz_madd(cid, xyzz(A), pack(B), out) == 0
        assert results(out) == C.add(A, B)
    assert shim.t_xyzz_madd(cid, xyzz(P), pack(P), out) == 1            # P == Q: reported, accumulator unchanged
    assert results(out) == P
    acc = None
    for j in range(6):                                                  # a running sum fed back in its own coordinates
        B = C.mul(1000 + 37 * j, C.G)
        buf = xyzz(acc)
        assert shim.t_xyzz_madd(cid, buf, pack(B), out) == 0
        acc = C.add(acc, B)
        assert results(out) == acc


@pytest.mark.parametrize("fid,F,curve", [(4, pyref.P4, "mnt4753_g1"), (6, pyref.P6, "mnt6753_g1")])
def test_host_fold_field(shim, fid, F, curve):
    """the 12 x u64 host field used by the window fold (host_math.h HF1) and G1 add/double on it"""
    import numpy as np
    p = F.p
    rng = pyref.Rng(55 + fid)
    U64 = ctypes.c_uint64

    def w64(x, n=12):
        return (U64 * n)(*[(x >> (64 * i)) & (2**64 - 1) for i in range(n)])

    def i64(w):
        return sum(int(v) << (64 * i) for i, v in enumerate(w))
    for it in range(200):
        a, b = rng.field_elem(p), rng.field_elem(p)
        if it == 0:
            a = 0
        if it == 1:
            a = b = p - 1
        out = (U64 * 12)()
        for op, e in ((0, a * b * F.Rinv % p), (2, (a + b) % p), (3, (a - b) % p), (4, -a % p), (5, 2 * a % p), (8, 11 * a % p)):
            shim.t_h64_op(fid, op, w64(a), w64(b), out)
            assert i64(out) == e, (op, it)
    C = pyref.CURVES[curve]
    P, Q = C.mul(777, C.G), C.mul(31337, C.G)

    def proj(Pt, z):
        if Pt is None:
            return w64(0 | (F.to_mont(1) << 768), 36)
        X, Y = (Pt[0][0] * z) % p, (Pt[1][0] * z) % p
        return w64(F.to_mont(X) | (F.to_mont(Y) << 768) | (F.to_mont(z) << 1536), 36)

    def aff(out):
        v = i64(out)
        X, Y, Z = (F.from_mont((v >> (768 * k)) & (2**768 - 1)) for k in range(3))
        return C.proj_to_affine((X,), (Y,), (Z,))
    out = (U64 * 36)()
    for A, B in ((P, Q), (P, P), (P, C.neg(P)), (None, Q), (P, None)):
        shim.t_h64_ec(fid, 0, proj(A, 5), proj(B, 9), out)
        assert aff(out) == C.add(A, B)
    shim.t_h64_ec(fid, 1, proj(P, 3), proj(P, 3), out)
    assert aff(out) == C.add(P, P)


@pytest.mark.parametrize("cid,name", [(1, "mnt4753_g2"), (3, "mnt6753_g2")])
def test_host_fold_towers(shim, cid, name):
    """G2 add / double on the 64-bit-limb host towers (host_math.h HF2 / HF3) used by the window fold"""
    C = pyref.CURVES[name]
    F, E, k = C.F, C.E, C.deg
    U64 = ctypes.c_uint64

    def proj(Pt, z):
        if Pt is None:
            X, Y, Z = E.zero(), E.one(), E.zero()
        else:
            X, Y, Z = E.mul(Pt[0], z), E.mul(Pt[1], z), z
        limbs = pyref.ext_to_abi(F, X) + pyref.ext_to_abi(F, Y) + pyref.ext_to_abi(F, Z)
        return (U64 * (36 * k))(*limbs)

    def aff(out):
        v = [int(x) for x in out]
        return C.proj_to_affine(pyref.ext_from_abi(F, v[:12 * k], k), pyref.ext_from_abi(F, v[12 * k:24 * k], k),
                                pyref.ext_from_abi(F, v[24 * k:], k))
    P, Q = C.mul(4242, C.G), C.mul(99991, C.G)
    z1, z2 = tuple(range(3, 3 + k)), tuple(range(11, 11 + k))
    out = (U64 * (36 * k))()
    for A, B in ((P, Q), (P, P), (P, C.neg(P)), (None, Q), (P, None)):
        shim.t_h64_ec_curve(cid, 0, proj(A, z1), proj(B, z2), out)
        assert aff(out) == C.add(A, B)
    shim.t_h64_ec_curve(cid, 1, proj(P, z1), proj(P, z1), out)
    assert aff(out) == C.add(P, P)
