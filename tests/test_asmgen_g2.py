"""The G2 affine-round kernels of ginger-lib_amd/asmgen/g2_rounds.py executed on the CPU by asmgen/sim.py -- no GPU needed.

Round 0 (table rows named by a signed list, gathered by both kernels) and a later round (inputs in the previous round's T64 list), forward
kernel -> tower inversion of the per-lane-group running products (Python here; aff_inv_kernel on the device) -> backward
kernel, on MNT4-753 G2 (lane pairs, Fq2) and MNT6-753 G2 (lane triples, Fq3): every output element against the textbook
affine group law of tests/pyref.py (the addition add_assign_mixed performs, swp.rs:481-519, in affine form), copies of odd
leftovers, a wave with a ragged tail, a wave with nothing to do, and the flag that hands a round with x1 == x2 or an
infinity marker to the C++ kernel.  Every load and store is checked against the buffers it may touch."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ginger-lib_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref                                                      # noqa: E402
from asmgen import g2_rounds as G2                                # noqa: E402
from asmgen.field import NL, limbs, unlimbs                       # noqa: E402
from asmgen.sim import Memory, Wave                               # noqa: E402

R = 1 << 754
_PROGS = {}


def _prog(cname, fwd, r0):
    key = (cname, fwd, r0)
    if key not in _PROGS:
        C = pyref.CURVES[cname]
        cfg = G2.Cfg(C.deg, int(C.E.nr or 1), C.F.p, R % C.F.p)
        _PROGS[key] = (G2.build("k_%s_%d%d" % (cname, fwd, r0), cfg, fwd, r0), cfg)
    return _PROGS[key]


class T64:
    """a T64 list: chunk c of lane slot s of tile t at ((t * nch + c) * 64 + s) * 16 bytes (aff_kernels.h)"""

    def __init__(self, tiles, nch, fill=0xDEADBEEF):
        self.nch = nch
        self.a = np.full((tiles, nch, 64, 4), fill, dtype=np.uint32)

    def words(self, t, s):
        return self.a[t, :, s, :].reshape(-1)

    def put_pt(self, t, s, x, y):
        w = self.a[t, :, s, :].reshape(-1)
        w[0:26] = limbs(x)
        w[26:52] = limbs(y)
        self.a[t, :, s, :] = w.reshape(self.nch, 4)

    def get_pt(self, t, s):
        w = self.words(t, s)
        return unlimbs(w[0:26]), unlimbs(w[26:52])

    def get_fp(self, t, s):
        return unlimbs(self.words(t, s)[0:26])

    def put_fp(self, t, s, v):
        w = self.a[t, :, s, :].reshape(-1)
        w[0:26] = limbs(v)
        self.a[t, :, s, :] = w.reshape(self.nch, 4)


ARG_ORDER = ("in", "sorted", "desc", "prefix", "stage1", "stage2", "out", "accs", "flag")
WRITABLE = ("prefix", "stage1", "stage2", "out", "accs", "flag")


def sim_runner(g, bufs, scalars, waves):
    """runs kernel g, one simulated wave at a time; bufs: name -> uint32 array (updated in place)"""
    mem = Memory()
    a = {k: mem.add(k, v, writable=k in WRITABLE) for k, v in bufs.items()}
    karg = np.zeros(22, dtype=np.uint32)
    for j, k in enumerate(ARG_ORDER):
        karg[2 * j], karg[2 * j + 1] = a[k] & 0xFFFFFFFF, a[k] >> 32
    karg[18], karg[19], karg[20] = scalars
    a_karg = mem.add("karg", karg)
    for wv in range(waves + 1):                                   # one wave more than the work: it must leave at once
        w = Wave(g, mem, lds_words=max(g.lds_bytes // 4, 1))
        w.S[0], w.S[1], w.S[2] = a_karg & 0xFFFFFFFF, a_karg >> 32, wv // 4
        w.V[0] = np.arange(64, dtype=np.uint32) + 64 * (wv % 4)
        w.lds[:] = 0xDEADBEEF
        w.run()
    for k in WRITABLE:
        bufs[k][...] = mem.get(k).reshape(bufs[k].shape)


def _run_round(cname, r0, n_out, B, waves, desc, rows=None, sorted_l=None, in_list=None, in_base=0, expect_bad=(),
               runner=sim_runner):
    C = pyref.CURVES[cname]
    p, L = C.F.p, C.deg
    gf, cfg = _prog(cname, True, r0)
    gb, _ = _prog(cname, False, r0)
    TPW = cfg.TPW
    tiles = waves * B + 1
    prefix, stage1, stage2, out = T64(tiles, 7), T64(tiles, 13), T64(tiles, 13), T64(tiles, 13)
    accs = T64(waves + 1, 7)
    bufs = {"prefix": prefix.a, "stage1": stage1.a, "stage2": stage2.a, "out": out.a, "accs": accs.a,
            "flag": np.zeros(16 + G2.FIX_CAP, dtype=np.uint32), "desc": np.array(desc, dtype=np.uint32),
            "sorted": np.array(sorted_l if sorted_l is not None else [0], dtype=np.uint32)}
    if r0:
        enc = np.zeros((len(rows), 2 * L, 26), dtype=np.uint32)
        for i, (x, y) in enumerate(rows):
            for j in range(L):
                enc[i, j] = limbs(x[j] * R % p)
                enc[i, L + j] = limbs(y[j] * R % p)
        bufs["in"] = enc
    else:
        bufs["in"] = in_list.a
    runner(gf, bufs, (n_out, in_base, B), waves)
    # the control block: word 0 = the whole round falls back (list overflow), word 1 = entries of the exception list (from word 16)
    ctl = bufs["flag"]
    assert int(ctl[0]) == 0 and int(ctl[1]) == len(expect_bad) and sorted(int(x) for x in ctl[16:16 + int(ctl[1])]) == sorted(expect_bad)
    # the inversion between the kernels (aff_inv_kernel on the device): per lane group, in the tower
    ri = pow(R, -1, p)
    for wv in range(waves):
        if wv * TPW * B >= n_out:
            continue
        for gidx in range(TPW):
            v = tuple(accs.get_fp(wv, gidx * L + j) * ri % p for j in range(L))
            assert not C.E.is_zero(v)                            # exceptions stay out of the running product
            iv = C.E.inv(v)
            for j in range(L):
                accs.put_fp(wv, gidx * L + j, iv[j] * R % p)
    runner(gb, bufs, (n_out, in_base, B), waves)
    assert int(ctl[1]) == len(expect_bad)
    return out, stage1, stage2, gf, gb


def _tower_pt(lst, t, g, L, p):
    ri = pow(R, -1, p)
    xs, ys = [], []
    for j in range(L):
        x, y = lst.get_pt(t, g * L + j)
        assert x < p and y < p, "unreduced output"
        xs.append(x * ri % p)
        ys.append(y * ri % p)
    return (tuple(xs), tuple(ys))


@pytest.mark.parametrize("cname", ["mnt4753_g2", "mnt6753_g2", "mnt6753_g1"])
def test_g2_round_kernels_in_the_simulator(cname):
    C = pyref.CURVES[cname]
    p, L = C.F.p, C.deg
    TPW = 64 // L
    rnd = random.Random(3)
    h = C.mul(rnd.randrange(1, 1 << 40), C.G)
    pts, pt = [], C.mul(rnd.randrange(1, 1 << 40), C.G)
    for _ in range(12):
        pts.append(pt)
        pt = C.add(pt, h)
    # ---- round 0: two waves of B = 2 elements per lane group, the second wave ragged
    B, waves = 2, 2
    n_out = 2 * TPW * B - 5
    sorted_l, desc, exp = [], [], []
    for o in range(n_out):
        pair = rnd.randrange(4) != 0
        i1, s1 = rnd.randrange(len(pts)), rnd.randrange(2)
        i2, s2 = rnd.randrange(len(pts)), rnd.randrange(2)
        while i2 == i1:
            i2 = rnd.randrange(len(pts))
        desc.append(len(sorted_l) | (0x80000000 if pair else 0))
        sorted_l.append(i1 | (s1 << 31))
        P1 = C.neg(pts[i1]) if s1 else pts[i1]
        if pair:
            sorted_l.append(i2 | (s2 << 31))
            exp.append(C.add(P1, C.neg(pts[i2]) if s2 else pts[i2]))
        else:
            exp.append(P1)
    out0, st1, st2, gf, gb = _run_round(cname, True, n_out, B, waves, desc, rows=pts, sorted_l=sorted_l + [0])
    for o in range(n_out):
        wv, rem = divmod(o, TPW * B)
        k, g = divmod(rem, TPW)
        assert _tower_pt(out0, wv * B + k, g, L, p) == exp[o], o
    for g_ in (gf, gb):
        assert g_.max_v <= 256 and g_.max_s <= 102 and g_.max_a < 0
    assert gb.lds_bytes * 2 <= 160 * 1024 and gf.lds_bytes == 0
    # ---- a later round over that output: list indices are global (in_base), pairs of neighbours and copies
    in_base = 1000
    n_in = n_out
    desc2, exp2, i = [], [], 0
    while i < n_in:
        if i + 1 < n_in and rnd.randrange(5) != 0 and exp[i][0] != exp[i + 1][0]:      # (equal x is the flagged case, below)
            desc2.append((in_base + i) | 0x80000000)
            exp2.append(C.add(exp[i], exp[i + 1]))
            i += 2
        else:
            desc2.append(in_base + i)
            exp2.append(exp[i])
            i += 1
    # the input list of a chunk starts at tile 0: re-pack out0 (its tiles are per wave) densely
    dense = T64((n_in + TPW - 1) // TPW + 1, 13)
    for o in range(n_in):
        wv, rem = divmod(o, TPW * B)
        k, g = divmod(rem, TPW)
        for j in range(L):
            x, y = out0.get_pt(wv * B + k, g * L + j)
            dense.put_pt(o // TPW, (o % TPW) * L + j, x, y)
    n2 = len(desc2)
    B2 = 3
    waves2 = (n2 + TPW * B2 - 1) // (TPW * B2)
    out1, _, _, _, _ = _run_round(cname, False, n2, B2, waves2, desc2, in_list=dense, in_base=in_base)
    for o in range(n2):
        wv, rem = divmod(o, TPW * B2)
        k, g = divmod(rem, TPW)
        assert _tower_pt(out1, wv * B2 + k, g, L, p) == exp2[o], o


@pytest.mark.parametrize("cname", ["mnt4753_g2", "mnt6753_g2", "mnt4753_g1"])
def test_g2_round_kernels_list_the_rare_cases(cname):
    """P + P (the reference's doubling branch, swp.rs:492), P - P and an infinity marker among a pair's inputs do not enter the
    shared inversion: the forward kernel appends such elements to the exception list (aff_fix_kernel recomputes them on the
    device), both kernels treat them as copies, and every other element of the round is still the group law's sum."""
    C = pyref.CURVES[cname]
    p, L = C.F.p, C.deg
    TPW = 64 // L
    pts = [C.G, C.add(C.G, C.G), C.mul(5, C.G), C.mul(7, C.G)]
    desc = [0 | 0x80000000, 2 | 0x80000000, 4 | 0x80000000, 6, 7 | 0x80000000]
    sorted_l = [0, 1, 2, 2, 3, 3 | 0x80000000, 1, 0, 2, 0]            # G + 2G | 5G + 5G | 7G - 7G | copy 2G | G + 5G
    out, _, _, _, _ = _run_round(cname, True, 5, 1, 1, desc, rows=pts, sorted_l=sorted_l, expect_bad=(1, 2))
    assert _tower_pt(out, 0, 0, L, p) == C.mul(3, C.G)
    assert _tower_pt(out, 0, 3, L, p) == pts[1]
    assert _tower_pt(out, 0, 4, L, p) == C.mul(6, C.G)
    assert _tower_pt(out, 0, 1, L, p) == pts[2]                       # exceptions leave as copies of their first input
    # a later round with an infinity marker (x.l[0] = 0xFFFFFFFF) among its inputs
    lst = T64(2, 13)
    vals = [C.G, None, C.mul(5, C.G), C.mul(7, C.G), C.mul(9, C.G), None]
    for o, v in enumerate(vals):
        for j in range(L):
            if v is None:
                lst.put_pt(0, o * L + j, 0, 0)
                lst.words(0, o * L + j)      # (view)
                lst.a[0, 0, o * L + j, 0] = 0xFFFFFFFF
            else:
                lst.put_pt(0, o * L + j, v[0][j] * R % p, v[1][j] * R % p)
    desc = [0 | 0x80000000, 2 | 0x80000000, 4, 5]                    # G + marker | 5G + 7G | copy 9G | copy marker
    out, _, _, _, _ = _run_round(cname, False, 4, 1, 1, desc, in_list=lst, expect_bad=(0,))
    assert _tower_pt(out, 0, 1, L, p) == C.mul(12, C.G)
    assert _tower_pt(out, 0, 2, L, p) == C.mul(9, C.G)
    assert int(out.a[0, 0, 3 * L, 0]) == 0xFFFFFFFF                   # a copied marker stays a marker
