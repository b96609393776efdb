"""The generated gfx950 assembly (ginger-lib_amd/asmgen) executed on the CPU by asmgen/sim.py -- no GPU, no assembler needed.

* the field routines (Montgomery product, square, dual product with one reduction, sub, conditional negation, zero test)
  against Python integers: the values fp29.h defines, which restate algebra/src/fields/models/fp_768.rs:1009-1185 (mul_assign +
  mont_reduce), :339-548 (square_in_place), :939-949 (sub_assign), :870-883 (neg);
* the whole G1 bucket-accumulation kernel on small task lists against the textbook group law (tests/pyref.py), including the
  cases the reference branches on (swp.rs:481-519): the sum is infinity, P + P (doubling: the salt detour, both salts), P - P,
  restart after infinity, empty lists -- and every load / store checked against the buffers it may touch.
"""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ginger-lib_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref                                                      # noqa: E402
from asmgen import g1_xyzz                                        # noqa: E402
from asmgen.field import Chain, FieldGen, NL, interleave, limbs, run, unlimbs   # noqa: E402
from asmgen.isa import Prog, S, V, module_text                    # noqa: E402
from asmgen.sim import Memory, Wave                               # noqa: E402

R = 1 << 754


def _slot(i):
    return V(40 + NL * i, NL)


def _put(w, s, vals):
    for l in range(64):
        for i, x in enumerate(limbs(vals[l])):
            w.V[s.idx + i][l] = x


def _get(w, s):
    return [unlimbs(w.V[s.idx:s.idx + NL, l]) for l in range(64)]


@pytest.mark.parametrize("tag", ["p4", "p6"])
def test_field_routines_in_the_simulator(tag):
    p = pyref.FIELDS[tag].p
    rnd = random.Random(5)
    g = Prog("t")
    f = FieldGen(g, p, 24, 50, 21, 20)
    chA = Chain(V(248, 2), V(252), V(253), S(14, 2), S(16, 2))
    chB = Chain(V(250, 2), V(254), V(255), S(18, 2), S(76, 2))
    f.load_constants()
    interleave(f.mul(chA, _slot(0), _slot(1), _slot(4), _slot(0)), f.mul(chB, _slot(2), _slot(3), _slot(5), _slot(2)))
    interleave(f.sqr(chA, _slot(1), _slot(6), _slot(0)), f.sqr(chB, _slot(3), _slot(7), _slot(2)))
    run(f.dual(chA, chB, _slot(4), _slot(5), _slot(0), _slot(2), _slot(6), _slot(7)))
    run(f.sub(chA, _slot(4), _slot(5), _slot(7)))
    run(f.sub(chB, _slot(5), _slot(4), _slot(1)))
    g.s_mov_b32(S(78), 0xF0F0F0F0)
    g.s_mov_b32(S(79), 0x0000FFFF)
    run(f.neg_sel(chA, _slot(3), V(39), S(78, 2)))
    run(f.is_zero_mask(chA, _slot(7), S(80, 2)))
    g.s_endpgm()
    a, b, c, d = ([rnd.randrange(p) for _ in range(64)] for _ in range(4))
    a[0] = b[0] = c[0] = d[0] = p - 1          # largest operands: the column bound of the single-reduction dual product
    a[1] = 0
    c[2] = 0
    a[3] = b[3] = 1
    c[5], d[5] = a[5], b[5]                    # equal products on lane 5: the difference is zero
    w = Wave(g, Memory())
    for s, v in ((0, a), (1, b), (2, c), (3, d)):
        _put(w, _slot(s), v)
    w.run()
    ri = pow(R, -1, p)
    e4 = [a[l] * b[l] * ri % p for l in range(64)]
    e5 = [c[l] * d[l] * ri % p for l in range(64)]
    e0 = [b[l] * b[l] * ri % p for l in range(64)]
    e2 = [d[l] * d[l] * ri % p for l in range(64)]
    assert _get(w, _slot(4)) == e4 and _get(w, _slot(5)) == e5
    assert _get(w, _slot(0)) == e0 and _get(w, _slot(2)) == e2
    assert _get(w, _slot(6)) == [(e4[l] * e5[l] + e0[l] * e2[l]) * ri % p for l in range(64)]
    e7 = [(e4[l] - e5[l]) % p for l in range(64)]
    assert _get(w, _slot(7)) == e7
    assert _get(w, _slot(1)) == [(e5[l] - e4[l]) % p for l in range(64)]
    msk = 0x0000FFFFF0F0F0F0
    assert _get(w, _slot(3)) == [((p - d[l]) if (msk >> l) & 1 else d[l]) for l in range(64)]
    assert w.rd_smask(S(80, 2)) == sum(1 << l for l in range(64) if e7[l] == 0) and (w.rd_smask(S(80, 2)) >> 5) & 1
    # 2 products, 2 squares, one dual product: exactly the multiplier instructions fp29.h counts
    assert w.hist["v_mad_u64_u32"] == 2 * 1352 + 2 * 1027 + 2028


def _run_kernel(cname, lists, pts, seed=0):
    C = pyref.CURVES[cname]
    p = C.F.p
    r = R % p
    ri = pow(r, -1, p)
    g = g1_xyzz.build("acc_" + cname, p, r)

    def enc(pt):
        return limbs(pt[0][0] * r % p) + limbs(pt[1][0] * r % p)
    mem = Memory()
    a_bases = mem.add("bases", np.array([enc(pt) for pt in pts], dtype=np.uint32))
    sorted_l, tasks = [], []
    for l in lists:
        tasks.append((len(sorted_l), len(l)))
        sorted_l += [i | (s << 31) for (i, s) in l]
    a_sorted = mem.add("sorted", np.array(sorted_l + [0], dtype=np.uint32))
    nt = len(lists)
    a_out = mem.add("out", np.zeros((nt, 78), dtype=np.uint32), writable=True)
    tk = np.zeros((nt, 4), dtype=np.uint32)
    for t, (b, c) in enumerate(tasks):
        d = a_out + t * 312
        tk[t] = (b, c, d & 0xFFFFFFFF, d >> 32)
    a_tasks = mem.add("tasks", tk)
    a_salts = mem.add("salts", np.array([enc(C.G), enc(C.add(C.G, C.G))], dtype=np.uint32))
    karg = np.zeros(10, dtype=np.uint32)
    for j, a in enumerate((a_bases, a_sorted, a_tasks, a_salts)):
        karg[2 * j], karg[2 * j + 1] = a & 0xFFFFFFFF, a >> 32
    karg[8] = nt
    a_karg = mem.add("karg", karg)
    for blk in range((nt + 255) // 256):
        for wv in range(4):
            if blk * 256 + wv * 64 >= nt:
                break
            w = Wave(g, mem, lds_words=g.lds_bytes // 4)
            w.S[0], w.S[1], w.S[2] = a_karg & 0xFFFFFFFF, a_karg >> 32, blk
            w.V[0] = np.arange(64, dtype=np.uint32) + 64 * wv
            w.lds[:] = 0xDEADBEEF
            w.run()
    out = mem.get("out").reshape(nt, 78)
    res = []
    for t in range(nt):
        xyz = [unlimbs(out[t, 26 * k:26 * k + 26]) for k in range(3)]
        assert max(xyz) < p, "unreduced output"
        X, Y, Z = (v * ri % p for v in xyz)
        res.append(((X, Y, Z), C.proj_to_affine((X,), (Y,), (Z,))))
    return g, res


@pytest.mark.parametrize("cname,ntasks", [("mnt4753_g1", 70), ("mnt6753_g1", 12)])
def test_g1_accumulation_kernel_in_the_simulator(cname, ntasks):
    C = pyref.CURVES[cname]
    rnd = random.Random(11)
    g1, g2 = C.G, C.add(C.G, C.G)
    h = C.mul(rnd.randrange(1, 1 << 60), C.G)
    pts = [g1, g2]
    pt = C.mul(rnd.randrange(1, 1 << 60), C.G)
    for _ in range(30):
        pts.append(pt)
        pt = C.add(pt, h)
    special = [
        [],                                              # empty list -> infinity
        [(5, 0)],
        [(5, 0), (5, 0)],                                # P + P: the detour through a salt point (swp.rs:492 doubles here)
        [(5, 0), (5, 1)],                                # P - P -> infinity
        [(5, 0), (5, 1), (7, 0)],                        # ... and a restart from infinity
        [(0, 0), (0, 0)],                                # G + G: the salt must be 2G
        [(1, 0), (1, 0), (3, 1)],                        # 2G + 2G: the salt must be G
        [(4, 1), (4, 1), (4, 1)],
        [(6, 0), (7, 0), (6, 1), (7, 1), (9, 0)],        # cancels after four entries
    ]
    lists = []
    for t in range(ntasks):
        if t < len(special):
            lists.append(special[t])
        else:
            lists.append([(rnd.randrange(len(pts)), rnd.randrange(2)) for _ in range(rnd.randrange(1, 6))])
    g, res = _run_kernel(cname, lists, pts)
    for t, l in enumerate(lists):
        exp = None
        for (i, s) in l:
            exp = C.add(exp, C.neg(pts[i]) if s else pts[i])
        xyz, got = res[t]
        assert got == exp, (t, l)
        if exp is None:
            assert xyz == (0, 1, 0)                      # GroupProjective::zero() (swp.rs:372-378)
    assert g.max_v == 256 and g.max_s <= 102 and g.max_a < 0     # two waves per SIMD, no AGPRs, inside the SGPR file


def test_generated_module_text_is_well_formed():
    p = pyref.FIELDS["p4"].p
    g = g1_xyzz.build("k", p, R % p)
    text, info = module_text([g])
    assert ".amdhsa_kernel k" in text and ".amdhsa_private_segment_fixed_size 0" in text
    assert info["k"]["vgprs"] == 256 and g.lds_bytes * 2 <= 160 * 1024          # two blocks per CU
    for line in text.splitlines():
        if "ds_read_b32" in line or "ds_write_b32" in line:
            assert int(line.split("offset:")[1].split()[0]) < 65536


def test_hazard_padding_finds_the_gfx950_cases_and_pads_them():
    """isa.fix_hazards: the wait states gfx940 / gfx950 do not interlock (LLVM GCNHazardRecognizer::checkVALUHazards,
    hasVDecCoExecHazard) -- VALU writes SGPR -> VALU reads it: 2; VALU writes VGPR -> v_readfirstlane reads it: 1.  The second one
    was met on the card: a kernel read its wave index as 0."""
    from asmgen.isa import hazard_scan, fix_hazards
    g = Prog("h")
    g.v_lshrrev_b32(V(15), 6, V(0))
    g.v_readfirstlane_b32(S(3), V(15))                  # needs 1 wait state
    g.v_cmp_eq_u32(S(96, 2), 0, V(4))
    g.v_cndmask_b32(V(6), V(16), V(17), S(96, 2))       # needs 2
    g.v_cmp_eq_u32(S(98, 2), 0, V(4))
    g.v_mov_b32(V(20), 1)
    g.v_cndmask_b32(V(7), V(16), V(17), S(98, 2))       # one instruction in between: 1 more
    g.v_sub_co_u32(V(1), S(16, 2), V(2), V(3))
    g.v_subb_co_u32(V(4), S(16, 2), V(5), V(6), S(16, 2))   # carry chain: 2
    g.v_cmp_eq_u32(S(90, 2), 0, V(4))
    g.s_and_b64(S(92, 2), S(90, 2), S(94, 2))           # SALU reads are interlocked: nothing
    g.v_mov_b32(V(30), 2)
    g.v_mov_b32(V(31), 3)
    g.v_cndmask_b32(V(8), V(16), V(17), S(90, 2))       # two instructions in between: nothing
    g.label("join")
    g.v_cndmask_b32(V(9), V(16), V(17), S(80, 2))       # first instruction behind a label: the worst predecessor is assumed
    g.s_endpgm()
    found, _ = hazard_scan(g, False)
    assert [(op, need) for _, op, need in found] == [("v_readfirstlane_b32", 1), ("v_cndmask_b32", 2), ("v_cndmask_b32", 1),
                                                     ("v_subb_co_u32", 2), ("v_cndmask_b32", 2)]
    n = fix_hazards(g)
    assert n == 5 and hazard_scan(g, False)[0] == []
    assert [i.args[0] for i in g.ins if i.op == "s_nop"] == [0, 1, 0, 1, 1]          # s_nop N = N + 1 wait states
    # every shipped kernel is hazard-free after its own padding
    from asmgen import build
    for prog in build.programs():
        assert hazard_scan(prog, False)[0] == [], prog.name


def test_simulator_refuses_a_register_with_a_result_in_flight():
    """sim.py: a register a load / ds instruction will write may not be touched before the s_waitcnt that covers it"""
    mem = Memory()
    a = mem.add("buf", np.arange(64, dtype=np.uint32))
    def prog(wait):
        g = Prog("w")
        g.v_lshlrev_b32(V(1), 2, V(0))
        g.global_load_dword(V(2), V(1), S(4, 2))
        if wait:
            g.s_waitcnt(vmcnt=0)
        g.v_add_u32(V(3), 1, V(2))
        g.s_endpgm()
        return g
    for wait in (True, False):
        w = Wave(prog(wait), mem)
        w.S[4], w.S[5] = a & 0xFFFFFFFF, a >> 32
        w.V[0] = np.arange(64, dtype=np.uint32)
        if wait:
            w.run()
            assert list(w.V[3]) == list(range(1, 65))
        else:
            with pytest.raises(RuntimeError, match="in flight"):
                w.run()


@pytest.mark.parametrize("tag", ["p4", "p6"])
def test_tower_field_routines_in_the_simulator(tag):
    """field.py routines of the G2 round kernels against Python integers: `triple` (three products with ONE reduction, fp29.h
    fp_mul3 = the per-lane form of fields/models/fp3.rs:453-477) with the largest reduced operands, `mul_small` (k x - q p, the
    non-residue factor: value = k x mod p or that plus p, k = 1 the identity) and the lane exchange `bperm`."""
    p = pyref.FIELDS[tag].p
    rnd = random.Random(7)
    g = Prog("t")
    f = FieldGen(g, p, 24, 50, 21, 20)
    chA = Chain(V(248, 2), V(252), V(253), S(14, 2), S(16, 2))
    chB = Chain(V(250, 2), V(254), V(255), S(18, 2), S(22, 2))
    sl = lambda i: V(20 + NL * i, NL)
    f.load_constants()
    lo, hi = f.invc_bits()
    g.s_mov_b32(S(76), lo)
    g.s_mov_b32(S(77), hi)
    run(f.triple(chA, chB, [(sl(0), sl(1)), (sl(2), sl(3)), (sl(4), sl(5))], sl(6), sl(7)))
    run(f.mul_small(chA, sl(0), V(19), sl(7), S(76, 2)))
    run(f.bperm(sl(5), V(18), sl(1)))
    g.s_waitcnt(lgkmcnt=0)
    g.s_endpgm()
    vals = [[rnd.randrange(p) for _ in range(64)] for _ in range(6)]
    for s_ in range(6):
        vals[s_][0] = p - 1                                     # the column bound and the 2.33 p value bound
    vals[0][1] = 0
    ks = [13 if (l & 1) == 0 else 1 for l in range(64)]
    ks[3] = 15
    vals[0][3] = p - 1
    w = Wave(g, Memory())
    for s_ in range(6):
        for l in range(64):
            for i, x in enumerate(limbs(vals[s_][l])):
                w.V[sl(s_).idx + i][l] = x
    for l in range(64):
        w.V[19][l] = ks[l]
        w.V[18][l] = 4 * (l ^ 1)
    w.run()
    get = lambda s_: [unlimbs(w.V[s_.idx:s_.idx + NL, l]) for l in range(64)]
    ri = pow(R, -1, p)
    assert get(sl(6)) == [(vals[0][l] * vals[1][l] + vals[2][l] * vals[3][l] + vals[4][l] * vals[5][l]) * ri % p for l in range(64)]
    for l, v in enumerate(get(sl(7))):
        assert v % p == ks[l] * vals[0][l] % p and v < p + 27 * (1 << 725) and (ks[l] != 1 or v == vals[0][l]), l
    assert get(sl(5)) == [vals[1][l ^ 1] for l in range(64)]
