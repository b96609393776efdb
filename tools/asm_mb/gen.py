"""Micro-benchmarks of the assembly building blocks (ginger-lib_amd/asmgen): every kernel repeats ONE block `iters` times on
the register plan of the G1 accumulation kernel, at two waves per SIMD (LDS pinned to 79 872 B per block).  The driver
(tools/asm_mb/run.hip) loads the code object and times each kernel:  python tools/asm_mb/gen.py OUTDIR
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ginger-lib_amd"))
from asmgen.isa import Prog, V, S, VCC, EXEC, OFF, module_text, fix_hazards          # noqa: E402
from asmgen.field import FieldGen, Chain, interleave, run, NL, LM      # noqa: E402
from asmgen import g1_xyzz                                             # noqa: E402
from asmgen.build import P4, LLVM                                      # noqa: E402

E = g1_xyzz.E
BLOCKS = {}


def block(name):
    def deco(fn):
        BLOCKS[name] = fn
        return fn
    return deco


@block("mul_pair")
def _(g, f, A, B):
    interleave(f.mul(A, E[2], E[0], E[4], E[2]), f.mul(B, E[3], E[1], E[5], E[3]))
    return 2


@block("mul_pair_noreduce")
def _(g, f, A, B):
    interleave(f.mul(A, E[2], E[0], E[4], E[2], reduce=False), f.mul(B, E[3], E[1], E[5], E[3], reduce=False))
    return 2


@block("mul_single")
def _(g, f, A, B):
    run(f.mul(A, E[2], E[0], E[4], E[2]))
    return 1


@block("mul_seq2")
def _(g, f, A, B):
    run(f.mul(A, E[2], E[0], E[4], E[2]))
    run(f.mul(B, E[3], E[1], E[5], E[3]))
    return 2


@block("sqr_pair")
def _(g, f, A, B):
    interleave(f.sqr(A, E[4], E[2], E[6]), f.sqr(B, E[5], E[3], E[7]))
    return 2


@block("dual")
def _(g, f, A, B):
    run(f.dual(A, B, E[5], E[2], E[6], E[4], E[7], E[6]))
    return 1


@block("sub_pair")
def _(g, f, A, B):
    interleave(f.sub(A, E[4], E[2], E[4]), f.sub(B, E[5], E[3], E[5]))
    return 2


@block("sub_single")
def _(g, f, A, B):
    run(f.sub(A, E[4], E[2], E[4]))
    return 1


@block("condsub_pair")
def _(g, f, A, B):
    interleave(f.cond_sub(A, E[4], E[2], E[4]), f.cond_sub(B, E[5], E[3], E[5]))
    return 2


@block("neg_sel")
def _(g, f, A, B):
    run(f.neg_sel(A, E[3], V(39), S(78, 2)))
    return 1


@block("mads_only_pair")
def _(g, f, A, B):
    # 2 x 1352 mads alternating on two accumulators, nothing else
    for k in range(1352):
        g.v_mad_u64_u32(A.acc, A.sdum, E[2].sub(k % 26), E[0].sub((k * 7) % 26), A.acc)
        g.v_mad_u64_u32(B.acc, B.sdum, E[3].sub(k % 26), E[1].sub((k * 7) % 26), B.acc)
    return 2


@block("mads_only_pair_sgpr")
def _(g, f, A, B):
    for k in range(1352):
        g.v_mad_u64_u32(A.acc, A.sdum, E[2].sub(k % 26), f.sP(k % 26), A.acc)
        g.v_mad_u64_u32(B.acc, B.sdum, E[3].sub(k % 26), f.sP((k + 3) % 26), B.acc)
    return 2


@block("mads_only_pair_vcc")
def _(g, f, A, B):
    for k in range(1352):
        g.v_mad_u64_u32(A.acc, VCC, E[2].sub(k % 26), E[0].sub((k * 7) % 26), A.acc)
        g.v_mad_u64_u32(B.acc, VCC, E[3].sub(k % 26), E[1].sub((k * 7) % 26), B.acc)
    return 2


@block("mads_only_single")
def _(g, f, A, B):
    for k in range(2704):
        g.v_mad_u64_u32(A.acc, A.sdum, E[2].sub(k % 26), E[0].sub((k * 7) % 26), A.acc)
    return 2


@block("mads_only_quad")
def _(g, f, A, B):
    # four accumulators
    accs = [A.acc, B.acc, V(30, 2), V(32, 2)]
    for k in range(676):
        for j, a in enumerate(accs):
            g.v_mad_u64_u32(a, A.sdum, E[2 + (j & 1)].sub(k % 26), E[j & 1].sub((k * 7) % 26), a)
    return 2


@block("close_low_pair")
def _(g, f, A, B):
    # 26 column closings (mul_lo, and, mad, alignbit, lshr) per chain, interleaved; x 10 to make the block long enough
    for _ in range(10):
        for k in range(NL):
            interleave(f._close_low(A, E[4], k), f._close_low(B, E[5], k))
    return 20


@block("park_roundtrip")
def _(g, f, A, B):
    for _ in range(10):
        for w in range(NL):
            g.ds_write_b32(V(1), E[4].sub(w), offset=w * 1024)
        for w in range(NL):
            g.ds_read_b32(E[5].sub(w), V(1), offset=w * 1024)
        g.s_waitcnt(lgkmcnt=0)
    return 10


@block("sqr_seq2")
def _(g, f, A, B):
    run(f.sqr(A, E[4], E[2], E[6]))
    run(f.sqr(B, E[5], E[3], E[7]))
    return 2


@block("mul_seq2_sameA")
def _(g, f, A, B):
    run(f.mul(A, E[2], E[0], E[4], E[2]))
    run(f.mul(A, E[3], E[1], E[5], E[3]))
    return 2


@block("mul_pair_samedum")
def _(g, f, A, B):
    B2 = Chain(B.acc, B.t0, B.t1, A.sdum, B.scar)
    interleave(f.mul(A, E[2], E[0], E[4], E[2]), f.mul(B2, E[3], E[1], E[5], E[3]))
    return 2


@block("mul_pair_oddbank")
def _(g, f, A, B):
    # chain B's three slots start at registers = 1 mod 4 (chain A's at 0 mod 4)
    a, b, m = V(117, 26), V(65, 26), V(169, 26)
    interleave(f.mul(A, E[2], E[0], E[4], E[2], reduce=False), f.mul(B, a, b, m, a, reduce=False))
    return 2


@block("mul_pair_sharedb")
def _(g, f, A, B):
    # both products share the operand b (as PPP || ZZ3 do)
    interleave(f.mul(A, E[2], E[0], E[4], E[2], reduce=False), f.mul(B, E[3], E[0], E[5], E[3], reduce=False))
    return 2


@block("mul_pair_percolumn")
def _(g, f, A, B):
    # coarser interleave: chain A does column k, then chain B does column k
    ga = f.mont_columns(A, f.mul_terms(E[2], E[0]), E[4])
    gb = f.mont_columns(B, f.mul_terms(E[3], E[1]), E[5])
    def ncol(k):
        lo = max(0, k - NL + 1); hi = min(k, NL - 1)
        n = hi - lo + 1
        n += (k if k < NL else 2 * NL - 1 - k)
        n += 5 if k < NL else (1 if k == 2 * NL - 1 else 3)
        return n
    for k in range(2 * NL):
        for gen in (ga, gb):
            for _ in range(ncol(k)):
                next(gen)
    for gen in (ga, gb):
        for _ in gen:
            raise RuntimeError("column count mismatch")
    return 2


@block("mul_seq2_noreduce")
def _(g, f, A, B):
    run(f.mul(A, E[2], E[0], E[4], E[2], reduce=False))
    run(f.mul(B, E[3], E[1], E[5], E[3], reduce=False))
    return 2


# ---- building blocks of the G2 round kernels (asmgen/g2_rounds.py).  kernel() sets up: v8 = 4 (lane ^ 1), v10 / v11 = 4 x the
#      even / odd lane of the pair, v9 / v12 = 4 x other lanes of a triple, v6 / v7 = 13 on even lanes else 1, s[88:89] = 1 / (p_25 + 1)
@block("g2_triple")
def _(g, f, A, B):
    run(f.triple(A, B, [(E[0], E[3]), (E[1], E[4]), (E[2], E[5])], E[6], E[7]))
    return 1


@block("g2_mul_small_single")
def _(g, f, A, B):
    run(f.mul_small(A, E[0], V(6), E[6], S(88, 2)))
    return 1


@block("g2_mul_small_pair")
def _(g, f, A, B):
    interleave(f.mul_small(A, E[0], V(6), E[6], S(88, 2)), f.mul_small(B, E[1], V(7), E[7], S(88, 2)))
    return 2


@block("g2_bperm3_wait")
def _(g, f, A, B):
    run(f.bperm(E[5], V(10), E[0]))
    run(f.bperm(E[6], V(11), E[0]))
    run(f.bperm(E[7], V(8), E[1]))
    g.s_waitcnt(lgkmcnt=0)
    return 1


@block("g2_tower2")
def _(g, f, A, B):
    # one Fq2 tower product as the round kernels issue it: broadcasts of b, swap + scale of a, dual product
    run(f.bperm(E[5], V(10), E[2]))
    run(f.bperm(E[6], V(11), E[2]))
    run(f.bperm(E[3], V(8), E[0]))
    g.s_waitcnt(lgkmcnt=0)
    run(f.mul_small(A, E[3], V(6), E[3], S(88, 2)))
    g.s_waitcnt(lgkmcnt=0)
    run(f.dual(A, B, E[0], E[5], E[3], E[6], E[1], E[5]))
    return 1


@block("g2_tower3")
def _(g, f, A, B):
    for dst, addr in ((E[5], V(10)), (E[6], V(11)), (E[7], V(12))):
        run(f.bperm(dst, addr, E[2]))
    run(f.bperm(E[3], V(8), E[0]))
    run(f.bperm(E[4], V(9), E[0]))
    g.s_waitcnt(lgkmcnt=0)
    interleave(f.mul_small(A, E[3], V(6), E[3], S(88, 2)), f.mul_small(B, E[4], V(7), E[4], S(88, 2)))
    g.s_waitcnt(lgkmcnt=0)
    run(f.triple(A, B, [(E[0], E[5]), (E[3], E[6]), (E[4], E[7])], E[1], E[5]))
    return 1


@block("g2_dual_seq5_subs")
def _(g, f, A, B):
    # the arithmetic of one backward iteration without memory, exchanges or parks: 5 dual products + 6 subtractions
    for _ in range(5):
        run(f.dual(A, B, E[0], E[5], E[3], E[6], E[1], E[5]))
    for _ in range(6):
        run(f.sub(A, E[4], E[2], E[4]))
    return 1


# ---- single instruction classes (independent instructions on eight register sets: what one instruction costs to issue)
def _class_block(emit, n=2048):
    def blk(g, f, A, B):
        for k in range(n):
            emit(g, f, k)
        return 1
    return blk


BLOCKS["ic_mad_u64"] = _class_block(lambda g, f, k: g.v_mad_u64_u32(V(30 + 2 * (k % 4), 2), S(14, 2), E[2].sub(k % 26), E[0].sub((k * 7) % 26), V(30 + 2 * (k % 4), 2)))
BLOCKS["ic_mul_lo"] = _class_block(lambda g, f, k: g.v_mul_lo_u32(V(30 + (k % 8)), E[2].sub(k % 26), E[0].sub((k * 7) % 26)))
BLOCKS["ic_mul_lo_sgpr"] = _class_block(lambda g, f, k: g.v_mul_lo_u32(V(30 + (k % 8)), E[2].sub(k % 26), S(g1_xyzz.S_INV)))
BLOCKS["ic_lshr64"] = _class_block(lambda g, f, k: g.v_lshrrev_b64(V(30 + 2 * (k % 4), 2), 29, V(E[2].idx + 2 * (k % 12), 2)))
BLOCKS["ic_alignbit"] = _class_block(lambda g, f, k: g.v_alignbit_b32(V(30 + (k % 8)), E[2].sub(k % 26), E[0].sub((k * 7) % 26), 29))
BLOCKS["ic_and"] = _class_block(lambda g, f, k: g.v_and_b32(V(30 + (k % 8)), S(g1_xyzz.S_LM), E[0].sub((k * 7) % 26)))
BLOCKS["ic_add3"] = _class_block(lambda g, f, k: g.v_add3_u32(V(30 + (k % 8)), E[2].sub(k % 26), E[0].sub((k * 7) % 26), E[1].sub(k % 26)))
BLOCKS["ic_lshl_add_u64"] = _class_block(lambda g, f, k: g.v_lshl_add_u64(V(30 + 2 * (k % 4), 2), V(E[2].idx + 2 * (k % 12), 2), 0, V(E[3].idx + 2 * (k % 12), 2)))
BLOCKS["ic_cndmask"] = _class_block(lambda g, f, k: g.v_cndmask_b32(V(30 + (k % 8)), E[2].sub(k % 26), E[0].sub((k * 7) % 26), S(78, 2)))
BLOCKS["ic_bfe"] = _class_block(lambda g, f, k: g.v_bfe_u32(V(30 + (k % 8)), E[2].sub(k % 26), 3, 29))


@block("close_low_pair_mad")
def _(g, f, A, B):
    # the column closing with the m digit from a v_mad_u64_u32 (low word) instead of v_mul_lo_u32, and one 64-bit shift
    def close(ch, m, k, tmp):
        g.v_mad_u64_u32(tmp, ch.sdum, ch.acc.lo(), S(f.s_inv), 0); yield
        g.v_and_b32(m.sub(k), S(f.s_lm), tmp.lo()); yield
        g.v_mad_u64_u32(ch.acc, ch.sdum, m.sub(k), f.sP(0), ch.acc); yield
        g.v_lshrrev_b64(ch.acc, 29, ch.acc); yield
    for _ in range(10):
        for k in range(NL):
            interleave(close(A, E[4], k, V(30, 2)), close(B, E[5], k, V(32, 2)))
    return 20


class _OldCloseGen(FieldGen):
    """field.py before the 64-bit shifts: v_alignbit_b32 + v_lshrrev_b32 per column (A/B block)"""
    def _close_low(self, ch, m, k):
        g = self.g
        g.v_mul_lo_u32(ch.t0, ch.acc.lo(), S(self.s_inv)); yield
        g.v_and_b32(m.sub(k), S(self.s_lm), ch.t0); yield
        g.v_mad_u64_u32(ch.acc, ch.sdum, m.sub(k), self.sP(0), ch.acc); yield
        g.v_alignbit_b32(ch.acc.lo(), ch.acc.hi(), ch.acc.lo(), 29); yield
        g.v_lshrrev_b32(ch.acc.hi(), 29, ch.acc.hi()); yield

    def _close_high(self, ch, r, k, last_unmasked=False):
        g = self.g
        if k == 2 * NL - 1:
            g.v_and_b32(r.sub(k - NL), S(self.s_lm), ch.acc.lo()); yield
            return
        g.v_and_b32(r.sub(k - NL), S(self.s_lm), ch.acc.lo()); yield
        g.v_alignbit_b32(ch.acc.lo(), ch.acc.hi(), ch.acc.lo(), 29); yield
        g.v_lshrrev_b32(ch.acc.hi(), 29, ch.acc.hi()); yield


@block("mul_pair_oldclose")
def _(g, f, A, B):
    fo = _OldCloseGen(g, f.p, f.s_p, f.s_np, f.s_inv, f.s_lm)
    interleave(fo.mul(A, E[2], E[0], E[4], E[2]), fo.mul(B, E[3], E[1], E[5], E[3]))
    return 2


@block("mul_single_oldclose")
def _(g, f, A, B):
    fo = _OldCloseGen(g, f.p, f.s_p, f.s_np, f.s_inv, f.s_lm)
    run(fo.mul(A, E[2], E[0], E[4], E[2]))
    return 1


@block("dual_one_chain")
def _(g, f, A, B):
    # a b + c d on ONE accumulator: every mad takes its addend from its predecessor
    a, b, c, d, m, dd = E[5], E[2], E[6], E[4], E[7], E[6]
    def terms(k):
        lo, hi = max(0, k - NL + 1), min(k, NL - 1)
        return [(a.sub(i), b.sub(k - i)) for i in range(lo, hi + 1)] + [(c.sub(i), d.sub(k - i)) for i in range(lo, hi + 1)]
    run(f.mont_columns(A, terms, m))
    run(f.cond_sub(A, m, dd, m))
    return 1


@block("g2_triple_seq")
def _(g, f, A, B):
    # the triple product with its two chains issued one after the other inside a column instead of alternating
    import types
    pairs = [(E[0], E[3]), (E[1], E[4]), (E[2], E[5])]
    m, dd = E[6], E[7]
    (a, b), (c, d), (e, ff) = pairs
    X, Y = A.acc, B.acc
    fx = True
    for k in range(2 * NL):
        lo, hi = max(0, k - NL + 1), min(k, NL - 1)
        xs, ys = [], []
        for i in range(lo, hi + 1):
            xs.append((a.sub(i), b.sub(k - i))); xs.append((c.sub(i), d.sub(k - i))); ys.append((e.sub(i), ff.sub(k - i)))
        mlo = 0 if k < NL else k - NL + 1
        mhi = k - 1 if k < NL else NL - 1
        for i in range(mlo, mhi + 1):
            ys.append((m.sub(i), f.sP(k - i)))
        fy = True
        for x, y in xs:
            g.v_mad_u64_u32(X, A.sdum, x, y, 0 if fx else X); fx = False
        for x, y in ys:
            g.v_mad_u64_u32(Y, B.sdum, x, y, 0 if fy else Y); fy = False
        if k == 2 * NL - 1:
            g.v_and_b32(m.sub(NL - 1), S(f.s_lm), X.lo())
            g.v_lshrrev_b32(B.t1, 29, X.lo())
            break
        g.v_and_b32(A.t0, S(f.s_lm), X.lo())
        g.v_lshrrev_b64(X, 29, X)
        g.v_mad_u64_u32(Y, B.sdum, A.t0, 1, Y)
        if k < NL:
            g.v_mul_lo_u32(A.t0, Y.lo(), S(f.s_inv))
            g.v_and_b32(m.sub(k), S(f.s_lm), A.t0)
            g.v_mad_u64_u32(Y, B.sdum, m.sub(k), f.sP(0), Y)
        else:
            g.v_and_b32(m.sub(k - NL), S(f.s_lm), Y.lo())
        g.v_lshrrev_b64(Y, 29, Y)
        g.v_lshl_add_u64(X, Y, 0, X)
    return 1


def _phase_block(pa, pb, pm, beta, kind="mul"):
    # slots with base register = pa / pb / pm (mod 4), accumulator pair at bank beta
    bases = {0: [40, 68, 96], 2: [126, 154, 182], 1: [41 + 170, 0, 0], 3: [0, 0, 0]}
    def blk(g, f, A, B):
        used = {0: 0, 2: 0}
        def take(ph):
            r = V(bases[ph][used[ph]], 26); used[ph] += 1
            return r
        a, b, m = take(pa), take(pb), take(pm)
        ch = Chain(V(248 + beta, 2), V(252), V(253), S(14, 2), S(16, 2))
        if kind == "mul":
            run(f.mul(ch, a, b, m, a, reduce=False))
        else:
            for i in range(NL):
                g.v_lshlrev_b32(b.sub(i), 1, a.sub(i))
            run(f.mont_columns(ch, f.sqr_terms(a, b), m))
        return 1
    return blk


for _pa in (0, 2):
    for _pb in (0, 2):
        for _pm in (0, 2):
            for _beta in (0, 2):
                BLOCKS["ph_mul_a%d_b%d_m%d_acc%d" % (_pa, _pb, _pm, _beta)] = _phase_block(_pa, _pb, _pm, _beta)
for _pa in (0, 2):
    for _pm in (0, 2):
        for _beta in (0, 2):
            BLOCKS["ph_sqr_a%d_a2%d_m%d_acc%d" % (_pa, _pa ^ 2, _pm, _beta)] = _phase_block(_pa, _pa ^ 2, _pm, _beta, "sqr")
            BLOCKS["ph_sqr_a%d_a2%d_m%d_acc%d" % (_pa, _pa, _pm, _beta)] = _phase_block(_pa, _pa, _pm, _beta, "sqr")


def _rep_block(n):
    def blk(g, f, A, B):
        for _ in range(n):
            interleave(f.mul(A, E[2], E[0], E[4], E[2]), f.mul(B, E[3], E[1], E[5], E[3]))
        return 2 * n
    return blk


for _n in (1, 2, 3, 4, 5, 6, 8):
    BLOCKS["codesize_mulpair_x%d" % _n] = _rep_block(_n)


def kernel(name, fn):
    g = Prog("mb_" + name)
    g.lds_bytes = g1_xyzz.LDS_BYTES
    g.add_arg(4, "val")
    g.add_arg(4, "val")
    f = FieldGen(g, P4, g1_xyzz.S_P, g1_xyzz.S_NP, g1_xyzz.S_INV, g1_xyzz.S_LM)
    A = Chain(V(248, 2), V(252), V(253), S(14, 2), S(16, 2))
    B = Chain(V(250, 2), V(254), V(255), S(18, 2), S(22, 2))
    g.s_load_dword(S(3), S(0, 2), 0)
    f.load_constants()
    g.v_lshlrev_b32(V(1), 2, V(0))
    g.s_mov_b32(S(78), 0x0F0F3355); g.s_mov_b32(S(79), 0xAAAA00FF)
    lo_, hi_ = f.invc_bits()
    g.s_mov_b32(S(88), lo_); g.s_mov_b32(S(89), hi_)
    g.v_and_b32(V(34), 63, V(0))
    g.v_xor_b32(V(35), 1, V(34)); g.v_lshlrev_b32(V(8), 2, V(35))
    g.v_and_b32(V(35), 62, V(34)); g.v_lshlrev_b32(V(10), 2, V(35))
    g.v_or_b32(V(35), 1, V(34)); g.v_lshlrev_b32(V(11), 2, V(35))
    g.v_xor_b32(V(35), 2, V(34)); g.v_lshlrev_b32(V(9), 2, V(35)); g.v_lshlrev_b32(V(12), 2, V(35))
    g.v_and_b32(V(35), 1, V(34)); g.v_cmp_eq_u32(S(80, 2), 0, V(35))
    g.v_mov_b32(V(36), 1); g.v_mov_b32(V(37), 13)
    g.v_cndmask_b32(V(6), V(36), V(37), S(80, 2)); g.v_cndmask_b32(V(7), V(36), V(37), S(80, 2))
    # operands: limbs derived from the lane id, below 2^29, top limb small so that values stay below p
    for i in range(8):
        for w in range(NL):
            g.v_mov_b32(V(34), 0x1234567 + 977 * w + i)
            g.v_mul_u32_u24(V(35), 0x9E37 + 131 * i + w, V(0))
            g.v_add_u32(E[i].sub(w), V(34), V(35))
            g.v_and_b32(E[i].sub(w), (LM >> 3) if w == NL - 1 else LM, E[i].sub(w))
    g.v_mov_b32(A.acc.lo(), 0); g.v_mov_b32(A.acc.hi(), 0); g.v_mov_b32(B.acc.lo(), 0); g.v_mov_b32(B.acc.hi(), 0)
    g.v_mov_b32(V(30), 0); g.v_mov_b32(V(31), 0); g.v_mov_b32(V(32), 0); g.v_mov_b32(V(33), 0)
    g.s_waitcnt(lgkmcnt=0)
    L = g.uniq("loop")
    g.label(L)
    n0 = g.count()
    units = fn(g, f, A, B)
    g.body_instr = g.count() - n0
    g.s_sub_u32(S(3), S(3), 1)
    g.s_cmp_lg_u32(S(3), 0)
    Lx = g.uniq("exit")
    g.s_cbranch_scc0(Lx)
    g.long_branch(L, S(94, 2))
    g.label(Lx)
    g.s_endpgm()
    fix_hazards(g)
    return g, units


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    progs, meta = [], []
    only = os.environ.get('MB_ONLY')
    for name, fn in BLOCKS.items():
        if only and not any(name == o or (o.endswith('*') and name.startswith(o[:-1])) for o in only.split(',')):
            continue
        g, units = kernel(name, fn)
        progs.append(g)
        meta.append("%s %d %d" % (g.name, units, g.body_instr))
    text, _ = module_text(progs)
    s = os.path.join(outdir, "asm_mb.s")
    open(s, "w").write(text)
    subprocess.run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s,
                    "-o", os.path.join(outdir, "asm_mb.o")], check=True)
    subprocess.run([os.path.join(LLVM, "ld.lld"), "-shared", os.path.join(outdir, "asm_mb.o"), "-o", os.path.join(outdir, "asm_mb.hsaco")], check=True)
    open(os.path.join(outdir, "asm_mb.txt"), "w").write("\n".join(meta) + "\n")
    print("\n".join(meta))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "asm_mb"))
