"""field.py -- instruction sequences for Fp arithmetic in the rr29 representation of csrc/fp29.h
(26 limbs of 29 bits, Montgomery radix 2^754, every value fully reduced into [0, p)), as generators.

The results are limb for limb those of fp29.h's fp_mul / fp_sqr / fp_mul2s / fp_sub / fp_neg (which restate
algebra/src/fields/models/fp_768.rs:1009-1185 mul_assign + mont_reduce, :339-548 square_in_place, :929-949 add / sub,
:870-883 neg): tests/test_asmgen.py runs them in sim.py against Python integers.

A "slot" is a Reg range of 26 VGPRs.  Every routine is a Python generator that yields after each emitted instruction, so
that two independent routines can be interleaved instruction by instruction (`interleave`) -- worth it for the short carry
chains through SGPRs (sub, masks: they wait on the SGPR hazard), NOT for products: at two waves per SIMD a v_mad_u64_u32
that takes its addend from its predecessor issues in 4 cycles, one that alternates with another accumulator in 5
(tools/asm_mb: mul_seq2 / dual_one_chain / g2_triple_seq against their interleaved forms, 1.6 - 5.4 %; the round-2
measurement that said otherwise, tools/microbench/lone_wave.hip, had one wave on the card).  Each chain owns an
accumulator pair, two temporaries, a dummy carry-out pair for the mads and a carry pair for borrow chains (VCC is never
used inside a routine, so two interleaved chains cannot disturb each other).

gfx9 constant-bus rule (ONE SGPR or literal per VALU instruction, an SGPR carry-in included) shapes the sequences: a
borrow chain cannot take p_i from an SGPR, so the conditional subtraction runs on signed limbs with the borrow in a VGPR.
"""
from .isa import Reg, V, S

NL = 26
LB = 29
LM = (1 << LB) - 1


def limbs(x, n=NL):
    return [(x >> (LB * i)) & LM for i in range(n)]


def unlimbs(l):
    return sum(int(v) << (LB * i) for i, v in enumerate(l))


class Chain:
    """Per-chain scratch registers."""

    def __init__(self, acc, t0, t1, sdum, scar):
        self.acc = acc        # V pair
        self.t0 = t0          # V
        self.t1 = t1          # V
        self.sdum = sdum      # S pair: carry-out of the mads (never read)
        self.scar = scar      # S pair: borrow chains / selects


class FieldGen:
    def __init__(self, prog, p, s_p, s_np, s_inv, s_lm):
        self.g = prog
        self.p = p
        self.pl = limbs(p)
        self.inv = (-pow(p, -1, 1 << LB)) % (1 << LB)
        self.s_p = s_p          # S index of p_0 (26 SGPRs)
        self.s_np = s_np        # S index of -p_0 mod 2^32 (26 SGPRs)
        self.s_inv = s_inv
        self.s_lm = s_lm

    def sP(self, i):
        return S(self.s_p + i)

    def sNP(self, i):
        return S(self.s_np + i)

    def load_constants(self):
        g = self.g
        for i in range(NL):
            g.s_mov_b32(self.sP(i), self.pl[i])
        for i in range(NL):
            g.s_mov_b32(self.sNP(i), (-self.pl[i]) & 0xFFFFFFFF)
        g.s_mov_b32(S(self.s_inv), self.inv)
        g.s_mov_b32(S(self.s_lm), LM)

    # ------------------------------------------------------------------ products
    def _mad(self, ch, x, y, first):
        self.g.v_mad_u64_u32(ch.acc, ch.sdum, x, y, 0 if first else ch.acc)

    def _close_low(self, ch, m, k):
        """column k < 26 of a Montgomery product: m_k, + m_k p_0, shift"""
        g = self.g
        g.v_mul_lo_u32(ch.t0, ch.acc.lo(), S(self.s_inv)); yield
        g.v_and_b32(m.sub(k), S(self.s_lm), ch.t0); yield
        g.v_mad_u64_u32(ch.acc, ch.sdum, m.sub(k), self.sP(0), ch.acc); yield
        g.v_lshrrev_b64(ch.acc, LB, ch.acc); yield                  # one full-rate 64-bit shift (tools/asm_mb: ic_lshr64)

    def _close_high(self, ch, r, k, last_unmasked=False):
        g = self.g
        if k == 2 * NL - 1:
            if last_unmasked:
                g.v_mov_b32(r.sub(k - NL), ch.acc.lo())
            else:
                g.v_and_b32(r.sub(k - NL), S(self.s_lm), ch.acc.lo())
            yield
            return
        g.v_and_b32(r.sub(k - NL), S(self.s_lm), ch.acc.lo()); yield
        g.v_lshrrev_b64(ch.acc, LB, ch.acc); yield

    def mont_columns(self, ch, ab_terms, m, r=None):
        """The 52 columns of a Montgomery product.  ab_terms(k) -> list of (x, y) register pairs whose products form
        column k of the integer product; m: slot for the m digits; r: slot the (unreduced, < 2p) result limbs go to
        (default: over m -- r_j is written after the last use of m_j, a_j and b_j, so r may also be an operand's slot)."""
        if r is None:
            r = m
        first = True
        for k in range(2 * NL):
            for (x, y) in ab_terms(k):
                self._mad(ch, x, y, first); first = False; yield
            lo = 0 if k < NL else k - NL + 1
            hi = k - 1 if k < NL else NL - 1
            for i in range(lo, hi + 1):
                self._mad(ch, m.sub(i), self.sP(k - i), first); first = False; yield
            if k < NL:
                yield from self._close_low(ch, m, k)
            else:
                yield from self._close_high(ch, r, k)

    @staticmethod
    def mul_terms(a, b):
        def terms(k):
            lo = max(0, k - NL + 1)
            hi = min(k, NL - 1)
            return [(a.sub(i), b.sub(k - i)) for i in range(lo, hi + 1)]
        return terms

    @staticmethod
    def sqr_terms(a, a2):
        """a2 = 2 a (limbs < 2^30): off-diagonal products once against the doubled operand (fp29.h fp_sqr)"""
        def terms(k):
            lo = max(0, k - NL + 1)
            t = [(a.sub(i), a2.sub(k - i)) for i in range(lo, NL) if 2 * i < k and k - i < NL]
            if k % 2 == 0 and k // 2 < NL:
                t.append((a.sub(k // 2), a.sub(k // 2)))
            return t
        return terms

    def cond_sub(self, ch, r, d, dst):
        """r (normalised limbs, value < 2p) -> r mod p into dst.  d: scratch slot (may not alias r; dst may be r or d)."""
        g = self.g
        bw = ch.t1
        x = ch.t0
        for i in range(NL):
            if i == 0:
                g.v_add_u32(x, self.sNP(0), r.sub(0)); yield
            else:
                g.v_add3_u32(x, r.sub(i), self.sNP(i), bw); yield
            g.v_ashrrev_i32(bw, 31, x); yield
            g.v_and_b32(d.sub(i), S(self.s_lm), x); yield
        # bw == 0 -> r >= p -> take d
        g.v_cmp_eq_u32(ch.scar, 0, bw); yield
        for i in range(NL):
            g.v_cndmask_b32(dst.sub(i), r.sub(i), d.sub(i), ch.scar); yield

    def mul(self, ch, a, b, m, d, dst=None, reduce=True):
        """dst = a b 2^-754 mod p.  m: free slot (m digits, then the unreduced result); d: slot that is dead once the product's
        columns are done (typically a or b) for the conditional subtraction; dst: m (default) or d."""
        if not reduce:
            # the unreduced result (< a b / R + p: below 2 p if at most one operand is itself unreduced) straight into dst: an
            # operand of later products only, each time paired with a fully reduced one
            yield from self.mont_columns(ch, self.mul_terms(a, b), m, m if dst is None else dst)
            return
        yield from self.mont_columns(ch, self.mul_terms(a, b), m)
        yield from self.cond_sub(ch, m, d, m if dst is None else dst)

    def sqr(self, ch, a, a2, m, dst=None):
        """dst = a^2 2^-754 mod p.  a2: free slot for the doubled operand (dead afterwards: used by the reduction)."""
        g = self.g
        for i in range(NL):
            g.v_lshlrev_b32(a2.sub(i), 1, a.sub(i)); yield
        yield from self.mont_columns(ch, self.sqr_terms(a, a2), m)
        yield from self.cond_sub(ch, m, a2, m if dst is None else dst)

    def dual(self, chx, chy, a, b, c, d, m, dd, dst=None):
        """dst = (a b + c d) 2^-754 mod p with ONE reduction (fp29.h fp_mul2s: all four operands fully reduced, which
        bounds every column below 2^64), on ONE accumulator chain: the a b terms of a column, then its c d terms, then
        m p.  (Until round 4 the two products ran on two alternating chains, joined when a column closed: 2 % slower on the card,
        tools/asm_mb dual against dual_one_chain -- a mad that takes its addend from its predecessor is the cheaper one.)  chy is
        unused; dd: scratch slot for the conditional subtraction (e.g. c, dead after the columns)."""
        def terms(k):
            lo = max(0, k - NL + 1)
            hi = min(k, NL - 1)
            return ([(a.sub(i), b.sub(k - i)) for i in range(lo, hi + 1)] +
                    [(c.sub(i), d.sub(k - i)) for i in range(lo, hi + 1)])
        yield from self.mont_columns(chx, terms, m)
        yield from self.cond_sub(chx, m, dd, m if dst is None else dst)

    def triple(self, chx, chy, pairs, m, dd, dst=None):
        """dst = (a b + c d + e f) 2^-754 mod p for pairs = [(a, b), (c, d), (e, f)] with ONE reduction (fp29.h fp_mul3, which
        restates the Karatsuba-free schoolbook form of fields/models/fp3.rs:453-477 coefficient by coefficient).  A column holds up to
        104 products of 58 bits: it does not fit 64 bits, so two chains each take half (X = a b + c d + carry-in, Y = e f + m p,
        at most 52 products + 2^36 each); when the column closes X's low limb moves over to Y, which closes it as a single chain does.  Operands need normalised limbs
        only (< 2^29); the value is below (3 p^2 + R p) / R < 2.33 p, which can exceed 2^754 by one bit: `top`, folded into
        the first of two conditional subtractions.  dd: scratch slot for them (may be an operand: the columns are done)."""
        g = self.g
        (a, b), (c, d), (e, f) = pairs
        X, Y = chx.acc, chy.acc
        fx = True
        for k in range(2 * NL):
            lo = max(0, k - NL + 1)
            hi = min(k, NL - 1)
            xs, ys = [], []
            for i in range(lo, hi + 1):
                xs.append((a.sub(i), b.sub(k - i)))
                xs.append((c.sub(i), d.sub(k - i)))
                ys.append((e.sub(i), f.sub(k - i)))
            mlo = 0 if k < NL else k - NL + 1
            mhi = k - 1 if k < NL else NL - 1
            for i in range(mlo, mhi + 1):
                ys.append((m.sub(i), self.sP(k - i)))
            fy = True
            # one chain after the other: a mad whose addend is its predecessor's result is the cheap one (alternating the chains
            # measured 5 % slower: tools/asm_mb g2_triple against g2_triple_seq)
            for x, y in xs:
                g.v_mad_u64_u32(X, chx.sdum, x, y, 0 if fx else X); fx = False; yield
            for x, y in ys:
                g.v_mad_u64_u32(Y, chy.sdum, x, y, 0 if fy else Y); fy = False; yield
            if k == 2 * NL - 1:                                     # only the carry of column 50
                g.v_and_b32(m.sub(NL - 1), S(self.s_lm), X.lo()); yield
                g.v_lshrrev_b32(chy.t1, LB, X.lo()); yield          # top
                break
            # Close the column without a 65-bit sum (and without carries through SGPRs, which cost wait states on gfx950):
            # X's low 29 bits move over to Y (52 products + 2^29 still fit 64 bits), X >> 29 waits; Y closes the column as a
            # single chain does; the next column starts from (X >> 29) + (Y >> 29).
            g.v_and_b32(chx.t0, S(self.s_lm), X.lo()); yield
            g.v_lshrrev_b64(X, LB, X); yield
            g.v_mad_u64_u32(Y, chy.sdum, chx.t0, 1, Y); yield
            if k < NL:
                g.v_mul_lo_u32(chx.t0, Y.lo(), S(self.s_inv)); yield
                g.v_and_b32(m.sub(k), S(self.s_lm), chx.t0); yield
                g.v_mad_u64_u32(Y, chy.sdum, m.sub(k), self.sP(0), Y); yield
            else:
                g.v_and_b32(m.sub(k - NL), S(self.s_lm), Y.lo()); yield
            g.v_lshrrev_b64(Y, LB, Y); yield
            g.v_lshl_add_u64(X, Y, 0, X); yield
        # value = m + top 2^754 < 2.33 p: subtract p where top is set or m >= p, then the usual conditional subtraction
        bw, x = chx.t1, chx.t0
        for i in range(NL):
            if i == 0:
                g.v_add_u32(x, self.sNP(0), m.sub(0)); yield
            else:
                g.v_add3_u32(x, m.sub(i), self.sNP(i), bw); yield
            g.v_ashrrev_i32(bw, 31, x); yield
            g.v_and_b32(dd.sub(i), S(self.s_lm), x); yield
        g.v_cmp_eq_u32(chx.scar, 0, bw); yield
        g.v_cmp_ne_u32(chy.scar, 0, chy.t1); yield
        g.s_or_b64(chx.scar, chx.scar, chy.scar); yield
        for i in range(NL):
            g.v_cndmask_b32(m.sub(i), m.sub(i), dd.sub(i), chx.scar); yield
        yield from self.cond_sub(chx, m, dd, m if dst is None else dst)

    def mul_small(self, ch, x, k, dst, s_invc):
        """dst = k x - q p with q = floor(k x_25 / (p_25 + 1)), k a per-lane VGPR in [1, 16) (fp29.h fp_mul_small_rt without its
        final conditional subtraction): the value is k x mod p or that plus p, below p + 27 2^725 -- an operand for products
        only (their column bounds have 2^57 of slack for a top limb that exceeds p_25 by 27; a sum of two such products stays
        below 2 p R).  k = 1 returns x.  s_invc: S pair holding the double 1 / (p_25 + 1).  dst may alias x."""
        g = self.g
        acc, t0, nq = ch.acc, ch.t0, ch.t1
        g.v_mul_lo_u32(t0, x.sub(NL - 1), k); yield
        g.v_cvt_f64_u32(acc, t0); yield
        g.v_add_f64(acc, acc, "0.5"); yield                         # (v + 1/2) / c is never within 2^-29 of an integer: the rounded
        g.v_mul_f64(acc, acc, s_invc); yield                        # product truncates to floor(v / c) exactly
        g.v_cvt_u32_f64(nq, acc); yield
        g.v_sub_u32(nq, 0, nq); yield
        for i in range(NL):
            g.v_mad_i64_i32(acc, ch.sdum, x.sub(i), k, 0 if i == 0 else acc); yield
            g.v_mad_i64_i32(acc, ch.sdum, nq, self.sP(i), acc); yield
            g.v_and_b32(dst.sub(i), S(self.s_lm), acc.lo()); yield
            if i + 1 < NL:
                g.v_ashrrev_i64(acc, LB, acc); yield

    def invc_bits(self):
        """the double 1 / (p_25 + 1) as (lo, hi) words"""
        import struct
        lo, hi = struct.unpack("<II", struct.pack("<d", 1.0 / (self.pl[NL - 1] + 1)))
        return lo, hi

    def bperm(self, dst, addr, src):
        """dst = src of the lane addr / 4 (ds_bpermute_b32: LDS crossbar, no LDS memory); the caller waits on lgkmcnt"""
        for i in range(NL):
            self.g.ds_bpermute_b32(dst.sub(i), addr, src.sub(i)); yield

    # ------------------------------------------------------------------ add / sub / neg
    def sub(self, ch, a, b, dst):
        """dst = a - b mod p (a, b in [0, p)).  dst may alias a or b."""
        g = self.g
        car = ch.scar
        for i in range(NL):
            if i == 0:
                g.v_sub_co_u32(dst.sub(0), car, a.sub(0), b.sub(0)); yield
            else:
                g.v_subb_co_u32(dst.sub(i), car, a.sub(i), b.sub(i), car); yield
        # car = lanes with a < b: there subtract 2^754 - p = sum (LM - p_i) 2^(29 i) + 1 (the + 1 is the borrow-in), elsewhere 0
        bml = ch.t1
        g.v_cndmask_b32(bml, 0, -1, car); yield                              # (VOP3 takes no literal on gfx9: mask in two steps)
        g.v_and_b32(bml, S(self.s_lm), bml); yield
        for i in range(NL):
            g.v_bfi_b32(ch.t0, self.sP(i), 0, bml); yield                     # ~p_i & bml
            g.v_and_b32(dst.sub(i), S(self.s_lm), dst.sub(i)); yield
            g.v_subb_co_u32(dst.sub(i), car, dst.sub(i), ch.t0, car); yield
            g.v_and_b32(dst.sub(i), S(self.s_lm), dst.sub(i)); yield

    def sub_plus_p(self, ch, a, b, dst):
        """dst = a - b + p as normalised limbs (a, b in [0, p): the value is in (0, 2 p) -- an operand for a product whose other
        operand is fully reduced).  Borrow in a VGPR, no select: 4 instructions per limb.  dst may alias a or b."""
        g = self.g
        t, x, c = ch.t0, ch.acc.lo(), ch.t1
        for i in range(NL):
            g.v_sub_u32(t, a.sub(i), b.sub(i)); yield
            if i == 0:
                g.v_add_u32(x, self.sP(0), t); yield
            else:
                g.v_add3_u32(x, t, self.sP(i), c); yield
            if i + 1 < NL:
                g.v_ashrrev_i32(c, LB, x); yield
            g.v_and_b32(dst.sub(i), S(self.s_lm), x); yield

    def add_mod(self, ch, a, b):
        """a = a + b mod p, fully reduced (a, b in [0, p)); b is destroyed (it holds the digits of a + b - p).  Two carry chains in
        VGPRs (the sum, and the sum minus p on signed limbs) and one select: 8 instructions per limb."""
        g = self.g
        t, u, c1, bw = ch.t0, ch.acc.lo(), ch.acc.hi(), ch.t1
        for i in range(NL):
            g.v_add_u32(t, a.sub(i), b.sub(i)); yield
            if i == 0:
                g.v_and_b32(a.sub(0), S(self.s_lm), t); yield
                g.v_lshrrev_b32(c1, LB, t); yield
                g.v_add_u32(u, self.sNP(0), t); yield
            else:
                g.v_add_u32(u, t, c1); yield
                g.v_lshrrev_b32(c1, LB, u); yield
                g.v_and_b32(a.sub(i), S(self.s_lm), u); yield
                g.v_add3_u32(u, t, self.sNP(i), bw); yield
            g.v_ashrrev_i32(bw, LB, u); yield
            g.v_and_b32(b.sub(i), S(self.s_lm), u); yield
        # bw == 0 -> a + b >= p -> take the second chain's digits
        g.v_cmp_eq_u32(ch.scar, 0, bw); yield
        for i in range(NL):
            g.v_cndmask_b32(a.sub(i), a.sub(i), b.sub(i), ch.scar); yield

    def neg_sel(self, ch, y, tmp, sel):
        """y = sel ? p - y : y  in place (sel: S pair lane mask; y in [0, p); y == 0 gives p on negated lanes, a harmless
        unreduced zero: products accept it and a - p == a).  tmp: one more scratch VGPR.  Uses the chain's accumulator pair
        as scratch (no product of this chain may be in flight)."""
        g = self.g
        m, t0, lmm, x, c = ch.t1, ch.t0, ch.acc.lo(), ch.acc.hi(), tmp
        g.v_cndmask_b32(m, 0, -1, sel); yield
        g.v_and_b32(lmm, S(self.s_lm), m); yield
        g.v_and_b32(c, 1, m); yield                                   # p - y = p + (2^754 - 1 - y) + 1 - 2^754
        for i in range(NL):
            g.v_and_b32(t0, self.sP(i), m); yield
            g.v_xor_b32(x, y.sub(i), lmm); yield
            g.v_add3_u32(x, x, t0, c); yield
            g.v_lshrrev_b32(c, LB, x); yield
            g.v_and_b32(y.sub(i), S(self.s_lm), x); yield

    def is_zero_mask(self, ch, a, smask):
        """smask (S pair) = lanes where a == 0"""
        g = self.g
        t = ch.t0
        g.v_or3_b32(t, a.sub(0), a.sub(1), a.sub(2)); yield
        i = 3
        while i < NL:
            if i + 1 < NL:
                g.v_or3_b32(t, t, a.sub(i), a.sub(i + 1)); i += 2
            else:
                g.v_or_b32(t, t, a.sub(i)); i += 1
            yield
        g.v_cmp_eq_u32(smask, 0, t); yield

    def set_const(self, slot, value):
        for i, l in enumerate(limbs(value)):
            self.g.v_mov_b32(slot.sub(i), l)
            yield

    def copy(self, dst, src):
        for i in range(NL):
            self.g.v_mov_b32(dst.sub(i), src.sub(i))
            yield


def run(gen):
    for _ in gen:
        pass


def interleave(*gens):
    gens = list(gens)
    while gens:
        for g in list(gens):
            try:
                next(g)
            except StopIteration:
                gens.remove(g)
