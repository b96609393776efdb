"""g2_rounds.py -- the affine bucket-sum rounds of the G2 MSMs (MNT4-753 G2 over Fq2, MNT6-753 G2 over Fq3) as gfx950 assembly.

What they compute is csrc/aff_kernels.h AffRoundLane::run (which replaces the n x W calls of add_assign_mixed in the bucket
loop of algebra/src/msm/variable_base.rs:36-59 by pairwise affine additions with Montgomery's shared inversion,
curves/models/short_weierstrass_projective.rs:402-442): round r turns the bucket-ordered list P_r into P_(r+1), output o =
P_r[i] + P_r[i + 1] (or a copy of an odd leftover), named by a descriptor (first input | pair << 31).  Same lists, same
descriptors, same T64 layouts as the C++ kernel, which stays in the library as the fallback for the rare cases.

Split into two kernels around a small C++ inversion kernel (aff_inv_kernel):
  fwd : per lane group, the running product acc of the denominators x2 - x1 over its B elements; parks every prefix product;
        round 0 also gathers the table rows named by the sorted list, applies the signs and stages them (T64) for bwd.
        Writes acc per lane group.  Raises `flag` if an element needs the group law's rare branches (x1 == x2: doubling or
        cancellation, swp.rs:492; an infinity marker among the inputs) -- the whole round is then redone by the C++ kernel.
  bwd : from acc^-1 backwards: 1 / (x2 - x1), lambda, x3 = lambda^2 - x1 - x2, y3 = lambda (x1 - x3) - y1 (5 M + 1 S in the
        tower per addition), the output list.

A tower element lives in LANES adjacent lanes, one Fp coefficient per lane (F2S / F3S of msm_kernels.h).  A tower product is,
per lane, ONE multi-product with a single Montgomery reduction:
  Fq2 (u^2 = 13): c_j = a_j b_0 + [13 if j = 0] a_(j-1) b_1                                   (fields/models/fp2.rs:389-400)
  Fq3 (u^3 = 11): c_j = a_j b_0 + [11 if j = 0] a_(j-1) b_1 + [11 if j < 2] a_(j-2) b_2        (fields/models/fp3.rs:453-477)
(indices mod LANES).  The partner coefficients travel by ds_bpermute_b32 (the LDS crossbar: no VALU slot, no LDS memory).

Register plan: 256 VGPRs (two waves per SIMD), no scratch.  Eight 26-register slots E0..E7; x1, x2, y1 of the element in flight
are parked in LDS (3 x 26 KiB per block of 256 lanes).  p, -p, the Montgomery constant and the limb mask live in SGPRs.
"""
from .isa import Prog, V, S, VCC, EXEC, OFF, fix_hazards
from .field import FieldGen, Chain, interleave, run, NL, LM, limbs

PARK_STRIDE = 256 * 4
LDS_BYTES = 3 * NL * PARK_STRIDE
PT_TILE = 13 * 1024          # bytes of one T64 tile of points (13 chunks of 64 x 16 bytes)
FP_TILE = 7 * 1024           # ... of prefix products
MARK = 0xFFFFFFFF            # x.l[0] of the infinity marker (aff_kernels.h AFF_MARK)
FIX_CAP = 16384              # entries of the exception list (control block: 64 bytes + 4 FIX_CAP)

# kernel argument block (bytes)
ARG_IN, ARG_SORTED, ARG_DESC, ARG_PREFIX, ARG_STAGE1, ARG_STAGE2, ARG_OUT, ARG_ACCS, ARG_FLAG = 0, 8, 16, 24, 32, 40, 48, 56, 64
ARG_NOUT, ARG_INBASE, ARG_B = 72, 76, 80
ARG_BYTES = 88

# SGPRs
S_KARG = S(0, 2)
S_WG = S(2)
S_TMP = S(3)
S_IN, S_SORTED, S_DESC, S_PREFIX, S_STAGE1 = S(4, 2), S(6, 2), S(8, 2), S(10, 2), S(12, 2)
S_LM, S_INV = 20, 21
S_P, S_NP = 24, 50
S_STAGE2, S_OUT, S_ACCS, S_FLAGP = S(76, 2), S(78, 2), S(80, 2), S(82, 2)
S_NOUT, S_INBASE, S_B, S_K = S(84), S(85), S(86), S(87)
S_INVC = S(88, 2)
S_LIVE, S_ACT, S_PAIR, S_T0, S_T1, S_FLAG = S(90, 2), S(92, 2), S(94, 2), S(96, 2), S(98, 2), S(100, 2)

# VGPRs
V_TID, V_LDS, V_LDS2, V_LANE16, V_COMP, V_G = V(0), V(1), V(2), V(3), V(4), V(5)
V_K1, V_K2, V_AP, V_AN = V(6), V(7), V(8), V(9)
V_BC = [V(10), V(11), V(12)]
V_DE, V_T0, V_T1, V_T2, V_T3, V_ONEM, V_O = V(13), V(14), V(15), V(16), V(17), V(18), V(19)
V_ADDR1, V_ADDR2 = V(228, 2), V(230, 2)
V_OFF = [V(232), V(233), V(234)]
V_POFF = [V(235), V(236)]
V_E1, V_E2 = V(238), V(239)
V_ROWB = V(240, 2)
V_A1B, V_A1C, V_A2B, V_A2C = V(242, 2), V(244, 2), V(246, 2), V(240, 2)    # + 4096 / + 12288 of the two list addresses (later rounds;
                                                                             # V_A2C shares its pair with V_ROWB, round 0 only)
V_TMP = V(237)


def slot(i):
    return V(20 + NL * i, NL)


E = [slot(i) for i in range(8)]      # v20 .. v227


class Cfg:
    def __init__(self, lanes, nr, p, one_mont):     # lanes: 1 (G1), 2 (Fq2 lane pairs), 3 (Fq3 lane triples)
        self.L = lanes
        self.NR = nr
        self.p = p
        self.one = one_mont
        self.TPW = 64 // lanes
        self.row_bytes = 2 * lanes * 104


import os as _os
# parts of the wave that gather their rows one after the other.  The C++ round kernel and the G1 update gather a quarter of the
# wave at a time (address translations: aff_kernels.h gather_row); here ONE pass is faster -- a quarter of the load instructions
# (MNT4-753 G2 2^20: 66.7 -> 64.7 ms per MSM, MNT6-753 G2 2^19: 74.8 -> 73.0; tools/ab_split.sh)
GATHER_SPLIT = int(_os.environ.get("GH_ASM_GATHER_SPLIT", "1"))
ABLATE = set(x for x in _os.environ.get("GH_ASM_ABLATE", "").split(",") if x)      # timing experiments only: results are wrong


def build(name, cfg, fwd, r0, debug=False):
    """fwd: forward (True) or backward (False) kernel; r0: round 0 (inputs are table rows named by the sorted list, staged by
    the forward kernel) or a later round (inputs in the previous round's T64 list)."""
    L, TPW = cfg.L, cfg.TPW
    g = Prog(name)
    g.lds_bytes = 0 if fwd else LDS_BYTES
    g.add_arg(ARG_BYTES, "val")
    f = FieldGen(g, cfg.p, S_P, S_NP, S_INV, S_LM)
    chA = Chain(V(248, 2), V(252), V(253), S(14, 2), S(16, 2))
    chB = Chain(V(250, 2), V(254), V(255), S(18, 2), S(22, 2))
    L_START, L_LOOP, L_NEXT, L_END, L_BODY = (g.uniq(s) for s in ("start", "loop", "next", "end", "body"))
    V_DEN = V(242) if r0 else V(238)      # the NEXT iteration's descriptor, fetched an iteration ahead (registers this kernel variant does
                                          # not use otherwise: the + 4096 list address of later rounds / the list entries of round 0)

    def fetch_desc(o_reg):
        """V_DEN = desc[o] on the lanes whose element o exists (the others keep what they have: never used)"""
        g.v_cmp_gt_u32(S_T0, S_NOUT, o_reg)
        g.s_and_b64(EXEC, S_LIVE, S_T0)
        g.v_lshlrev_b32(V_T0, 2, o_reg)
        g.global_load_dword(V_DEN, V_T0, S_DESC)
        g.s_mov_b64(EXEC, S_LIVE)

    def dbg(stage):
        """debug builds: flag[1 + wave of the block] = stage (tools/asm_g2_check.py)"""
        if not debug:
            return
        g.s_mov_b64(S_T1, EXEC)
        g.s_mov_b64(EXEC, 1)
        g.v_lshrrev_b32(V_T0, 6, V_TID)
        g.v_lshlrev_b32(V_T0, 2, V_T0)
        g.v_mov_b32(V_T1, stage)
        g.global_store_dword(V_T0, V_T1, S_FLAGP, offset=4)
        g.s_mov_b64(EXEC, S_T1)

    def dbg_s(idx, sreg):
        """debug builds: flag[8 + 8 wave + idx] = an SGPR"""
        if not debug:
            return
        g.s_mov_b64(S_T1, EXEC)
        g.s_mov_b64(EXEC, 1)
        g.v_lshrrev_b32(V_T0, 6, V_TID)
        g.v_lshlrev_b32(V_T0, 5, V_T0)
        g.v_mov_b32(V_T1, sreg)
        g.global_store_dword(V_T0, V_T1, S_FLAGP, offset=32 + 4 * idx)
        g.s_mov_b64(EXEC, S_T1)

    # ------------------------------------------------------------ helpers
    def park_addr(which, w):
        if which < 2:
            return V_LDS, (which * NL + w) * PARK_STRIDE
        return V_LDS2, ((which - 2) * NL + w) * PARK_STRIDE

    def park_put(which, sl):
        if "parks" in ABLATE and not fwd:
            return
        for w in range(NL):
            a, off = park_addr(which, w)
            g.ds_write_b32(a, sl.sub(w), offset=off)

    def park_get(which, sl):
        if "parks" in ABLATE and not fwd:
            return
        for w in range(NL):
            a, off = park_addr(which, w)
            g.ds_read_b32(sl.sub(w), a, offset=off)

    def list_chunk(c):
        """(index of the offset register, immediate) for chunk c of a T64 point"""
        if c < 4:
            return 0, c * 1024
        if c < 8:
            return 1, (c - 4) * 1024
        return 2, (c - 12) * 1024

    def ld_x_list(sl, saddr, offs):
        """x of a T64 point: chunks 0..5 and the low half of chunk 6.  saddr: S pair + per-lane 32-bit offsets, or None: offs are
        64-bit address pairs"""
        if "loads" in ABLATE and not fwd:
            return
        for c in range(6):
            r, imm = list_chunk(c)
            g.global_load_dwordx4(V(sl.idx + 4 * c, 4), offs[r], saddr if saddr is not None else OFF, offset=imm)
        r, imm = list_chunk(6)
        g.global_load_dwordx2(V(sl.idx + 24, 2), offs[r], saddr if saddr is not None else OFF, offset=imm)

    def ld_y_list(sl, saddr, offs):
        if "loads" in ABLATE and not fwd:
            return
        r, imm = list_chunk(6)
        g.global_load_dwordx2(V(sl.idx, 2), offs[r], saddr if saddr is not None else OFF, offset=imm + 8)
        for c in range(6):
            r, imm = list_chunk(7 + c)
            g.global_load_dwordx4(V(sl.idx + 2 + 4 * c, 4), offs[r], saddr if saddr is not None else OFF, offset=imm)

    def st_x_list(sl, saddr, offs):
        if "stores" in ABLATE and not fwd:
            return
        for c in range(6):
            r, imm = list_chunk(c)
            g.global_store_dwordx4(offs[r], V(sl.idx + 4 * c, 4), saddr, offset=imm)
        r, imm = list_chunk(6)
        g.global_store_dwordx2(offs[r], V(sl.idx + 24, 2), saddr, offset=imm)

    def st_y_list(sl, saddr, offs):
        if "stores" in ABLATE and not fwd:
            return
        r, imm = list_chunk(6)
        g.global_store_dwordx2(offs[r], V(sl.idx, 2), saddr, offset=imm + 8)
        for c in range(6):
            r, imm = list_chunk(7 + c)
            g.global_store_dwordx4(offs[r], V(sl.idx + 2 + 4 * c, 4), saddr, offset=imm)

    def ld_fp_list(sl, saddr, offs):
        if "loads" in ABLATE and not fwd:
            return
        for c in range(6):
            g.global_load_dwordx4(V(sl.idx + 4 * c, 4), offs[c // 4], saddr, offset=(c % 4) * 1024)
        g.global_load_dwordx2(V(sl.idx + 24, 2), offs[1], saddr, offset=2 * 1024)

    def st_fp_list(sl, saddr, offs):
        if "stores" in ABLATE and not fwd:
            return
        for c in range(6):
            g.global_store_dwordx4(offs[c // 4], V(sl.idx + 4 * c, 4), saddr, offset=(c % 4) * 1024)
        g.global_store_dwordx2(offs[1], V(sl.idx + 24, 2), saddr, offset=2 * 1024)

    def set_one(sl):
        """the tower's one: (2^754 mod p, 0 [, 0])"""
        for w, l in enumerate(limbs(cfg.one)):
            g.v_mov_b32(V_T0, l)
            g.v_and_b32(sl.sub(w), V_T0, V_ONEM)

    def prep_a(a, a1, a2):
        """the rotated / non-residue-scaled copies of the left operand: a1 = K1 a_(j-1), a2 = K2 a_(j-2)"""
        if L == 1:
            return
        if "bperm" in ABLATE and not fwd:
            if L == 3:
                interleave(f.mul_small(chA, a1, V_K1, a1, S_INVC), f.mul_small(chB, a2, V_K2, a2, S_INVC))
            else:
                run(f.mul_small(chA, a1, V_K1, a1, S_INVC))
            return
        run(f.bperm(a1, V_AP, a))
        if L == 3:
            run(f.bperm(a2, V_AN, a))
        g.s_waitcnt(lgkmcnt=0)
        if L == 3:
            interleave(f.mul_small(chA, a1, V_K1, a1, S_INVC), f.mul_small(chB, a2, V_K2, a2, S_INVC))
        else:
            run(f.mul_small(chA, a1, V_K1, a1, S_INVC))

    def prep_b(b, bs):
        """the coefficients of the right operand, broadcast over the lane group (G1: the operand itself, copied: the product's
        result may take b's slot while bs[0] is still read)"""
        if L == 1:
            run(f.copy(bs[0], b))
            return
        if "bperm" in ABLATE and not fwd:
            return
        for i in range(L):
            run(f.bperm(bs[i], V_BC[i], b))

    def tower_mul(a, a1, a2, bs, m, dd):
        """m = a (x) b in the tower (this lane's coefficient); a1, a2 from prep_a, bs from prep_b (waited for here)"""
        g.s_waitcnt(lgkmcnt=0)
        if L == 1:
            run(f.mul(chA, a, bs[0], m, dd))
        elif L == 2:
            run(f.dual(chA, chB, a, bs[0], a1, bs[1], m, dd))
        else:
            run(f.triple(chA, chB, [(a, bs[0]), (a1, bs[1]), (a2, bs[2])], m, dd))

    def select(dst, a, b, smask):
        """dst = smask ? b : a"""
        for w in range(NL):
            g.v_cndmask_b32(dst.sub(w), a.sub(w), b.sub(w), smask)

    def list_addr(idx, addr, addr_b, addr_c):
        """64-bit addresses of element idx (per lane) of the input list: + 0, + 4096, + 12288"""
        if L == 1:
            g.v_lshrrev_b32(V_T2, 6, idx)
            g.v_and_b32(V_T3, 63, idx)
        elif L == 2:
            g.v_lshrrev_b32(V_T2, 5, idx)
            g.v_and_b32(V_T3, 31, idx)
        else:
            g.s_mov_b32(S_TMP, 0xC30C30C4)                       # ceil(2^36 / 21): exact for idx < 2^31 / 5
            g.v_mul_hi_u32(V_T2, idx, S_TMP)
            g.v_lshrrev_b32(V_T2, 4, V_T2)
            g.v_mul_u32_u24(V_T3, V_T2, 21)
            g.v_sub_u32(V_T3, idx, V_T3)
        g.v_mad_u32_u24(V_T3, V_T3, L, V_COMP)                    # slot = (idx % TPW) L + comp
        g.v_mov_b32(V_TMP, PT_TILE)
        g.v_mad_u64_u32(addr, chA.sdum, V_T2, V_TMP, S_IN)
        g.v_mad_u64_u32(addr, chA.sdum, V_T3, 16, addr)
        g.v_add_co_u32(addr_b.lo(), VCC, 4096, addr.lo())
        g.v_addc_co_u32(addr_b.hi(), VCC, 0, addr.hi(), VCC)
        g.v_add_co_u32(addr_c.lo(), VCC, 12288, addr.lo())
        g.v_addc_co_u32(addr_c.hi(), VCC, 0, addr.hi(), VCC)

    def bad_elements(x1, x2, d, make_d=None):
        """S_FLAG = the lanes of pairs that need the group law's rare branches -- an infinity marker among the inputs, or x1 == x2
        (doubling / cancellation, swp.rs:492): any lane of the group sees its marker limb or a zero coefficient of d = x2 - x1 (a
        superset of d == 0 in the tower).  Uniform over a lane group, and the same in the forward and the backward kernel.  Such an
        element is treated as a copy here (S_PAIR loses it) and recomputed by aff_fix_kernel from the exception list."""
        g.v_cmp_eq_u32(S_T0, -1, x1.sub(0))
        g.v_cmp_eq_u32(S_T1, -1, x2.sub(0))
        g.s_or_b64(S_T0, S_T0, S_T1)
        if make_d is not None:                                       # d may take x2's registers
            make_d()
        run(f.is_zero_mask(chA, d, S_T1))
        g.s_or_b64(S_T0, S_T0, S_T1)
        g.s_and_b64(S_T0, S_T0, S_PAIR)
        g.s_mov_b64(S_FLAG, 0)
        g.s_cmp_eq_u64(S_T0, 0)
        L_NOBAD = g.uniq("nobad")
        g.s_cbranch_scc1(L_NOBAD)
        if L == 1:
            g.s_mov_b64(S_FLAG, S_T0)
        else:
            g.v_cndmask_b32(V_T0, 0, 1, S_T0)
            for i in range(L):                                       # or over the lane group
                g.ds_bpermute_b32(V(V_T1.idx + i), V_BC[i], V_T0)
            g.s_waitcnt(lgkmcnt=0)
            g.v_or_b32(V_T0, V_T1, V_T2)
            if L == 3:
                g.v_or_b32(V_T0, V_T0, V_T3)
            g.v_cmp_ne_u32(S_FLAG, 0, V_T0)
        g.s_andn2_b64(S_PAIR, S_PAIR, S_FLAG)
        g.label(L_NOBAD)

    # ------------------------------------------------------------ prologue
    g.s_load_dwordx8(S(4, 8), S_KARG, ARG_IN)
    g.s_load_dwordx2(S_STAGE1, S_KARG, ARG_STAGE1)
    g.s_load_dwordx8(S(76, 8), S_KARG, ARG_STAGE2)
    g.s_load_dwordx4(S(84, 4), S_KARG, ARG_NOUT)
    f.load_constants()
    lo, hi = f.invc_bits()
    g.s_mov_b32(S_INVC.lo(), lo)
    g.s_mov_b32(S_INVC.hi(), hi)
    g.v_lshlrev_b32(V_LDS, 2, V_TID)
    g.v_add_u32(V_LDS2, 2 * NL * PARK_STRIDE, V_LDS)
    g.v_and_b32(V_T0, 63, V_TID)                                    # lane
    g.v_lshlrev_b32(V_LANE16, 4, V_T0)
    g.v_lshrrev_b32(V_T1, 6, V_TID)
    g.v_readfirstlane_b32(S_TMP, V_T1)                              # wave of the block
    if L == 1:                                                      # G1: one lane per element, no exchange
        g.v_mov_b32(V_COMP, 0)
        g.v_mov_b32(V_G, V_T0)
        g.s_mov_b64(S_LIVE, -1)
    elif L == 2:
        g.v_and_b32(V_COMP, 1, V_T0)
        g.v_lshrrev_b32(V_G, 1, V_T0)
        g.s_mov_b64(S_LIVE, -1)
    else:
        g.v_mul_u32_u24(V_G, 171, V_T0)
        g.v_lshrrev_b32(V_G, 9, V_G)
        g.v_mul_u32_u24(V_T1, V_G, 3)
        g.v_sub_u32(V_COMP, V_T0, V_T1)
        g.s_mov_b32(S_LIVE.lo(), 0xFFFFFFFF)
        g.s_mov_b32(S_LIVE.hi(), 0x7FFFFFFF)                        # lane 63 idles
    g.s_mov_b64(EXEC, S_LIVE)
    g.v_sub_u32(V_T1, V_T0, V_COMP)                                 # first lane of the group
    for i in range(L):
        g.v_add_u32(V_T2, i, V_T1)
        g.v_lshlrev_b32(V_BC[i], 2, V_T2)
    # lane holding a_(j-1): j = 0 -> base + L - 1, else lane - 1;  a_(j-2) (L = 3): j = 0 -> base + 1, 1 -> base + 2, 2 -> base
    g.v_cmp_eq_u32(S_T0, 0, V_COMP)
    g.v_add_u32(V_T2, L - 1, V_T1)
    g.v_subrev_u32(V_T3, 1, V_T0)
    g.v_cndmask_b32(V_T2, V_T3, V_T2, S_T0)
    g.v_lshlrev_b32(V_AP, 2, V_T2)
    g.v_mov_b32(V_T2, 1)
    g.v_mov_b32(V_T3, cfg.NR)
    g.v_cndmask_b32(V_K1, V_T2, V_T3, S_T0)
    g.v_cndmask_b32(V_ONEM, 0, -1, S_T0)
    if L == 3:
        g.v_add_u32(V_T2, 1, V_COMP)                                # (j + 1) mod 3
        g.v_cmp_eq_u32(S_T1, 3, V_T2)
        g.v_cndmask_b32(V_T2, V_T2, 0, S_T1)
        g.v_add_u32(V_T2, V_T2, V_T1)
        g.v_lshlrev_b32(V_AN, 2, V_T2)
        g.v_cmp_gt_u32(S_T1, 2, V_COMP)
        g.v_mov_b32(V_T2, 1)
        g.v_cndmask_b32(V_K2, V_T2, V_T3, S_T1)
    g.s_waitcnt(lgkmcnt=0)
    # wave -> its chunk of B elements per lane group
    g.s_lshl_b32(S_K, S_WG, 2)
    g.s_add_u32(S_K, S_K, S_TMP)                                    # global wave index
    g.s_mul_i32(S_TMP, S_K, TPW)
    g.s_mul_hi_u32(S_T0.lo(), S_TMP, S_B)
    g.s_mul_i32(S_TMP, S_TMP, S_B)                                  # chunk = wave TPW B
    g.s_cmp_lg_u32(S_T0.lo(), 0)
    g.s_cbranch_scc1(L_END + "_exit")
    g.s_cmp_ge_u32(S_TMP, S_NOUT)
    g.s_cbranch_scc1(L_END + "_exit")
    g.s_branch(L_START)
    g.label(L_END + "_exit")
    g.s_endpgm()
    g.label(L_START)
    if not fwd:                                                     # a flagged round is redone by the C++ kernel
        g.s_load_dword(S_T0.lo(), S_FLAGP, 0)
        g.s_waitcnt(lgkmcnt=0)
        g.s_cmp_lg_u32(S_T0.lo(), 0)
        g.s_cbranch_scc1(L_END + "_exit")
    g.v_add_u32(V_O, S_TMP, V_G)                                    # element of iteration 0
    dbg(1)
    dbg_s(0, S_TMP)
    dbg_s(1, S_K)
    dbg_s(2, S_NOUT)
    dbg_s(3, S_B)
    dbg_s(4, S_WG)
    # list bases of this wave: tile0 = wave B
    g.s_mul_i32(S_T0.lo(), S_K, S_B)                                # tile0 (< 2^32)

    def advance_base(sp, tile_bytes):
        g.s_mul_hi_u32(S_T1.hi(), S_T0.lo(), tile_bytes)
        g.s_mul_i32(S_T1.lo(), S_T0.lo(), tile_bytes)
        g.s_add_u32(sp.lo(), sp.lo(), S_T1.lo())
        g.s_addc_u32(sp.hi(), sp.hi(), S_T1.hi())

    g.s_mov_b32(S_TMP, FP_TILE)
    advance_base(S_PREFIX, S_TMP)
    g.s_mov_b32(S_TMP, PT_TILE)
    advance_base(S_STAGE1, S_TMP)
    advance_base(S_STAGE2, S_TMP)
    advance_base(S_OUT, S_TMP)
    # accs / invs: tile = wave
    g.s_mul_hi_u32(S_T1.hi(), S_K, FP_TILE)
    g.s_mul_i32(S_T1.lo(), S_K, FP_TILE)
    g.s_add_u32(S_ACCS.lo(), S_ACCS.lo(), S_T1.lo())
    g.s_addc_u32(S_ACCS.hi(), S_ACCS.hi(), S_T1.hi())
    if r0:
        g.v_mul_u32_u24(V_T1, 104, V_COMP)
        g.v_add_co_u32(V_ROWB.lo(), VCC, S_IN.lo(), V_T1)
        g.v_mov_b32(V_T2, S_IN.hi())
        g.v_addc_co_u32(V_ROWB.hi(), VCC, 0, V_T2, VCC)

    ACC = E[0]
    if fwd:
        set_one(ACC)
        g.s_mov_b32(S_K, 0)
        g.v_mov_b32(V_OFF[0], V_LANE16)
        g.v_mov_b32(V_POFF[0], V_LANE16)
        fetch_desc(V_O)
    else:
        g.v_mov_b32(V_POFF[0], V_LANE16)
        g.v_add_u32(V_POFF[1], 4096, V_LANE16)
        ld_fp_list(ACC, S_ACCS, V_POFF)
        g.s_sub_u32(S_K, S_B, 1)
        g.s_mul_i32(S_TMP, S_K, TPW)
        g.v_add_u32(V_O, S_TMP, V_O)                                # element of iteration B - 1
        g.s_mul_i32(S_TMP, S_K, PT_TILE)
        g.v_add_u32(V_OFF[0], S_TMP, V_LANE16)
        g.s_mul_i32(S_TMP, S_K, FP_TILE)
        g.v_add_u32(V_POFF[0], S_TMP, V_LANE16)
        g.v_subrev_u32(V_POFF[0], FP_TILE, V_POFF[0])               # prefix of element k - 1
        g.s_waitcnt(vmcnt=0)
        fetch_desc(V_O)
    g.s_branch(L_LOOP)

    # ------------------------------------------------------------ end
    g.label(L_END)
    g.s_mov_b64(EXEC, S_LIVE)
    if fwd:
        g.v_mov_b32(V_POFF[0], V_LANE16)
        g.v_add_u32(V_POFF[1], 4096, V_LANE16)
        st_fp_list(ACC, S_ACCS, V_POFF)
    g.s_endpgm()

    # ------------------------------------------------------------ loop head
    g.label(L_LOOP)
    g.s_mov_b64(EXEC, S_LIVE)
    g.s_waitcnt(vmcnt=0)                                            # the descriptor fetched an iteration ago (and this wave's stores)
    g.v_mov_b32(V_DE, V_DEN)
    # the next iteration's descriptor goes on its way now
    L_NOPF = g.uniq("nopf")
    if fwd:
        g.s_add_u32(S_TMP, S_K, 1)
        g.s_cmp_ge_u32(S_TMP, S_B)
        g.s_cbranch_scc1(L_NOPF)
        g.v_add_u32(V_T1, TPW, V_O)
    else:
        g.s_cmp_eq_u32(S_K, 0)
        g.s_cbranch_scc1(L_NOPF)
        g.v_subrev_u32(V_T1, TPW, V_O)
    fetch_desc(V_T1)
    g.label(L_NOPF)
    g.v_cmp_gt_u32(S_ACT, S_NOUT, V_O)
    g.s_and_b64(EXEC, EXEC, S_ACT)
    if fwd:
        g.s_cbranch_execnz(L_BODY)                                  # elements only get fewer: nothing left for this wave
        g.long_branch(L_END, S_T1)
    else:
        g.s_cbranch_execnz(L_BODY)
        g.long_branch(L_NEXT, S_T1)
    g.label(L_BODY)
    dbg(2)
    g.v_add_u32(V_OFF[1], 4096, V_OFF[0])
    g.v_add_u32(V_OFF[2], 12288, V_OFF[0])
    g.v_add_u32(V_POFF[1], 4096, V_POFF[0])
    g.v_cmp_gt_i32(S_PAIR, 0, V_DE)                                 # bit 31: two inputs
    g.v_and_b32(V_DE, 0x7FFFFFFF, V_DE)
    g.v_cndmask_b32(V_T1, 0, 1, S_PAIR)

    X1, X2 = E[1], E[6]
    # backward kernel on lane pairs: E4 and E7 are free until lambda is formed, so y1 and y2 travel with x1 and x2 (one exposed
    # memory latency less per element); lane triples need all eight slots for the two products in between
    early_y = (not fwd) and L <= 2
    def r0_rows():
        """round 0: the two list entries of the element (V_E1, V_E2: row | sign << 31) and the addresses of this lane's coefficient in
        their table rows (V_ADDR1, V_ADDR2)"""
        g.v_mad_u64_u32(V_ADDR1, chA.sdum, V_DE, 4, S_SORTED)
        g.v_add_u32(V_T2, V_DE, V_T1)
        g.v_mad_u64_u32(V_ADDR2, chA.sdum, V_T2, 4, S_SORTED)
        g.global_load_dword(V_E1, V_ADDR1, OFF)
        g.global_load_dword(V_E2, V_ADDR2, OFF)
        g.s_mov_b32(S_TMP, cfg.row_bytes)
        g.s_waitcnt(vmcnt=0)
        g.v_and_b32(V_T2, 0x7FFFFFFF, V_E1)
        g.v_mad_u64_u32(V_ADDR1, chA.sdum, V_T2, S_TMP, V_ROWB)
        g.v_and_b32(V_T2, 0x7FFFFFFF, V_E2)
        g.v_mad_u64_u32(V_ADDR2, chA.sdum, V_T2, S_TMP, V_ROWB)

    def r0_gather(parts):
        """parts: (slot, address pair, byte offset in the row) -- every lane group reads its own rows: a quarter of the wave at a time,
        all loads of its rows back to back, so that the address translations of its pages stay resident (aff_kernels.h gather_row:
        13 TLB misses per row without it).  The caller waits."""
        if "loads" in ABLATE and not fwd:
            return
        g.s_mov_b64(S_T0, EXEC)
        for q in range(GATHER_SPLIT):
            if GATHER_SPLIT > 1:
                g.s_bfm_b64(S_T1, 64 // GATHER_SPLIT, (64 // GATHER_SPLIT) * q)
                g.s_and_b64(EXEC, S_T0, S_T1)
            for sl, addr, off in parts:
                for j in range(NL // 2):
                    g.global_load_dwordx2(V(sl.idx + 2 * j, 2), addr, OFF, offset=off + 8 * j)
        g.s_mov_b64(EXEC, S_T0)

    def r0_signs(y1, y2):
        """the signs of the two list entries on the ordinates"""
        g.v_cmp_gt_i32(S_T0, 0, V_E1)
        run(f.neg_sel(chA, y1, V_TMP, S_T0))
        g.v_cmp_gt_i32(S_T0, 0, V_E2)
        run(f.neg_sel(chB, y2, V_TMP, S_T0))

    # Round 0 reads the table rows in BOTH kernels (no staged copy: the forward kernel needs the abscissae only, and staging
    # 832 B per element made it the one memory-bound kernel of an MSM -- 39 GB in 11.8 ms at 2^20 MNT4-753 G2 pairs)
    if fwd and r0:
        r0_rows()
        r0_gather([(X1, V_ADDR1, 0), (X2, V_ADDR2, 0)])
        g.s_waitcnt(vmcnt=0)
    elif r0:
        r0_rows()
        if early_y:
            r0_gather([(X1, V_ADDR1, 0), (E[4], V_ADDR1, L * 104), (X2, V_ADDR2, 0), (E[7], V_ADDR2, L * 104)])
            g.s_waitcnt(vmcnt=0)
            r0_signs(E[4], E[7])
        else:
            r0_gather([(X1, V_ADDR1, 0), (X2, V_ADDR2, 0)])
            g.s_waitcnt(vmcnt=0)
    else:
        g.v_subrev_u32(V_T0, S_INBASE, V_DE)
        list_addr(V_T0, V_ADDR1, V_A1B, V_A1C)
        g.v_add_u32(V_T0, V_T0, V_T1)
        list_addr(V_T0, V_ADDR2, V_A2B, V_A2C)
        ld_x_list(X1, None, [V_ADDR1, V_A1B, V_A1C])
        ld_x_list(X2, None, [V_ADDR2, V_A2B, V_A2C])
        if early_y:
            ld_y_list(E[4], None, [V_ADDR1, V_A1B, V_A1C])
            ld_y_list(E[7], None, [V_ADDR2, V_A2B, V_A2C])
        g.s_waitcnt(vmcnt=0)

    if fwd:
        bad_elements(X1, X2, X2, lambda: run(f.sub(chA, X2, X1, X2)))     # d = x2 - x1 (over x2)
        # exception list: word 1 of the control block counts, the element indices follow from word 16; one lane per group appends
        L_NOAPP = g.uniq("noappend")
        g.s_cmp_eq_u64(S_FLAG, 0)
        g.s_cbranch_scc1(L_NOAPP)
        g.s_mov_b64(S_T0, EXEC)
        g.v_cmp_eq_u32(S_T1, 0, V_COMP)
        g.s_and_b64(S_T1, S_T1, S_FLAG)
        g.s_and_b64(EXEC, EXEC, S_T1)
        g.v_mov_b32(V_T0, 1)
        g.v_mov_b32(V_T1, 0)
        g.global_atomic_add(V_T2, V_T1, V_T0, S_FLAGP, offset=4, sc0=True)         # sc0: return the count before the add
        g.s_waitcnt(vmcnt=0)
        g.s_mov_b32(S_TMP, FIX_CAP)
        g.v_cmp_gt_u32(S_T1, S_TMP, V_T2)                            # room in the list
        g.s_andn2_b64(VCC, EXEC, S_T1)                               # lanes without: the whole round falls back (word 0)
        g.s_and_b64(EXEC, EXEC, S_T1)
        g.v_lshlrev_b32(V_T2, 2, V_T2)
        g.global_store_dword(V_T2, V_O, S_FLAGP, offset=64)
        g.s_mov_b64(EXEC, VCC)
        g.global_store_dword(V_T1, V_T0, S_FLAGP)
        g.global_store_dword(V_T1, V_T0, S_FLAGP, offset=16)        # sticky for the MSM: the host takes the key off the assembly rounds
        g.s_mov_b64(EXEC, S_T0)
        g.label(L_NOAPP)
        A1, A2, BS, M = E[1], E[2], [E[3], E[4], E[5]], E[7]
        prep_b(X2, BS)
        prep_a(ACC, A1, A2)
        tower_mul(ACC, A1, A2, BS, M, X2)
        select(ACC, ACC, M, S_PAIR)
        st_fp_list(ACC, S_PREFIX, V_POFF)
        # advance
        g.s_add_u32(S_K, S_K, 1)
        g.s_mov_b64(EXEC, S_LIVE)
        g.v_add_u32(V_O, TPW, V_O)
        g.v_add_u32(V_OFF[0], PT_TILE, V_OFF[0])
        g.v_add_u32(V_POFF[0], FP_TILE, V_POFF[0])
        g.s_cmp_lt_u32(S_K, S_B)
        g.s_cbranch_scc0(L_NEXT)
        g.long_branch(L_LOOP, S_T1)
        g.label(L_NEXT)
        g.long_branch(L_END, S_T1)
        g.hazard_nops = fix_hazards(g)
        return g

    # ------------------------------------------------------------ backward body
    INV = ACC
    PX1, PX2, PY1 = 0, 1, 2
    dbg(3)
    park_put(PX1, X1)
    park_put(PX2, X2)
    D = E[2]
    run(f.sub(chA, X2, X1, D))
    bad_elements(X1, X2, D)
    A1, A2, BS = E[3], E[4], [E[5], E[6], E[7]]
    g.s_waitcnt(lgkmcnt=0)                                          # X2 (= E6) is a broadcast target below: parked first
    prep_b(D, BS)
    # prefix_(k-1) into d's slot (its broadcasts are on their way): in flight during the first product
    PK = E[2]
    L_PKE = g.uniq("pkearly")
    g.s_cmp_eq_u32(S_K, 0)
    g.s_cbranch_scc1(L_PKE)
    ld_fp_list(PK, S_PREFIX, V_POFF)
    g.label(L_PKE)
    prep_a(INV, A1, A2)
    M1 = E[1]
    dbg(4)
    tower_mul(INV, A1, A2, BS, M1, BS[0])                           # inv (x) d: the inverse of the elements before this one
    dbg(5)
    # 1 / d_k = inv (x) prefix_(k-1)
    L_PK0, L_PKD = g.uniq("pk0"), g.uniq("pkd")
    g.s_cmp_eq_u32(S_K, 0)
    g.s_cbranch_scc1(L_PK0)
    g.s_waitcnt(vmcnt=0)
    g.s_branch(L_PKD)
    g.label(L_PK0)
    set_one(PK)
    g.label(L_PKD)
    prep_b(PK, BS)
    DI = E[2]
    tower_mul(INV, A1, A2, BS, DI, BS[0])
    select(INV, INV, M1, S_PAIR)                                    # a copy leaves the running inverse alone
    # lambda = (y2 - y1) / d
    if early_y:
        Y1, NUM = E[4], E[7]
    else:
        Y1, NUM = E[1], E[3]
        if r0:
            r0_gather([(Y1, V_ADDR1, L * 104), (NUM, V_ADDR2, L * 104)])
            g.s_waitcnt(vmcnt=0)
            r0_signs(Y1, NUM)
        else:
            ld_y_list(Y1, None, [V_ADDR1, V_A1B, V_A1C])
            ld_y_list(NUM, None, [V_ADDR2, V_A2B, V_A2C])
            g.s_waitcnt(vmcnt=0)
    park_put(PY1, Y1)
    run(f.sub(chA, NUM, Y1, NUM))
    A1, A2 = E[4], E[1]
    prep_b(NUM, BS)
    prep_a(DI, A1, A2)
    LAM = E[3]
    tower_mul(DI, A1, A2, BS, LAM, BS[0])
    # x3 = lambda^2 - x1 - x2
    A1, A2 = E[1], E[2]
    prep_b(LAM, BS)
    prep_a(LAM, A1, A2)
    SQ = E[4]
    tower_mul(LAM, A1, A2, BS, SQ, BS[0])
    park_get(PX1, E[5])
    park_get(PX2, E[6])
    g.s_waitcnt(lgkmcnt=0)
    run(f.sub(chA, SQ, E[5], SQ))
    run(f.sub(chA, SQ, E[6], SQ))                                   # x3
    T = E[6]
    run(f.sub(chA, E[5], SQ, T))                                    # x1 - x3
    select(SQ, E[5], SQ, S_PAIR)                                    # copy: x1
    st_x_list(SQ, S_OUT, V_OFF)
    # y3 = lambda (x1 - x3) - y1
    BS2 = [E[5], E[7], E[4]]
    prep_b(T, BS2)
    Y3 = E[6]
    tower_mul(LAM, A1, A2, BS2, Y3, BS2[0])
    park_get(PY1, E[4])
    g.s_waitcnt(lgkmcnt=0)
    run(f.sub(chA, Y3, E[4], Y3))
    select(Y3, E[4], Y3, S_PAIR)
    st_y_list(Y3, S_OUT, V_OFF)
    dbg(6)
    # ------------------------------------------------------------ next (smaller) k
    g.label(L_NEXT)
    g.s_mov_b64(EXEC, S_LIVE)
    g.v_subrev_u32(V_O, TPW, V_O)
    g.v_subrev_u32(V_OFF[0], PT_TILE, V_OFF[0])
    g.v_subrev_u32(V_POFF[0], FP_TILE, V_POFF[0])
    g.s_cmp_eq_u32(S_K, 0)
    g.s_cbranch_scc1(L_NEXT + "_done")
    g.s_sub_u32(S_K, S_K, 1)
    g.long_branch(L_LOOP, S_T1)
    g.label(L_NEXT + "_done")
    g.s_endpgm()
    g.hazard_nops = fix_hazards(g)      # isa.py: the gfx950 VALU -> SGPR -> VALU wait states
    return g
