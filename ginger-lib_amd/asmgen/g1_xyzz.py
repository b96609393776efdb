"""g1_xyzz.py -- the G1 bucket accumulation kernel (MNT4-753 / MNT6-753 G1) as gfx950 assembly.

Same task list, list walk, salt detour and output as csrc/msm_kernels.h msm_accumulate_xyzz_kernel (which restates the
bucket loop of algebra/src/msm/variable_base.rs:36-59 with add_assign_mixed, swp.rs:481-519, replaced by madd-2008-s on
(X, Y, ZZ, ZZZ) accumulators): one thread per task, every list entry one mixed addition.

Register plan (256 VGPRs = two waves per SIMD, 0 bytes of scratch):
  v0 tid | v1 LDS address | v[2:3] list cursor | v4 entries left | v5 phase | v6 sigma | v7 salt id | v[8:9] dst
  v[10:11] gather address | v12 list entry | v13 guard | v14..v19 temporaries | v[20:21] bases | v[22:23] salts | v39 temp
  E0..E7 = v[40 + 26 i ..]: eight field-element slots; E0 = ZZ, E1 = ZZZ for the whole task
  chain A: v[248:249] accumulator, v252, v253;  chain B: v[250:251], v254, v255
  X, V = sigma Y and (between two steps) R^2 of the running sum are parked in LDS, word-major: park[3][26][256].
The prime, -p, the Montgomery constant and the limb mask live in SGPRs.

Order of the products (each one chain of mads, pairs back to back -- `seq`; the subtractions of a pair stay interleaved):
  U2, S2  P, W  PP, RR  park RR  PPP, ZZ3  Q, ZZZ3  X3, T  dual(W T + V PPP)
"""
from .isa import Prog, V, S, VCC, EXEC, OFF, fix_hazards
from .field import FieldGen, Chain, interleave, run, NL, LM

AFF_BYTES = 2 * NL * 4          # 208
PROJ_WORDS = 3 * NL             # 78
PARK_STRIDE = 256 * 4           # bytes between consecutive words of a parked element
PARK_X, PARK_V, PARK_RR = 0, 1, 2
LDS_BYTES = 3 * NL * PARK_STRIDE

# SGPR map
S_KARG = S(0, 2)
S_WG = S(2)
S_BASES, S_SORTED, S_TASKS, S_SALTS = S(4, 2), S(6, 2), S(8, 2), S(10, 2)
S_NTASKS = S(12)
S_208 = S(13)
S_LM, S_INV = 20, 21
S_P, S_NP = 24, 50
S_ACTIVE = S(76, 2)
S_NEGSEL = S(78, 2)
S_T0, S_T1 = S(80, 2), S(82, 2)
S_LAUNCH = S(84, 2)
S_SAVE = S(86, 2)
S_DETOUR = S(88, 2)
S_T2 = S(90, 2)
S_ZERO = S(92, 2)

V_TID, V_LDS = V(0), V(1)
V_CUR = V(2, 2)
V_LEFT, V_PHASE, V_SNEG, V_SALT = V(4), V(5), V(6), V(7)
V_DST = V(8, 2)
V_ADDR = V(10, 2)
V_E = V(12)
V_GUARD = V(13)
V_T = [V(14), V(15), V(16), V(17), V(18), V(19)]
V_BASES, V_SALTS = V(20, 2), V(22, 2)
V_TMP = V(39)
V_LDS2 = V(24)
V_ENEXT = V(25)
V_TOUCH = V(26)
V_NADDR = V(28, 2)


def slot(i):
    return V(40 + NL * i, NL)


def seq(*gens):
    """The products of a pair one after the other.  (Rounds 2-4 interleaved them instruction by instruction on two accumulator
    chains; measured on the card, tools/asm_mb mul_seq2 / sqr_seq2 against mul_pair / sqr_pair, the sequential form is 1.6 % /
    2.6 % faster at two waves per SIMD: a v_mad_u64_u32 whose 64-bit addend is its predecessor's result does not read it from
    the register file.)"""
    for gen in gens:
        run(gen)


E = [slot(i) for i in range(8)]


def build(name, p, one_mont, prefetch=False, split=4):
    """p: the base prime; one_mont = 2^754 mod p (internal Montgomery one)."""
    g = Prog(name)
    g.lds_bytes = LDS_BYTES
    for _ in range(4):
        g.add_arg(8, "ptr")
    g.add_arg(4, "val")
    f = FieldGen(g, p, S_P, S_NP, S_INV, S_LM)
    chA = Chain(V(248, 2), V(252), V(253), S(14, 2), S(16, 2))
    chB = Chain(V(250, 2), V(254), V(255), S(18, 2), S(22, 2))

    def park_addr(which, w):
        # DS offsets are 16 bits: the third parked element is addressed from a second base register
        if which < 2:
            return V_LDS, (which * NL + w) * PARK_STRIDE
        return V_LDS2, ((which - 2) * NL + w) * PARK_STRIDE

    def park_put(which, sl):
        for w in range(NL):
            a, off = park_addr(which, w)
            g.ds_write_b32(a, sl.sub(w), offset=off)

    def park_get(which, sl):
        for w in range(NL):
            a, off = park_addr(which, w)
            g.ds_read_b32(sl.sub(w), a, offset=off)

    L_LOOP, L_INIT, L_INIT_RET, L_DETOUR, L_DET_RET, L_ADV, L_DONE, L_START = (
        g.uniq(s) for s in ("loop", "init", "init_ret", "detour", "det_ret", "adv", "done", "start"))
    S_JMP = S(94, 2)                                                # scratch of the long jumps

    # ------------------------------------------------------------ prologue
    g.s_load_dwordx8(S(4, 8), S_KARG, 0)
    g.s_load_dword(S_NTASKS, S_KARG, 32)
    f.load_constants()
    g.s_mov_b32(S_208, AFF_BYTES)
    g.v_lshlrev_b32(V_LDS, 2, V_TID)
    g.v_add_u32(V_LDS2, 2 * NL * PARK_STRIDE, V_LDS)
    g.v_lshl_or_b32(V_T[0], S_WG, 8, V_TID)
    g.s_waitcnt(lgkmcnt=0)
    g.v_cmp_gt_u32(VCC, S_NTASKS, V_T[0])
    g.s_and_saveexec_b64(S_LAUNCH, VCC)
    g.s_cbranch_execnz(L_START)
    g.s_endpgm()
    g.label(L_START)
    g.s_mov_b64(S_LAUNCH, EXEC)
    g.v_lshlrev_b32(V_T[1], 4, V_T[0])
    g.global_load_dwordx4(V(16, 4), V_T[1], S_TASKS)                # beg, cnt, dst
    g.v_mov_b32(V_BASES.lo(), S_BASES.lo()); g.v_mov_b32(V_BASES.hi(), S_BASES.hi())
    g.v_mov_b32(V_SALTS.lo(), S_SALTS.lo()); g.v_mov_b32(V_SALTS.hi(), S_SALTS.hi())
    g.v_mov_b32(V_CUR.lo(), S_SORTED.lo()); g.v_mov_b32(V_CUR.hi(), S_SORTED.hi())
    g.v_mov_b32(V_PHASE, 0); g.v_mov_b32(V_SNEG, 0); g.v_mov_b32(V_SALT, 0)
    for w in range(NL):                                             # ZZ = ZZZ = 0: the running sum is the point at infinity
        g.v_mov_b32(E[0].sub(w), 0)
        g.v_mov_b32(E[1].sub(w), 0)
    g.s_waitcnt(vmcnt=0)
    g.v_mov_b32(V_LEFT, V(17))
    g.v_mov_b32(V_DST.lo(), V(18)); g.v_mov_b32(V_DST.hi(), V(19))
    g.v_mad_u64_u32(V_CUR, chA.sdum, V(16), 4, V_CUR)
    g.v_lshl_add_u32(V_GUARD, V_LEFT, 2, 8)                         # at most 4 cnt + 8 iterations (the C++ kernel's guard)

    g.s_branch(L_LOOP)

    # ------------------------------------------------------------ epilogue: (X ZZZ : Y ZZ : ZZ ZZZ), infinity -> (0, 1, 0)
    g.label(L_DONE)
    g.s_mov_b64(EXEC, S_LAUNCH)
    run(f.is_zero_mask(chA, E[0], S_ZERO))
    # lanes that never parked anything (empty task) read whatever LDS holds: their result is replaced below
    park_get(PARK_X, E[2])
    park_get(PARK_V, E[3])
    g.v_cmp_ne_u32(S_NEGSEL, 0, V_SNEG)
    g.s_waitcnt(lgkmcnt=0)
    for w in range(NL):                                             # garbage from LDS must still be a normalised element
        g.v_and_b32(E[2].sub(w), S(S_LM), E[2].sub(w))
        g.v_and_b32(E[3].sub(w), S(S_LM), E[3].sub(w))
    run(f.neg_sel(chA, E[3], V_TMP, S_NEGSEL))
    seq(f.mul(chA, E[2], E[1], E[4], E[2]),                         # x = X ZZZ -> E4
        f.mul(chB, E[3], E[0], E[5], E[3]))                         # y = Y ZZ  -> E5
    run(f.cond_sub(chA, E[0], E[6], E[0]))                          # ZZ, ZZZ leave the loop below 2 p: one of them reduced for z
    run(f.mul(chA, E[0], E[1], E[6], E[2]))                         # z = ZZ ZZZ -> E6
    g.s_mov_b64(S_SAVE, EXEC)
    g.s_and_b64(EXEC, EXEC, S_ZERO)
    for w in range(NL):
        g.v_mov_b32(E[4].sub(w), 0)
        g.v_mov_b32(E[6].sub(w), 0)
    run(f.set_const(E[5], one_mont))
    g.s_mov_b64(EXEC, S_SAVE)
    for j in range(PROJ_WORDS // 2):                                # E4, E5, E6 are contiguous: Proj = x[26] y[26] z[26]
        g.global_store_dwordx2(V_DST, V(E[4].idx + 2 * j, 2), OFF, offset=8 * j)
    g.s_endpgm()

    # ------------------------------------------------------------ loop head
    g.label(L_LOOP)
    g.s_mov_b64(EXEC, S_LAUNCH)
    g.v_cmp_ne_u32(S_ACTIVE, 0, V_LEFT)
    g.v_cmp_ne_u32(S_T0, 0, V_GUARD)
    g.s_and_b64(S_ACTIVE, S_ACTIVE, S_T0)
    g.s_mov_b64(S_DETOUR, 0)
    g.s_and_b64(EXEC, S_ACTIVE, EXEC)
    g.s_cbranch_execz(L_DONE)
    # the entry of phases 0 / 2 (read by every active lane: the cursor stays inside the list while entries are left)
    g.global_load_dword(V_E, V_CUR, OFF)
    if prefetch:                                                    # the entry after this one (the same again at the end of the list)
        g.v_cmp_lt_u32(S_T1, 1, V_LEFT)
        g.v_cndmask_b32(V_T[2], 0, 4, S_T1)
        g.v_add_co_u32(V_NADDR.lo(), VCC, V_CUR.lo(), V_T[2])
        g.v_addc_co_u32(V_NADDR.hi(), VCC, 0, V_CUR.hi(), VCC)
        g.global_load_dword(V_ENEXT, V_NADDR, OFF)
    g.v_and_b32(V_T[1], 1, V_PHASE)
    g.v_cmp_ne_u32(S_T0, 0, V_T[1])                                 # S_T0 = lanes in a salt phase (1, 3)
    g.v_mad_u64_u32(V(16, 2), chA.sdum, V_SALT, S_208, V_SALTS)
    g.s_waitcnt(vmcnt=0)
    g.v_and_b32(V_T[0], 0x7FFFFFFF, V_E)
    g.v_mad_u64_u32(V_ADDR, chA.sdum, V_T[0], S_208, V_BASES)
    g.v_cndmask_b32(V_ADDR.lo(), V_ADDR.lo(), V(16), S_T0)
    g.v_cndmask_b32(V_ADDR.hi(), V_ADDR.hi(), V(17), S_T0)
    # E2 = q.x, E3 = q.y (contiguous).  Every lane reads its own row: one wave-wide load touches 64 pages, more than the L1 TLB
    # holds, so each of a row's 13 loads would miss all of its translations again (profiles/r02_affine_rounds_counters.txt:
    # TCP_UTCL1_TRANSLATION_MISS 13 per row).  A part of the wave at a time, all 13 loads of its rows back to back.
    if split > 1:
        g.s_mov_b64(S_SAVE, EXEC)
    for q in range(split):
        if split > 1:
            g.s_bfm_b64(S_T1, 64 // split, (64 // split) * q)
            g.s_and_b64(EXEC, S_SAVE, S_T1)
        for j in range(AFF_BYTES // 16):
            g.global_load_dwordx4(V(E[2].idx + 4 * j, 4), V_ADDR, OFF, offset=16 * j)
    if split > 1:
        g.s_mov_b64(EXEC, S_SAVE)
    # negate q.y on lanes where (entry sign, or phase == 3 for the salt) != sigma
    g.v_lshrrev_b32(V_T[0], 31, V_E)
    g.v_lshrrev_b32(V_T[1], 1, V_PHASE)
    g.v_cndmask_b32(V_T[0], V_T[0], V_T[1], S_T0)
    g.v_xor_b32(V_T[0], V_T[0], V_SNEG)
    g.v_cmp_ne_u32(S_NEGSEL, 0, V_T[0])
    g.s_waitcnt(vmcnt=0)
    if prefetch:
        # Touch the NEXT entry's row (its four 64-byte lines) now, a whole update ahead: the gathers are one row per lane out of a
        # multi-GB table, i.e. 64 address translations per wave instruction; with the lines and translations warm the
        # gather at the top of the next iteration returns from L2 (tools/asm_mb/acc_run: 74.5 K -> 68.7 K cycles per update
        # is the cost of the cold gather).  The loaded words are never read.
        g.v_and_b32(V_T[2], 0x7FFFFFFF, V_ENEXT)
        g.v_mad_u64_u32(V_NADDR, chA.sdum, V_T[2], S_208, V_BASES)
        for j in range(4):
            g.global_load_dword(V_TOUCH, V_NADDR, OFF, offset=min(64 * j, AFF_BYTES - 4))
    run(f.neg_sel(chA, E[3], V_TMP, S_NEGSEL))
    run(f.is_zero_mask(chA, E[0], S_ZERO))
    g.s_cmp_lg_u64(S_ZERO, 0)
    g.s_cbranch_scc0(L_INIT_RET)
    g.long_branch(L_INIT, S_JMP)
    g.label(L_INIT_RET)

    # ------------------------------------------------------------ the update
    seq(f.mul(chA, E[2], E[0], E[4], E[2]),                         # U2 = q.x ZZ   -> E4
        f.mul(chB, E[3], E[1], E[5], E[3]))                         # S2 = q.y ZZZ  -> E5
    park_get(PARK_X, E[2])
    park_get(PARK_V, E[3])
    g.s_waitcnt(lgkmcnt=0)
    interleave(f.sub(chA, E[4], E[2], E[4]),                        # P = U2 - X    -> E4
               f.sub(chB, E[5], E[3], E[5]))                        # W = S2 - V    -> E5
    # acc == q (P == 0 and W == 0 in phase 0): the salt detour
    interleave(f.is_zero_mask(chA, E[4], S_T0), f.is_zero_mask(chB, E[5], S_T1))
    g.v_cmp_eq_u32(S_T2, 0, V_PHASE)
    g.s_and_b64(S_T0, S_T0, S_T1)
    g.s_and_b64(S_DETOUR, S_T0, S_T2)
    g.s_cbranch_scc0(L_DET_RET)
    g.long_branch(L_DETOUR, S_JMP)
    g.label(L_DET_RET)
    seq(f.sqr(chA, E[4], E[2], E[6]),                               # PP = P^2      -> E6
        f.sqr(chB, E[5], E[3], E[7]))                               # RR = W^2      -> E7
    park_put(PARK_RR, E[7])
    seq(f.mul(chA, E[4], E[6], E[2], E[4], dst=E[4]),               # PPP = P PP    -> E4
        f.mul(chB, E[0], E[6], E[3], E[0], dst=E[0], reduce=False))   # ZZ3 = ZZ PP -> E0, below 2 p (see the epilogue)
    park_get(PARK_X, E[2])
    g.s_waitcnt(lgkmcnt=0)
    seq(f.mul(chA, E[2], E[6], E[3], E[2], dst=E[2]),               # Q = X PP      -> E2
        f.mul(chB, E[1], E[4], E[7], E[1], dst=E[1], reduce=False))   # ZZZ3 = ZZZ PPP -> E1, below 2 p
    park_get(PARK_RR, E[3])
    g.s_waitcnt(lgkmcnt=0)
    run(f.sub(chA, E[3], E[4], E[3]))                               # X3 = RR - PPP - 2 Q -> E3
    run(f.sub(chA, E[3], E[2], E[3]))
    run(f.sub(chA, E[3], E[2], E[3]))
    park_put(PARK_X, E[3])
    park_get(PARK_V, E[6])
    run(f.sub(chA, E[3], E[2], E[2]))                               # T = X3 - Q    -> E2
    g.s_waitcnt(lgkmcnt=0)
    run(f.dual(chA, chB, E[5], E[2], E[6], E[4], E[7], E[6]))       # W T + V PPP = -sigma Y3 -> E7: the new V, sigma flips
    park_put(PARK_V, E[7])
    g.v_xor_b32(V_SNEG, 1, V_SNEG)

    # ------------------------------------------------------------ advance
    g.label(L_ADV)
    g.s_andn2_b64(EXEC, S_ACTIVE, S_DETOUR)
    g.v_cmp_eq_u32(S_T0, 0, V_PHASE)
    g.v_cmp_eq_u32(S_T1, 3, V_PHASE)
    g.s_or_b64(S_T0, S_T0, S_T1)                                    # lanes that move on to the next entry
    g.v_add_u32(V_T[0], 1, V_PHASE)
    g.v_cndmask_b32(V_PHASE, V_T[0], 0, S_T0)
    g.v_cndmask_b32(V_T[0], 0, 1, S_T0)
    g.v_sub_u32(V_LEFT, V_LEFT, V_T[0])
    g.v_lshlrev_b32(V_T[0], 2, V_T[0])
    g.v_add_co_u32(V_CUR.lo(), VCC, V_CUR.lo(), V_T[0])
    g.v_addc_co_u32(V_CUR.hi(), VCC, 0, V_CUR.hi(), VCC)
    g.s_mov_b64(EXEC, S_ACTIVE)
    g.v_subrev_u32(V_GUARD, 1, V_GUARD)
    g.long_branch(L_LOOP, S_JMP)

    # ------------------------------------------------------------ rare: the running sum is infinity -> take q
    g.label(L_INIT)
    g.s_mov_b64(S_SAVE, EXEC)
    g.s_mov_b64(EXEC, S_ZERO)
    park_put(PARK_X, E[2])
    park_put(PARK_V, E[3])
    run(f.set_const(E[0], one_mont))
    run(f.set_const(E[1], one_mont))
    g.s_andn2_b64(EXEC, S_SAVE, S_ZERO)
    g.s_cbranch_execz(L_ADV)
    g.long_branch(L_INIT_RET, S_JMP)

    # ------------------------------------------------------------ rare: acc == q -> ((q + S) + q) - S through a salt point
    g.label(L_DETOUR)
    g.s_mov_b64(S_SAVE, EXEC)
    g.s_mov_b64(EXEC, S_DETOUR)
    # salt id = (q.x == salts[0].x) ? 1 : 0 ; q.x is read again from the lane's gather address
    for j in range(NL // 2):
        g.global_load_dwordx2(V(E[2].idx + 2 * j, 2), V_ADDR, OFF, offset=8 * j)
    for j in range(NL // 2):
        g.global_load_dwordx2(V(E[3].idx + 2 * j, 2), V_SALTS, OFF, offset=8 * j)
    g.s_waitcnt(vmcnt=0)
    for w in range(NL):
        g.v_xor_b32(E[2].sub(w), E[2].sub(w), E[3].sub(w))
    run(f.is_zero_mask(chA, E[2], S_T0))
    g.v_cndmask_b32(V_SALT, 0, 1, S_T0)
    g.v_mov_b32(V_PHASE, 1)
    g.s_andn2_b64(EXEC, S_SAVE, S_DETOUR)
    g.s_cbranch_execz(L_ADV)
    g.long_branch(L_DET_RET, S_JMP)
    g.hazard_nops = fix_hazards(g)      # isa.py: the gfx950 VALU -> SGPR -> VALU wait states
    return g
