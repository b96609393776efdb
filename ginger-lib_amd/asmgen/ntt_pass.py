"""ntt_pass.py -- one Stockham pass of the radix-2 NTT (csrc/ntt_kernels.h ntt_pass_kernel) as gfx950 assembly.

Same pass as the C++ kernel, index for index (which restates algebra/src/fft/domain.rs:262-317 serial_fft / best_fft in
self-sorting form): column j of the N / 2^k columns reads x[j + t N / 2^k], t < 2^k, applies the
optional coset factor and the inter-pass twiddle w^(kk t << sh), runs the k radix-2 DIF stages and writes row bitrev(t)
to jbase + (bitrev_k(t) << log_ns), times the optional final factor.  The two kernels can be mixed pass by pass.

Why assembly: the C++ pass exchanges the butterflies' operands through LDS (26 words out, 26 back per element and stage,
a block barrier per stage) and spills 160 bytes per lane.  Here ONE WAVE owns 256 elements = 2^k rows x 2^(8-k) columns,
four per lane, and a stage is two in-lane butterflies: the two top row bits start in-lane, every further stage first
swaps one in-lane bit with a lane bit (two ds_bpermute_b32 + two selects per limb: no LDS memory, no barrier, waves are
independent) -- and the arithmetic is the hand-allocated product of field.py.

Fewer products than the C++ pass (16 per lane and pass instead of 18): the inter-pass twiddle w^(c t), c = kk << sh, is not
multiplied in.  A column's transform is the evaluation of its polynomial at theta w_R^u, theta = w^c -- a DFT on a coset -- and
splitting by the top row bit keeps that shape: the even outputs are the half-size transform of a + theta^h b with the same
shift, the odd ones that of a - theta^h b with shift theta w_R.  So every stage is (a, b) <- (a + v b, a - v b) with
v = (the sub-transform's shift)^h = w^e, e = (kk << (sh + beta)) + (rev << (ssh + beta)), rev = the bit-reversed row bits above
beta: one product per butterfly, the inter-pass factor included; in pass 0 (c = 0) the first stage and half of the second
have v = 1.  Same outputs at the same indices as the C++ pass.

Register plan (232 VGPRs, two waves per SIMD, no scratch, no LDS allocation):
  v0 tid | v1 lane | v2 tl = lane >> log_c | v3 column j | v4 kk | v5..v19 indices and temporaries
  E0..E3 = v[20 + 26 i ..]: the lane's four elements | TWA, TWB: twiddles / factors, prefetched alternately
  M, D: product scratch (m digits / the difference) -- all four also stage the raw 24-word ABI elements on the way in and out
  chain: v[228:229] accumulator, v230, v231
Data stays in the ABI layout in HBM (24 words, 2^768 Montgomery form): the transform is linear, twiddles are in the internal
2^754 form, so products keep the data's scaling (ntt_kernels.h header).
"""
from .isa import Prog, V, S, VCC, EXEC, OFF, fix_hazards
from .field import FieldGen, Chain, run, NL, LB, LM

# kernarg block (NttAsmArgs in csrc/asm_kernels.h)
ARG_IN, ARG_OUT, ARG_TW, ARG_PRE, ARG_POST = 0, 8, 16, 24, 32
ARG_LOGN, ARG_LOGNS, ARG_INVERSE, ARG_POST_STRIDE, ARG_NWAVES = 40, 44, 48, 52, 56
ARG_BYTES = 64

S_KARG, S_WG = S(0, 2), S(2)
S_IN, S_OUT, S_TW, S_PRE, S_POST = S(4, 2), S(6, 2), S(8, 2), S(10, 2), S(12, 2)
S_LOGN, S_LOGNS, S_INVERSE, S_PSTRIDE, S_NWAVES = S(14), S(15), S(16), S(17), S(18)
S_LM, S_INV = 20, 21
S_P, S_NP = 24, 50
S_DUM, S_CAR, S_MASK = S(22, 2), S(76, 2), S(78, 2)
S_SSH, S_SH, S_NMASK, S_N, S_NSMASK, S_96, S_104, S_WAVE, S_T0, S_T1 = (S(80), S(81), S(82), S(83), S(84), S(85), S(86), S(87),
                                                                         S(88), S(89))

V_TID, V_LANE, V_TL, V_J, V_KK = V(0), V(1), V(2), V(3), V(4)
V_OFF = [V(5), V(6), V(7), V(8)]          # byte offsets of the four elements (input, then output)
V_IDX = [V(9), V(10), V(11), V(12)]       # element indices (ia, then oa)
V_T = [V(13), V(14), V(15), V(16), V(17), V(18), V(19)]


def slot(i):
    return V(20 + NL * i, NL)


E = [slot(i) for i in range(4)]
TWA, TWB, M, D = slot(4), slot(5), slot(6), slot(7)
STAGE = [TWA, TWB, M, D]


def build(name, p, k):
    """p: the scalar field's prime (the NTT runs over Fr of the pairing); k: stages of the pass, 6..8 (a wave owns 2^(8-k) columns)"""
    assert 4 <= k <= 8
    log_c = 8 - k
    g = Prog(name)
    for _ in range(5):
        g.add_arg(8, "ptr")
    for _ in range(6):
        g.add_arg(4, "val")
    assert g.kernarg_bytes == ARG_BYTES
    f = FieldGen(g, p, S_P, S_NP, S_INV, S_LM)
    ch = Chain(V(228, 2), V(230), V(231), S_DUM, S_CAR)
    L_END, L_GO = g.uniq("end"), g.uniq("go")

    # ------------------------------------------------------------ helpers
    def load_fp(dst, voff, sbase):
        """26 limbs (104 bytes, 8-byte aligned) at sbase + voff"""
        for q in range(NL // 2):
            g.global_load_dwordx2(V(dst.idx + 2 * q, 2), voff, sbase, offset=8 * q)

    def load_raw(st, voff):
        for q in range(6):
            g.global_load_dwordx4(V(st.idx + 4 * q, 4), voff, S_IN, offset=16 * q)

    def unpack(dst, st):
        """24 words -> 26 limbs of 29 bits (fp29.h fp_unpack)"""
        for i in range(NL):
            bit = LB * i
            wi, sh = bit >> 5, bit & 31
            if sh + LB <= 32:
                if sh == 0:
                    g.v_and_b32(dst.sub(i), S(S_LM), st.sub(wi))
                else:
                    g.v_bfe_u32(dst.sub(i), st.sub(wi), sh, LB)
            elif wi + 1 < 24:
                g.v_alignbit_b32(dst.sub(i), st.sub(wi + 1), st.sub(wi), sh)
                g.v_and_b32(dst.sub(i), S(S_LM), dst.sub(i))
            else:
                g.v_lshrrev_b32(dst.sub(i), sh, st.sub(wi))

    def pack(st, src):
        """26 limbs -> 24 words (fp29.h fp_pack)"""
        for j in range(24):
            lo = (32 * j) // LB
            sh = 32 * j - LB * lo
            w = st.sub(j)
            if sh == 0:
                prev = src.sub(lo)
            else:
                g.v_lshrrev_b32(w, sh, src.sub(lo))
                prev = w
            if lo + 1 < NL:
                g.v_lshl_or_b32(w, src.sub(lo + 1), LB - sh, prev)
                prev = w
            if lo + 2 < NL and 2 * LB - sh < 32:
                g.v_lshl_or_b32(w, src.sub(lo + 2), 2 * LB - sh, prev)
                prev = w
            assert prev is w

    def tw_offset(dst, e):
        """dst = byte offset of twiddle e (the index, a VGPR): w^e forward, w^(N - e) inverse"""
        g.v_sub_u32(V_T[6], S_N, e)
        g.v_and_b32(V_T[6], S_NMASK, V_T[6])
        g.v_cndmask_b32(dst, e, V_T[6], S_T0X)
        g.v_mul_lo_u32(dst, dst, S_104)

    def butterfly(x, y, tw):
        """(x, y) <- (x + tw y, x - tw y); tw None: the factor is one"""
        if tw is None:
            run(f.sub(ch, x, y, D))
            run(f.add_mod(ch, x, y))
            run(f.copy(y, D))
        else:
            run(f.mul(ch, y, tw, M, D, dst=D))
            run(f.sub(ch, x, D, y))
            run(f.add_mod(ch, x, D))

    def times(x, tw):
        """x <- x tw (tw is dead afterwards: it serves the conditional subtraction)"""
        run(f.mul(ch, x, tw, M, tw, dst=x))

    S_T0X = S(90, 2)                         # lanes of an inverse transform (all or none)

    # ------------------------------------------------------------ prologue
    g.s_load_dwordx8(S(4, 8), S_KARG, 0)
    g.s_load_dwordx8(S(12, 8), S_KARG, 32)
    f.load_constants()
    g.s_mov_b32(S_96, 96)
    g.s_mov_b32(S_104, 104)
    g.v_and_b32(V_LANE, 63, V_TID)
    g.v_lshrrev_b32(V_T[0], 6, V_TID)
    g.v_readfirstlane_b32(S_WAVE, V_T[0])
    g.s_lshl_b32(S_T0, S_WG, 2)
    g.s_add_u32(S_WAVE, S_WAVE, S_T0)
    g.s_waitcnt(lgkmcnt=0)
    g.s_cmp_lt_u32(S_WAVE, S_NWAVES)
    g.s_cbranch_scc1(L_GO)
    g.s_endpgm()
    g.label(L_GO)
    g.s_sub_u32(S_SSH, S_LOGN, k)                                   # log2 of the row stride N / 2^k
    g.s_sub_u32(S_SH, S_SSH, S_LOGNS)                               # shift of the inter-pass twiddle exponent
    g.s_lshl_b32(S_N, 1, S_LOGN)
    g.s_sub_u32(S_NMASK, S_N, 1)
    g.s_lshl_b32(S_NSMASK, 1, S_LOGNS)
    g.s_sub_u32(S_NSMASK, S_NSMASK, 1)
    g.s_cmp_lg_u32(S_INVERSE, 0)
    g.s_cselect_b64(S_T0X, -1, 0)
    if log_c:
        g.v_and_b32(V_T[0], (1 << log_c) - 1, V_LANE)
        g.v_lshrrev_b32(V_TL, log_c, V_LANE)
        g.s_lshl_b32(S_T0, S_WAVE, log_c)
        g.v_add_u32(V_J, S_T0, V_T[0])
    else:
        g.v_mov_b32(V_TL, V_LANE)
        g.v_mov_b32(V_J, S_WAVE)
    g.v_and_b32(V_KK, S_NSMASK, V_J)
    # rows of the lane's elements: t_a = a 2^(k-2) + tl; input index ia = j + (t_a << ssh)
    for a in range(4):
        g.v_or_b32(V_T[a], a << (k - 2), V_TL)                      # t_a (kept for the twiddle exponent)
        g.v_lshlrev_b32(V_IDX[a], S_SSH, V_T[a])
        g.v_add_u32(V_IDX[a], V_IDX[a], V_J)
        g.v_mul_lo_u32(V_OFF[a], V_IDX[a], S_96)
    for a in range(4):
        load_raw(STAGE[a], V_OFF[a])
    for a in range(4):
        g.s_waitcnt(vmcnt=6 * (3 - a))
        unpack(E[a], STAGE[a])

    # ------------------------------------------------------------ optional coset factor
    L_NOPRE = g.uniq("nopre")
    S_JMP = S(92, 2)
    g.s_cmp_eq_u64(S_PRE, 0)
    g.s_cbranch_scc0(g_pre := g.uniq("pre"))
    g.long_branch(L_NOPRE, S_JMP)
    g.label(g_pre)
    for a in range(4):
        g.v_mul_lo_u32(V_OFF[a], V_IDX[a], S_104)
    load_fp(TWA, V_OFF[0], S_PRE)
    for a in range(4):
        cur, nxt = (TWA, TWB) if a % 2 == 0 else (TWB, TWA)
        if a + 1 < 4:
            load_fp(nxt, V_OFF[a + 1], S_PRE)
            g.s_waitcnt(vmcnt=NL // 2)
        else:
            g.s_waitcnt(vmcnt=0)
        times(E[a], cur)
    g.label(L_NOPRE)


    # ------------------------------------------------------------ the k stages
    # inl[b] = the row bit that in-lane bit b holds; lane bit q >= log_c holds row bit q - log_c until it is swapped in
    inl = [k - 2, k - 1]
    V_BP, V_E, V_REV = V_T[4], V_T[5], V_T[3]
    S_T2 = S(94)
    g.v_bfrev_b32(V_REV, V_TL)
    g.v_lshrrev_b32(V_REV, 32 - (k - 2), V_REV)                     # the row bits the lanes hold (tl has k - 2 bits), reversed
    for beta in range(k - 1, -1, -1):
        last = beta == 0
        if beta in inl:
            b = inl.index(beta)
            swap_q = None
        else:
            b = 0 if inl[0] > inl[1] else 1                         # the in-lane bit done longest ago goes out to the lanes
            swap_q = beta + log_c
        pairs = [(0, 2), (1, 3)] if b == 1 else [(0, 1), (2, 3)]
        # the stage's factors w^e, e = (kk << (sh + beta)) + (rev << (ssh + beta)); rev = reversed row bits above beta: those in the
        # lanes (V_REV, its low k - 2 - beta bits) and the other in-lane bit (bit k - 2 - beta: clear for the first pair, set for the second)
        g.s_add_u32(S_T1, S_SH, beta)
        g.s_add_u32(S_T2, S_SSH, beta)
        g.v_lshlrev_b32(V_E, S_T1, V_KK)
        tws = [TWA]
        if beta <= k - 2:
            nrev = k - 2 - beta
            if nrev > 0:
                g.v_and_b32(V_T[0], (1 << nrev) - 1, V_REV)
                g.v_lshlrev_b32(V_T[0], S_T2, V_T[0])
                g.v_add_u32(V_T[0], V_T[0], V_E)
            else:
                g.v_mov_b32(V_T[0], V_E)
            g.s_lshl_b32(S_T1, 1 << nrev, S_T2)
            g.v_add_u32(V_T[1], S_T1, V_T[0])
            g.v_mov_b32(V_E, V_T[0])
            tw_offset(V_T[1], V_T[1])
            load_fp(TWB, V_T[1], S_TW)
            tws.append(TWB)
        tw_offset(V_T[0], V_E)
        load_fp(TWA, V_T[0], S_TW)
        if swap_q is not None:
            # in-lane bit b <-> lane bit swap_q: lanes with the bit clear give their upper element and take the partner's lower one
            g.v_xor_b32(V_BP, 1 << swap_q, V_LANE)
            g.v_lshlrev_b32(V_BP, 2, V_BP)
            g.v_and_b32(V_T[6], 1 << swap_q, V_LANE)
            g.v_cmp_ne_u32(S_MASK, 0, V_T[6])
            for (x, y) in pairs:
                run(f.bperm(M, V_BP, E[x]))
                run(f.bperm(D, V_BP, E[y]))
                g.s_waitcnt(lgkmcnt=0)
                for w in range(NL):
                    g.v_cndmask_b32(E[y].sub(w), M.sub(w), E[y].sub(w), S_MASK)
                    g.v_cndmask_b32(E[x].sub(w), E[x].sub(w), D.sub(w), S_MASK)
            inl[b] = beta
        g.s_waitcnt(vmcnt=0)
        for n, (x, y) in enumerate(pairs):
            tw = tws[min(n, len(tws) - 1)]
            if beta == k - 1 or (beta == k - 2 and n == 0):
                # rev == 0: in pass 0 (log_ns == 0, kk == 0) the factor is one
                L_FULL, L_NEXT = g.uniq("full"), g.uniq("next")
                g.s_cmp_eq_u32(S_LOGNS, 0)
                g.s_cbranch_scc0(L_FULL)
                butterfly(E[x], E[y], None)
                g.long_branch(L_NEXT, S_JMP)
                g.label(L_FULL)
                butterfly(E[x], E[y], tw)
                g.label(L_NEXT)
            else:
                butterfly(E[x], E[y], tw)
    # ------------------------------------------------------------ output indices, optional final factor, store
    # row t = (tl << 2) | in-lane bits; output index oa = jbase + (bitrev_k(t) << log_ns), jbase = ((j - kk) << k) + kk
    g.v_sub_u32(V_T[0], V_J, V_KK)
    g.v_lshlrev_b32(V_T[0], k, V_T[0])
    g.v_add_u32(V_T[0], V_T[0], V_KK)                               # jbase
    g.v_lshlrev_b32(V_T[1], 2, V_TL)
    g.v_bfrev_b32(V_T[1], V_T[1])
    g.v_lshrrev_b32(V_T[1], 32 - k, V_T[1])                         # bitrev_k(tl << 2)
    for a in range(4):
        t_low = (((a >> 0) & 1) << inl[0]) | (((a >> 1) & 1) << inl[1])
        rev = int(format(t_low, "0%db" % k)[::-1], 2)
        g.v_or_b32(V_IDX[a], rev, V_T[1])
        g.v_lshlrev_b32(V_IDX[a], S_LOGNS, V_IDX[a])
        g.v_add_u32(V_IDX[a], V_IDX[a], V_T[0])
    L_STORE = g.uniq("store")
    g.s_cmp_eq_u64(S_POST, 0)
    g.s_cbranch_scc0(g_post := g.uniq("post"))
    g.long_branch(L_STORE, S_JMP)
    g.label(g_post)
    for a in range(4):
        g.v_mul_lo_u32(V_OFF[a], V_IDX[a], S_PSTRIDE)
    load_fp(TWA, V_OFF[0], S_POST)
    for a in range(4):
        cur, nxt = (TWA, TWB) if a % 2 == 0 else (TWB, TWA)
        if a + 1 < 4:
            load_fp(nxt, V_OFF[a + 1], S_POST)
            g.s_waitcnt(vmcnt=NL // 2)
        else:
            g.s_waitcnt(vmcnt=0)
        times(E[a], cur)
    g.label(L_STORE)
    for a in range(4):
        g.v_mul_lo_u32(V_OFF[a], V_IDX[a], S_96)
    for a in range(4):
        pack(STAGE[a], E[a])
        for q in range(6):
            g.global_store_dwordx4(V_OFF[a], V(STAGE[a].idx + 4 * q, 4), S_OUT, offset=16 * q)
    g.s_endpgm()
    g.label(L_END)
    g.s_endpgm()
    g.hazard_nops = fix_hazards(g)
    return g
