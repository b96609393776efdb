"""microbench.py -- the Montgomery product peak of the card, measured by the library itself (gh_measure_fpmul_peak).

One kernel: every lane runs `iters` x two independent rr29 products (field.py mul: 2 x 26^2 v_mad_u64_u32 + one conditional
subtraction each, fp29.h fp_mul = algebra/src/fields/models/fp_768.rs:1009-1185 mul_assign), one after the other (the fastest form measured: tools/asm_mb,
mul_seq2 against mul_pair), on the register plan of the hot kernels (256 VGPRs, two waves per SIMD).  bench.py
reports the rate next to the round-1 constant (23.4 G products/s, tools/microbench/mb.hip) that `valu.frac` is priced against.
"""
from .isa import Prog, V, S, fix_hazards
from .field import FieldGen, Chain, run, NL, LM


def build(name, p):
    g = Prog(name)
    g.lds_bytes = 3 * NL * 1024            # as the accumulation kernels: two blocks of 256 lanes per CU
    g.add_arg(4, "val")
    g.add_arg(4, "val")
    f = FieldGen(g, p, 24, 50, 21, 20)
    A = Chain(V(248, 2), V(252), V(253), S(14, 2), S(16, 2))
    B = Chain(V(250, 2), V(254), V(255), S(18, 2), S(22, 2))
    E = [V(40 + NL * i, NL) for i in range(8)]
    g.s_load_dword(S(3), S(0, 2), 0)
    f.load_constants()
    for i in range(6):                     # operands: limbs derived from the lane id, top limb small (values below p)
        for w in range(NL):
            g.v_mov_b32(V(34), 0x1234567 + 977 * w + i)
            g.v_mul_u32_u24(V(35), 0x9E37 + 131 * i + w, V(0))
            g.v_add_u32(E[i].sub(w), V(34), V(35))
            g.v_and_b32(E[i].sub(w), (LM >> 3) if w == NL - 1 else LM, E[i].sub(w))
    g.s_waitcnt(lgkmcnt=0)
    L, Lx = g.uniq("loop"), g.uniq("exit")
    g.label(L)
    run(f.mul(A, E[2], E[0], E[4], E[2]))          # two products, one after the other: the fastest form measured (tools/asm_mb)
    run(f.mul(B, E[3], E[1], E[5], E[3]))
    g.s_sub_u32(S(3), S(3), 1)
    g.s_cmp_lg_u32(S(3), 0)
    g.s_cbranch_scc0(Lx)
    g.long_branch(L, S(94, 2))
    g.label(Lx)
    g.s_endpgm()
    fix_hazards(g)
    return g
