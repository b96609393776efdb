"""The MSM stage of create_proof (proof-systems/src/groth16/prover.rs:273-345) replayed from buffers
(SURVEY.md section 8d, config 5 without a Rust toolchain): ginger-lib_amd/groth16.py over the C ABI
(device-resident query tails, one pipelined batch for the four large G1 MSMs, shift tables) against a
literal replay of the same lines on the CPU oracle (multi_scalar_mul, mul, add_assign, sub_assign,
into_affine).  Bit-exact on the three affine proof elements.  The witness map that feeds this stage is
covered in test_gpu_parity.py::test_witness_map_vs_oracle."""
import importlib

import numpy as np
import pytest

import pyref
import support as S

pytestmark = pytest.mark.gpu


def _oracle_stage(pairing, pk, ni, inp, aux, h_inp, h_aux, r, s):
    g1, g2 = pairing + "_g1", pairing + "_g2"
    O = S.oracle()

    def ec(curve, op, p, q=None, flag=0):
        C = pyref.CURVES[curve]
        out = np.zeros(36 * C.deg, dtype=np.uint64)
        p = np.ascontiguousarray(p, dtype=np.uint64)
        qq = None if q is None else np.ascontiguousarray(q, dtype=np.uint64)
        O.oracle_ec_op(S.CURVE_ID[curve], op, S.ptr(p), None if qq is None else S.ptr(qq), flag, S.ptr(out))
        return out

    def zero(curve):                                  # GroupProjective::zero() = k * P with k = 0
        C = pyref.CURVES[curve]
        return ec(curve, 3, S.proj_array(C, None), np.zeros(12, dtype=np.uint64))

    def from_affine(curve, xy, inf=0):                # GroupProjective::from(affine): zero + affine (mixed add)
        return ec(curve, 2, zero(curve), xy, int(inf))
    first = lambda name: (pk[name][0], 0 if name + "_inf" not in pk else int(pk[name + "_inf"][0]))

    add = lambda curve, a, b: ec(curve, 0, a, b)
    mul = lambda curve, a, k: ec(curve, 3, a, k)
    # infinity flags of a query (keys made by the generator hold GroupAffine::zero() entries), sliced like the query
    def msm(curve, bases, scal, inf=None):
        return S.oracle_msm(curve, bases, inf, scal, 8)
    q = lambda name, lo, hi=None: (pk[name][lo:hi], None if name + "_inf" not in pk else pk[name + "_inf"][lo:hi])
    neg = lambda curve, a: mul(curve, a, S.scalar_array([pyref.CURVES[curve].order - 1])[0])

    # prover.rs:273-284
    a_inputs_acc = msm(g1, q("a_query", 1, ni)[0], inp, q("a_query", 1, ni)[1])
    a_aux_acc = msm(g1, q("a_query", ni)[0], aux, q("a_query", ni)[1])
    g_a = mul(g1, from_affine(g1, pk["delta_g1"]), r)
    for t in (from_affine(g1, *first("a_query")), a_inputs_acc, a_aux_acc, from_affine(g1, pk["alpha_g1"])):
        g_a = add(g1, g_a, t)
    # :287-300
    b_inputs_acc = msm(g1, q("b_g1_query", 1, ni)[0], inp, q("b_g1_query", 1, ni)[1])
    b_aux_acc = msm(g1, q("b_g1_query", ni)[0], aux, q("b_g1_query", ni)[1])
    g1_b = mul(g1, from_affine(g1, pk["delta_g1"]), s)
    for t in (from_affine(g1, *first("b_g1_query")), b_inputs_acc, b_aux_acc, from_affine(g1, pk["beta_g1"])):
        g1_b = add(g1, g1_b, t)
    # :303-316
    b2_inputs_acc = msm(g2, q("b_g2_query", 1, ni)[0], inp, q("b_g2_query", 1, ni)[1])
    b2_aux_acc = msm(g2, q("b_g2_query", ni)[0], aux, q("b_g2_query", ni)[1])
    g2_b = mul(g2, from_affine(g2, pk["delta_g2"]), s)
    for t in (from_affine(g2, *first("b_g2_query")), b2_inputs_acc, b2_aux_acc, from_affine(g2, pk["beta_g2"])):
        g2_b = add(g2, g2_b, t)
    # :319-337
    h_inputs_acc = msm(g1, q("h_query", 0, ni)[0], h_inp, q("h_query", 0, ni)[1])
    h_aux_acc = msm(g1, q("h_query", ni)[0], h_aux, q("h_query", ni)[1])
    l_aux_acc = msm(g1, q("l_query", 0)[0], aux, q("l_query", 0)[1])
    s_g_a = mul(g1, g_a, s)
    r_g1_b = mul(g1, g1_b, r)
    r_s_delta = mul(g1, mul(g1, from_affine(g1, pk["delta_g1"]), r), s)
    g_c = add(g1, s_g_a, r_g1_b)
    g_c = add(g1, g_c, neg(g1, r_s_delta))
    for t in (l_aux_acc, h_inputs_acc, h_aux_acc):
        g_c = add(g1, g_c, t)
    return S.oracle_affine(g1, g_a), S.oracle_affine(g2, g2_b), S.oracle_affine(g1, g_c)


@pytest.mark.parametrize("pairing,num_inputs,num_aux,precompute,extra_aux", [("mnt4753", 4, 700, True, 0), ("mnt6753", 3, 300, True, 0),
                                                                              ("mnt4753", 2, 150, False, 0), ("mnt4753", 3, 120, True, 2)])
def test_create_proof_msm_stage_vs_oracle(gpu, pairing, num_inputs, num_aux, precompute, extra_aux):
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    rng = pyref.Rng(2024 + num_aux)
    n = num_inputs + num_aux                     # variables; h has n - 1 ... here: any length >= num_inputs
    pool1 = S.chain_points(C1, 96, rng)
    pool2 = S.chain_points(C2, 24, rng)
    def q1(m, k):
        return S.bases_array(C1, [pool1[(i * k + 1) % 96] for i in range(m)])[0]
    pk = {"a_query": q1(n, 5), "b_g1_query": q1(n, 7), "h_query": q1(n + 5, 11), "l_query": q1(num_aux, 13),
          "b_g2_query": S.bases_array(C2, [pool2[(i * 5 + 2) % 24] for i in range(n)])[0],
          "alpha_g1": q1(1, 1)[0], "beta_g1": q1(2, 3)[1], "delta_g1": q1(3, 17)[2],
          "beta_g2": S.bases_array(C2, [pool2[3]])[0][0], "delta_g2": S.bases_array(C2, [pool2[9]])[0][0]}
    r_ord = C1.order
    inp = S.scalar_array([rng.field_elem(r_ord) for _ in range(num_inputs - 1)])
    # witness-like: 0 / 1 / 2 mixed in; extra_aux: an aux vector longer than the queries (zip truncation, variable_base.rs:36)
    aux = S.scalar_array([rng.field_elem(r_ord) if i % 5 else i % 3 for i in range(num_aux + extra_aux)])
    h_inp = S.scalar_array([rng.field_elem(r_ord) for _ in range(num_inputs)])
    h_aux = S.scalar_array([rng.field_elem(r_ord) for _ in range(n + 5 - num_inputs)])
    r, s = S.scalar_array([rng.field_elem(r_ord), rng.field_elem(r_ord)])
    key = groth16.ResidentProvingKey(gpu, pairing, pk, num_inputs, precompute=precompute)
    try:
        got = key.create_proof_msms(inp, aux, h_inp, h_aux, r, s)
    finally:
        key.free()
    exp = _oracle_stage(pairing, pk, num_inputs, inp, aux, h_inp, h_aux, r, s)
    for name, (gxy, ginf), (exy, einf) in zip("ABC", got, exp):
        assert ginf == einf and (np.asarray(gxy) == np.asarray(exy)).all(), (pairing, name)


def test_proj_mul_neg_vs_oracle(gpu):
    for curve in ("mnt4753_g1", "mnt4753_g2", "mnt6753_g1", "mnt6753_g2"):
        C = pyref.CURVES[curve]
        rng = pyref.Rng(len(curve))
        P = S.chain_points(C, 1, rng)[0]
        xyz = S.proj_array(C, P)
        for k in (0, 1, 2, C.order - 1, C.order, rng.field_elem(C.order)):
            ks = S.scalar_array([k])[0]
            got = gpu.proj_to_affine(curve, gpu.proj_mul(curve, xyz, ks))
            exp = None if k % C.order == 0 else C.mul(k % C.order, P)
            assert S.affine_of_xyz(C, gpu.proj_mul(curve, xyz, ks)) == exp, (curve, k)
            assert got[1] == (exp is None)
        assert S.affine_of_xyz(C, gpu.proj_neg(curve, xyz)) == C.neg(P)
        assert (gpu.field_one(curve)[:12] == np.array(pyref.fe_to_abi(C.F, 1), dtype=np.uint64)).all()
