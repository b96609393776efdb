"""The GM17 prover over a device-resident key: ginger-lib_amd/gm17.py against a LITERAL replay of
proof-systems/src/gm17/prover.rs:267-352 on the CPU oracle -- nine multi_scalar_mul calls with the reference's slices
(gm17/mod.rs:237-330), every `mul`, `add_assign` and `into_affine` as written -- and, end to end, against that replay fed by
the oracle's R1CStoSAP::witness_map (r1cs_to_sap.rs:99-245).  Bit-exact on the three affine proof elements, both pairings,
with and without shift tables.  (GM17's Proof::write is unimplemented upstream; the elements are compared as
GroupAffine::write would serialise them.)"""
import importlib

import numpy as np
import pytest

import pyref
import support as S

pytestmark = pytest.mark.gpu


def _oracle_gm17_stage(pairing, pk, ni, inp, aux, h_inp, h_aux, d1, d2, r):
    g1, g2 = pairing + "_g1", pairing + "_g2"
    O = S.oracle()
    mod = pyref.CURVES[g1].order

    def ec(curve, op, p, q=None, flag=0):
        C = pyref.CURVES[curve]
        out = np.zeros(36 * C.deg, dtype=np.uint64)
        p = np.ascontiguousarray(p, dtype=np.uint64)
        qq = None if q is None else np.ascontiguousarray(q, dtype=np.uint64)
        O.oracle_ec_op(S.CURVE_ID[curve], op, S.ptr(p), None if qq is None else S.ptr(qq), flag, S.ptr(out))
        return out

    def zero(curve):
        return ec(curve, 3, S.proj_array(pyref.CURVES[curve], None), np.zeros(12, dtype=np.uint64))

    proj = lambda curve, xy: ec(curve, 2, zero(curve), xy, 0)          # into_projective
    add = lambda curve, a, b: ec(curve, 0, a, b)
    sc = lambda k: S.scalar_array([k % mod])[0]
    mul = lambda curve, a, k: ec(curve, 3, a, sc(k))                    # .mul(k)
    msm = lambda curve, bases, scal: S.oracle_msm(curve, bases, None, scal, 8)
    # Compute A (:268-279)
    a_inputs_acc = msm(g1, pk["a_query"][1:ni], inp)
    a_aux_acc = msm(g1, pk["a_query"][ni:], aux)
    g_a = mul(g1, proj(g1, pk["g_gamma_z"]), r)
    for t in (proj(g1, pk["a_query"][0]), mul(g1, proj(g1, pk["g_gamma_z"]), d1), a_inputs_acc, a_aux_acc):
        g_a = add(g1, g_a, t)
    # Compute B (:284-296)
    b_inputs_acc = msm(g2, pk["b_query"][1:ni], inp)
    b_aux_acc = msm(g2, pk["b_query"][ni:], aux)
    g_b = mul(g2, proj(g2, pk["h_gamma_z"]), r)
    for t in (proj(g2, pk["b_query"][0]), mul(g2, proj(g2, pk["h_gamma_z"]), d1), b_inputs_acc, b_aux_acc):
        g_b = add(g2, g_b, t)
    # Compute C (:300-343)
    r_2, r2 = 2 * r, r * r
    d1_r_2 = d1 * r_2
    c1_acc = msm(g1, pk["c_query_1"], aux)                               # get_c_query_1(0): (.., c_query_1[0..])
    c2_acc = add(g1, msm(g1, pk["c_query_2"][1:ni], inp), msm(g1, pk["c_query_2"][ni:], aux))
    g_acc = add(g1, msm(g1, pk["g_gamma2_z_t"][0:ni], h_inp), msm(g1, pk["g_gamma2_z_t"][ni:], h_aux))
    g_c = c1_acc
    for t in (mul(g1, proj(g1, pk["g_gamma2_z2"]), r2), mul(g1, proj(g1, pk["g_ab_gamma_z"]), r), mul(g1, proj(g1, pk["g_ab_gamma_z"]), d1),
              mul(g1, proj(g1, pk["c_query_2"][0]), r), mul(g1, proj(g1, pk["g_gamma2_z2"]), d1_r_2), mul(g1, c2_acc, r),
              mul(g1, proj(g1, pk["g_gamma2_z_t"][0]), d2), g_acc):
        g_c = add(g1, g_c, t)
    return S.oracle_affine(g1, g_a), S.oracle_affine(g2, g_b), S.oracle_affine(g1, g_c)


def _synthetic_key(pairing, ni, n_var, n_h, seed):
    """any points do for the equality of the two prover paths (a key that verifies needs the generator, out of scope)"""
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    rng = pyref.Rng(seed)
    pool1, pool2 = S.chain_points(C1, 128, rng), S.chain_points(C2, 32, rng)
    q1 = lambda m, k: S.bases_array(C1, [pool1[(i * k + 1) % 128] for i in range(m)])[0]
    return {"a_query": q1(n_var + 1, 5), "c_query_2": q1(n_var + 1, 7), "c_query_1": q1(n_var + 1 - ni, 11), "g_gamma2_z_t": q1(n_h, 13),
            "b_query": S.bases_array(C2, [pool2[(i * 5 + 2) % 32] for i in range(n_var + 1)])[0],
            "g_gamma_z": q1(2, 17)[1], "g_ab_gamma_z": q1(3, 19)[2], "g_gamma2_z2": q1(4, 23)[3],
            "h_gamma_z": S.bases_array(C2, [pool2[9]])[0][0]}


@pytest.mark.parametrize("pairing,num_inputs,num_aux,precompute", [("mnt4753", 4, 900, True), ("mnt6753", 3, 260, True), ("mnt4753", 2, 5000, False),
                                                                   ("mnt6753", 2, 4400, True)])
def test_gm17_msm_stage_vs_literal_oracle_replay(gpu, pairing, num_inputs, num_aux, precompute):
    gm17 = importlib.import_module("ginger_lib_amd.gm17")
    C1 = pyref.CURVES[pairing + "_g1"]
    mod = C1.order
    rng = pyref.Rng(77 + num_aux)
    n_var = (num_inputs - 1) + num_aux
    n_h = n_var + 9
    pk = _synthetic_key(pairing, num_inputs, n_var, n_h, 5 + num_aux)
    inp = S.scalar_array([rng.field_elem(mod) for _ in range(num_inputs - 1)])
    aux = S.scalar_array([rng.field_elem(mod) if i % 5 else i % 3 for i in range(num_aux)])          # witness-like: 0 / 1 / 2 mixed in
    h = S.scalar_array([rng.field_elem(mod) for _ in range(n_h)])
    d1, d2, r = (rng.field_elem(mod) for _ in range(3))
    key = gm17.ResidentGm17Key(gpu, pairing, pk, num_inputs, precompute=precompute)
    try:
        got = key.create_proof_msms(inp, aux, h, d1, d2, r)
        got0 = key.create_proof_msms(inp, aux, h, 0, 0, r)                # d1 = d2 = 0: the blinding terms vanish
    finally:
        key.free()
        gpu.dev_trim()
    exp = _oracle_gm17_stage(pairing, pk, num_inputs, inp, aux, h[:num_inputs], h[num_inputs:], d1, d2, r)
    exp0 = _oracle_gm17_stage(pairing, pk, num_inputs, inp, aux, h[:num_inputs], h[num_inputs:], 0, 0, r)
    for g, e in ((got, exp), (got0, exp0)):
        for name, (gxy, ginf), (exy, einf) in zip("ABC", g, e):
            assert ginf == einf and (np.asarray(gxy) == np.asarray(exy)).all(), (pairing, name)
    assert gm17.proof_bytes(pairing, got) != gm17.proof_bytes(pairing, got0)


@pytest.mark.parametrize("pairing,n_con,precompute", [("mnt4753", 253, True), ("mnt6753", 125, False)])
def test_gm17_create_proof_end_to_end_vs_oracle(gpu, pairing, n_con, precompute):
    """rows of the `Benchmark` circuit -> SAP witness map (host rows + device transforms) -> into_repr -> MSM stage, against the
    oracle's sap_witness_map + the literal replay above"""
    gm17 = importlib.import_module("ginger_lib_amd.gm17")
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    C1 = pyref.CURVES[pairing + "_g1"]
    mod = C1.order
    field = pairing + "_fr"
    F = S.FIELD_OF[field]
    rows = groth16.benchmark_circuit_rows(pairing, n_con)
    ni = rows[0]
    full, a, c, log_n = gm17.sap_rows_from_r1cs(pairing, *rows)
    size = 1 << log_n
    # the extended assignment satisfies the SAP: a_k^2 = c_k on the constraint rows (r1cs_to_sap.rs:14-98 builds exactly this system)
    assert all(x * x % mod == y for x, y in zip(a, c))
    n_var = len(full) - 1
    assert n_var == 2 * (ni - 1) + (len(rows[1]) - ni) + n_con               # sap_num_variables (:36-37)
    pk = _synthetic_key(pairing, ni, n_var, size + 1, 31)
    rng = pyref.Rng(9)
    d1, d2, r = (rng.field_elem(mod) for _ in range(3))
    key = gm17.ResidentGm17Key(gpu, pairing, pk, ni, precompute=precompute)
    try:
        got = key.create_proof(rows, d1, d2, r)
    finally:
        key.free()
        gpu.dev_trim()
    h = S.oracle_sap_witness_map(field, S.fe_array(F, a), S.fe_array(F, c), S.fe_array(F, [d1])[0], S.fe_array(F, [d2])[0])
    h_ints = S.scalar_array(S.fe_list(F, h))                                  # into_repr
    scal = S.scalar_array(full)
    exp = _oracle_gm17_stage(pairing, pk, ni, scal[1:ni], scal[ni:], h_ints[:ni], h_ints[ni:], d1, d2, r)
    for name, (gxy, ginf), (exy, einf) in zip("ABC", got, exp):
        assert ginf == einf and (np.asarray(gxy) == np.asarray(exy)).all(), (pairing, name)


@pytest.mark.gpu
def test_gm17_key_generated_on_device_and_proof_in_closed_form(gpu):
    """GM17 to Groth16's standard of evidence: the proving key of the `Benchmark` circuit with 2^16 - 3
    constraints (SAP domain 2^17) is GENERATED on the device -- gm17.generate_parameters mirrors generator.rs:146-335 with its
    five FixedBaseMSM::multi_scalar_mul calls on gh_fixed_base_msm_affine --, made resident, and used by create_proof
    (prover.rs:201-352: SAP witness map on the device, five MSMs).  A, B, C are then compared with GM17's equations
    evaluated IN THE EXPONENT from the toxic waste with Python integers and three textbook scalar multiplications:
        A = gamma (sum a_i(t) x_i + r Z(t)) g,   B = the same scalar on h,
        C = (sum_aux x_i (gamma c_i(t) + (alpha + beta) a_i(t)) + r^2 gamma^2 Z^2 + r (alpha + beta) gamma Z
             + 2 r gamma^2 Z sum a_i x_i + gamma^2 ((sum a_i x_i)^2 - sum c_i x_i)) g
    (d1 = d2 = 0; the last term is gamma^2 Z(t) H(t) by the SAP's divisibility) -- no MSM or FFT code path involved."""
    import time
    gm17 = importlib.import_module("ginger_lib_amd.gm17")
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    pairing = "mnt4753"
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    r = C1.order
    n_con = (1 << 16) - 3
    rng = pyref.Rng(417)
    alpha, beta, gamma, t, r_ = (rng.field_elem(r) for _ in range(5))
    g1, g2 = C1.mul(rng.next_u64() | 1, C1.G), C2.mul(rng.next_u64() | 1, C2.G)
    t0 = time.perf_counter()
    lcs = groth16.benchmark_circuit_lcs(n_con)
    pk, info = gm17.generate_parameters(gpu, pairing, lcs, alpha, beta, gamma, t, S.proj_array(C1, g1), S.proj_array(C2, g2))
    t_gen = time.perf_counter() - t0
    assert info["log_n"] == 17 and info["fixed_base"]["fixed_base_calls"] == 5
    assert int(pk["a_query_inf"].sum()) == sum(1 for v in info["sap"][0] if v == 0) > 0        # the extra SAP variables do not occur in A
    rows = groth16.benchmark_circuit_rows(pairing, n_con)
    key = gm17.ResidentGm17Key(gpu, pairing, pk, 3)
    try:
        t0 = time.perf_counter()
        proof = key.create_proof(rows, 0, 0, r_)
        t_proof = time.perf_counter() - t0
    finally:
        key.free()
        gpu.dev_trim()
    print("gm17 generate %.1f s (%d fixed-base scalar-muls in 5 calls, windows g %d h %d), create_proof %.2f s" % (
        t_gen, info["fixed_base"]["fixed_base_scalars"], info["g_window"], info["h_window"], t_proof))
    a, c, zt = info["sap"]
    full, _, _, _ = gm17.sap_rows_from_r1cs(pairing, 3, rows[1], rows[2], rows[3], rows[4])
    assert len(full) == len(a) == info["sap_num_variables"] + 1
    sa = sum(x * y for x, y in zip(full, a)) % r
    sc = sum(x * y for x, y in zip(full, c)) % r
    ab = (alpha + beta) % r
    A_s = gamma * (sa + r_ * zt) % r
    C_s = (sum(full[i] * (gamma * c[i] + ab * a[i]) for i in range(3, len(full))) + r_ * r_ * gamma * gamma % r * zt * zt
           + r_ * ab * gamma % r * zt + 2 * r_ * gamma * gamma % r * zt * sa + gamma * gamma * (sa * sa - sc)) % r
    import groth16_ref as G
    exp = G.wire(C1, C1.mul(A_s, g1)) + G.wire(C2, C2.mul(A_s, g2)) + G.wire(C1, C1.mul(C_s, g1))
    assert gm17.proof_bytes(pairing, proof) == exp
