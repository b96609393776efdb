"""The config-5 test infrastructure checked against first principles on the CPU (no GPU): the Parameters stream made by
tests/groth16_ref.py parses back, the `Benchmark` rows satisfy the constraint system, and the oracle's replay of
create_proof yields exactly the group elements the Groth16 equations prescribe when everything is computed in the exponent
with the toxic waste (A = alpha + sum a_i(t) x_i + r delta, ... -- textbook, independent of the MSM code paths)."""
import importlib

import pyref
import support as S
import groth16_ref as G


def test_parameters_roundtrip_and_proof_in_the_exponent(gl):
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    pairing = "mnt4753"
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    r = C1.order
    blob, info = G.generate_parameters(pairing, 29, seed=11)
    pk = groth16.parse_parameters(pairing, blob)
    key = info["key"]
    assert pk["alpha_g1"] == G.wire(C1, key["alpha_g1"]) and pk["delta_g2"] == G.wire(C2, key["delta_g2"])
    assert pk["b_g2_query"] == b"".join(G.wire(C2, P) for P in key["b_g2_query"])
    assert len(pk["l_query"]) == 193 * (len(info["assignment"]) - info["num_inputs"])
    rows = groth16.benchmark_circuit_rows(pairing, 29)
    asg = info["assignment"]
    assert rows[0] == info["num_inputs"] and rows[1] == asg
    ev = lambda row: sum(cf * asg[ix] for cf, ix in row) % r
    assert [ev(x) for x in info["at"]] == rows[2] and [ev(x) for x in info["bt"]] == rows[3] and [ev(x) for x in info["ct"]] == rows[4]
    assert all(a * b % r == c for a, b, c in zip(rows[2], rows[3], rows[4]))
    # the proof in the exponent
    rng = pyref.Rng(4)
    d1, d2, d3, r_, s_ = (rng.field_elem(r) for _ in range(5))
    proof = G.oracle_create_proof(pairing, info, d1, d2, d3, r_, s_)
    alpha, beta, gamma, delta, t = info["toxic"]
    a, b, c, l, zt = info["qap"]
    g1, g2 = info["generators"]
    ni = info["num_inputs"]
    h = info["last_h"]
    size = 1 << info["log_n"]
    A_s = (alpha + sum(x * y for x, y in zip(asg, a)) + r_ * delta) % r
    B_s = (beta + sum(x * y for x, y in zip(asg, b)) + s_ * delta) % r
    di = pow(delta, -1, r)
    H = sum(h[j] * (zt * di % r) * pow(t, j, r) for j in range(size - 1)) % r
    C_s = (sum(asg[i] * l[i] for i in range(ni, len(asg))) + H + s_ * A_s + r_ * B_s - r_ * s_ * delta) % r
    exp = G.wire(C1, C1.mul(A_s, g1)) + G.wire(C2, C2.mul(B_s, g2)) + G.wire(C1, C1.mul(C_s, g1))
    assert proof == exp
    # and the QAP divisibility that makes it a valid proof: (sum a_i x_i)(sum b_i x_i) - sum c_i x_i == h(t) zt with d1 = d2 = d3 = 0
    proof0 = G.oracle_create_proof(pairing, info, 0, 0, 0, r_, s_)
    h0 = info["last_h"]
    lhs = (sum(x * y for x, y in zip(asg, a)) * sum(x * y for x, y in zip(asg, b)) - sum(x * y for x, y in zip(asg, c))) % r
    assert lhs == sum(h0[j] * pow(t, j, r) for j in range(size - 1)) * zt % r
    assert len(proof0) == 771
