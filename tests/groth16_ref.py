"""Test infrastructure for BASELINE config 5 (end-to-end create_proof): a first-principles Groth16 key generator for the
`Benchmark` circuit and the oracle-side replay of create_proof, both above tests/pyref.py and the C++ oracle.

  benchmark_lcs            the circuit's constraint system as linear combinations, following
                           proof-systems/src/groth16/examples/snark-scalability/constraints.rs:20-92 statement by statement
  generate_parameters      generator.rs:149-345 (instance_map_with_evaluation r1cs_to_qap.rs:20-85, Lagrange coefficients
                           domain.rs:183-219, queries = FixedBaseMSM of the oracle) -> Parameters::write bytes (mod.rs:188-208).
                           vk.alpha_g1_beta_g2 is a pairing value: opaque filler bytes here (pairings are out of scope and the
                           prover never reads it).
  oracle_create_proof      prover.rs:201-345 replayed literally on the CPU oracle -> Proof::write bytes (mod.rs:35-42)
Nothing here is shipped; the product side is ginger-lib_amd/groth16.py."""
import ctypes
import struct

import numpy as np

import pyref
import support as S

G2_DEG = {"mnt4753": 2, "mnt6753": 3}
FQK_BYTES = {"mnt4753": 4 * 96, "mnt6753": 6 * 96}


def benchmark_lcs(num_constraints, r):
    """-> (num_inputs, assignment, at, bt, ct): rows are lists of (coeff, index into the full assignment)"""
    num_inputs = 1                       # the "one" input variable (prover.rs:229 / generator.rs:172)
    assignment_in, assignment_aux = [1], []
    at, bt, ct = [], [], []

    def alloc_input(v):
        assignment_in.append(v)
        return ("in", len(assignment_in) - 1)

    def alloc(v):
        assignment_aux.append(v)
        return ("aux", len(assignment_aux) - 1)

    rec = []
    a_val = 1
    a_var = alloc_input(a_val)
    rec.append((a_val, a_var))
    b_val = 1
    b_var = alloc_input(b_val)
    rec.append((a_val, a_var))          # sic (:35)
    one = ("in", 0)
    for i in range(num_constraints - 1):
        if i % 2 != 0:
            c_val = a_val * b_val % r
            c_var = alloc(c_val)
            at.append([(1, a_var)]); bt.append([(1, b_var)]); ct.append([(1, c_var)])
        else:
            c_val = (a_val + b_val) % r
            c_var = alloc(c_val)
            at.append([(1, a_var), (1, b_var)]); bt.append([(1, one)]); ct.append([(1, c_var)])
        rec.append((c_val, c_var))
        a_val, a_var, b_val, b_var = b_val, b_var, c_val, c_var
    c_val = pow(sum(v for v, _ in rec) % r, 2, r)
    c_var = alloc(c_val)
    lc = [(1, var) for _, var in rec]
    at.append(list(lc)); bt.append(list(lc)); ct.append([(1, c_var)])
    ni = len(assignment_in)
    idx = lambda var: var[1] if var[0] == "in" else ni + var[1]
    conv = lambda rows: [[(cf, idx(v)) for cf, v in row] for row in rows]
    return ni, assignment_in + assignment_aux, conv(at), conv(bt), conv(ct)


def lagrange_coefficients(F, log_n, tau):
    """EvaluationDomain::evaluate_all_lagrange_coefficients (domain.rs:183-219), tau outside the domain"""
    p, n = F.p, 1 << log_n
    w = pyref.domain_params(F, log_n)
    t_size = pow(tau, n, p)
    assert t_size != 1
    l = (t_size - 1) * pow(n, -1, p) % p
    u, rr = [], 1
    for _ in range(n):
        u.append(l * pow((tau - rr) % p, -1, p) % p)
        l = l * w % p
        rr = rr * w % p
    return u


def _fixed(curve, g, ks, F):
    """[k * g for k in ks] as affine points (None = infinity), through the oracle's FixedBaseMSM restatement"""
    C = pyref.CURVES[curve]
    O = S.oracle()
    O.oracle_fixed_base_msm.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t,
                                        ctypes.c_void_p, ctypes.c_int]
    n = len(ks)
    out = np.zeros((max(n, 1), 36 * C.deg), dtype=np.uint64)
    sc = S.fe_array(F, ks)
    O.oracle_fixed_base_msm(S.CURVE_ID[curve], S.ptr(S.proj_array(C, g)), 753, 0, S.ptr(sc), n, S.ptr(out), 8)
    return [S.affine_of_xyz(C, out[i]) for i in range(n)]


def wire(C, P):
    x, y = (tuple([0] * C.deg), tuple([1] + [0] * (C.deg - 1))) if P is None else P
    out = bytearray()
    for coord in (x, y):
        for c in coord:
            out += int(c).to_bytes(96, "little")
    out.append(1 if P is None else 0)
    return bytes(out)


def generate_parameters(pairing, num_constraints, seed=1):
    """-> (Parameters::write bytes, dict with the toxic waste and the key as Python points for the oracle replay)"""
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    F = S.FIELD_OF[pairing + "_fr"]
    r = F.p
    rng = pyref.Rng(seed)
    ni, assignment, at, bt, ct = benchmark_lcs(num_constraints, r)
    n_con = len(at)
    n_aux = len(assignment) - ni
    size = 1
    while size < n_con + (ni - 1) + 1:
        size <<= 1
    log_n = size.bit_length() - 1
    alpha, beta, gamma, delta, t = (rng.field_elem(r) for _ in range(5))
    u = lagrange_coefficients(F, log_n, t)
    zt = (pow(t, size, r) - 1) % r
    nv = (ni - 1) + n_aux
    a, b, c = [0] * (nv + 1), [0] * (nv + 1), [0] * (nv + 1)
    for i in range(ni):
        a[i] = u[n_con + i]
    for i in range(n_con):
        for cf, ix in at[i]:
            a[ix] = (a[ix] + u[i] * cf) % r
        for cf, ix in bt[i]:
            b[ix] = (b[ix] + u[i] * cf) % r
        for cf, ix in ct[i]:
            c[ix] = (c[ix] + u[i] * cf) % r
    gi, di = pow(gamma, -1, r), pow(delta, -1, r)
    comb = [(beta * x + alpha * y + z) % r for x, y, z in zip(a, b, c)]
    gamma_abc = [v * gi % r for v in comb[:ni]]
    l = [v * di % r for v in comb]
    g1 = C1.mul(rng.next_u64() | 1, C1.G)
    g2 = C2.mul(rng.next_u64() | 1, C2.G)
    key = {
        "alpha_g1": C1.mul(alpha, g1), "beta_g1": C1.mul(beta, g1), "beta_g2": C2.mul(beta, g2),
        "delta_g1": C1.mul(delta, g1), "delta_g2": C2.mul(delta, g2), "gamma_g2": C2.mul(gamma, g2),
        "a_query": _fixed(pairing + "_g1", g1, a, F), "b_g1_query": _fixed(pairing + "_g1", g1, b, F),
        "b_g2_query": _fixed(pairing + "_g2", g2, b, F),
        "h_query": _fixed(pairing + "_g1", g1, [zt * di % r * pow(t, i, r) % r for i in range(size - 1)], F),
        "l_query": _fixed(pairing + "_g1", g1, l, F)[ni:],
        "gamma_abc_g1": _fixed(pairing + "_g1", g1, gamma_abc, F),
    }
    blob = bytearray()
    blob += bytes((i * 37 + 5) & 0xFF for i in range(FQK_BYTES[pairing]))      # vk.alpha_g1_beta_g2: opaque here
    blob += wire(C2, key["gamma_g2"]) + wire(C2, key["delta_g2"])
    blob += struct.pack(">I", len(key["gamma_abc_g1"])) + b"".join(wire(C1, P) for P in key["gamma_abc_g1"])
    blob += wire(C1, key["alpha_g1"]) + wire(C1, key["beta_g1"]) + wire(C2, key["beta_g2"]) + wire(C1, key["delta_g1"]) + wire(C2, key["delta_g2"])
    for name, Cq in (("a_query", C1), ("b_g1_query", C1), ("b_g2_query", C2), ("h_query", C1), ("l_query", C1)):
        blob += struct.pack(">I", len(key[name])) + b"".join(wire(Cq, P) for P in key[name])
    info = {"num_inputs": ni, "assignment": assignment, "at": at, "bt": bt, "ct": ct, "log_n": log_n, "key": key, "toxic": (alpha, beta, gamma, delta, t),
            "qap": (a, b, c, l, zt), "generators": (g1, g2)}
    return bytes(blob), info


def oracle_create_proof(pairing, info, d1, d2, d3, r_, s_):
    """prover.rs:201-345 on the CPU oracle: rows -> witness_map (oracle) -> into_repr -> the nine MSMs (oracle) -> Proof::write"""
    from test_groth16_stage import _oracle_stage
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    field = pairing + "_fr"
    F = S.FIELD_OF[field]
    r = F.p
    ni, asg, at, bt, ct = info["num_inputs"], info["assignment"], info["at"], info["bt"], info["ct"]
    n_con, size = len(at), 1 << info["log_n"]
    ev = lambda row: sum(cf * asg[ix] for cf, ix in row) % r
    a = [ev(row) for row in at] + [0] * (size - n_con)
    b = [ev(row) for row in bt] + [0] * (size - n_con)
    c = [ev(row) for row in ct] + [0] * (size - n_con)
    for i in range(ni):
        a[n_con + i] = asg[i] if i > 0 else 1
    h = S.oracle_witness_map(field, S.fe_array(F, a), S.fe_array(F, b), S.fe_array(F, c), *(S.fe_array(F, [d])[0] for d in (d1, d2, d3)))
    h_ints = S.fe_list(F, h)                                             # into_repr
    key = info["key"]
    pk = {}
    for name, Cq in (("a_query", C1), ("b_g1_query", C1), ("b_g2_query", C2), ("h_query", C1), ("l_query", C1)):
        pk[name], pk[name + "_inf"] = S.bases_array(Cq, key[name])
    for name, Cq in (("alpha_g1", C1), ("beta_g1", C1), ("delta_g1", C1), ("beta_g2", C2), ("delta_g2", C2)):
        pk[name] = S.bases_array(Cq, [key[name]])[0][0]
    sc = lambda vals: S.scalar_array(vals)
    A, B, Cc = _oracle_stage(pairing, pk, ni, sc(asg[1:ni]), sc(asg[ni:]), sc(h_ints[:ni]), sc(h_ints[ni:]), sc([r_])[0], sc([s_])[0])

    def to_wire(Cq, res):
        xy, inf = res
        k = Cq.deg
        v = [int(x) for x in np.asarray(xy).ravel()]
        P = None if inf else (pyref.ext_from_abi(Cq.F, v[:12 * k], k), pyref.ext_from_abi(Cq.F, v[12 * k:], k))
        return wire(Cq, P)
    info["last_h"] = h_ints
    return to_wire(C1, A) + to_wire(C2, B) + to_wire(C1, Cc)
