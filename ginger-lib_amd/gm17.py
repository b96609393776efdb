"""The GM17 prover over a device-resident proving key.

Mirror of proof-systems/src/gm17/prover.rs:201-352 (create_proof): the R1CS -> SAP witness map
(gm17/r1cs_to_sap.rs:99-245; its transforms are gh_sap_witness_map_dev) followed by the MSM stage
(prover.rs:267-343): nine multi_scalar_mul calls over a_query, b_query, c_query_1, c_query_2 and
g_gamma2_z_t (gm17/mod.rs:136-147, getters :237-330), a dozen single scalar multiplications and additions,
then into_affine() of A, B, C.  Same restructuring as groth16.py: every query is ONE resident vector that
carries the single points the reference adds by hand, so a proof is five MSMs -- four on G1 as one pipelined
batch (gh_msm_resident_dev_batch), one on G2 -- and the sums are the same group elements:

  A = MSM(a_query[1..] || a_query[0] || g_gamma_z ; input || aux || 1 || r + d1)                       (:268-279)
  B = MSM(b_query[1..] || b_query[0] || h_gamma_z ; the same scalars)                                  (:284-296)
  C = MSM(c_query_1 || g_gamma2_z2 || g_ab_gamma_z ; aux || r^2 + 2 r d1 || r + d1)                   (:306-308, :327-331)
      + r * MSM(c_query_2[1..] || c_query_2[0] ; input || aux || 1)                                    (:313-317, :330, :332-333)
      + MSM(g_gamma2_z_t || g_gamma2_z_t[0] ; h || d2)                                                 (:322-325, :331)

GM17's Proof / Parameters have no byte format in the reference (gm17/mod.rs:59-69, :186-196: unimplemented), so the
proof is returned as three affine points in the C ABI's form; `proof_bytes` writes them the way GroupAffine::write
would (short_weierstrass_projective.rs:185-192), which is what the parity tests compare.
"""
import numpy as np

from .groth16 import _MODULUS, _canon_rows, _canon_rows_fast, _ints_from_mont_rows, _mont_rows, affine_to_wire


def sap_rows_from_r1cs(pairing, num_inputs, assignment, A, B, C):
    """The host part of R1CStoSAP::witness_map (r1cs_to_sap.rs:123-148, :159-184, :198-219): from the evaluated R1CS rows
    A_i = <at_i, x>, B_i, C_i and the assignment x (Python integers) to the extended assignment and the two vectors the
    transforms start from.  -> (full_assignment, a, c, log_n): a and c have 2^log_n entries."""
    r = _MODULUS[pairing]
    n_con = len(A)
    ni = num_inputs
    full = list(assignment)
    n_var = len(full)                                            # num_inputs + num_aux
    extra = [(x - y) * (x - y) % r for x, y in zip(A, B)]        # :127-141
    full += extra
    for i in range(1, ni):                                       # :144-149
        full.append((full[i] - 1) * (full[i] - 1) % r)
    size = 1
    while size < 2 * n_con + 2 * (ni - 1) + 1:                   # :151-155
        size <<= 1
    off = 2 * n_con
    var_off, var_off2 = n_var, n_var + n_con - 1
    a = [0] * size
    c = [0] * size
    for i in range(n_con):
        a[2 * i] = (A[i] + B[i]) % r
        a[2 * i + 1] = (A[i] - B[i]) % r
        c[2 * i] = (4 * C[i] + full[var_off + i]) % r            # :200-212
        c[2 * i + 1] = full[var_off + i]
    a[off] = 1
    c[off] = 1
    for i in range(1, ni):
        a[off + 2 * i - 1] = (full[i] + 1) % r
        a[off + 2 * i] = (full[i] - 1) % r
        c[off + 2 * i - 1] = (4 * full[i] + full[var_off2 + i]) % r
        c[off + 2 * i] = full[var_off2 + i]
    return full, a, c, size.bit_length() - 1


class ResidentGm17Key:
    """pk: dict of numpy arrays in the ABI formats (Montgomery x || y rows, all points finite): a_query, c_query_2 (n x 24),
    c_query_1 (n - num_inputs rows), g_gamma2_z_t (m x 24), b_query (n x 24*deg), and the single points g_gamma_z,
    g_ab_gamma_z, g_gamma2_z2 (24 u64), h_gamma_z (24*deg u64)."""

    def __init__(self, gl, pairing, pk, num_inputs, precompute=True):
        assert pairing in ("mnt4753", "mnt6753")
        self.gl, self.pk, self.num_inputs, self.pairing = gl, pk, int(num_inputs), pairing
        self.g1, self.g2 = pairing + "_g1", pairing + "_g2"
        row = lambda v: np.asarray(v, dtype=np.uint64).reshape(1, -1)
        cat = lambda *parts: np.ascontiguousarray(np.concatenate(parts), dtype=np.uint64)
        vectors = {
            "a": (self.g1, cat(pk["a_query"][1:], row(pk["a_query"][0]), row(pk["g_gamma_z"]))),
            "b": (self.g2, cat(pk["b_query"][1:], row(pk["b_query"][0]), row(pk["h_gamma_z"]))),
            "c1": (self.g1, cat(pk["c_query_1"], row(pk["g_gamma2_z2"]), row(pk["g_ab_gamma_z"]))),
            "c2": (self.g1, cat(pk["c_query_2"][1:], row(pk["c_query_2"][0]))),
            "g": (self.g1, cat(pk["g_gamma2_z_t"], row(pk["g_gamma2_z_t"][0]))),
        }
        # keys made by generate_parameters hold GroupAffine::zero() wherever a variable does not occur in A (the extra SAP
        # variables: r1cs_to_sap.rs:68, :85-91): pk["<query>_inf"] flags them (absent = all finite)
        def flags(q, head_last, tail):
            f = pk.get(q + "_inf")
            if f is None:
                return None
            f = np.asarray(f, dtype=np.uint8)
            f = np.concatenate([f[1:], f[:1]]) if head_last else f
            return np.ascontiguousarray(np.concatenate([f, np.zeros(tail, dtype=np.uint8)]))
        inf = {"a": flags("a_query", True, 1), "b": flags("b_query", True, 1), "c1": flags("c_query_1", False, 2),
               "c2": flags("c_query_2", True, 0), "g": None}
        self.keys = {}
        for name, (curve, rows) in vectors.items():
            rb = gl.ResidentBases(curve, rows, infinity=inf[name])
            if precompute and rb.n:
                try:
                    rb.precompute(0)
                except gl.GingerHipError:
                    pass                      # no memory for the table / a point of 2-power order: per-window path
            self.keys[name] = rb

    def free(self):
        for rb in self.keys.values():
            rb.free()

    def create_proof_msms(self, input_assignment, aux_assignment, h, d1, d2, r, h_dev=None):
        """prover.rs:267-352 after the witness map.  input_assignment: num_inputs - 1 rows, aux_assignment: the rest of
        the EXTENDED assignment (aux, then the extra SAP variables), h: its coefficients -- all canonical 12-u64 rows
        (into_repr, :224-252); d1, d2, r: Python integers.  h_dev = (DeviceBuffer, rows): h already on the device as canonical
        scalars.  Returns (A, B, C) as (xy, is_infinity) pairs."""
        gl, ni = self.gl, self.num_inputs
        g1, g2 = self.g1, self.g2
        mod = _MODULUS[self.pairing]
        inp = np.ascontiguousarray(input_assignment, dtype=np.uint64).reshape(-1, 12)
        aux = np.ascontiguousarray(aux_assignment, dtype=np.uint64).reshape(-1, 12)
        assert len(inp) == ni - 1
        k = self.keys
        n_var = k["a"].n - 2                                    # variables the a / b / c_2 queries pair with: input || aux
        assert k["b"].n - 2 == n_var and k["c2"].n - 1 == n_var and k["c1"].n - 2 == n_var - (ni - 1)
        aux_used = aux[:n_var - len(inp)]                        # the reference's zip cuts a longer vector here (variable_base.rs:36)
        pad = np.zeros((n_var - len(inp) - len(aux_used), 12), dtype=np.uint64)
        one = np.zeros((1, 12), dtype=np.uint64)
        one[0, 0] = 1
        r_d1 = _canon_rows([(r + d1) % mod])
        c1_tail = _canon_rows([(r * r + 2 * r * d1) % mod, (r + d1) % mod])
        # ONE device vector  input || aux || 1 || (r + d1) || (r^2 + 2 r d1) || (r + d1):
        #   a / b keys take rows [0, n_var + 2), c_2 rows [0, n_var + 1), c_1 rows [ni - 1, n_var) followed by its own two
        # -- c_1's tail is not contiguous with the aux part, so c_1 gets the aux rows again behind the shared vector
        rows = [inp, aux_used, pad, one, r_d1]
        d_s = gl.DeviceBuffer((n_var + 2) * 96)
        d_c1 = gl.DeviceBuffer((n_var - (ni - 1) + 2) * 96)
        import ctypes
        lib = gl.load_library()

        def fill(buf, parts):
            row = 0
            for part in parts:
                if len(part):
                    part = np.ascontiguousarray(part, dtype=np.uint64)
                    gl._check(lib.gh_dev_upload(ctypes.c_void_p(buf.ptr.value + row * 96), gl._ptr(part), part.nbytes))
                    row += len(part)
            return row
        assert fill(d_s, rows) == n_var + 2
        assert fill(d_c1, [aux_used, pad, c1_tail]) == n_var - (ni - 1) + 2
        own_h = h_dev is None
        if own_h:
            h_all = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 12)
            n_h = min(len(h_all), k["g"].n - 1)
            hv = np.concatenate([h_all[:n_h], np.zeros((k["g"].n - 1 - n_h, 12), dtype=np.uint64), _canon_rows([d2 % mod])])
            d_h = gl.DeviceBuffer(hv.nbytes).upload(hv)
        else:
            d_h, n_h = h_dev[0], int(h_dev[1])
            assert n_h == k["g"].n - 1                           # the caller left room for one more row behind h
            row_d2 = _canon_rows([d2 % mod])
            gl._check(lib.gh_dev_upload(ctypes.c_void_p(d_h.ptr.value + n_h * 96), gl._ptr(row_d2), 96))
        try:
            g_a, c1, c2, g_acc = gl.msm_batch_dev([(k["a"], d_s, n_var + 2), (k["c1"], d_c1, n_var - (ni - 1) + 2),
                                                   (k["c2"], d_s, n_var + 1), (k["g"], d_h, k["g"].n)])
            g_b = k["b"].msm_dev(d_s, n_var + 2)
        finally:
            d_s.free()
            d_c1.free()
            if own_h:
                d_h.free()
        g_c = gl.proj_add(g1, c1, gl.proj_mul(g1, c2, _canon_rows([r % mod])[0]))
        g_c = gl.proj_add(g1, g_c, g_acc)
        return gl.proj_to_affine(g1, g_a), gl.proj_to_affine(g2, g_b), gl.proj_to_affine(g1, g_c)

    def create_proof(self, circuit_rows, d1, d2, r):
        """create_proof (prover.rs:201-352) for evaluated constraint rows (groth16.benchmark_circuit_rows form): the host part
        of the SAP witness map, its transforms on the device (gh_sap_witness_map_dev), into_repr on the device, the MSM stage."""
        gl, pairing = self.gl, self.pairing
        mod = _MODULUS[pairing]
        num_inputs, assignment, A, B, C = circuit_rows
        assert num_inputs == self.num_inputs
        full, a, c, log_n = sap_rows_from_r1cs(pairing, num_inputs, assignment, A, B, C)
        size = 1 << log_n
        field = "mnt4753_fr" if pairing == "mnt4753" else "mnt6753_fr"
        dd = _mont_rows([d1, d2], mod)
        lib = gl.load_library()
        bufs = [gl.DeviceBuffer(size * 96 + 192) for _ in range(3)]
        try:
            bufs[0].upload(_mont_rows(a, mod))
            bufs[1].upload(_mont_rows(c, mod))
            gl._check(lib.gh_sap_witness_map_dev(gl.FIELDS[field], bufs[0].ptr, bufs[1].ptr, log_n, gl._ptr(dd[0]), gl._ptr(dd[1]), bufs[2].ptr))
            one_plain = np.zeros(12, dtype=np.uint64)
            one_plain[0] = 1
            gl._check(lib.gh_vec_scale_dev(gl.FIELDS[field], bufs[2].ptr, gl._ptr(one_plain), size + 1))      # into_repr of h (:237-252)
            scal = _canon_rows(full)
            n_h = self.keys["g"].n - 1
            assert n_h <= size + 1
            return self.create_proof_msms(scal[1:num_inputs], scal[num_inputs:], None, d1, d2, r, h_dev=(bufs[2], n_h))
        finally:
            for buf in bufs:
                buf.free()


def generate_parameters(gl, pairing, lcs, alpha, beta, gamma, t, g1_xyz, g2_xyz):
    """generate_parameters (proof-systems/src/gm17/generator.rs:146-335) above the C ABI, for a constraint system given as linear
    combinations (groth16.benchmark_circuit_lcs form); the toxic waste alpha, beta, gamma, the evaluation point t
    (sample_element_outside_domain, :183) and the generators g, h (`rand`, :41-42) are arguments instead of RNG draws.
      * Lagrange coefficients at t on the device (gh_lagrange_coefficients = evaluate_all_lagrange_coefficients, domain.rs:183-219);
      * R1CStoSAP::instance_map_with_evaluation (r1cs_to_sap.rs:14-96): host integers, as in the reference;
      * the FIVE FixedBaseMSM::multi_scalar_mul calls (:222-229 a_query, :245-254 g_gamma2_z_t, :259-270 verifier query ||
        c_query_1, :274-285 c_query_2, :297-302 b_query on h^gamma) with the reference's window rule (:198-213, :290) followed
        by batch_normalization + into_affine (:323-333): gh_fixed_base_msm_affine;
      * the single points (:233-241) by host-side scalar multiplications (gh_proj_mul).
    Returns (pk, info): pk in ResidentGm17Key's form (Montgomery x || y rows + `<query>_inf` flags; GM17's Parameters::write is
    unimplemented upstream, gm17/mod.rs:186-196), info = the SAP evaluations a, c, Z(t) for a check in the exponent."""
    r = _MODULUS[pairing]
    field = "mnt4753_fr" if pairing == "mnt4753" else "mnt6753_fr"
    g1c, g2c = pairing + "_g1", pairing + "_g2"
    num_inputs, num_aux, at, bt, ct = lcs
    n_con = len(at)
    ni1 = num_inputs - 1
    size = 1
    while size < 2 * n_con + 2 * ni1 + 1:                                          # r1cs_to_sap.rs:18-21
        size <<= 1
    log_n = size.bit_length() - 1
    zt = (pow(t, size, r) - 1) % r
    u_rows = np.zeros((size, 12), dtype=np.uint64)
    gl._check(gl.load_library().gh_lagrange_coefficients(gl.FIELDS[field], log_n, gl._ptr(_mont_rows([t], r)[0]), gl._ptr(u_rows)))
    u = _ints_from_mont_rows(u_rows, r)
    sap_nv = 2 * ni1 + num_aux + n_con                                             # :30-31
    xvo, xco, xvo2 = ni1 + num_aux + 1, 2 * n_con, ni1 + num_aux + n_con           # :32-35
    a, c = [0] * (sap_nv + 1), [0] * (sap_nv + 1)
    for i in range(n_con):                                                         # :40-73
        u0, u1 = u[2 * i], u[2 * i + 1]
        ua, us = (u0 + u1) % r, (u0 - u1) % r
        for cf, ix in at[i]:
            a[ix] = (a[ix] + ua * cf) % r
        for cf, ix in bt[i]:
            a[ix] = (a[ix] + us * cf) % r
        for cf, ix in ct[i]:
            c[ix] = (c[ix] + 4 * u0 * cf) % r
        c[xvo + i] = (c[xvo + i] + ua) % r
    a[0] = (a[0] + u[xco]) % r                                                     # :75-76
    c[0] = (c[0] + u[xco]) % r
    for i in range(1, ni1 + 1):                                                    # :78-94
        uo, ue = u[xco + 2 * i - 1], u[xco + 2 * i]
        a[i] = (a[i] + uo + ue) % r
        a[0] = (a[0] + uo - ue) % r
        c[i] = (c[i] + 4 * uo) % r
        c[xvo2 + i] = (c[xvo2 + i] + uo + ue) % r
    non_zero_a = sum(1 for v in a[:sap_nv] if v)                                   # generator.rs:191-195
    m_raw = size
    FB = gl.FixedBaseMSM
    g_window = FB.get_mul_window_size(num_inputs + non_zero_a + (sap_nv - ni1) + sap_nv + 1 + m_raw + 1)      # :198-211
    h_window = FB.get_mul_window_size(non_zero_a)                                  # :290
    ab = (alpha + beta) % r
    gz = gamma * zt % r
    stats = {"fixed_base_scalars": 0, "fixed_base_calls": 0}

    def rows(table, vals):
        xy, inf = table.multi_scalar_mul_affine(_canon_rows_fast(vals), canonical=False)
        stats["fixed_base_scalars"] += len(vals)
        stats["fixed_base_calls"] += 1
        return np.ascontiguousarray(xy, dtype=np.uint64), np.asarray(inf, dtype=np.uint8)

    def point(curve, xyz, k):
        xy, inf = gl.proj_to_affine(curve, gl.proj_mul(curve, xyz, _canon_rows([k % r])[0]))
        assert not inf
        return np.asarray(xy, dtype=np.uint64).reshape(-1)

    pk = {}
    tg = FB(g1c, g1_xyz, 753, g_window)
    try:
        pk["a_query"], pk["a_query_inf"] = rows(tg, [v * gamma % r for v in a])
        g2zt = gz * gamma % r
        pw, cur = [], g2zt
        for _ in range(m_raw + 1):
            pw.append(cur)
            cur = cur * t % r
        pk["g_gamma2_z_t"], ginf = rows(tg, pw)
        assert not ginf.any()
        res, rinf = rows(tg, [(cv * gamma + av * ab) % r for av, cv in zip(a, c)])
        pk["verifier_query"], pk["c_query_1"], pk["c_query_1_inf"] = res[:num_inputs], res[num_inputs:], rinf[num_inputs:]
        pk["c_query_2"], pk["c_query_2_inf"] = rows(tg, [v * (2 * gamma * gamma % r) % r * zt % r for v in a])
    finally:
        tg.free()
    h_gamma = gl.proj_mul(g2c, g2_xyz, _canon_rows([gamma % r])[0])
    th = FB(g2c, h_gamma, 753, h_window)
    try:
        pk["b_query"], pk["b_query_inf"] = rows(th, a)
    finally:
        th.free()
    pk["g_gamma_z"] = point(g1c, g1_xyz, gz)
    pk["h_gamma_z"] = point(g2c, h_gamma, zt)
    pk["g_ab_gamma_z"] = point(g1c, g1_xyz, ab * gz)
    pk["g_gamma2_z2"] = point(g1c, g1_xyz, gz * gz)
    info = {"num_inputs": num_inputs, "log_n": log_n, "sap": (a, c, zt), "sap_num_variables": sap_nv, "g_window": g_window,
            "h_window": h_window, "non_zero_a": non_zero_a, "fixed_base": stats}
    return pk, info


def proof_bytes(pairing, proof):
    """A || B || C as GroupAffine::write records (GM17's own Proof::write is unimplemented upstream: gm17/mod.rs:59-69)"""
    (a, ai), (b, bi), (c, ci) = proof
    return affine_to_wire(pairing, "g1", a, ai) + affine_to_wire(pairing, "g2", b, bi) + affine_to_wire(pairing, "g1", c, ci)
