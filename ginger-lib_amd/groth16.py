"""The MSM stage of the Groth16 prover over a device-resident proving key.

Mirror of proof-systems/src/groth16/prover.rs:273-345 (create_proof after the witness map): nine
multi_scalar_mul calls over the key's five query vectors (groth16/mod.rs:158-170, getters :318-370), a
few single scalar multiplications and additions, then into_affine() of A, B, C.  The Rust shim keeps
doing this composition in Rust; this module is the same sequence above the C ABI for
tests and for the replay of BASELINE config 5 from exported buffers:

  * the five query vectors live on the device (ResidentBases, optional shift tables), each extended by
    the few single points the reference adds by hand (ResidentProvingKey below);
  * five MSMs per proof instead of nine: four on G1 as ONE pipelined batch (gh_msm_resident_dev_batch),
    one on G2.
"""
import struct

import numpy as np

# base-field bytes of one coordinate coefficient / degree of the G2 coordinate field / bytes of the target-field element
# vk.alpha_g1_beta_g2 (Fq4 resp. Fq6: carried as opaque bytes -- pairings are out of scope, the prover never reads it)
_FQ_BYTES = 96
_G2_DEG = {"mnt4753": 2, "mnt6753": 3}
_FQK_BYTES = {"mnt4753": 4 * 96, "mnt6753": 6 * 96}
_MODULUS = {   # r = scalar field of the pairing = base field of the other curve
    "mnt4753": 0x1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db26c5c28c859a99b3eebca9429212636b9dff97634993aa4d6c381bc3f0057974ea099170fa13a4fd90776e240000001,
    "mnt6753": 0x1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db117e776f218059db80f0da5cb537e38685acce9767254a4638810719ac425f0e39d54522cdd119f5e9063de245e8001,
}


def parse_parameters(pairing, blob):
    """Parameters::write stream (proof-systems/src/groth16/mod.rs:188-208; VerifyingKey::write :104-114) -> dict of byte
    strings: single points as GroupAffine::write records (x || y || infinity byte, canonical little-endian coefficients:
    short_weierstrass_projective.rs:185-192), queries as their records back to back.  Lengths are big-endian u32."""
    g1 = 2 * _FQ_BYTES + 1
    g2 = 2 * _FQ_BYTES * _G2_DEG[pairing] + 1
    pos = 0

    def take(nbytes):
        nonlocal pos
        if pos + nbytes > len(blob):
            raise ValueError("Parameters stream is truncated")
        out = bytes(blob[pos:pos + nbytes])
        pos += nbytes
        return out

    def vec(rec):
        (count,) = struct.unpack(">I", take(4))
        return take(count * rec)

    pk = {"vk_alpha_g1_beta_g2": take(_FQK_BYTES[pairing]), "vk_gamma_g2": take(g2), "vk_delta_g2": take(g2)}
    pk["vk_gamma_abc_g1"] = vec(g1)
    for name, rec in (("alpha_g1", g1), ("beta_g1", g1), ("beta_g2", g2), ("delta_g1", g1), ("delta_g2", g2)):
        pk[name] = take(rec)
    for name, rec in (("a_query", g1), ("b_g1_query", g1), ("b_g2_query", g2), ("h_query", g1), ("l_query", g1)):
        pk[name] = vec(rec)
    if pos != len(blob):
        raise ValueError("trailing bytes after the Parameters stream")
    return pk


def benchmark_circuit_rows(pairing, num_constraints):
    """The `Benchmark` circuit of proof-systems/src/groth16/examples/snark-scalability/constraints.rs:20-92 evaluated at its
    own witness: returns (num_inputs, full assignment, A, B, C) as Python integer lists -- the vectors `a`, `b`, `c`
    R1CStoQAP::witness_map forms from the constraint rows (r1cs_to_qap.rs:105-119, :141-151; before padding to the domain).
    Host scalar code, as in the reference (synthesis is not on the device path).  The circuit has two public inputs
    a = b = 1, alternating a + b = c / a * b = c steps and one closing constraint (sum of the recorded assignments)^2;
    the recording pushes (a, a) twice and never b -- reproduced as written (:31-36)."""
    r = _MODULUS[pairing]
    num_inputs = 3                                           # one, a, b
    inputs = [1, 1, 1]
    aux = []
    A, B, C = [], [], []
    recorded = [(1, ("in", 1)), (1, ("in", 1))]              # (a_val, a_var) twice
    a_val, a_var, b_val, b_var = 1, ("in", 1), 1, ("in", 2)
    for i in range(num_constraints - 1):
        if i % 2 != 0:
            c_val = a_val * b_val % r
            A.append(a_val); B.append(b_val); C.append(c_val)
        else:
            c_val = (a_val + b_val) % r
            A.append(c_val); B.append(1); C.append(c_val)    # (a + b) * one = c
        c_var = ("aux", len(aux))
        aux.append(c_val)
        recorded.append((c_val, c_var))
        a_val, a_var, b_val, b_var = b_val, b_var, c_val, c_var
    total = sum(v for v, _ in recorded) % r
    c_val = total * total % r
    aux.append(c_val)
    A.append(total); B.append(total); C.append(c_val)
    return num_inputs, inputs + aux, A, B, C


def benchmark_circuit_lcs(num_constraints):
    """The same circuit as linear combinations, the form the key generator consumes (KeypairAssembly: at / bt / ct rows of
    (coefficient, index) with inputs first, generator.rs:28-110): constraints.rs:20-92 statement by statement.
    -> (num_inputs, num_aux, at, bt, ct); index 0 is the constant one."""
    num_inputs = 3
    var = [1, 2]                                             # v_0 = a (input 1), v_1 = b (input 2); v_(k+2) = aux k
    at, bt, ct = [], [], []
    n_aux = 0
    recorded = [1, 1]                                        # (a, a): sic, :31-36
    for i in range(num_constraints - 1):
        c_var = num_inputs + n_aux
        n_aux += 1
        a_var, b_var = var[-2], var[-1]
        if i % 2 != 0:
            at.append([(1, a_var)]); bt.append([(1, b_var)]); ct.append([(1, c_var)])
        else:
            at.append([(1, a_var), (1, b_var)]); bt.append([(1, 0)]); ct.append([(1, c_var)])
        var = [b_var, c_var]
        recorded.append(c_var)
    c_var = num_inputs + n_aux
    n_aux += 1
    lc = [(1, v) for v in recorded]
    at.append(lc); bt.append(list(lc)); ct.append([(1, c_var)])
    return num_inputs, n_aux, at, bt, ct


def _ints_from_mont_rows(rows, modulus):
    rinv = pow(1 << 768, -1, modulus)
    raw = np.ascontiguousarray(rows, dtype=np.uint64).tobytes()
    return [int.from_bytes(raw[96 * i:96 * i + 96], "little") * rinv % modulus for i in range(len(raw) // 96)]


def _canon_rows_fast(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(96, "little") for v in vals), dtype=np.uint64).reshape(-1, 12) if len(vals) else np.zeros((0, 12), np.uint64)


def generate_parameters(gl, pairing, lcs, alpha, beta, gamma, delta, t, g1_xyz, g2_xyz, vk_pairing_bytes=None):
    """generate_parameters (proof-systems/src/groth16/generator.rs:146-335) above the C ABI, for a constraint system given as
    linear combinations; the toxic waste, the evaluation point t (sample_element_outside_domain, :181) and the two
    generators (:225-226, `rand`) are arguments instead of RNG draws.  Returns (Parameters::write bytes, info).
      * Lagrange coefficients at t on the device (gh_lagrange_coefficients = evaluate_all_lagrange_coefficients, domain.rs:183-219);
      * instance_map_with_evaluation (r1cs_to_qap.rs:14-69): host integers, as in the reference;
      * the five queries and gamma_abc_g1: FixedBaseMSM on the device with the reference's window rule (:233-311), followed
        by batch_normalization + into_affine (:318-335) -- gh_fixed_base_msm_affine, straight into the serialised form;
      * vk.alpha_g1_beta_g2 is a pairing value (:313): pairings are out of scope, the prover never reads it -- filler bytes
        unless the caller supplies them."""
    r = _MODULUS[pairing]
    field = "mnt4753_fr" if pairing == "mnt4753" else "mnt6753_fr"
    g1c, g2c = pairing + "_g1", pairing + "_g2"
    num_inputs, num_aux, at, bt, ct = lcs
    n_con = len(at)
    size = 1
    while size < n_con + (num_inputs - 1) + 1:
        size <<= 1
    log_n = size.bit_length() - 1
    zt = (pow(t, size, r) - 1) % r                                               # evaluate_vanishing_polynomial
    tau = _mont_rows([t], r)[0]
    u_rows = np.zeros((size, 12), dtype=np.uint64)
    gl._check(gl.load_library().gh_lagrange_coefficients(gl.FIELDS[field], log_n, gl._ptr(tau), gl._ptr(u_rows)))
    u = _ints_from_mont_rows(u_rows, r)
    nv = (num_inputs - 1) + num_aux
    a, b, c = [0] * (nv + 1), [0] * (nv + 1), [0] * (nv + 1)
    for i in range(num_inputs):
        a[i] = u[n_con + i]
    for rows, acc in ((at, a), (bt, b), (ct, c)):
        for i, row in enumerate(rows):
            ui = u[i]
            for cf, ix in row:
                acc[ix] = (acc[ix] + (ui if cf == 1 else ui * cf)) % r
    non_zero_a = sum(1 for v in a[:nv] if v)
    non_zero_b = sum(1 for v in b[:nv] if v)
    gi, di = pow(gamma, -1, r), pow(delta, -1, r)
    comb = [(beta * x + alpha * y + z) % r for x, y, z in zip(a, b, c)]
    gamma_abc = [v * gi % r for v in comb[:num_inputs]]
    l = [v * di % r for v in comb]
    hs, cur = [], zt * di % r
    for _ in range(size - 1):
        hs.append(cur)
        cur = cur * t % r
    FB = gl.FixedBaseMSM
    g1_window = FB.get_mul_window_size(non_zero_a + non_zero_b + nv + size + 1)
    g2_window = FB.get_mul_window_size(non_zero_b)
    stats = {"fixed_base_scalars": 0, "fixed_base_s": 0.0}

    def wire_rows(table, deg, vals):
        import time
        rows = _canon_rows_fast(vals)                                          # host integers -> 96-byte rows: not part of the call
        t0 = time.perf_counter()
        xy, inf = table.multi_scalar_mul_affine(rows, canonical=True)
        stats["fixed_base_s"] += time.perf_counter() - t0
        stats["fixed_base_scalars"] += len(vals)
        rec = np.empty((len(vals), 192 * deg + 1), dtype=np.uint8)
        rec[:, :192 * deg] = xy.view(np.uint8).reshape(len(vals), 192 * deg)
        rec[:, 192 * deg] = inf
        return rec.tobytes()

    def wire_point(curve, deg, g_xyz, k):
        xy, inf = gl.proj_to_affine(curve, gl.proj_mul(curve, g_xyz, _canon_rows([k])[0]))
        return affine_to_wire(pairing, "g1" if deg == 1 else "g2", xy, inf)

    deg2 = _G2_DEG[pairing]
    t1 = FB(g1c, g1_xyz, 753, g1_window)
    try:
        a_q = wire_rows(t1, 1, a)
        b1_q = wire_rows(t1, 1, b)
        h_q = wire_rows(t1, 1, hs)
        l_q = wire_rows(t1, 1, l)[193 * num_inputs:]                             # l_query[num_inputs..]  (:298)
        abc_q = wire_rows(t1, 1, gamma_abc)
    finally:
        t1.free()
    t2 = FB(g2c, g2_xyz, 753, g2_window)
    try:
        b2_q = wire_rows(t2, deg2, b)
    finally:
        t2.free()
    blob = bytearray()
    blob += vk_pairing_bytes if vk_pairing_bytes is not None else bytes((i * 37 + 5) & 0xFF for i in range(_FQK_BYTES[pairing]))
    blob += wire_point(g2c, deg2, g2_xyz, gamma) + wire_point(g2c, deg2, g2_xyz, delta)
    blob += struct.pack(">I", num_inputs) + abc_q
    blob += wire_point(g1c, 1, g1_xyz, alpha) + wire_point(g1c, 1, g1_xyz, beta) + wire_point(g2c, deg2, g2_xyz, beta)
    blob += wire_point(g1c, 1, g1_xyz, delta) + wire_point(g2c, deg2, g2_xyz, delta)
    for q, rec in ((a_q, 193), (b1_q, 193), (b2_q, 192 * deg2 + 1), (h_q, 193), (l_q, 193)):
        blob += struct.pack(">I", len(q) // rec) + q
    info = {"num_inputs": num_inputs, "log_n": log_n, "qap": (a, b, c, l, zt), "g1_window": g1_window, "g2_window": g2_window,
            "non_zero": (non_zero_a, non_zero_b), "fixed_base": stats}
    return bytes(blob), info


def _mont_rows(vals, modulus):
    """Python integers -> n x 12 u64 Montgomery rows (x * 2^768 mod p), the in-memory form of Fp768 (fp_768.rs:24-30)"""
    R = (1 << 768) % modulus
    out = np.zeros((len(vals), 12), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        x = v * R % modulus
        for k in range(12):
            out[i, k] = (x >> (64 * k)) & mask
    return out


def _canon_rows(vals):
    out = np.zeros((len(vals), 12), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        for k in range(12):
            out[i, k] = (v >> (64 * k)) & mask
    return out


def affine_to_wire(pairing, group, xy, is_infinity):
    """GroupAffine::write of an affine result (Montgomery x || y limbs from gh_proj_to_affine): canonical little-endian
    coefficients + infinity byte; zero() is written as (0, 1, true) (swp.rs:130-132)."""
    p = _MODULUS["mnt6753" if pairing == "mnt4753" else "mnt4753"]      # base field of the pairing's curves
    deg = 1 if group == "g1" else _G2_DEG[pairing]
    rinv = pow(1 << 768, -1, p)
    v = [int(x) for x in np.asarray(xy, dtype=np.uint64).ravel()]
    out = bytearray()
    for c in range(2 * deg):
        m = sum(v[12 * c + k] << (64 * k) for k in range(12))
        out += (m * rinv % p).to_bytes(_FQ_BYTES, "little")
    out.append(1 if is_infinity else 0)
    return bytes(out)


class ResidentProvingKey:
    """pk: dict of numpy arrays in the ABI formats (Montgomery x||y rows, all points finite):
    alpha_g1, beta_g1, delta_g1 (24 u64), beta_g2, delta_g2 (24*deg u64), a_query, b_g1_query, h_query,
    l_query (n x 24), b_g2_query (n x 24*deg).

    The reference issues nine MSMs because the input and aux assignments are separate vectors and adds
    r * delta, the [0] entries and alpha / beta by hand (prover.rs:273-316).  The sums are the same group
    elements when each query is ONE resident vector
        a_query[1..] || a_query[0] || alpha_g1 || delta_g1 || infinity     with scalars  input || aux || 1 || 1 || r || s
        b_query[1..] || b_query[0] || beta     || infinity || delta         with the SAME scalar vector
    (the infinity base swallows the other query's blinding scalar; h_query as a whole), so a proof needs five MSMs
    -- four on G1, issued as one pipelined batch, one on G2 --, ONE upload of the assignment, and no 2-pair MSM pays
    the latency of a full bucket reduction."""

    def __init__(self, gl, pairing, pk, num_inputs, precompute=True):
        assert pairing in ("mnt4753", "mnt6753")
        self.gl, self.pk, self.num_inputs = gl, pk, int(num_inputs)
        self.g1, self.g2 = pairing + "_g1", pairing + "_g2"
        row = lambda v: np.asarray(v, dtype=np.uint64).reshape(1, -1)
        def ext(q, c, d, delta_last):       # q[1..] || q[0] || c || (delta, infinity) or (infinity, delta)
            dl = row(pk[d])
            tail = [np.zeros_like(dl), dl] if delta_last else [dl, np.zeros_like(dl)]
            rows = np.ascontiguousarray(np.concatenate([pk[q][1:], row(pk[q][0]), row(pk[c])] + tail), dtype=np.uint64)
            inf = np.zeros(len(rows), dtype=np.uint8)
            inf[-2 if delta_last else -1] = 1
            return rows, inf
        vectors = {"a": (self.g1,) + ext("a_query", "alpha_g1", "delta_g1", False),
                   "b1": (self.g1,) + ext("b_g1_query", "beta_g1", "delta_g1", True),
                   "b2": (self.g2,) + ext("b_g2_query", "beta_g2", "delta_g2", True),
                   "h": (self.g1, np.ascontiguousarray(pk["h_query"], dtype=np.uint64), None),
                   "l": (self.g1, np.ascontiguousarray(pk["l_query"], dtype=np.uint64), None)}
        self.keys = {}
        for name, (curve, rows, inf) in vectors.items():
            rb = gl.ResidentBases(curve, rows, infinity=inf)
            if precompute and rb.n:
                try:
                    rb.precompute(0)
                except gl.GingerHipError:
                    pass                      # no memory for the table / a point of 2-power order: per-window path
            self.keys[name] = rb

    @classmethod
    def from_parameters(cls, gl, pairing, blob, num_inputs, precompute=True):
        """The same resident key straight from a Parameters::write stream (a proving-key file): every query goes to the
        device in its serialised form (gh_bases_upload_wire: canonical coefficients, infinity flags honoured -- keys made by
        the reference generator contain GroupAffine::zero() entries wherever a variable does not occur in A or B)."""
        self = cls.__new__(cls)
        self.gl, self.num_inputs = gl, int(num_inputs)
        self.g1, self.g2 = pairing + "_g1", pairing + "_g2"
        pk = parse_parameters(pairing, blob)
        self.pk = pk
        self.pairing = pairing
        r1, r2 = 2 * _FQ_BYTES + 1, 2 * _FQ_BYTES * _G2_DEG[pairing] + 1
        def ext(q, c, d, rec, delta_last):  # q[1..] || q[0] || c || (delta, infinity) or (infinity, delta), GroupAffine::write records
            zero = bytes(rec - 1) + b"\x01"                                             # an infinity record: flag byte set
            tail = zero + pk[d] if delta_last else pk[d] + zero
            return pk[q][rec:] + pk[q][:rec] + pk[c] + tail
        vectors = {"a": (self.g1, ext("a_query", "alpha_g1", "delta_g1", r1, False)),
                   "b1": (self.g1, ext("b_g1_query", "beta_g1", "delta_g1", r1, True)),
                   "b2": (self.g2, ext("b_g2_query", "beta_g2", "delta_g2", r2, True)),
                   "h": (self.g1, pk["h_query"]), "l": (self.g1, pk["l_query"])}
        self.keys = {}
        for name, (curve, data) in vectors.items():
            rb = gl.ResidentBases.from_wire(curve, data)
            if precompute and rb.n:
                try:
                    rb.precompute(0)
                except gl.GingerHipError:
                    pass
            self.keys[name] = rb
        # delta_g1 as Montgomery limbs for the host-side r * s * delta term: through the device converter (one point)
        one = gl.ResidentBases.from_wire(self.g1, pk["delta_g1"])
        self.pk_delta_g1 = one.download(0, 1)[0]
        one.free()
        return self

    def free(self):
        for rb in self.keys.values():
            rb.free()

    def prepare_rows(self, circuit_rows, d1, d2, d3):
        """The host side of create_proof up to the device boundary: the evaluated constraint rows (Python integers) as
        Montgomery / canonical limb arrays, the way the reference holds them before the witness map (r1cs_to_qap.rs:100-120).
        Separate from prove_prepared so that a caller (bench.py's `prover` object) can time the device stage alone."""
        pairing = self.pairing
        modulus = _MODULUS[pairing]
        num_inputs, assignment, A, B, C = circuit_rows
        assert num_inputs == self.num_inputs
        n_con = len(A)
        size = 1
        while size < n_con + (num_inputs - 1) + 1:               # EvaluationDomain::new(num_constraints + num_inputs) (r1cs_to_qap.rs:100)
            size <<= 1
        a = np.zeros((size, 12), dtype=np.uint64)
        b = np.zeros((size, 12), dtype=np.uint64)
        c = np.zeros((size, 12), dtype=np.uint64)
        a[:n_con] = _mont_rows(A, modulus)
        b[:n_con] = _mont_rows(B, modulus)
        c[:n_con] = _mont_rows(C, modulus)
        a[n_con:n_con + num_inputs] = _mont_rows([1] + list(assignment[1:num_inputs]), modulus)     # :116-118
        return {"size": size, "a": a, "b": b, "c": c, "dd": _mont_rows([d1, d2, d3], modulus), "scal": _canon_rows(assignment)}

    def prove_prepared(self, prep, r, s, timing=None):
        """create_proof (prover.rs:201-345) from prepare_rows' arrays: rows to the device, gh_witness_map_dev, into_repr on the
        device, the five MSMs over the resident key, Proof::write bytes.  timing (a dict) receives wall-clock ms per stage."""
        import time
        gl, pairing = self.gl, self.pairing
        size, num_inputs = prep["size"], self.num_inputs
        log_n = size.bit_length() - 1
        field = "mnt4753_fr" if pairing == "mnt4753" else "mnt6753_fr"
        dd, scal = prep["dd"], prep["scal"]
        lib = gl.load_library()
        t0 = time.perf_counter()
        bufs = [gl.DeviceBuffer(size * 96 + 96) for _ in range(4)]
        try:
            for buf, arr in zip(bufs, (prep["a"], prep["b"], prep["c"])):
                buf.upload(arr)
            lib.gh_dev_sync()
            t1 = time.perf_counter()
            gl._check(lib.gh_witness_map_dev(gl.FIELDS[field], bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, log_n, gl._ptr(dd[0]), gl._ptr(dd[1]),
                                             gl._ptr(dd[2]), bufs[3].ptr))
            one_plain = np.zeros(12, dtype=np.uint64)
            one_plain[0] = 1
            gl._check(lib.gh_vec_scale_dev(gl.FIELDS[field], bufs[3].ptr, gl._ptr(one_plain), size + 1))      # into_repr of h (:256-267)
            lib.gh_dev_sync()
            t2 = time.perf_counter()
            self.pk = dict(self.pk, delta_g1=self.pk_delta_g1)
            A_, B_, C_ = self.create_proof_msms(scal[1:num_inputs], scal[num_inputs:], None, None, _canon_rows([r])[0], _canon_rows([s])[0],
                                                h_dev=(bufs[3], self.keys["h"].n))
            t3 = time.perf_counter()
        finally:
            for buf in bufs:
                buf.free()
        if timing is not None:
            timing.update(rows_upload_ms=(t1 - t0) * 1e3, witness_map_ms=(t2 - t1) * 1e3, msm_stage_ms=(t3 - t2) * 1e3)
        return affine_to_wire(pairing, "g1", *A_) + affine_to_wire(pairing, "g2", *B_) + affine_to_wire(pairing, "g1", *C_)

    def create_proof(self, circuit_rows, d1, d2, d3, r, s):
        """create_proof (prover.rs:201-345) for evaluated constraint rows: circuit_rows = (num_inputs, assignment, A, B, C) as
        Python integers (benchmark_circuit_rows); d1, d2, d3, r, s integers.  Returns Proof::write bytes (mod.rs:35-42)."""
        return self.prove_prepared(self.prepare_rows(circuit_rows, d1, d2, d3), r, s)

    def create_proof_msms(self, input_assignment, aux_assignment, h_input_assignment, h_aux_assignment, r, s, h_dev=None):
        """All arguments are canonical 12-u64 scalars (rows).  Returns (A, B, C) as (xy, is_infinity) pairs:
        exactly Proof { a: g_a.into_affine(), b: g2_b.into_affine(), c: g_c.into_affine() } (prover.rs:340-344).
        h_dev = (DeviceBuffer, rows): the coefficients of h already on the device as canonical scalars (the output
        of gh_witness_map_dev after into_repr); h_input_assignment / h_aux_assignment are then ignored."""
        gl, pk, ni = self.gl, self.pk, self.num_inputs
        g1, g2 = self.g1, self.g2
        add, mul = gl.proj_add, gl.proj_mul
        inp = np.ascontiguousarray(input_assignment, dtype=np.uint64).reshape(-1, 12)
        aux = np.ascontiguousarray(aux_assignment, dtype=np.uint64).reshape(-1, 12)
        r = np.ascontiguousarray(r, dtype=np.uint64).reshape(1, 12)
        s = np.ascontiguousarray(s, dtype=np.uint64).reshape(1, 12)
        assert len(inp) == ni - 1
        one = np.zeros((1, 12), dtype=np.uint64)
        one[0, 0] = 1
        # the variable part must line up with the query: len(a_query) - 1 scalars (a real key has exactly
        # len(input) + len(aux) of them; a longer aux vector is cut where the reference's zip cuts it, :36)
        n_var = self.keys["a"].n - 4
        assert self.keys["b1"].n - 4 == n_var and self.keys["b2"].n - 4 == n_var and n_var >= len(inp)
        aux_used = aux[:n_var - len(inp)]
        pad = np.zeros((n_var - len(inp) - len(aux_used), 12), dtype=np.uint64)
        # ONE scalar vector  input || aux || 1 || 1 || r || s  for the a, b_g1 and b_g2 queries (each key pairs the blinding scalar
        # it does not use with an infinity base); the aux part alone is the l_query's vector.
        # (uploaded piecewise: concatenating 100 MB host vectors first costs more than the MSM stage's sort)
        import ctypes
        lib = gl.load_library()
        rows_total = n_var + 4
        d_r = gl.DeviceBuffer(rows_total * 96)
        row = 0
        for part in (inp, aux_used, pad, np.concatenate([one, one, r, s])):
            if len(part):
                part = np.ascontiguousarray(part, dtype=np.uint64)
                gl._check(lib.gh_dev_upload(ctypes.c_void_p(d_r.ptr.value + row * 96), gl._ptr(part), part.nbytes))
                row += len(part)
        assert row == rows_total
        d_s = d_r

        class _View:                          # aux_assignment inside d_r
            def __init__(self, buf, row0):
                import ctypes
                self.ptr = ctypes.c_void_p(buf.ptr.value + row0 * 96)
        if h_dev is not None:
            d_h, n_h, own_h = h_dev[0], int(h_dev[1]), False
        else:
            h_all = np.concatenate([np.ascontiguousarray(h_input_assignment, dtype=np.uint64).reshape(-1, 12),
                                    np.ascontiguousarray(h_aux_assignment, dtype=np.uint64).reshape(-1, 12)])
            d_h, n_h, own_h = gl.DeviceBuffer(max(96, h_all.nbytes)).upload(h_all), len(h_all), True
        # l_query pairs with the whole aux vector: normally the aux part of d_r, else (aux longer than the a / b
        # queries use) its own upload
        own_l = len(aux_used) != len(aux)
        d_l = gl.DeviceBuffer(max(96, aux.nbytes)).upload(aux) if own_l else _View(d_r, ni - 1)
        k = self.keys
        g_a, g1_b, h_acc, l_acc = gl.msm_batch_dev([
            (k["a"], d_r, n_var + 4), (k["b1"], d_s, n_var + 4), (k["h"], d_h, n_h), (k["l"], d_l, len(aux))])
        g2_b = k["b2"].msm_dev(d_s, n_var + 4)
        d_r.free()
        if own_l:
            d_l.free()
        if own_h:
            d_h.free()
        # Compute C  (prover.rs:319-337):  s * g_a + r * g1_b - (r s) * delta_g1 + l + h
        delta = np.zeros(36, dtype=np.uint64)
        delta[:24] = np.asarray(pk["delta_g1"], dtype=np.uint64).ravel()
        delta[24:] = gl.field_one(g1)
        g_c = add(g1, mul(g1, g_a, s), mul(g1, g1_b, r))
        g_c = add(g1, g_c, gl.proj_neg(g1, mul(g1, mul(g1, delta, r), s)))
        g_c = add(g1, g_c, l_acc)
        g_c = add(g1, g_c, h_acc)
        return gl.proj_to_affine(g1, g_a), gl.proj_to_affine(g2, g2_b), gl.proj_to_affine(g1, g_c)
