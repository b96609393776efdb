"""The MSM stage of the Groth16 prover over a device-resident proving key.

Mirror of proof-systems/src/groth16/prover.rs:273-345 (create_proof after the witness map): nine
multi_scalar_mul calls over the key's five query vectors (groth16/mod.rs:158-170, getters :318-370), a
few single scalar multiplications and additions, then into_affine() of A, B, C.  The Rust shim keeps
doing this composition in Rust; this module is the same sequence above the C ABI for
tests and for the replay of BASELINE config 5 from exported buffers:

  * the five large query tails (a_query[num_inputs..], b_g1_query[..], b_g2_query[..],
    h_query[num_inputs..], l_query) live on the device (ResidentBases, optional shift tables);
  * the four large G1 MSMs go to the library as ONE pipelined batch (gh_msm_resident_dev_batch);
  * the short "inputs" MSMs (num_inputs - 1 pairs) use gh_msm.
"""
import numpy as np


class ResidentProvingKey:
    """pk: dict of numpy arrays in the ABI formats (Montgomery x||y rows, all points finite unless an
    *_inf array says otherwise): alpha_g1, beta_g1, delta_g1 (24 u64), beta_g2, delta_g2 (24*deg u64),
    a_query, b_g1_query, h_query, l_query (n x 24), b_g2_query (n x 24*deg)."""

    def __init__(self, gl, pairing, pk, num_inputs, precompute=True):
        assert pairing in ("mnt4753", "mnt6753")
        self.gl, self.pk, self.num_inputs = gl, pk, int(num_inputs)
        self.g1, self.g2 = pairing + "_g1", pairing + "_g2"
        self.deg2 = gl.CURVE_DEG[self.g2]
        ni = self.num_inputs
        self.tails = {}
        for name, curve, lo in (("a_query", self.g1, ni), ("b_g1_query", self.g1, ni), ("b_g2_query", self.g2, ni),
                                ("h_query", self.g1, ni), ("l_query", self.g1, 0)):
            rows = np.ascontiguousarray(pk[name][lo:], dtype=np.uint64)
            inf = pk.get(name + "_inf")
            rb = gl.ResidentBases(curve, rows, None if inf is None else inf[lo:])
            if precompute and rb.n:
                try:
                    rb.precompute(0)
                except gl.GingerHipError:
                    pass                      # no memory for the table / a point of 2-power order: per-window path
            self.tails[name] = rb

    def free(self):
        for rb in self.tails.values():
            rb.free()

    def create_proof_msms(self, input_assignment, aux_assignment, h_input_assignment, h_aux_assignment, r, s, h_dev=None):
        """All arguments are canonical 12-u64 scalars (rows).  Returns (A, B, C) as (xy, is_infinity) pairs:
        exactly Proof { a: g_a.into_affine(), b: g2_b.into_affine(), c: g_c.into_affine() } (prover.rs:340-344).
        h_dev = (DeviceBuffer, rows): the coefficients of h already on the device as canonical scalars (the output
        of gh_witness_map_dev after into_repr); h_input_assignment / h_aux_assignment are then taken from it."""
        gl, pk, ni = self.gl, self.pk, self.num_inputs
        g1, g2 = self.g1, self.g2
        msm = gl.VariableBaseMSM.multi_scalar_mul
        add = gl.proj_add
        mul = gl.proj_mul
        inp = np.ascontiguousarray(input_assignment, dtype=np.uint64).reshape(-1, 12)
        aux = np.ascontiguousarray(aux_assignment, dtype=np.uint64).reshape(-1, 12)
        if h_dev is not None:
            import ctypes
            h_buf, h_rows = h_dev
            head = np.empty(ni * 12, dtype=np.uint64)
            gl._check(gl.load_library().gh_dev_download(gl._ptr(head), h_buf.ptr, head.nbytes))
            h_inp = head.reshape(ni, 12)
            n_haux = int(h_rows) - ni

            class _View:                      # the tail of h on the device: h[num_inputs..]  (prover.rs:262-267)
                ptr = ctypes.c_void_p(h_buf.ptr.value + ni * 96)
            d_haux, own_haux = _View, False
        else:
            h_inp = np.ascontiguousarray(h_input_assignment, dtype=np.uint64).reshape(-1, 12)
            h_aux = np.ascontiguousarray(h_aux_assignment, dtype=np.uint64).reshape(-1, 12)
            n_haux = len(h_aux)
            d_haux, own_haux = gl.DeviceBuffer(max(96, h_aux.nbytes)).upload(h_aux), True
        r = np.ascontiguousarray(r, dtype=np.uint64)
        s = np.ascontiguousarray(s, dtype=np.uint64)

        def proj(curve, xy):                 # From<GroupAffine> for GroupProjective (swp.rs:651-660), finite points
            deg = gl.CURVE_DEG[curve]
            out = np.zeros(36 * deg, dtype=np.uint64)
            out[:24 * deg] = np.asarray(xy, dtype=np.uint64).ravel()
            out[24 * deg:36 * deg] = gl.field_one(curve)
            return out

        # the four large G1 MSMs as one pipelined batch; the G2 one on its own (a batch stays on one curve)
        d_aux = gl.DeviceBuffer(max(96, aux.nbytes)).upload(aux)
        t = self.tails
        a_aux_acc, b1_aux_acc, h_aux_acc, l_aux_acc = gl.msm_batch_dev([
            (t["a_query"], d_aux, len(aux)), (t["b_g1_query"], d_aux, len(aux)),
            (t["h_query"], d_haux, n_haux), (t["l_query"], d_aux, len(aux))])
        b2_aux_acc = t["b_g2_query"].msm_dev(d_aux, len(aux))
        d_aux.free()
        if own_haux:
            d_haux.free()
        # Compute A  (prover.rs:273-284)
        a_inputs_acc = msm(g1, pk["a_query"][1:ni], inp)
        g_a = mul(g1, proj(g1, pk["delta_g1"]), r)
        g_a = add(g1, g_a, proj(g1, pk["a_query"][0]))
        g_a = add(g1, g_a, a_inputs_acc)
        g_a = add(g1, g_a, a_aux_acc)
        g_a = add(g1, g_a, proj(g1, pk["alpha_g1"]))
        # Compute B in G1  (:287-300)
        b_inputs_acc = msm(g1, pk["b_g1_query"][1:ni], inp)
        g1_b = mul(g1, proj(g1, pk["delta_g1"]), s)
        g1_b = add(g1, g1_b, proj(g1, pk["b_g1_query"][0]))
        g1_b = add(g1, g1_b, b_inputs_acc)
        g1_b = add(g1, g1_b, b1_aux_acc)
        g1_b = add(g1, g1_b, proj(g1, pk["beta_g1"]))
        # Compute B in G2  (:303-316)
        b2_inputs_acc = msm(g2, pk["b_g2_query"][1:ni], inp)
        g2_b = mul(g2, proj(g2, pk["delta_g2"]), s)
        g2_b = add(g2, g2_b, proj(g2, pk["b_g2_query"][0]))
        g2_b = add(g2, g2_b, b2_inputs_acc)
        g2_b = add(g2, g2_b, b2_aux_acc)
        g2_b = add(g2, g2_b, proj(g2, pk["beta_g2"]))
        # Compute C  (:319-337)
        h_inputs_acc = msm(g1, pk["h_query"][0:ni], h_inp)
        s_g_a = mul(g1, g_a, s)
        r_g1_b = mul(g1, g1_b, r)
        r_s_delta = mul(g1, mul(g1, proj(g1, pk["delta_g1"]), r), s)
        g_c = add(g1, s_g_a, r_g1_b)
        g_c = add(g1, g_c, gl.proj_neg(g1, r_s_delta))
        g_c = add(g1, g_c, l_aux_acc)
        g_c = add(g1, g_c, h_inputs_acc)
        g_c = add(g1, g_c, h_aux_acc)
        return gl.proj_to_affine(g1, g_a), gl.proj_to_affine(g2, g2_b), gl.proj_to_affine(g1, g_c)
