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
import numpy as np


class ResidentProvingKey:
    """pk: dict of numpy arrays in the ABI formats (Montgomery x||y rows, all points finite):
    alpha_g1, beta_g1, delta_g1 (24 u64), beta_g2, delta_g2 (24*deg u64), a_query, b_g1_query, h_query,
    l_query (n x 24), b_g2_query (n x 24*deg).

    The reference issues nine MSMs because the input and aux assignments are separate vectors and adds
    r * delta, the [0] entries and alpha / beta by hand (prover.rs:273-316).  The sums are the same group
    elements when each query is ONE resident vector
        a_query[1..] || a_query[0] || alpha_g1 || delta_g1        with scalars  input || aux || 1 || 1 || r
    (likewise b_g1 / b_g2 with beta and s, and h_query as a whole), so a proof needs five MSMs -- four on G1,
    issued as one pipelined batch, one on G2 -- and no 2-pair MSM pays the latency of a full bucket reduction."""

    def __init__(self, gl, pairing, pk, num_inputs, precompute=True):
        assert pairing in ("mnt4753", "mnt6753")
        self.gl, self.pk, self.num_inputs = gl, pk, int(num_inputs)
        self.g1, self.g2 = pairing + "_g1", pairing + "_g2"
        row = lambda v: np.asarray(v, dtype=np.uint64).reshape(1, -1)
        ext = lambda q, c, d: np.ascontiguousarray(np.concatenate([pk[q][1:], row(pk[q][0]), row(pk[c]), row(pk[d])]), dtype=np.uint64)
        vectors = {"a": (self.g1, ext("a_query", "alpha_g1", "delta_g1")),
                   "b1": (self.g1, ext("b_g1_query", "beta_g1", "delta_g1")),
                   "b2": (self.g2, ext("b_g2_query", "beta_g2", "delta_g2")),
                   "h": (self.g1, np.ascontiguousarray(pk["h_query"], dtype=np.uint64)),
                   "l": (self.g1, np.ascontiguousarray(pk["l_query"], dtype=np.uint64))}
        self.keys = {}
        for name, (curve, rows) in vectors.items():
            rb = gl.ResidentBases(curve, rows)
            if precompute and rb.n:
                try:
                    rb.precompute(0)
                except gl.GingerHipError:
                    pass                      # no memory for the table / a point of 2-power order: per-window path
            self.keys[name] = rb

    def free(self):
        for rb in self.keys.values():
            rb.free()

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
        n_var = self.keys["a"].n - 3
        assert self.keys["b1"].n - 3 == n_var and self.keys["b2"].n - 3 == n_var and n_var >= len(inp)
        aux_used = aux[:n_var - len(inp)]
        pad = np.zeros((n_var - len(inp) - len(aux_used), 12), dtype=np.uint64)
        # scalars  input || aux || 1 || 1 || r   and the same with s; the aux part alone is the l_query's vector
        sc_r = np.concatenate([inp, aux_used, pad, one, one, r])
        d_r = gl.DeviceBuffer(sc_r.nbytes).upload(sc_r)
        d_s = gl.DeviceBuffer(sc_r.nbytes)
        lib = gl.load_library()
        gl._check(lib.gh_dev_upload(d_s.ptr, gl._ptr(np.concatenate([inp, aux_used, pad, one, one, s])), sc_r.nbytes))

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
            (k["a"], d_r, n_var + 3), (k["b1"], d_s, n_var + 3), (k["h"], d_h, n_h), (k["l"], d_l, len(aux))])
        g2_b = k["b2"].msm_dev(d_s, n_var + 3)
        d_r.free()
        d_s.free()
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
