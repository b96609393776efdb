"""GPU parity at every BASELINE.json config size (the sizes test_gpu_parity.py does not reach):
  config 3  MNT4-753 G1 at 2^24 pairs (shift table c = 23 and per-window path),
  config 4  MNT6-753 G1 and G2 at the 8-GPU shard size 2^19 and at 2^22 on one GPU; MNT4-753 G2 at 2^20,
  config 5  the device halves of create_proof at 2^20 constraints (witness map + MSM stage),
plus the G2 lane-pair / lane-triple kernels on inputs with heavy buckets, L1 = 16 segments, several
pseudo-windows and pipelined batches, and the C ABI called from two threads.
Full-size results are checked through size-independent properties (table path == per-window path == other
window sizes in affine, linearity on a slice) and an oracle spot check on a slice of the same inputs
(reference semantics: algebra/src/msm/variable_base.rs:10-90, proof-systems/src/groth16/prover.rs:273-345)."""
import importlib
import threading

import numpy as np
import pytest

import pyref
import support as S

pytestmark = pytest.mark.gpu


def affine(gl, curve, xyz):
    xy, inf = gl.proj_to_affine(curve, xyz)
    return inf, xy.tobytes()


def oracle_affine(curve, xyz):
    xy, inf = S.oracle_affine(curve, xyz)
    return inf, xy.tobytes()


def tiled_bases(curve, n, pool_n, seed):
    C = pyref.CURVES[curve]
    pool = S.chain_points(C, pool_n, pyref.Rng(seed))
    pb, _ = S.bases_array(C, pool)
    return np.tile(pb, (n // pool_n, 1)), pool


def add_mod(a, b, r):
    out = np.empty_like(a)
    for i in range(len(a)):
        out[i] = pyref.int_to_limbs((pyref.limbs_to_int([int(v) for v in a[i]]) + pyref.limbs_to_int([int(v) for v in b[i]])) % r)
    return out


def skew(s, r):
    """witness-like scalars: a third 1, some 0 / 2 / r - 1 / one repeated value: long buckets, the scalar == 1 shortcut"""
    s = s.copy()
    s[0::3, :] = 0
    s[0::3, 0] = 1
    s[1::15, :] = 0
    s[4::15, :] = 0
    s[4::15, 0] = 2
    s[7::15] = np.array(pyref.int_to_limbs(r - 1), dtype=np.uint64)
    s[10::15] = s[10]
    return s


# ------------------------------------------------------------------------------ config 3 at 2^24
def test_cfg3_mnt4_g1_2p24(gpu, no_dedup):
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    r = C.order
    n = 1 << 24
    bases, _ = tiled_bases(curve, n, 4096, 2024)
    s = S.random_scalars_np(n, seed=31, below=r)
    rb = gpu.ResidentBases(curve, bases)
    ds = gpu.DeviceBuffer(s.nbytes).upload(s)
    try:
        plain = affine(gpu, curve, rb.msm_dev(ds, n))                   # per-window path, c from n
        tm = gpu.msm_last_timing()
        assert tm["num_windows"] == 752 // tm["window_bits"] + 1 and tm["accumulate_madds"] > 0.85 * n * tm["num_windows"]
        # linearity on a 2^17 slice and the oracle on a 2^14 slice of the same inputs
        m, k = 1 << 17, 1 << 14
        t = S.random_scalars_np(m, seed=32, below=r)
        st = add_mod(s[:m], t, r)
        lhs = gpu.proj_add(curve, rb.msm(s[:m]), rb.msm(t))
        assert affine(gpu, curve, lhs) == affine(gpu, curve, rb.msm(st))
        exp = S.oracle_msm(curve, bases[:k], None, s[:k], 16)
        assert affine(gpu, curve, rb.msm(s[:k])) == oracle_affine(curve, exp)
        # shift table: c = 23, 33 rows, 115 GB
        assert rb.precompute(0) == 23
        table = affine(gpu, curve, rb.msm_dev(ds, n))
        assert gpu.msm_last_timing()["num_windows"] == 33
        assert table == plain
        assert affine(gpu, curve, rb.msm(s[:k])) == oracle_affine(curve, exp)
        # projective bucket sums give the same point as the affine rounds (whichever the automatic mode picked)
        gpu.msm_set_affine(0)
        assert affine(gpu, curve, rb.msm_dev(ds, n)) == plain
    finally:
        gpu.msm_set_affine(2)
        ds.free()
        rb.free()
        gpu.dev_trim()


# ------------------------------------------------------------------------------ full sizes against a closed form
def chain_key(gl, curve, n, seed):
    """n distinct resident bases P_0 + i H (gh_bases_generate_chain, the bench's key) and the two points as pyref values"""
    C = pyref.CURVES[curve]
    rng = pyref.Rng(seed)
    P0, H = C.mul(rng.next_u64() | 1, C.G), C.mul(rng.next_u64() | 1, C.G)
    xy, _ = S.bases_array(C, [P0, H])
    return gl.ResidentBases.chain(curve, xy[0], xy[1], n), P0, H


def expect_affine(curve, P):
    xy, inf = S.affine_abi_of_point(pyref.CURVES[curve], P)
    return inf, xy.tobytes()


@pytest.mark.parametrize("curve,log_n,with_table", [("mnt4753_g1", 20, True), ("mnt4753_g1", 24, True), ("mnt6753_g1", 22, True),
                                                    ("mnt6753_g1", 19, True), ("mnt4753_g2", 20, True), ("mnt6753_g2", 19, True),
                                                    ("mnt6753_g2", 22, False)])
def test_full_size_msm_against_closed_form(gpu, curve, log_n, with_table):
    """Every BASELINE size with an answer that no MSM code path produced: on a chain key P_i = P_0 + i H
    sum s_i P_i = (sum s_i) P_0 + (sum i s_i) H -- two scalar multiplications in Python integers (the reference's own test
    is Pippenger == naive sum, variable_base.rs:102-151).  Per-window path, shift table, a pipelined batch with a ragged
    second length, and the first few bases of the device-generated chain against P_0 + i H itself."""
    C = pyref.CURVES[curve]
    n = 1 << log_n
    rb, P0, H = chain_key(gpu, curve, n, 1000 + log_n)
    s = S.random_scalars_np(n, seed=600 + log_n, below=C.order)
    s[5] = 0
    s[6, :] = 0
    s[6, 0] = 1
    s[7] = np.array(pyref.int_to_limbs(C.order - 1), dtype=np.uint64)
    m = n - 12345
    exp = expect_affine(curve, S.chain_msm_closed_form(C, P0, H, s))
    exp_m = expect_affine(curve, S.chain_msm_closed_form(C, P0, H, s[:m]))
    ds = gpu.DeviceBuffer(s.nbytes).upload(s)
    try:
        head = rb.download(0, 3)
        want = S.bases_array(C, [P0, C.add(P0, H), C.add(C.add(P0, H), H)])[0]
        assert (head == want).all()
        tail = rb.download(n - 1, 1)
        assert (tail == S.bases_array(C, [C.add(P0, C.mul(n - 1, H))])[0]).all()
        assert affine(gpu, curve, rb.msm_dev(ds, n)) == exp                      # per-window path
        if with_table:
            rb.precompute(0)
            assert affine(gpu, curve, rb.msm_dev(ds, n)) == exp
        outs = gpu.msm_batch_dev([(rb, ds, n), (rb, ds, m), (rb, ds, n)])
        assert [affine(gpu, curve, o) for o in outs] == [exp, exp_m, exp]
    finally:
        ds.free()
        rb.free()
        gpu.dev_trim()


# ------------------------------------------------------------------------------ config 4 shapes
@pytest.mark.parametrize("curve,log_n,pool_n,with_table", [("mnt6753_g1", 19, 4096, True), ("mnt6753_g1", 22, 4096, True),
                                                          ("mnt6753_g2", 19, 512, True), ("mnt6753_g2", 22, 512, False),
                                                          ("mnt4753_g2", 20, 512, True)])
def test_cfg4_shapes(gpu, no_dedup, curve, log_n, pool_n, with_table):
    C = pyref.CURVES[curve]
    r = C.order
    n = 1 << log_n
    bases, _ = tiled_bases(curve, n, pool_n, 77 + log_n)
    s = S.random_scalars_np(n, seed=41 + log_n, below=r)
    rb = gpu.ResidentBases(curve, bases)
    ds = gpu.DeviceBuffer(s.nbytes).upload(s)
    try:
        ref = affine(gpu, curve, rb.msm_dev(ds, n))
        c0 = gpu.msm_last_timing()["window_bits"]
        gpu.msm_set_window(c0 - 2)                                      # window-size invariance
        assert affine(gpu, curve, rb.msm_dev(ds, n)) == ref
        assert gpu.msm_last_timing()["window_bits"] == c0 - 2
        gpu.msm_set_window(0)
        k = 1 << (13 if C.deg == 1 else 12)                             # oracle spot check on a slice
        exp = S.oracle_msm(curve, bases[:k], None, s[:k], 16)
        assert affine(gpu, curve, rb.msm(s[:k])) == oracle_affine(curve, exp)
        if with_table:
            c = rb.precompute(0)
            assert affine(gpu, curve, rb.msm_dev(ds, n)) == ref
            assert gpu.msm_last_timing()["window_bits"] == c
            assert affine(gpu, curve, rb.msm(s[:k])) == oracle_affine(curve, exp)
    finally:
        gpu.msm_set_window(0)
        ds.free()
        rb.free()
        gpu.dev_trim()


@pytest.mark.parametrize("curve,log_n", [("mnt4753_g2", 15), ("mnt6753_g2", 14), ("mnt6753_g1", 16)])
def test_split_kernels_heavy_buckets_batch_vs_oracle(gpu, no_dedup, curve, log_n):
    """mid-size oracle parity with skewed scalars (heavy buckets and chunks), duplicate and opposite bases, with and
    without the shift table (several pseudo-windows, L1 = 16 segments), and as a pipelined batch of 4 MSMs"""
    C = pyref.CURVES[curve]
    r = C.order
    n = 1 << log_n
    bases, pool = tiled_bases(curve, n, 256, 5 + log_n)
    # duplicate and opposite bases next to each other, with equal scalars: P + P and P - P inside one bucket
    negp = S.bases_array(C, [C.neg(p) for p in pool[:64]])[0]
    bases[64:128] = negp
    s = skew(S.random_scalars_np(n, seed=51 + log_n, below=r), r)
    s[64:128] = s[0:64]
    s[256:320] = s[0:64]
    t = S.random_scalars_np(n, seed=52 + log_n, below=r)
    exp_s = oracle_affine(curve, S.oracle_msm(curve, bases, None, s, 16))
    exp_t = oracle_affine(curve, S.oracle_msm(curve, bases, None, t, 16))
    half = n // 2 + 3
    exp_h = oracle_affine(curve, S.oracle_msm(curve, bases, None, s[:half], 16))
    rb = gpu.ResidentBases(curve, bases)
    ds, dt = gpu.DeviceBuffer(s.nbytes).upload(s), gpu.DeviceBuffer(t.nbytes).upload(t)
    try:
        for table in (False, True):
            if table:
                rb.precompute(0)
            for mode in (0, 1):       # projective bucket sums (heavy buckets cut into chunks) and affine rounds (extra rounds)
                gpu.msm_set_affine(mode)
                assert affine(gpu, curve, rb.msm_dev(ds, n)) == exp_s, (curve, table, mode)
                if mode == 0 and not table:
                    assert gpu.msm_last_timing()["heavy_buckets"] > 0
                outs = gpu.msm_batch_dev([(rb, ds, n), (rb, dt, n), (rb, ds, half), (rb, dt, n)])
                got = [affine(gpu, curve, o) for o in outs]
                assert got == [exp_s, exp_t, exp_h, exp_t], (curve, table, mode)
    finally:
        gpu.msm_set_affine(2)
        ds.free(); dt.free()
        rb.free()
        gpu.dev_trim()


def test_partition_sort_path_skewed_vs_oracle(gpu, no_dedup):
    """the bucket lists of large inputs come from the two-level counting sort (msm_kernels.h 2a: W n >= 2^22 entries): oracle parity
    at 2^17 pairs with skewed scalars (zeros, ones, one value repeated thousands of times: heavy buckets, bins with a single
    bucket holding most entries), infinity bases, per-window path and shift table, one MSM and a batch with a ragged length"""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    r = C.order
    n = 1 << 17
    bases, pool = tiled_bases(curve, n, 256, 77)
    s = skew(S.random_scalars_np(n, seed=91, below=r), r)
    s[1000:9000] = s[999]                                    # 8 000 equal scalars: the same bucket in every window
    inf = np.zeros(n, dtype=np.uint8)
    inf[5::97] = 1
    t = S.random_scalars_np(n, seed=92, below=r)
    exp_s = oracle_affine(curve, S.oracle_msm(curve, bases, inf, s, 16))
    exp_t = oracle_affine(curve, S.oracle_msm(curve, bases, inf, t, 16))
    m = n - 12345
    exp_m = oracle_affine(curve, S.oracle_msm(curve, bases[:m], inf[:m], s[:m], 16))
    rb = gpu.ResidentBases(curve, bases, infinity=inf)
    ds, dt = gpu.DeviceBuffer(s.nbytes).upload(s), gpu.DeviceBuffer(t.nbytes).upload(t)
    try:
        for table in (False, True):
            if table:
                rb.precompute(0)
            assert affine(gpu, curve, rb.msm_dev(ds, n)) == exp_s, table
            assert gpu.msm_last_timing()["heavy_buckets"] > 0
            outs = gpu.msm_batch_dev([(rb, dt, n), (rb, ds, m), (rb, ds, n)])
            assert [affine(gpu, curve, o) for o in outs] == [exp_t, exp_m, exp_s], table
    finally:
        ds.free(); dt.free()
        rb.free()
        gpu.dev_trim()


# ------------------------------------------------------------------------------ config 5 at 2^20
def _replay(gpu, log_n, with_oracle):
    """device halves of create_proof for a 2^log_n-constraint MNT4-753 circuit (tools/prover_replay.py as a test)"""
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    from test_groth16_stage import _oracle_stage
    pairing = "mnt4753"
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    N = 1 << log_n
    ni, nv = 3, N - 1
    rng = pyref.Rng(5)
    b1 = S.bases_array(C1, S.chain_points(C1, 1 << 10, rng))[0]
    b2 = S.bases_array(C2, S.chain_points(C2, 1 << 6, rng))[0]
    tile1 = lambda m, shift: np.roll(np.tile(b1, (m // len(b1) + 1, 1)), shift, axis=0)[:m]
    pk = {"a_query": tile1(nv, 0), "b_g1_query": tile1(nv, 3), "h_query": tile1(N - 1, 7), "l_query": tile1(nv - ni, 11),
          "b_g2_query": np.tile(b2, (nv // len(b2) + 1, 1))[:nv],
          "alpha_g1": b1[5], "beta_g1": b1[6], "delta_g1": b1[7], "beta_g2": b2[5], "delta_g2": b2[7]}
    r_ord = C1.order
    F = "mnt4753_fr"
    a, b, c = (S.random_scalars_np(N, seed=s0, below=r_ord) for s0 in (1, 2, 3))
    d = S.random_scalars_np(3, seed=4, below=r_ord)
    assign = S.random_scalars_np(nv - 1, seed=9, below=r_ord)
    assign[::7] = 0
    assign[1::7, 1:] = 0
    assign[1::7, 0] = 1
    r, s = S.random_scalars_np(2, seed=10, below=r_ord)
    lib = gpu.load_library()
    one_plain = np.zeros(12, dtype=np.uint64)
    one_plain[0] = 1
    results = []
    h = None
    for precompute in (True, False):
        key = groth16.ResidentProvingKey(gpu, pairing, pk, ni, precompute=precompute)
        da, db, dc, dh = (gpu.DeviceBuffer(N * 96 + 96) for _ in range(4))
        try:
            da.upload(a); db.upload(b); dc.upload(c)
            gpu._check(lib.gh_witness_map_dev(gpu.FIELDS[F], da.ptr, db.ptr, dc.ptr, log_n, gpu._ptr(d[0]), gpu._ptr(d[1]), gpu._ptr(d[2]), dh.ptr))
            gpu._check(lib.gh_vec_scale_dev(gpu.FIELDS[F], dh.ptr, gpu._ptr(one_plain), N + 1))       # into_repr (prover.rs:256-267)
            proof = key.create_proof_msms(assign[:ni - 1], assign[ni - 1:], None, None, r, s, h_dev=(dh, N - 1))
            results.append([(inf, np.asarray(xy).tobytes()) for xy, inf in proof])
            if h is None:
                h = dh.download()[:(N - 1) * 12].reshape(N - 1, 12).copy()
        finally:
            for x in (da, db, dc, dh):
                x.free()
            key.free()
    assert results[0] == results[1]                    # table path == per-window path, A, B and C
    if with_oracle:
        hm = S.oracle_witness_map(F, a, b, c, d[0], d[1], d[2])
        # the oracle's h is in Montgomery form; into_repr on the oracle side: multiply by the plain integer 1
        exp = _oracle_stage(pairing, pk, ni, assign[:ni - 1], assign[ni - 1:], h[:ni], h[ni:], r, s)
        assert results[0] == [(inf, np.asarray(xy).tobytes()) for xy, inf in exp]
        hm_plain = np.array([pyref.int_to_limbs(v) for v in S.fe_list(S.FIELD_OF[F], hm[:N - 1])], dtype=np.uint64)
        assert (hm_plain == h).all()
    gpu.dev_trim()


def test_cfg5_replay_2p16_vs_oracle(gpu, no_dedup):
    """witness map + into_repr + MSM stage at 2^16 constraints against the oracle's literal replay (nine MSMs)"""
    _replay(gpu, 16, True)


def test_cfg5_replay_2p20(gpu, no_dedup):
    """the same at BASELINE config 5's size: A, B, C from the table path and from the per-window path agree"""
    _replay(gpu, 20, False)


# ------------------------------------------------------------------------------ the C ABI from two threads
def test_abi_two_threads(gpu):
    """include/ginger_hip.h: "calls are serialised by an internal lock and may come from any thread" -- two threads
    issue host-buffer MSMs and transforms with different inputs and sizes at the same time; each must get its own result"""
    curve = "mnt4753_g1"
    C = pyref.CURVES[curve]
    rng = pyref.Rng(8)
    pts = S.chain_points(C, 700, rng)
    b, inf = S.bases_array(C, pts)
    rb = gpu.ResidentBases(curve, b, inf)
    jobs = []
    for i, n in enumerate((700, 300, 650, 120)):
        s = S.scalar_array([rng.field_elem(C.order) for _ in range(n)])
        jobs.append((s, oracle_affine(curve, S.oracle_msm(curve, b, inf, s, 8))))
    ffts = []
    for i, lg in enumerate((10, 12, 9, 11)):
        a = S.random_scalars_np(1 << lg, seed=60 + i, below=pyref.P6.p)
        ffts.append((a, lg, S.oracle_fft("mnt4753_fr", a, lg, i & 3)))
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                j = (tid * 2 + rep) % 4
                s, exp = jobs[j]
                got = rb.msm(s) if rep % 2 else gpu.VariableBaseMSM.multi_scalar_mul(curve, b, s, inf)
                if affine(gpu, curve, got) != exp:
                    errors.append(("msm", tid, rep))
                a, lg, ref = ffts[j]
                dom = gpu.EvaluationDomain("mnt4753_fr", 1 << lg)
                out = dom._run(a, j & 3).reshape(-1, 12)
                if not (out == ref).all():
                    errors.append(("fft", tid, rep))
        except Exception as e:     # noqa
            errors.append(("exception", tid, repr(e)))

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    rb.free()
    assert not errors, errors


# ------------------------------------------------------------------------------ config 5 end to end from reference-format bytes
@pytest.mark.parametrize("pairing,num_constraints", [("mnt4753", 253), ("mnt6753", 125)])
def test_cfg5_create_proof_from_parameters_bytes(gpu, pairing, num_constraints):
    """create_proof end to end on reference-format bytes: a Parameters::write stream (groth16/mod.rs:188-208) made by the
    first-principles generator (tests/groth16_ref.py) is parsed into five resident queries (infinity entries included), the
    `Benchmark` circuit's rows are evaluated on the host, witness map and MSM stage run on the device, and the
    Proof::write bytes (mod.rs:35-42) equal the oracle's literal replay of prover.rs:201-345 -- with and without shift tables."""
    import groth16_ref as G
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    blob, info = G.generate_parameters(pairing, num_constraints, seed=7 + num_constraints)
    pk = groth16.parse_parameters(pairing, blob)
    C1 = pyref.CURVES[pairing + "_g1"]
    n_b_inf = sum(1 for P in info["key"]["b_g1_query"] if P is None)
    assert n_b_inf > 0                                           # the key does contain GroupAffine::zero() entries
    assert len(pk["a_query"]) == 193 * len(info["key"]["a_query"]) and len(pk["h_query"]) == 193 * ((1 << info["log_n"]) - 1)
    rng = pyref.Rng(99)
    r_mod = C1.order
    d1, d2, d3, r_, s_ = (rng.field_elem(r_mod) for _ in range(5))
    rows = groth16.benchmark_circuit_rows(pairing, num_constraints)
    assert rows[0] == info["num_inputs"] and rows[1] == info["assignment"]
    exp = G.oracle_create_proof(pairing, info, d1, d2, d3, r_, s_)
    for precompute in (False, True):
        key = groth16.ResidentProvingKey.from_parameters(gpu, pairing, blob, info["num_inputs"], precompute=precompute)
        try:
            got = key.create_proof(rows, d1, d2, d3, r_, s_)
        finally:
            key.free()
        assert got == exp, (pairing, precompute)
    with pytest.raises(ValueError):
        groth16.parse_parameters(pairing, blob[:-5])
    gpu.dev_trim()


def test_cfg5_benchmark_circuit_2p20_end_to_end(gpu):
    """BASELINE config 5 at its own size: the `Benchmark` circuit with 2^20 - 3 constraints (domain 2^20), rows evaluated on the
    host, device witness map + MSM stage over a synthetic resident key; proof bytes from the table path and from the per-window
    path are identical.  (A reference-format key of this size would take the CPU generator hours; the byte path is checked at
    small sizes above, the arithmetic at this size here.)"""
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    pairing = "mnt4753"
    n_con = (1 << 20) - 3
    rows = groth16.benchmark_circuit_rows(pairing, n_con)
    ni, assignment = rows[0], rows[1]
    assert ni == 3 and len(assignment) == 3 + n_con
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    nv = len(assignment)                                          # a_query has one entry per variable (incl. one)
    N = 1 << 20

    def chain(curve, n, seed):
        Cc = pyref.CURVES[curve]
        prng = pyref.Rng(seed)
        xy, _ = S.bases_array(Cc, [Cc.mul(prng.next_u64() | 1, Cc.G), Cc.mul(prng.next_u64() | 1, Cc.G)])
        rb = gpu.ResidentBases.chain(curve, xy[0], xy[1], n)
        rows_ = rb.download(0, n)
        rb.free()
        return rows_
    pk = {"a_query": chain(pairing + "_g1", nv, 1), "b_g1_query": chain(pairing + "_g1", nv, 2), "h_query": chain(pairing + "_g1", N - 1, 3),
          "l_query": chain(pairing + "_g1", nv - ni, 4), "b_g2_query": chain(pairing + "_g2", nv, 5)}
    pk.update({"alpha_g1": pk["h_query"][5], "beta_g1": pk["h_query"][6], "delta_g1": pk["h_query"][7],
               "beta_g2": pk["b_g2_query"][5], "delta_g2": pk["b_g2_query"][7]})
    rng = pyref.Rng(5)
    d1, d2, d3, r_, s_ = (rng.field_elem(C1.order) for _ in range(5))
    proofs = []
    for precompute in (True, False):
        key = groth16.ResidentProvingKey(gpu, pairing, pk, ni, precompute=precompute)
        key.pairing, key.pk_delta_g1 = pairing, pk["delta_g1"]
        try:
            proofs.append(key.create_proof(rows, d1, d2, d3, r_, s_))
        finally:
            key.free()
    assert proofs[0] == proofs[1] and len(proofs[0]) == 193 + 385 + 193
    gpu.dev_trim()


# ------------------------------------------------------------------------------ config 5: the key itself made on the device
@pytest.mark.parametrize("pairing,n_con", [("mnt4753", 61), ("mnt6753", 29)])
def test_cfg5_generator_on_device_equals_first_principles_generator(gpu, pairing, n_con):
    """groth16.generate_parameters (generator.rs:146-335 above the C ABI: Lagrange coefficients, FixedBaseMSM, batch
    normalisation on the device) emits byte for byte the Parameters::write stream of tests/groth16_ref.py's generator, which
    is pinned by the proof-in-the-exponent test (tests/test_groth16_ref.py) -- same toxic waste, same generators."""
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    import groth16_ref as G
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    blob_ref, info = G.generate_parameters(pairing, n_con, seed=3)
    alpha, beta, gamma, delta, t = info["toxic"]
    g1, g2 = info["generators"]
    lcs = groth16.benchmark_circuit_lcs(n_con)
    assert (lcs[0], lcs[2], lcs[3], lcs[4]) == (info["num_inputs"], info["at"], info["bt"], info["ct"])
    blob, ginfo = groth16.generate_parameters(gpu, pairing, lcs, alpha, beta, gamma, delta, t, S.proj_array(C1, g1), S.proj_array(C2, g2))
    assert ginfo["qap"][:4] == tuple(info["qap"][:4]) and ginfo["qap"][4] == info["qap"][4]
    assert blob == blob_ref
    gpu.dev_trim()


def test_cfg5_key_generated_on_device_at_2p20_and_proof_in_closed_form(gpu):
    """BASELINE config 5 at its own size from reference-format bytes: the proving key of the `Benchmark` circuit with
    2^20 - 3 constraints (domain 2^20) is GENERATED on the device (gh_lagrange_coefficients + gh_fixed_base_msm_affine, the
    reference's window rule), serialised as Parameters::write does, parsed back, made resident (gh_bases_upload_wire +
    shift tables) and used by create_proof; A, B and C are then compared with the Groth16 equations evaluated in the
    exponent from the toxic waste -- Python integers and two/three textbook scalar multiplications, no MSM involved."""
    import time
    groth16 = importlib.import_module("ginger_lib_amd.groth16")
    import groth16_ref as G
    pairing = "mnt4753"
    C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
    r = C1.order
    n_con = (1 << 20) - 3
    rng = pyref.Rng(2026)
    alpha, beta, gamma, delta, t, r_, s_ = (rng.field_elem(r) for _ in range(7))
    g1, g2 = C1.mul(rng.next_u64() | 1, C1.G), C2.mul(rng.next_u64() | 1, C2.G)
    t0 = time.perf_counter()
    lcs = groth16.benchmark_circuit_lcs(n_con)
    blob, info = groth16.generate_parameters(gpu, pairing, lcs, alpha, beta, gamma, delta, t, S.proj_array(C1, g1), S.proj_array(C2, g2))
    t_gen = time.perf_counter() - t0
    del lcs
    assert info["log_n"] == 20
    fb = info["fixed_base"]
    print("generate_parameters 2^20: %.1f s, %d fixed-base scalar-muls in %.2f s on the device (%.2f M/s), windows g1 %d g2 %d, %d MB" % (
        t_gen, fb["fixed_base_scalars"], fb["fixed_base_s"], fb["fixed_base_scalars"] / fb["fixed_base_s"] / 1e6, info["g1_window"], info["g2_window"], len(blob) >> 20))
    pk = groth16.parse_parameters(pairing, blob)
    assert len(pk["a_query"]) == 193 * (n_con + 3) and len(pk["h_query"]) == 193 * ((1 << 20) - 1) and len(pk["l_query"]) == 193 * n_con
    t0 = time.perf_counter()
    key = groth16.ResidentProvingKey.from_parameters(gpu, pairing, blob, 3)
    t_load = time.perf_counter() - t0
    del blob, pk
    try:
        rows = groth16.benchmark_circuit_rows(pairing, n_con)
        t0 = time.perf_counter()
        proof = key.create_proof(rows, 0, 0, 0, r_, s_)
        t_proof = time.perf_counter() - t0
    finally:
        key.free()
        gpu.dev_trim()
    print("key load %.1f s, create_proof %.2f s" % (t_load, t_proof))
    a, b, c, l, zt = info["qap"]
    asg = rows[1]
    assert len(asg) == len(a) == n_con + 3
    sa = sum(x * y for x, y in zip(asg, a)) % r
    sb = sum(x * y for x, y in zip(asg, b)) % r
    sc = sum(x * y for x, y in zip(asg, c)) % r
    di = pow(delta, -1, r)
    A_s = (alpha + sa + r_ * delta) % r
    B_s = (beta + sb + s_ * delta) % r
    H = (sa * sb - sc) * di % r                                  # h(t) Z(t) / delta with d1 = d2 = d3 = 0 (QAP divisibility)
    C_s = (sum(asg[i] * l[i] for i in range(3, len(asg))) + H + s_ * A_s + r_ * B_s - r_ * s_ * delta) % r
    exp = G.wire(C1, C1.mul(A_s, g1)) + G.wire(C2, C2.mul(B_s, g2)) + G.wire(C1, C1.mul(C_s, g1))
    assert proof == exp


# ------------------------------------------------------------------------------ the transforms at 2^24 and at the largest size of the assembly pass
@pytest.mark.parametrize("log_n", [24, 25, 26])
def test_ntt_full_size_sparse_input_against_closed_form(gpu, log_n):
    """2^24 (the bench's size), 2^25 (the largest domain whose passes run as generated assembly: 32-bit byte offsets) and 2^26 (the
    C++ pass with 64-bit offsets; 6.4 GB of data): a vector with a
    handful of non-zero coefficients has every output in closed form, X[k] = sum x_n (g^n) w^(n k) (domain.rs:113-179), whatever the
    size -- checked at the ends and at random indices for fft and coset_fft; ifft / coset_ifft then return the input bit for bit."""
    F = S.FIELD_OF["mnt4753_fr"]
    p = F.p
    n = 1 << log_n
    rnd = np.random.default_rng(500 + log_n)
    pos = sorted({0, 1, n - 1, n // 2 + 1} | {int(v) for v in rnd.integers(0, n, size=4)})
    vals = [int.from_bytes(rnd.bytes(96), "little") % p for _ in pos]
    a = np.zeros((n, 12), dtype=np.uint64)
    a[pos] = S.fe_array(F, vals)
    w = pyref.domain_params(F, log_n)
    g = F.generator
    ks = sorted({0, 1, n - 1, n // 2, n // 2 - 1} | {int(v) for v in rnd.integers(0, n, size=40)})
    dom = gpu.EvaluationDomain("mnt4753_fr", n)
    buf = gpu.DeviceBuffer(n * 96).upload(a)
    try:
        for fwd, inv, coset in ((0, 1, False), (2, 3, True)):
            dom.fft_dev(buf, fwd)
            out = buf.download().reshape(n, 12)
            got = S.fe_list(F, out[ks])
            for k, gk in zip(ks, got):
                exp = sum(v * (pow(g, q, p) if coset else 1) * pow(w, (q * k) % n, p) for q, v in zip(pos, vals)) % p
                assert gk == exp, (log_n, coset, k)
            del out
            dom.fft_dev(buf, inv)
            back = buf.download().reshape(n, 12)
            assert (back == a).all(), (log_n, coset)
            del back
    finally:
        buf.free()
