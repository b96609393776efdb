#!/usr/bin/env python3
"""BASELINE config 5 replayed from synthetic buffers (no Rust toolchain here): the device
halves of create_proof for a 2^log_n-constraint MNT4-753 circuit -- witness map (7 transforms + pointwise,
r1cs_to_qap.rs:121-166) and the MSM stage (prover.rs:273-345) over a device-resident proving key.
Circuit synthesis and the evaluation of the A/B/C rows are CPU scalar code in the reference and are not
part of the replay (the rows are random field elements).  --check compares A, B, C with the CPU oracle.
Usage: python tools/prover_replay.py [log_n] [--check]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyref
import support as S
from __graft_entry__ import _load_pkg

log_n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
check = "--check" in sys.argv
gl = _load_pkg()
gl.init()
groth16 = importlib.import_module("ginger_lib_amd.groth16")
pairing = "mnt4753"
C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
N = 1 << log_n
ni = 3                                   # public inputs incl. the constant one
nv = N - 1                               # variables such that the h query has N - 1 entries
rng = pyref.Rng(5)
pool1, pool2 = S.chain_points(C1, 1 << 10, rng), S.chain_points(C2, 1 << 6, rng)
b1 = S.bases_array(C1, pool1)[0]
b2 = S.bases_array(C2, pool2)[0]
tile1 = lambda m, shift: np.roll(np.tile(b1, (m // len(b1) + 1, 1)), shift, axis=0)[:m]
pk = {"a_query": tile1(nv, 0), "b_g1_query": tile1(nv, 3), "h_query": tile1(N - 1, 7), "l_query": tile1(nv - ni, 11),
      "b_g2_query": np.tile(b2, (nv // len(b2) + 1, 1))[:nv],
      "alpha_g1": b1[5], "beta_g1": b1[6], "delta_g1": b1[7], "beta_g2": b2[5], "delta_g2": b2[7]}
r_ord = C1.order
t0 = time.perf_counter()
key = groth16.ResidentProvingKey(gl, pairing, pk, ni, precompute=True)
print("proving key resident (5 query tails, shift tables): %.1f s" % (time.perf_counter() - t0), flush=True)

F = "mnt4753_fr"
a, b, c = (S.random_scalars_np(N, seed=s0, below=r_ord) for s0 in (1, 2, 3))
d = S.random_scalars_np(3, seed=4, below=r_ord)
da, db, dc, dh = (gl.DeviceBuffer(N * 96 + 96) for _ in range(4))
lib = gl.load_library()
assign = S.random_scalars_np(nv - 1, seed=9, below=r_ord)      # canonical scalars: inputs (ni - 1) || aux (nv - ni)
assign[::7] = 0
assign[1::7, 1:] = 0
assign[1::7, 0] = 1                                            # witness-like: many 0 / 1
r, s = S.random_scalars_np(2, seed=10, below=r_ord)


def run():
    t = {}
    t0 = time.perf_counter()
    da.upload(a); db.upload(b); dc.upload(c)
    t["upload_rows_ms"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    gl._check(lib.gh_witness_map_dev(gl.FIELDS[F], da.ptr, db.ptr, dc.ptr, log_n, gl._ptr(d[0]), gl._ptr(d[1]), gl._ptr(d[2]), dh.ptr))
    lib.gh_dev_sync()
    t["witness_map_ms"] = (time.perf_counter() - t0) * 1e3
    # into_repr() of every coefficient of h (prover.rs:256-267): Montgomery -> canonical is one product by the plain
    # integer 1, done in place on the device; the MSM stage then reads its h scalars where the witness map left them
    t0 = time.perf_counter()
    one_plain = np.zeros(12, dtype=np.uint64)
    one_plain[0] = 1
    gl._check(lib.gh_vec_scale_dev(gl.FIELDS[F], dh.ptr, gl._ptr(one_plain), N + 1))
    proof = key.create_proof_msms(assign[:ni - 1], assign[ni - 1:], None, None, r, s, h_dev=(dh, N - 1))
    lib.gh_dev_sync()
    t["into_repr_and_msm_stage_ms"] = (time.perf_counter() - t0) * 1e3
    h = None
    if check:
        h = dh.download()[:(N - 1) * 12].reshape(N - 1, 12)
    return proof, t, h


run()
proof, t, h = run()
print("replay 2^%d: " % log_n + ", ".join("%s %.1f" % kv for kv in t.items()), flush=True)
print("device halves of one proof: witness map + into_repr + MSM stage = %.1f ms" % (t["witness_map_ms"] + t["into_repr_and_msm_stage_ms"]), flush=True)
if check:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_groth16_stage import _oracle_stage
    t0 = time.perf_counter()
    exp = _oracle_stage(pairing, pk, ni, assign[:ni - 1], assign[ni - 1:], h[:ni], h[ni:], r, s)
    ok = all(gi == ei and bool((np.asarray(gx) == np.asarray(ex)).all()) for (gx, gi), (ex, ei) in zip(proof, exp))
    print("oracle replay of the MSM stage: %.1f s, A B C identical: %s" % (time.perf_counter() - t0, ok), flush=True)
    if not ok:
        sys.exit(1)
key.free()
