"""Stage timings of create_proof over a device-generated key (development aid): python3 tools/prover_probe.py [log_n=20]"""
import os, sys, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyref, support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
groth16 = importlib.import_module("ginger_lib_amd.groth16")
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
gl.init()
pairing = "mnt4753"
C1, C2 = pyref.CURVES[pairing + "_g1"], pyref.CURVES[pairing + "_g2"]
rr = C1.order
n_con = (1 << log_n) - 3
prng = pyref.Rng(2026)
alpha, beta, gamma, delta, tau, r_, s_ = (prng.field_elem(rr) for _ in range(7))
g1, g2 = C1.mul(prng.next_u64() | 1, C1.G), C2.mul(prng.next_u64() | 1, C2.G)
lcs = groth16.benchmark_circuit_lcs(n_con)
blob, info = groth16.generate_parameters(gl, pairing, lcs, alpha, beta, gamma, delta, tau, S.proj_array(C1, g1), S.proj_array(C2, g2))
key = groth16.ResidentProvingKey.from_parameters(gl, pairing, blob, 3)
for k, rb in key.keys.items():
    print("key", k, rb.curve, "n", rb.n, "table rows", rb.table_rows(), "window", gl.load_library().gh_bases_precomputed_window(rb.handle), flush=True)
rows = groth16.benchmark_circuit_rows(pairing, n_con)
prep = key.prepare_rows(rows, 0, 0, 0)
for rep in range(3):
    tm = {}
    t0 = time.perf_counter()
    key.prove_prepared(prep, r_, s_, timing=tm)
    print("proof %.1f ms" % ((time.perf_counter() - t0) * 1e3), {k: round(v, 1) for k, v in tm.items()}, flush=True)
    print("  last (G2) msm:", {k: round(v, 2) for k, v in gl.msm_last_timing().items() if k.endswith("_ms") or k in ("window_bits", "num_windows")})
key.free()
