"""python3 tools/g2_random_key_probe.py <curve> <log_n> [batch=10]
MSMs over a key of RANDOM points (k_i G with random k_i, made on the device by FixedBaseMSM) next to the chain key P0 + i S of the
bench: an arithmetic progression has equal partial sums in the later affine rounds (P_i + P_j = P_k + P_l whenever i + j = k + l),
which go through the exception list; a proving key has no such structure.  Closed form: sum s_i k_i G = (sum s_i k_i mod r) G."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyref, support as S
from __graft_entry__ import _load_pkg
gl = _load_pkg()
curve, log_n = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 10
gl.init()
n = 1 << log_n
C = pyref.CURVES[curve]
r = C.order
G = S.proj_array(C, C.G)


def to_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 12)
    return [sum(int(v) << (64 * i) for i, v in enumerate(row)) for row in a]


def run(rb, label, ds, check):
    rb.precompute(0)
    rb.msm_dev(ds, n)
    t0 = time.perf_counter(); out = rb.msm_dev(ds, n); one = (time.perf_counter() - t0) * 1e3
    gl.msm_batch_dev([(rb, ds, n)] * 2)
    t0 = time.perf_counter(); outs = gl.msm_batch_dev([(rb, ds, n)] * batch); dt = (time.perf_counter() - t0) / batch
    ok = check(out) and check(outs[-1])
    print("%s %s 2^%d: one MSM %.1f ms, pipelined batch of %d: %.2f ms per MSM = %.2f M scalar-muls/s, closed form %s" % (
        curve, label, log_n, one, batch, dt * 1e3, n / dt / 1e6, "ok" if ok else "MISMATCH"), flush=True)
    return ok


sc = S.random_scalars_np(n, seed=9, below=r)
ds = gl.DeviceBuffer(n * 96).upload(sc)
s_int = to_ints(sc)
# random key
ks = S.random_scalars_np(n, seed=77, below=r)
tab = gl.FixedBaseMSM(curve, G, 753, None, n)
xy, inf = tab.multi_scalar_mul_affine(ks)
tab.free()
assert not inf.any()
rb = gl.ResidentBases(curve, xy)
k_int = to_ints(ks)
tot = sum(a * b for a, b in zip(s_int, k_int)) % r
exp = C.mul(tot, C.G)


def check_random(out):
    g_xy, g_inf = gl.proj_to_affine(curve, out)
    e = S.bases_array(C, [exp])[0][0]
    return (not g_inf) and bool((np.asarray(g_xy).reshape(-1) == np.asarray(e).reshape(-1)).all())


ok1 = run(rb, "random key", ds, check_random)
rb.free()
# chain key (as bench.py / tools/g2_probe.py)
prng = pyref.Rng(5)
p0, step = C.mul(prng.next_u64() | 1, C.G), C.mul(prng.next_u64() | 1, C.G)
cxy, _ = S.bases_array(C, [p0, step])
rbc = gl.ResidentBases.chain(curve, cxy[0], cxy[1], n)
run(rbc, "chain key ", ds, lambda out: True)
sys.exit(0 if ok1 else 1)
