"""__graft_entry__.smoke(): one small invocation of each half of the hot path on cuda:0 through the
C ABI, checked against the CPU oracle."""
import numpy as np

import pyref
import support as S


def run(gl):
    gl.init(0)
    print("[smoke] device:", gl.device_name())
    rng = pyref.Rng(42)
    # --- MSM: MNT4-753 G1, 300 pairs incl. edge scalars
    C = pyref.CURVES["mnt4753_g1"]
    n = 300
    pts = S.chain_points(C, n, rng)
    scal = [rng.field_elem(C.order) for _ in range(n)]
    scal[0], scal[1], scal[2] = 0, 1, C.order - 1
    pts[5] = None
    b, inf = S.bases_array(C, pts)
    s = S.scalar_array(scal)
    got = gl.VariableBaseMSM.multi_scalar_mul("mnt4753_g1", b, s, inf)
    exp = S.oracle_msm("mnt4753_g1", b, inf, s, 4)
    g_xy, g_inf = gl.proj_to_affine("mnt4753_g1", got)
    e_xy, e_inf = S.oracle_affine("mnt4753_g1", exp)
    assert g_inf == e_inf and (g_xy == e_xy).all(), "MSM mismatch vs oracle"
    print("[smoke] msm ok", gl.msm_last_timing())
    # --- the same key resident with its shift table, two MSMs as one pipelined batch (the bench's path)
    rb = gl.ResidentBases("mnt4753_g1", b, inf)
    c = rb.precompute(0)
    ds = gl.DeviceBuffer(s.nbytes).upload(s)
    outs = gl.msm_batch_dev([(rb, ds, n), (rb, ds, n - 100)])
    exp2 = S.oracle_msm("mnt4753_g1", b, inf, s[:n - 100], 4)
    for o, e in ((outs[0], exp), (outs[1], exp2)):
        g_xy, g_inf = gl.proj_to_affine("mnt4753_g1", o)
        e_xy, e_inf = S.oracle_affine("mnt4753_g1", e)
        assert g_inf == e_inf and (g_xy == e_xy).all(), "table / batch MSM mismatch vs oracle"
    ds.free()
    rb.free()
    print("[smoke] msm on shift table (c = %d), pipelined batch ok" % c)
    # --- NTT: MNT4-753 Fr, 2^10, all four transforms
    F = pyref.P6
    a = S.fe_array(F, [rng.field_elem(F.p) for _ in range(1000)])   # padded to 1024
    dom = gl.EvaluationDomain("mnt4753_fr", 1000)
    for name, flags in (("fft", 0), ("ifft", 1), ("coset_fft", 2), ("coset_ifft", 3)):
        got = getattr(dom, name)(a).reshape(-1, 12)
        exp = S.oracle_fft("mnt4753_fr", a, 10, flags, 4)
        assert (got == exp).all(), name + " mismatch vs oracle"
    print("[smoke] ntt ok")
