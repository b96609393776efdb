#!/usr/bin/env python3
"""Generate first-principles golden vectors for the hot path (the reference
holds no MSM / FFT known-answer vectors on the 753-bit curves, so they are produced here from
textbook big-integer arithmetic -- tests/pyref.py: naive sum s_i P_i with affine chord-and-tangent,
O(n^2) DFT by definition -- and committed as fixtures).  Deterministic: seeds are fixed.
Writes tests/golden/msm_golden.json and tests/golden/ntt_golden.json.
"""
import json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import pyref


def gen_msm():
    out = {}
    rng = pyref.Rng(2024)
    for name, C in pyref.CURVES.items():
        r = C.order
        n = 20 if C.deg == 1 else 12
        H = C.mul(rng.next_u64() | 1, C.G)
        P = C.mul(rng.field_elem(r), C.G)
        pts = []
        for _ in range(n):
            pts.append(P)
            P = C.add(P, H)
        scal = [rng.field_elem(r) for _ in range(n + 3)]          # more scalars than bases: zip truncates
        scal[0] = 0
        scal[1] = 1
        scal[2] = r - 1
        scal[3] = 1 << 200
        scal[4] = scal[5]
        pts[6] = None                                             # infinity base
        pts[8] = pts[7]; scal[8] = scal[7]                        # same (base, scalar) twice: P + P branch
        pts[10] = C.neg(pts[9]); scal[10] = scal[9]               # P + (-P)
        exp = C.msm(pts, scal)
        out[name] = {
            "bases": [None if Q is None else [[hex(c) for c in Q[0]], [hex(c) for c in Q[1]]] for Q in pts],
            "scalars": [hex(s) for s in scal],
            "expected_affine": None if exp is None else [[hex(c) for c in exp[0]], [hex(c) for c in exp[1]]],
        }
        # a case whose total is the point at infinity: s*P + (r-s)*P
        s = rng.field_elem(r)
        out[name + "_zero_sum"] = {
            "bases": [[[hex(c) for c in pts[0][0]], [hex(c) for c in pts[0][1]]]] * 2,
            "scalars": [hex(s), hex(r - s)],
            "expected_affine": None,
        }
    json.dump(out, open(os.path.join(HERE, "msm_golden.json"), "w"), indent=0)


def gen_ntt():
    out = {}
    rng = pyref.Rng(777)
    for tag, F, sizes in (("mnt4753_fr", pyref.P6, (0, 1, 2, 5)), ("mnt6753_fr", pyref.P4, (3,))):
        for log_n in sizes:
            n = 1 << log_n
            for n_in in sorted({n, max(1, n - 1), n + 1}):
                a = [rng.field_elem(F.p) for _ in range(n_in)]
                case = {"log_n": log_n, "input": [hex(x) for x in a]}
                for nm, inv, cos in (("fft", False, False), ("ifft", True, False), ("coset_fft", False, True), ("coset_ifft", True, True)):
                    case[nm] = [hex(x) for x in pyref.dft(F, a, log_n, inverse=inv, coset=cos)]
                out["%s_log%d_in%d" % (tag, log_n, n_in)] = dict(case, field=tag)
    json.dump(out, open(os.path.join(HERE, "ntt_golden.json"), "w"), indent=0)


if __name__ == "__main__":
    gen_msm()
    gen_ntt()
    print("ok")
