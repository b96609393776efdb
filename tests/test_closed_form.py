"""The closed form the full-size GPU tests are judged by (tests/support.py: chain_sums, chain_msm_closed_form), checked on the
CPU against plain Python sums and against the oracle's literal msm_inner (variable_base.rs:10-83) on a chain key."""
import numpy as np
import pytest

import pyref
import support as S


def test_chain_sums_exact():
    r = pyref.CURVES["mnt4753_g1"].order
    s = S.random_scalars_np(3001, seed=3, below=r)
    s[17] = np.array(pyref.int_to_limbs(r - 1), dtype=np.uint64)
    ints = [S.to_int(row) for row in s]
    for chunk in (64, 1000, 1 << 18):
        a, b = S.chain_sums(s, r, chunk=chunk)
        assert a == sum(ints) % r
        assert b == sum(i * v for i, v in enumerate(ints)) % r


@pytest.mark.parametrize("curve,n", [("mnt4753_g1", 1 << 10), ("mnt6753_g1", 700), ("mnt4753_g2", 200), ("mnt6753_g2", 120)])
def test_closed_form_equals_oracle_msm(curve, n):
    C = pyref.CURVES[curve]
    rng = pyref.Rng(77)
    P0, H = C.mul(rng.next_u64() | 1, C.G), C.mul(rng.next_u64() | 1, C.G)
    pts, P = [], P0
    for _ in range(n):
        pts.append(P)
        P = C.add(P, H)
    bases, _ = S.bases_array(C, pts)
    s = S.random_scalars_np(n, seed=9, below=C.order)
    s[3] = 0
    s[4, :] = 0
    s[4, 0] = 1
    exp = S.chain_msm_closed_form(C, P0, H, s)
    xy, inf = S.oracle_affine(curve, S.oracle_msm(curve, bases, None, s, 8))
    want_xy, want_inf = S.affine_abi_of_point(C, exp)
    assert inf == want_inf and (xy == want_xy).all()
