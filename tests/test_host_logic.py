"""Host-side logic of the prover mirrors that needs no device (ginger-lib_amd/groth16.py, gm17.py): the `Benchmark` circuit as
linear combinations agrees with its evaluated rows, and the host part of R1CStoSAP::witness_map (gm17/r1cs_to_sap.rs:123-219)
produces an assignment that satisfies the SAP it is built for and feeds the oracle's transform half to a valid quotient."""
import importlib

import pytest

import pyref
import support as S


@pytest.fixture(scope="module")
def mods(gl):
    return importlib.import_module("ginger_lib_amd.groth16"), importlib.import_module("ginger_lib_amd.gm17")


@pytest.mark.parametrize("pairing,n_con", [("mnt4753", 30), ("mnt6753", 17)])
def test_benchmark_lcs_agree_with_rows(mods, pairing, n_con):
    groth16, _ = mods
    r = groth16._MODULUS[pairing]
    ni, n_aux, at, bt, ct = groth16.benchmark_circuit_lcs(n_con)
    ni2, asg, A, B, C = groth16.benchmark_circuit_rows(pairing, n_con)
    assert ni == ni2 == 3 and len(asg) == ni + n_aux and len(at) == len(bt) == len(ct) == n_con
    ev = lambda row: sum(cf * asg[ix] for cf, ix in row) % r
    assert [ev(x) for x in at] == A and [ev(x) for x in bt] == B and [ev(x) for x in ct] == C
    assert all(a * b % r == c for a, b, c in zip(A, B, C))          # the witness satisfies the R1CS


@pytest.mark.parametrize("pairing,n_con", [("mnt4753", 29), ("mnt6753", 13)])
def test_sap_rows_and_quotient(mods, pairing, n_con):
    groth16, gm17 = mods
    field = pairing + "_fr"
    F = S.FIELD_OF[field]
    r = F.p
    rows = groth16.benchmark_circuit_rows(pairing, n_con)
    ni = rows[0]
    full, a, c, log_n = gm17.sap_rows_from_r1cs(pairing, *rows)
    size = 1 << log_n
    assert size >= 2 * n_con + 2 * (ni - 1) + 1 and len(a) == len(c) == size
    assert len(full) == len(rows[1]) + n_con + (ni - 1)                     # aux, then one extra variable per constraint and per input
    assert all(x * x % r == y for x, y in zip(a, c))                        # every SAP constraint: (row of A)^2 = row of C
    # the extra variables are what r1cs_to_sap.rs:127-149 says
    nv = len(rows[1])
    assert full[nv:nv + n_con] == [(x - y) ** 2 % r for x, y in zip(rows[2], rows[3])]
    assert full[nv + n_con:] == [(full[i] - 1) ** 2 % r for i in range(1, ni)]
    # the oracle's transform half (restated witness_map :191-240) with d1 = d2 = 0: h is the quotient (a(X)^2 - c(X)) / Z(X)
    zero = S.fe_array(F, [0])[0]
    h = S.fe_list(F, S.oracle_sap_witness_map(field, S.fe_array(F, a), S.fe_array(F, c), zero, zero))
    assert len(h) == size + 1
    w = pyref.domain_params(F, log_n)
    # interpolate a and c from their values on the domain and compare at a random point t:  a(t)^2 - c(t) == h(t) (t^N - 1)
    rng = pyref.Rng(3)
    t = rng.field_elem(r)
    tn = pow(t, size, r)
    lag, wi = [], 1
    scale = (tn - 1) * pow(size, -1, r) % r
    for i in range(size):
        lag.append(scale * wi % r * pow((t - wi) % r, -1, r) % r)
        wi = wi * w % r
    at_t = sum(x * l for x, l in zip(a, lag)) % r
    ct_t = sum(x * l for x, l in zip(c, lag)) % r
    ht = sum(hj * pow(t, j, r) for j, hj in enumerate(h)) % r
    assert (at_t * at_t - ct_t) % r == ht * (tn - 1) % r
