"""The NTT pass kernels of ginger-lib_amd/asmgen/ntt_pass.py executed on the CPU by asmgen/sim.py -- no GPU needed.

A pass is checked wave by wave against its definition (csrc/ntt_kernels.h ntt_pass_kernel, which restates
algebra/src/fft/domain.rs:262-317): column j reads x[j + t N / 2^k], multiplies by the optional coset factor and the
inter-pass twiddle, takes the 2^k-point DFT over t with the root w^(N / 2^k) and writes DFT row u to jbase + (u << log_ns),
times the optional final factor -- all in Python integers here.  Whole transforms (two passes) are compared with the direct
evaluation of the polynomial at the powers of w.  Every load and store is checked against the buffers it may touch."""
import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ginger-lib_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref                                                      # noqa: E402
from asmgen import ntt_pass as NP                                 # noqa: E402
from asmgen.field import NL, limbs, unlimbs, FieldGen, Chain      # noqa: E402
from asmgen.isa import Prog, V, S, fix_hazards                    # noqa: E402
from asmgen.sim import Memory, Wave                               # noqa: E402

R = 1 << 754
_PROGS = {}


def _prog(p, k):
    if (p, k) not in _PROGS:
        _PROGS[(p, k)] = NP.build("ntt_k%d" % k, p, k)
    return _PROGS[(p, k)]


def abi_words(x):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(24)]


def from_abi(w):
    return sum(int(v) << (32 * i) for i, v in enumerate(w))


def root_of_unity(p, log_n):
    """an element of order exactly 2^log_n"""
    s = 0
    q = p - 1
    while q % 2 == 0:
        q //= 2
        s += 1
    assert log_n <= s
    g = 2
    while pow(g, (p - 1) // 2, p) == 1:
        g += 1
    return pow(pow(g, q, p), 1 << (s - log_n), p)


def run_pass(p, k, x, w, log_n, log_ns, inverse, pre=None, post=None, post_scalar=None, waves=None):
    """x: N residues (as the ABI words hold them); returns the output vector (None where no simulated wave wrote)"""
    N = 1 << log_n
    g = _prog(p, k)
    mem = Memory()
    xin = np.array([abi_words(v) for v in x], dtype=np.uint32)
    out = np.full((N, 24), 0xDEADBEEF, dtype=np.uint32)
    tw = np.array([limbs(pow(w, i, p) * R % p) for i in range(N)], dtype=np.uint32)
    a_in, a_out, a_tw = mem.add("in", xin), mem.add("out", out, writable=True), mem.add("tw", tw)
    a_pre = a_post = 0
    stride = 0
    if pre is not None:
        a_pre = mem.add("pre", np.array([limbs(v * R % p) for v in pre], dtype=np.uint32))
    if post is not None:
        a_post = mem.add("post", np.array([limbs(v * R % p) for v in post], dtype=np.uint32))
        stride = 104
    elif post_scalar is not None:
        a_post = mem.add("post", np.array(limbs(post_scalar * R % p), dtype=np.uint32))
    karg = np.zeros(16, dtype=np.uint32)
    for j, a in enumerate((a_in, a_out, a_tw, a_pre, a_post)):
        karg[2 * j], karg[2 * j + 1] = a & 0xFFFFFFFF, a >> 32
    n_waves = N >> 8
    karg[10], karg[11], karg[12], karg[13], karg[14] = log_n, log_ns, 1 if inverse else 0, stride, n_waves
    a_karg = mem.add("karg", karg)
    todo = list(range(n_waves)) if waves is None else list(waves)
    for wv in todo + [n_waves]:                                   # one wave beyond the work: it must leave at once
        wave = Wave(g, mem)
        wave.S[0], wave.S[1], wave.S[2] = a_karg & 0xFFFFFFFF, a_karg >> 32, wv // 4
        wave.V[0] = np.arange(64, dtype=np.uint32) + 64 * (wv % 4)
        wave.run()
    o = mem.get("out").reshape(N, 24)
    return [None if int(r[0]) == 0xDEADBEEF and int(r[1]) == 0xDEADBEEF else from_abi(r) for r in o]


def model_pass(p, k, x, w, log_n, log_ns, inverse, pre=None, post=None, post_scalar=None, columns=None):
    N = 1 << log_n
    Rk = 1 << k
    stride = N >> k
    sh = log_n - log_ns - k
    out = [None] * N
    wr = pow(w, N >> k, p)
    if inverse:
        wr = pow(wr, -1, p)
    for j in (range(stride) if columns is None else columns):
        kk = j & ((1 << log_ns) - 1)
        col = []
        for t in range(Rk):
            v = x[j + t * stride]
            if pre is not None:
                v = v * pre[j + t * stride] % p
            if log_ns > 0:
                e = (kk * t) << sh
                if inverse:
                    e = (N - e) % N
                v = v * pow(w, e, p) % p
            col.append(v)
        jbase = ((j - kk) << k) + kk
        for u in range(Rk):
            acc = 0
            for t in range(Rk):
                acc += col[t] * pow(wr, (u * t) % Rk, p)
            acc %= p
            o = jbase + (u << log_ns)
            if post is not None:
                acc = acc * post[o] % p
            elif post_scalar is not None:
                acc = acc * post_scalar % p
            out[o] = acc
    return out


def _fr(name):
    """the scalar field of the curve: the field the prover's transforms run over"""
    return int(pyref.CURVES[name].order)


def test_add_and_difference_routines_against_integers():
    p = _fr("mnt4753_g1")
    g = Prog("t")
    f = FieldGen(g, p, 24, 50, 21, 20)
    ch = Chain(V(228, 2), V(230), V(231), S(22, 2), S(76, 2))
    A, B, Dd = V(20, NL), V(46, NL), V(72, NL)
    f.load_constants()
    for _ in f.sub_plus_p(ch, A, B, Dd):
        pass
    for _ in f.add_mod(ch, A, B):
        pass
    g.s_endpgm()
    fix_hazards(g)
    rnd = random.Random(5)
    a = [rnd.randrange(p) for _ in range(64)]
    b = [rnd.randrange(p) for _ in range(64)]
    a[0], b[0] = 0, 0
    a[1], b[1] = p - 1, p - 1
    a[2], b[2] = 0, p - 1
    a[3], b[3] = p - 1, 0
    a[4], b[4] = 1, p - 1                       # a + b == p exactly
    a[5], b[5] = (p + 1) // 2, (p - 1) // 2     # a + b == p
    w = Wave(g, Memory())
    for l in range(64):
        la, lb = limbs(a[l]), limbs(b[l])
        for i in range(NL):
            w.V[20 + i][l] = la[i]
            w.V[46 + i][l] = lb[i]
    w.run()
    for l in range(64):
        s = unlimbs([int(w.V[20 + i][l]) for i in range(NL)])
        d = unlimbs([int(w.V[72 + i][l]) for i in range(NL)])
        assert all(int(w.V[20 + i][l]) < (1 << 29) and int(w.V[72 + i][l]) < (1 << 29) for i in range(NL))
        assert s == (a[l] + b[l]) % p
        assert d == a[l] - b[l] + p


@pytest.mark.parametrize("k", [8, 7, 6])
def test_first_pass_of_a_wave_against_the_dft(k):
    p = _fr("mnt4753_g1")
    log_n = k + 2 if k < 8 else 10
    N = 1 << log_n
    w = root_of_unity(p, log_n)
    rnd = random.Random(100 + k)
    x = [rnd.randrange(p) for _ in range(N)]
    x[0], x[1] = 0, p - 1
    C = 1 << (8 - k)
    waves = [0, (N >> 8) - 1]
    got = run_pass(p, k, x, w, log_n, 0, False, waves=waves)
    cols = [wv * C + c for wv in waves for c in range(C)]
    exp = model_pass(p, k, x, w, log_n, 0, False, columns=cols)
    assert [i for i in range(N) if exp[i] is not None] == [i for i in range(N) if got[i] is not None]
    assert all(got[i] == exp[i] for i in range(N) if exp[i] is not None)


@pytest.mark.parametrize("k,inverse,fac", [(8, False, None), (7, True, "scalar"), (6, True, "table"), (6, False, "pre")])
def test_later_pass_with_twiddles_and_factors(k, inverse, fac):
    p = _fr("mnt6753_g1")
    log_ns = 3
    log_n = k + log_ns
    N = 1 << log_n
    w = root_of_unity(p, log_n)
    rnd = random.Random(200 + k)
    x = [rnd.randrange(p) for _ in range(N)]
    kw = {}
    if fac == "scalar":
        kw["post_scalar"] = rnd.randrange(1, p)
    elif fac == "table":
        kw["post"] = [rnd.randrange(p) for _ in range(N)]
    elif fac == "pre":
        kw["pre"] = [rnd.randrange(p) for _ in range(N)]
    C = 1 << (8 - k)
    n_waves = N >> 8
    waves = sorted({0, n_waves - 1, n_waves // 2})
    got = run_pass(p, k, x, w, log_n, log_ns, inverse, waves=waves, **kw)
    cols = [wv * C + c for wv in waves for c in range(C)]
    exp = model_pass(p, k, x, w, log_n, log_ns, inverse, columns=cols, **kw)
    assert [i for i in range(N) if exp[i] is not None] == [i for i in range(N) if got[i] is not None]
    assert all(got[i] == exp[i] for i in range(N) if exp[i] is not None)


def test_two_passes_make_the_transform():
    p = _fr("mnt4753_g1")
    log_n = 12
    N = 1 << log_n
    w = root_of_unity(p, log_n)
    rnd = random.Random(7)
    x = [rnd.randrange(p) for _ in range(N)]
    y1 = run_pass(p, 6, x, w, log_n, 0, False)
    y2 = run_pass(p, 6, y1, w, log_n, 6, False)
    for i in rnd.sample(range(N), 24) + [0, 1, N - 1]:
        assert y2[i] == sum(x[t] * pow(w, (i * t) % N, p) for t in range(N)) % p
