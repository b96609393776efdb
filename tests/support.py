"""Shared test helpers: ctypes binding of the CPU oracle (oracle/liboracle.so), (de)serialisation
between Python ints and the ABI layouts, deterministic input generation.  Test infrastructure."""
import ctypes
import os
import subprocess

import numpy as np

import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
CURVE_ID = {"mnt4753_g1": 0, "mnt4753_g2": 1, "mnt6753_g1": 2, "mnt6753_g2": 3}
FIELD_ID = {"mnt4753_fr": 0, "mnt6753_fr": 1}
FIELD_OF = {"mnt4753_fr": pyref.P6, "mnt6753_fr": pyref.P4}

_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
        lib = ctypes.CDLL(ORACLE_SO)
        vp, sz, ci, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint32
        lib.oracle_fp_op.argtypes = [ci, ci, vp, vp, vp]
        lib.oracle_ext_op.argtypes = [ci, ci, vp, vp, vp]
        lib.oracle_ec_op.argtypes = [ci, ci, vp, vp, ci, vp]
        lib.oracle_msm.argtypes = [ci, vp, vp, sz, vp, sz, vp, ci]
        lib.oracle_fft.argtypes = [ci, vp, sz, u32, u32, ci]
        lib.oracle_fft_variant.argtypes = [ci, vp, u32, ci]
        lib.oracle_domain.argtypes = [ci, sz, vp, ctypes.POINTER(u32)]
        lib.oracle_vec_mul.argtypes = [ci, vp, vp, sz]
        lib.oracle_vanishing_inv_on_coset.argtypes = [ci, u32, vp]
        lib.oracle_witness_map.argtypes = [ci, vp, vp, vp, u32, vp, vp, vp, vp, ci]
        lib.oracle_batch_inversion.argtypes = [ci, vp, sz]
        lib.oracle_lagrange.argtypes = [ci, u32, vp, vp]
        lib.oracle_sap_witness_map.argtypes = [ci, vp, vp, u32, vp, vp, vp, ci]
        lib.oracle_fixed_base_msm.argtypes = [ci, vp, sz, sz, vp, sz, vp, ci]
        _oracle = lib
    return _oracle


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None


def u64(x, n=12):
    return np.array(pyref.int_to_limbs(x, n), dtype=np.uint64)


def to_int(arr):
    return pyref.limbs_to_int([int(v) for v in arr])


# ---- field element arrays (Montgomery, ABI layout)
def fe_array(F, vals):
    out = np.zeros((len(vals), 12), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = pyref.int_to_limbs(F.to_mont(v))
    return out


def fe_list(F, arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 12)
    return [F.from_mont(pyref.limbs_to_int([int(v) for v in r])) for r in arr]


def scalar_array(vals):
    out = np.zeros((len(vals), 12), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = pyref.int_to_limbs(v)
    return out


def bases_array(C, pts):
    k = C.deg
    out = np.zeros((len(pts), 24 * k), dtype=np.uint64)
    inf = np.zeros(len(pts), dtype=np.uint8)
    for i, P in enumerate(pts):
        if P is None:
            inf[i] = 1
            continue
        out[i, :12 * k] = pyref.ext_to_abi(C.F, P[0])
        out[i, 12 * k:] = pyref.ext_to_abi(C.F, P[1])
    return out, inf


def proj_array(C, P, z=None):
    """affine point (or None) -> projective ABI array with an arbitrary non-trivial Z"""
    k = C.deg
    E = C.E
    if P is None:
        X, Y, Z = E.zero(), E.one(), E.zero()
    else:
        z = z or tuple((7 + 3 * i) for i in range(k))
        X, Y, Z = E.mul(P[0], z), E.mul(P[1], z), z
    return np.array(pyref.ext_to_abi(C.F, X) + pyref.ext_to_abi(C.F, Y) + pyref.ext_to_abi(C.F, Z), dtype=np.uint64)


def affine_of_xyz(C, xyz):
    k = C.deg
    v = [int(x) for x in np.asarray(xyz).reshape(-1)]
    X = pyref.ext_from_abi(C.F, v[0:12 * k], k)
    Y = pyref.ext_from_abi(C.F, v[12 * k:24 * k], k)
    Z = pyref.ext_from_abi(C.F, v[24 * k:36 * k], k)
    return C.proj_to_affine(X, Y, Z)


def chain_points(C, n, rng):
    """n distinct curve points P_0 + i H (cheap: one affine addition each)."""
    H = C.mul(rng.next_u64() | 1, C.G)
    P = C.mul(rng.next_u64() | 1, C.G)
    pts = []
    for _ in range(n):
        pts.append(P)
        P = C.add(P, H)
    return pts


def random_scalars_np(n, modulus_bits_top=49, seed=1, below=None):
    """n uniform 12-limb integers with the top limb masked to 49 bits (algebra/src/fields/macros.rs:11-28);
    values >= `below` (if given) are resampled -- vectorised for the full-size GPU tests."""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 1 << 64, size=(n, 12), dtype=np.uint64)
    s[:, 11] &= np.uint64((1 << modulus_bits_top) - 1)
    if below is not None:
        top = np.uint64(below >> 704)
        bad = s[:, 11] >= top      # conservative: force the top limb strictly below the modulus' top limb
        while bad.any():
            s[bad, 11] = rng.integers(0, int(top), size=int(bad.sum()), dtype=np.uint64)
            bad = s[:, 11] >= top
    return s


# ---- oracle wrappers
def oracle_msm(curve, bases, inf, scalars, threads=8):
    C = pyref.CURVES[curve]
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    out = np.zeros(36 * C.deg, dtype=np.uint64)
    n_b = bases.size // (24 * C.deg)
    n_s = scalars.size // 12
    infp = None
    if inf is not None:
        inf = np.ascontiguousarray(inf, dtype=np.uint8)
        infp = ptr(inf)
    rc = oracle().oracle_msm(CURVE_ID[curve], ptr(bases), infp, n_b, ptr(scalars), n_s, ptr(out), threads)
    assert rc == 0
    return out


def oracle_affine(curve, xyz):
    C = pyref.CURVES[curve]
    xyz = np.ascontiguousarray(xyz, dtype=np.uint64)
    out = np.zeros(24 * C.deg, dtype=np.uint64)
    inf = oracle().oracle_ec_op(CURVE_ID[curve], 4, ptr(xyz), None, 0, ptr(out))
    return out, bool(inf)


def oracle_fft(field, a, log_n, flags, threads=8):
    """same contract as gh_fft: pad/truncate to 2^log_n, transform, return N x 12 u64"""
    n = 1 << log_n
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 12)
    n_in = min(len(a), n)
    buf = np.zeros((n, 12), dtype=np.uint64)
    buf[:n_in] = a[:n_in]
    rc = oracle().oracle_fft(FIELD_ID[field], ptr(buf), n_in, log_n, flags, threads)
    assert rc == 0
    return buf


def oracle_witness_map(field, a, b, c, d1, d2, d3, threads=8):
    """restated R1CStoQAP::witness_map (r1cs_to_qap.rs:121-166) from evaluated rows -> h (N + 1 elements)"""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 12)
    n = len(a)
    log_n = n.bit_length() - 1
    b = np.ascontiguousarray(b, dtype=np.uint64)
    c = np.ascontiguousarray(c, dtype=np.uint64)
    h = np.zeros((n + 1, 12), dtype=np.uint64)
    d1, d2, d3 = (np.ascontiguousarray(x, dtype=np.uint64) for x in (d1, d2, d3))
    rc = oracle().oracle_witness_map(FIELD_ID[field], ptr(a), ptr(b), ptr(c), log_n, ptr(d1), ptr(d2), ptr(d3), ptr(h), threads)
    assert rc == 0
    return h


def oracle_batch_inversion(field, a):
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 12).copy()
    assert oracle().oracle_batch_inversion(FIELD_ID[field], ptr(a), len(a)) == 0
    return a


def oracle_lagrange(field, log_n, tau12):
    out = np.zeros((1 << log_n, 12), dtype=np.uint64)
    tau = np.ascontiguousarray(tau12, dtype=np.uint64)
    assert oracle().oracle_lagrange(FIELD_ID[field], log_n, ptr(tau), ptr(out)) == 0
    return out


def oracle_sap_witness_map(field, a, c, d1, d2, threads=8):
    """restated R1CStoSAP::witness_map (gm17/r1cs_to_sap.rs:191-240) from evaluated rows -> h (N + 1 elements)"""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 12)
    n = len(a)
    c = np.ascontiguousarray(c, dtype=np.uint64)
    h = np.zeros((n + 1, 12), dtype=np.uint64)
    d1, d2 = (np.ascontiguousarray(x, dtype=np.uint64) for x in (d1, d2))
    assert oracle().oracle_sap_witness_map(FIELD_ID[field], ptr(a), ptr(c), n.bit_length() - 1, ptr(d1), ptr(d2), ptr(h), threads) == 0
    return h


# ---- closed form of an MSM over a chain key (first principles: independent of every MSM code path)
def chain_sums(scalars, modulus, chunk=1 << 18):
    """(sum_i s_i mod r, sum_i i * s_i mod r) for n x 12 u64 scalars, exactly, with numpy: the limbs are cut into 16-bit
    pieces and the index into 12-bit pieces, so that every dot product of a chunk (<= 2^18 rows) stays below
    2^12 * 2^16 * 2^18 = 2^46 and is exact in float64 (BLAS); the chunk results are added as Python integers."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 12)
    n = len(s)
    assert chunk <= 1 << 18 and n < 1 << 36
    s0 = 0
    s1 = 0
    weights = [1 << (16 * j) for j in range(48)]
    buf = np.empty((min(n, chunk), 48), dtype=np.float64)             # reused: a fresh 100 MB array per chunk costs more than the math
    for lo in range(0, n, chunk):
        blk = s[lo:lo + chunk]
        m = len(blk)
        pieces = buf[:m]
        np.copyto(pieces, blk.view(np.uint16))                        # m x 48, little endian: piece j has weight 2^(16 j)
        idx = np.arange(lo, lo + m, dtype=np.uint64)
        rows = [np.ones(m, dtype=np.float64)]
        shifts = []
        for part in range(0, 36, 12):                                 # the index in 12-bit pieces
            ip = (idx >> np.uint64(part)) & np.uint64(0xFFF)
            if ip.any():
                rows.append(ip.astype(np.float64))
                shifts.append(part)
        prod = pieces.T @ np.stack(rows).T                            # 48 x (1 + parts), every entry an exact integer < 2^46
        s0 += sum(int(prod[j, 0]) * weights[j] for j in range(48))
        for k, part in enumerate(shifts):
            s1 += sum(int(prod[j, 1 + k]) * weights[j] for j in range(48)) << part
    return s0 % modulus, s1 % modulus


def chain_msm_closed_form(C, P0, H, scalars):
    """sum_i s_i (P0 + i H) = (sum s_i) P0 + (sum i s_i) H, as an affine point (or None) -- two scalar multiplications with
    Python integers (pyref), the reference's own MSM test idea (variable_base.rs:102-151: Pippenger == naive sum)."""
    a, b = chain_sums(scalars, C.order)
    return C.add(C.mul(a, P0) if a else None, C.mul(b, H) if b else None)


def affine_abi_of_point(C, P):
    """(xy words, infinity flag) in the form gh_proj_to_affine returns, for an affine pyref point"""
    if P is None:
        xy, _ = bases_array(C, [None])
        k = C.deg
        one = pyref.ext_to_abi(C.F, C.E.one())
        out = np.zeros(24 * k, dtype=np.uint64)
        out[12 * k:] = one
        return out, True
    xy, _ = bases_array(C, [P])
    return xy[0], False
