"""ginger-lib_amd -- Python binding of libginger_hip.so (the C ABI in include/ginger_hip.h).

This is plumbing for tests, bench.py and the multi-GPU launcher: numpy arrays in the reference's
in-memory formats go straight through ctypes to the HIP library.  The names mirror the reference's
API for this path (algebra/src/msm/variable_base.rs:7-90, algebra/src/fft/domain.rs:24-179):

    VariableBaseMSM.multi_scalar_mul(curve, bases, scalars, infinity=None) -> projective xyz
    EvaluationDomain(field, num_coeffs).fft / ifft / coset_fft / coset_ifft (+ *_in_place)

There is deliberately no fallback: if the shared library is missing or no gfx950 device is
usable every call raises GingerHipError.
"""
import ctypes
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libginger_hip.so")

GH_OK = 0
FFT_INVERSE = 1
FFT_COSET = 2

CURVES = {"mnt4753_g1": 0, "mnt4753_g2": 1, "mnt6753_g1": 2, "mnt6753_g2": 3}
CURVE_DEG = {"mnt4753_g1": 1, "mnt4753_g2": 2, "mnt6753_g1": 1, "mnt6753_g2": 3}
FIELDS = {"mnt4753_fr": 0, "mnt6753_fr": 1}

# every symbol include/ginger_hip.h declares (checked by load_library and by tests/test_abi.py)
ABI_SYMBOLS = [
    "gh_init", "gh_shutdown", "gh_last_error", "gh_device_name",
    "gh_msm", "gh_bases_upload", "gh_bases_upload_wire", "gh_bases_generate_chain", "gh_bases_download", "gh_bases_free", "gh_bases_len", "gh_bases_precompute", "gh_bases_precomputed_window", "gh_bases_precompute_rows", "gh_bases_table_rows",
    "gh_msm_resident", "gh_msm_resident_dev", "gh_msm_resident_dev_batch",
    "gh_msm_cached", "gh_key_cache_config", "gh_key_cache_clear", "gh_key_cache_stats", "gh_bases_content_hash",
    "gh_bases_key_id", "gh_test_hooks",
    "gh_msm_set_window", "gh_msm_set_affine", "gh_msm_set_dedup", "gh_msm_get_window", "gh_msm_last_timing", "gh_msm_batch_timing",
    "gh_domain_supported", "gh_fft", "gh_fft_dev", "gh_vec_mul_dev", "gh_vec_sub_dev", "gh_vec_scale_dev",
    "gh_vec_mul", "gh_vec_scale", "gh_fft_last_kernel_ms", "gh_measure_fpmul_peak", "gh_kernel_resources", "gh_witness_map", "gh_witness_map_dev",
    "gh_sap_witness_map", "gh_sap_witness_map_dev", "gh_batch_inverse", "gh_batch_inverse_dev",
    "gh_lagrange_coefficients", "gh_lagrange_coefficients_dev",
    "gh_dev_alloc", "gh_dev_free", "gh_dev_upload", "gh_dev_download", "gh_dev_sync", "gh_dev_trim",
    "gh_proj_add", "gh_proj_mul", "gh_proj_neg", "gh_proj_to_affine",
    "gh_fixed_base_window", "gh_fixed_base_table", "gh_fixed_base_msm", "gh_fixed_base_msm_affine", "gh_fixed_base_free",
]
# include/ginger_hip_dist.h
DIST_SYMBOLS = ["gh_dist_unique_id", "gh_dist_probe_rccl", "gh_dist_init_rccl", "gh_dist_init_custom", "gh_dist_info", "gh_dist_transport",
                "gh_partials_allgather_fold", "gh_partials_allgather_fold_batch", "gh_dist_shutdown"]
ALLGATHER_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)


class GingerHipError(RuntimeError):
    pass


class MsmTiming(ctypes.Structure):
    _fields_ = [("sort_ms", ctypes.c_float), ("accumulate_ms", ctypes.c_float), ("heavy_ms", ctypes.c_float),
                ("reduce_ms", ctypes.c_float), ("fold_ms", ctypes.c_float), ("total_ms", ctypes.c_float),
                ("window_bits", ctypes.c_int), ("num_windows", ctypes.c_int), ("accumulate_madds", ctypes.c_ulonglong),
                ("heavy_buckets", ctypes.c_uint)]


class KeyCacheStats(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint64) for k in ("entries", "bytes", "hits", "misses", "evictions", "tables_built", "collisions")]


_lib = None


def load_library():
    """dlopen the HIP library and bind every ABI symbol (no device is touched)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GingerHipError("HIP library not built: %s (run `python __graft_entry__.py`)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    missing = [s for s in ABI_SYMBOLS + DIST_SYMBOLS if not hasattr(lib, s)]
    if missing:
        raise GingerHipError("libginger_hip.so lacks ABI symbols: %s" % missing)
    vp, sz, u32, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_int
    lib.gh_last_error.restype = ctypes.c_char_p
    lib.gh_device_name.restype = ctypes.c_char_p
    lib.gh_init.argtypes = [vp, ci]
    lib.gh_msm.argtypes = [ci, vp, vp, sz, vp, sz, vp]
    lib.gh_bases_upload.argtypes = [ci, vp, vp, sz, ctypes.POINTER(vp)]
    lib.gh_bases_upload_wire.argtypes = [ci, vp, sz, ctypes.POINTER(vp)]
    lib.gh_bases_generate_chain.argtypes = [ci, vp, vp, sz, ctypes.POINTER(vp)]
    lib.gh_bases_download.argtypes = [vp, sz, sz, vp]
    lib.gh_bases_free.argtypes = [vp]
    lib.gh_bases_len.argtypes = [vp]
    lib.gh_bases_len.restype = sz
    lib.gh_bases_precompute.argtypes = [vp, ci]
    lib.gh_bases_precomputed_window.argtypes = [vp]
    lib.gh_bases_precompute_rows.argtypes = [vp, ci, ci]
    lib.gh_bases_table_rows.argtypes = [vp]
    lib.gh_msm_resident.argtypes = [vp, vp, sz, vp]
    lib.gh_msm_resident_dev.argtypes = [vp, vp, sz, vp]
    lib.gh_msm_resident_dev_batch.argtypes = [vp, vp, vp, ci, vp]
    lib.gh_msm_cached.argtypes = [ci, vp, vp, sz, vp, sz, vp]
    lib.gh_key_cache_config.argtypes = [sz, ci]
    lib.gh_key_cache_stats.argtypes = [ctypes.POINTER(KeyCacheStats)]
    lib.gh_bases_content_hash.argtypes = [ci, vp, vp, sz, ctypes.POINTER(ctypes.c_uint64)]
    lib.gh_bases_content_hash.restype = ctypes.c_uint64
    lib.gh_measure_fpmul_peak.argtypes = [ctypes.POINTER(ctypes.c_double)]
    lib.gh_kernel_resources.argtypes = [ctypes.c_char_p, ctypes.POINTER(u32), ctypes.POINTER(u32), ctypes.POINTER(u32)]
    lib.gh_bases_key_id.argtypes = [ci, vp, vp, sz, ctypes.POINTER(ctypes.c_uint64)]
    lib.gh_test_hooks.argtypes = [ci]
    lib.gh_msm_set_window.argtypes = [ci]
    lib.gh_msm_set_affine.argtypes = [ci]
    lib.gh_msm_set_dedup.argtypes = [ci]
    lib.gh_msm_get_window.argtypes = [ci, sz]
    lib.gh_msm_last_timing.argtypes = [ctypes.POINTER(MsmTiming)]
    lib.gh_msm_batch_timing.argtypes = [ci, ctypes.POINTER(MsmTiming)]
    lib.gh_domain_supported.argtypes = [ci, sz, ctypes.POINTER(u32)]
    lib.gh_fft.argtypes = [ci, vp, sz, vp, u32, u32]
    lib.gh_fft_dev.argtypes = [ci, vp, u32, u32]
    lib.gh_vec_mul_dev.argtypes = [ci, vp, vp, sz]
    lib.gh_vec_sub_dev.argtypes = [ci, vp, vp, sz]
    lib.gh_vec_scale_dev.argtypes = [ci, vp, vp, sz]
    lib.gh_vec_mul.argtypes = [ci, vp, vp, sz]
    lib.gh_vec_scale.argtypes = [ci, vp, vp, sz]
    lib.gh_fft_last_kernel_ms.argtypes = [ctypes.POINTER(ctypes.c_float)]
    lib.gh_witness_map.argtypes = [ci, vp, vp, vp, u32, vp, vp, vp, vp]
    lib.gh_witness_map_dev.argtypes = [ci, vp, vp, vp, u32, vp, vp, vp, vp]
    lib.gh_sap_witness_map.argtypes = [ci, vp, vp, u32, vp, vp, vp]
    lib.gh_sap_witness_map_dev.argtypes = [ci, vp, vp, u32, vp, vp, vp]
    lib.gh_batch_inverse.argtypes = [ci, vp, sz]
    lib.gh_batch_inverse_dev.argtypes = [ci, vp, sz]
    lib.gh_lagrange_coefficients.argtypes = [ci, u32, vp, vp]
    lib.gh_lagrange_coefficients_dev.argtypes = [ci, u32, vp, vp]
    lib.gh_dev_alloc.argtypes = [ctypes.POINTER(vp), sz]
    lib.gh_dev_free.argtypes = [vp]
    lib.gh_dev_upload.argtypes = [vp, vp, sz]
    lib.gh_dev_download.argtypes = [vp, vp, sz]
    lib.gh_proj_add.argtypes = [ci, vp, vp]
    lib.gh_proj_mul.argtypes = [ci, vp, vp, vp]
    lib.gh_proj_neg.argtypes = [ci, vp]
    lib.gh_proj_to_affine.argtypes = [ci, vp, vp, vp]
    lib.gh_fixed_base_window.argtypes = [sz]
    lib.gh_fixed_base_table.argtypes = [ci, vp, sz, ci, ctypes.POINTER(vp)]
    lib.gh_fixed_base_msm.argtypes = [vp, vp, sz, vp]
    lib.gh_fixed_base_msm_affine.argtypes = [vp, vp, sz, vp, vp, ci]
    lib.gh_fixed_base_free.argtypes = [vp]
    lib.gh_dist_unique_id.argtypes = [vp]
    lib.gh_dist_init_rccl.argtypes = [vp, ci, ci]
    lib.gh_dist_init_custom.argtypes = [ALLGATHER_FN, vp, ci, ci]
    lib.gh_dist_info.argtypes = [ctypes.POINTER(ci), ctypes.POINTER(ci)]
    lib.gh_partials_allgather_fold.argtypes = [ci, vp, vp, ctypes.POINTER(ctypes.c_double)]
    lib.gh_partials_allgather_fold_batch.argtypes = [ci, vp, sz, vp, ctypes.POINTER(ctypes.c_double)]
    lib.gh_dist_transport.argtypes = [ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.c_char_p, sz]
    _lib = lib
    return lib


def _check(rc):
    if rc != GH_OK:
        raise GingerHipError("ginger_hip error %d: %s" % (rc, load_library().gh_last_error().decode()))


def _u64(a, words=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if words is not None and a.size % words:
        raise ValueError("array length is not a multiple of %d u64" % words)
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None and a.size else None


def init(device=None):
    lib = load_library()
    if device is None:
        _check(lib.gh_init(None, 0))
    else:
        arr = (ctypes.c_int * 1)(int(device))
        _check(lib.gh_init(arr, 1))


def shutdown():
    _check(load_library().gh_shutdown())


def device_name():
    return load_library().gh_device_name().decode()


# ------------------------------------------------------------------------------ MSM
class DeviceBuffer:
    """Plain device allocation obtained through the C ABI (gh_dev_alloc)."""

    def __init__(self, nbytes):
        self.ptr = ctypes.c_void_p()
        self.nbytes = int(nbytes)
        _check(load_library().gh_dev_alloc(ctypes.byref(self.ptr), self.nbytes))

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(load_library().gh_dev_upload(self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, dtype=np.uint64):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        _check(load_library().gh_dev_download(_ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            load_library().gh_dev_free(self.ptr)
            self.ptr = ctypes.c_void_p()


def msm_cached(curve, bases, scalars, infinity=None):
    """VariableBaseMSM::multi_scalar_mul for an unchanged caller (gh_msm_cached): a pure function of (bases, scalars) whose
    bases stay resident between calls, identified by a hash over all of their limbs -- never by address."""
    deg = CURVE_DEG[curve]
    bases = _u64(bases, 24 * deg)
    scalars = _u64(scalars, 12)
    inf = None if infinity is None else np.ascontiguousarray(infinity, dtype=np.uint8)
    out = np.zeros(36 * deg, dtype=np.uint64)
    _check(load_library().gh_msm_cached(CURVES[curve], _ptr(bases), _ptr(inf) if inf is not None else None, bases.size // (24 * deg),
                                        _ptr(scalars), scalars.size // 12, _ptr(out)))
    return out


def key_cache_config(max_bytes=None, table_after=2):
    """max_bytes None = GH_KEY_CACHE_AUTO: half of the device memory free at the next msm_cached call"""
    _check(load_library().gh_key_cache_config((1 << 64) - 1 if max_bytes is None else int(max_bytes), int(table_after)))


def key_cache_clear():
    _check(load_library().gh_key_cache_clear())


def key_cache_stats():
    st = KeyCacheStats()
    _check(load_library().gh_key_cache_stats(ctypes.byref(st)))
    return {k: int(getattr(st, k)) for k, _ in KeyCacheStats._fields_}


def bases_content_hash(curve, bases, infinity=None):
    """(lo, hi) of the 128-bit hash gh_msm_cached keys on; host-only"""
    deg = CURVE_DEG[curve]
    bases = _u64(bases, 24 * deg)
    inf = None if infinity is None else np.ascontiguousarray(infinity, dtype=np.uint8)
    hi = ctypes.c_uint64(0)
    lo = load_library().gh_bases_content_hash(CURVES[curve], _ptr(bases), _ptr(inf) if inf is not None else None, bases.size // (24 * deg), ctypes.byref(hi))
    return int(lo), int(hi.value)


def measure_fpmul_peak():
    """753-bit Montgomery products per second of this card, measured now (gh_measure_fpmul_peak)"""
    v = ctypes.c_double(0)
    _check(load_library().gh_measure_fpmul_peak(ctypes.byref(v)))
    return float(v.value)


def kernel_resources(which):
    """scratch bytes per lane, registers and LDS bytes of a generated kernel, from the loaded code object"""
    a, b, c = ctypes.c_uint32(0), ctypes.c_uint32(0), ctypes.c_uint32(0)
    _check(load_library().gh_kernel_resources(which.encode(), ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    return {"scratch_bytes_per_lane": int(a.value), "registers": int(b.value), "lds_bytes": int(c.value)}


def bases_key_id(curve, bases, infinity=None):
    """the four 64-bit lanes of a key's identity (gh_bases_key_id): [0:2] select the cache entry, [2:4] verify it; host-only"""
    deg = CURVE_DEG[curve]
    bases = _u64(bases, 24 * deg)
    inf = None if infinity is None else np.ascontiguousarray(infinity, dtype=np.uint8)
    out = (ctypes.c_uint64 * 4)()
    _check(load_library().gh_bases_key_id(CURVES[curve], _ptr(bases), _ptr(inf) if inf is not None else None, bases.size // (24 * deg), out))
    return tuple(int(x) for x in out)


def set_test_hooks(flags):
    """fault injection for the test-suite (gh_test_hooks): bit 0 = every key identity has constant selection lanes"""
    _check(load_library().gh_test_hooks(int(flags)))


class ResidentBases:
    """Device-resident MSM bases (the proving-key queries are static per circuit)."""

    def __init__(self, curve, bases, infinity=None):
        self.curve = curve
        deg = CURVE_DEG[curve]
        bases = _u64(bases, 24 * deg)
        n = bases.size // (24 * deg)
        inf = None
        if infinity is not None:
            inf = np.ascontiguousarray(infinity, dtype=np.uint8)
            assert inf.size == n
        self.handle = ctypes.c_void_p()
        _check(load_library().gh_bases_upload(CURVES[curve], _ptr(bases), _ptr(inf) if inf is not None else None, n,
                                               ctypes.byref(self.handle)))
        self.n = n

    @classmethod
    def from_wire(cls, curve, data):
        """data: bytes of n serialised affine points (GroupAffine::write: x || y || infinity byte, canonical LE)."""
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        rec = 192 * CURVE_DEG[curve] + 1
        if buf.size % rec:
            raise ValueError("wire data is not a multiple of %d bytes" % rec)
        self = cls.__new__(cls)
        self.curve = curve
        self.n = buf.size // rec
        self.handle = ctypes.c_void_p()
        _check(load_library().gh_bases_upload_wire(CURVES[curve], _ptr(buf), self.n, ctypes.byref(self.handle)))
        return self

    @classmethod
    def chain(cls, curve, p0_xy, step_xy, n):
        """n distinct resident bases P_0 + i * H generated on the device (synthetic key: gh_bases_generate_chain)."""
        self = cls.__new__(cls)
        self.curve, self.n = curve, int(n)
        self.handle = ctypes.c_void_p()
        p0, st = _u64(p0_xy), _u64(step_xy)
        _check(load_library().gh_bases_generate_chain(CURVES[curve], _ptr(p0), _ptr(st), self.n, ctypes.byref(self.handle)))
        return self

    def download(self, first, count):
        """count x 24*deg u64: the resident bases [first, first + count) as Montgomery x || y rows"""
        out = np.zeros((int(count), 24 * CURVE_DEG[self.curve]), dtype=np.uint64)
        _check(load_library().gh_bases_download(self.handle, int(first), int(count), _ptr(out)))
        return out

    def precompute(self, window_bits=0, max_rows=0):
        """Build the per-key shift table (gh_bases_precompute / gh_bases_precompute_rows: at most max_rows rows, a partial table
        with several bucket sets); returns the window size used."""
        _check(load_library().gh_bases_precompute_rows(self.handle, int(window_bits), int(max_rows)))
        return load_library().gh_bases_precomputed_window(self.handle)

    def table_rows(self):
        return load_library().gh_bases_table_rows(self.handle)

    def msm(self, scalars):
        scalars = _u64(scalars, 12)
        out = np.zeros(36 * CURVE_DEG[self.curve], dtype=np.uint64)
        _check(load_library().gh_msm_resident(self.handle, _ptr(scalars), scalars.size // 12, _ptr(out)))
        return out

    def msm_dev(self, d_scalars, n_scalars):
        out = np.zeros(36 * CURVE_DEG[self.curve], dtype=np.uint64)
        _check(load_library().gh_msm_resident_dev(self.handle, d_scalars.ptr, n_scalars, _ptr(out)))
        return out

    def free(self):
        if self.handle:
            load_library().gh_bases_free(self.handle)
            self.handle = ctypes.c_void_p()


def msm_batch_dev(jobs):
    """jobs: list of (ResidentBases, DeviceBuffer of scalars, n_scalars) on one curve -> list of projective sums.
    One gh_msm_resident_dev_batch call: the MSMs are pipelined over HIP streams."""
    if not jobs:
        return []
    k = len(jobs)
    deg = CURVE_DEG[jobs[0][0].curve]
    handles = (ctypes.c_void_p * k)(*[j[0].handle for j in jobs])
    scal = (ctypes.c_void_p * k)(*[j[1].ptr for j in jobs])
    ns = (ctypes.c_size_t * k)(*[int(j[2]) for j in jobs])
    out = np.zeros(k * 36 * deg, dtype=np.uint64)
    _check(load_library().gh_msm_resident_dev_batch(handles, scal, ns, k, _ptr(out)))
    return [out[i * 36 * deg:(i + 1) * 36 * deg].copy() for i in range(k)]


class VariableBaseMSM:
    """Mirror of algebra::msm::VariableBaseMSM (variable_base.rs:7-90)."""

    @staticmethod
    def multi_scalar_mul(curve, bases, scalars, infinity=None):
        """bases: n x 24*deg u64 (Montgomery x||y); scalars: m x 12 u64 canonical;
        returns the projective sum as 36*deg u64 (X||Y||Z, Montgomery; Z == 0 <=> infinity)."""
        deg = CURVE_DEG[curve]
        bases = _u64(bases, 24 * deg)
        scalars = _u64(scalars, 12)
        n_b, n_s = bases.size // (24 * deg), scalars.size // 12
        inf = None
        if infinity is not None:
            inf = np.ascontiguousarray(infinity, dtype=np.uint8)
            assert inf.size == n_b
        out = np.zeros(36 * deg, dtype=np.uint64)
        _check(load_library().gh_msm(CURVES[curve], _ptr(bases), _ptr(inf) if inf is not None else None, n_b,
                                      _ptr(scalars), n_s, _ptr(out)))
        return out


class FixedBaseMSM:
    """Mirror of algebra::msm::FixedBaseMSM (fixed_base.rs:7-79): get_mul_window_size / get_window_table /
    multi_scalar_mul for one base g; the window table lives on the device."""

    @staticmethod
    def get_mul_window_size(num_scalars):
        return load_library().gh_fixed_base_window(int(num_scalars))

    def __init__(self, curve, g_xyz, scalar_size=753, window=None, num_scalars=0):
        self.curve = curve
        self.window = int(window) if window else self.get_mul_window_size(num_scalars)
        g = _u64(g_xyz)
        assert g.size == 36 * CURVE_DEG[curve]
        self.handle = ctypes.c_void_p()
        _check(load_library().gh_fixed_base_table(CURVES[curve], _ptr(g), int(scalar_size), self.window, ctypes.byref(self.handle)))

    def multi_scalar_mul(self, scalars):
        """scalars: n x 12 u64 canonical integers (into_repr); returns n x 36*deg u64 projective points"""
        s = _u64(scalars, 12)
        n = s.size // 12
        out = np.zeros((n, 36 * CURVE_DEG[self.curve]), dtype=np.uint64)
        _check(load_library().gh_fixed_base_msm(self.handle, _ptr(s), n, _ptr(out)))
        return out

    def multi_scalar_mul_affine(self, scalars, canonical=False):
        """multi_scalar_mul followed by batch_normalization + into_affine on the device (generator.rs:247-335):
        (n x 24*deg u64 x || y rows, n infinity flags); canonical=True gives the integers GroupAffine::write serialises"""
        s = _u64(scalars, 12)
        n = s.size // 12
        xy = np.zeros((n, 24 * CURVE_DEG[self.curve]), dtype=np.uint64)
        inf = np.zeros(n, dtype=np.uint8)
        _check(load_library().gh_fixed_base_msm_affine(self.handle, _ptr(s), n, _ptr(xy), _ptr(inf), 1 if canonical else 0))
        return xy, inf

    def free(self):
        if self.handle:
            load_library().gh_fixed_base_free(self.handle)
            self.handle = ctypes.c_void_p()


def msm_set_window(c):
    _check(load_library().gh_msm_set_window(int(c)))


def msm_set_affine(mode):
    """0: projective bucket sums, 1: affine rounds always, 2: automatic (default)."""
    _check(load_library().gh_msm_set_affine(int(mode)))


def msm_set_dedup(on):
    """gh_msm_set_dedup: add up the scalars of a key's equal bases before its MSMs (tables built from now on); default on"""
    _check(load_library().gh_msm_set_dedup(1 if on else 0))


def dev_trim():
    _check(load_library().gh_dev_trim())


def msm_last_timing():
    t = MsmTiming()
    _check(load_library().gh_msm_last_timing(ctypes.byref(t)))
    return {k: getattr(t, k) for k, _ in MsmTiming._fields_}


def msm_batch_timing(index):
    t = MsmTiming()
    _check(load_library().gh_msm_batch_timing(int(index), ctypes.byref(t)))
    return {k: getattr(t, k) for k, _ in MsmTiming._fields_}


def proj_add(curve, acc, p):
    acc = _u64(acc).copy()
    p = _u64(p)
    _check(load_library().gh_proj_add(CURVES[curve], _ptr(acc), _ptr(p)))
    return acc


def proj_mul(curve, xyz, scalar12):
    """scalar * point for one projective point (host side): GroupProjective::mul_assign."""
    xyz = _u64(xyz)
    k = _u64(scalar12, 12)
    out = np.zeros_like(xyz)
    _check(load_library().gh_proj_mul(CURVES[curve], _ptr(xyz), _ptr(k), _ptr(out)))
    return out


def proj_neg(curve, xyz):
    out = _u64(xyz).copy()
    _check(load_library().gh_proj_neg(CURVES[curve], _ptr(out)))
    return out


_FIELD_ONE = {}


def field_one(curve):
    """Montgomery one of the curve's coordinate field (12*deg u64): the Y of the canonical zero (0, 1, 0)."""
    if curve not in _FIELD_ONE:
        deg = CURVE_DEG[curve]
        z = proj_mul(curve, np.zeros(36 * deg, dtype=np.uint64), np.zeros(12, dtype=np.uint64))
        _FIELD_ONE[curve] = z[12 * deg:24 * deg].copy()
    return _FIELD_ONE[curve]


def proj_to_affine(curve, xyz):
    """-> (xy as 24*deg u64 Montgomery, is_infinity) : the reference's into_affine()."""
    deg = CURVE_DEG[curve]
    xyz = _u64(xyz)
    out = np.zeros(24 * deg, dtype=np.uint64)
    inf = np.zeros(1, dtype=np.uint8)
    _check(load_library().gh_proj_to_affine(CURVES[curve], _ptr(xyz), _ptr(out), _ptr(inf)))
    return out, bool(inf[0])


# ------------------------------------------------------------------------------ FFT
class EvaluationDomain:
    """Mirror of algebra::fft::EvaluationDomain (domain.rs:24-179) for MNT4-753 Fr / MNT6-753 Fr.
    `EvaluationDomain.new(field, n)` returns None where the reference's `new` does (domain.rs:69-71)."""

    def __init__(self, field, num_coeffs):
        lg = ctypes.c_uint32()
        ok = load_library().gh_domain_supported(FIELDS[field], int(num_coeffs), ctypes.byref(lg))
        if not ok:
            raise GingerHipError("domain of 2^%d exceeds the 2-adicity of %s" % (lg.value, field))
        self.field = field
        self.log_size_of_group = lg.value
        self.size = 1 << lg.value

    @classmethod
    def new(cls, field, num_coeffs):
        try:
            return cls(field, num_coeffs)
        except GingerHipError:
            return None

    def _run(self, a, flags):
        a = _u64(a, 12)
        out = np.empty(self.size * 12, dtype=np.uint64)
        _check(load_library().gh_fft(FIELDS[self.field], _ptr(a), a.size // 12, _ptr(out), self.log_size_of_group, flags))
        return out

    def fft(self, coeffs):
        return self._run(coeffs, 0)

    def ifft(self, evals):
        return self._run(evals, FFT_INVERSE)

    def coset_fft(self, coeffs):
        return self._run(coeffs, FFT_COSET)

    def coset_ifft(self, evals):
        return self._run(evals, FFT_INVERSE | FFT_COSET)

    def fft_dev(self, dbuf, flags=0):
        _check(load_library().gh_fft_dev(FIELDS[self.field], dbuf.ptr, self.log_size_of_group, flags))

    def mul_polynomials_in_evaluation_domain(self, a, b):
        a = _u64(a, 12).copy()
        b = _u64(b, 12)
        assert a.size == b.size
        _check(load_library().gh_vec_mul(FIELDS[self.field], _ptr(a), _ptr(b), a.size // 12))
        return a


def witness_map(field, a, b, c, d1, d2, d3):
    """Transform part of R1CStoQAP::witness_map (r1cs_to_qap.rs:121-166): a, b, c are the 2^k
    evaluations of the A, B, C rows (Montgomery, 12 u64 each); returns the 2^k + 1 coefficients of h."""
    a, b, c = _u64(a, 12), _u64(b, 12), _u64(c, 12)
    n = a.size // 12
    assert n and (n & (n - 1)) == 0 and b.size == a.size and c.size == a.size
    log_n = n.bit_length() - 1
    d1, d2, d3 = _u64(d1, 12), _u64(d2, 12), _u64(d3, 12)
    h = np.empty((n + 1) * 12, dtype=np.uint64)
    _check(load_library().gh_witness_map(FIELDS[field], _ptr(a), _ptr(b), _ptr(c), log_n, _ptr(d1), _ptr(d2), _ptr(d3), _ptr(h)))
    return h


def sap_witness_map(field, a, c, d1, d2):
    """Transform part of R1CStoSAP::witness_map (gm17/r1cs_to_sap.rs:194-240): a, c are the 2^k evaluations; returns h (2^k + 1)."""
    a, c = _u64(a, 12), _u64(c, 12)
    n = a.size // 12
    assert n and (n & (n - 1)) == 0 and c.size == a.size
    d1, d2 = _u64(d1, 12), _u64(d2, 12)
    h = np.empty((n + 1) * 12, dtype=np.uint64)
    _check(load_library().gh_sap_witness_map(FIELDS[field], _ptr(a), _ptr(c), n.bit_length() - 1, _ptr(d1), _ptr(d2), _ptr(h)))
    return h


def batch_inversion(field, a):
    """algebra::fields::batch_inversion (fields/mod.rs:412-442) on Montgomery rows; zeros stay zero."""
    a = _u64(a, 12).copy()
    _check(load_library().gh_batch_inverse(FIELDS[field], _ptr(a), a.size // 12))
    return a


def evaluate_all_lagrange_coefficients(field, log_n, tau12):
    """EvaluationDomain::evaluate_all_lagrange_coefficients (domain.rs:183-219) -> 2^log_n Montgomery rows"""
    tau = _u64(tau12, 12)
    out = np.empty((1 << log_n) * 12, dtype=np.uint64)
    _check(load_library().gh_lagrange_coefficients(FIELDS[field], int(log_n), _ptr(tau), _ptr(out)))
    return out.reshape(-1, 12)


def vec_scale(field, a, scalar12):
    a = _u64(a, 12).copy()
    s = _u64(scalar12, 12)
    _check(load_library().gh_vec_scale(FIELDS[field], _ptr(a), _ptr(s), a.size // 12))
    return a


def fft_last_kernel_ms():
    ms = ctypes.c_float()
    _check(load_library().gh_fft_last_kernel_ms(ctypes.byref(ms)))
    return ms.value
