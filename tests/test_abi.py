"""The C-ABI library loads and exports every symbol include/ginger_hip.h declares; without a GPU
every compute entry point fails loudly (no CPU fallback).  No compute calls here."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(name="ginger_hip.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gh_\w+)\s*\(", src)))


def test_every_declared_symbol_is_exported(gl):
    lib = gl.load_library()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(gl.ABI_SYMBOLS) == names
    dist = header_functions("ginger_hip_dist.h")
    assert len(dist) == 9 and sorted(gl.DIST_SYMBOLS) == dist
    for n in dist:
        assert hasattr(lib, n), n


def test_library_has_no_oracle_dependency():
    # the product must not link or embed the checker
    import subprocess
    so = os.path.join(ROOT, "ginger-lib_amd", "libginger_hip.so")
    out = subprocess.run(["ldd", so], stdout=subprocess.PIPE, text=True).stdout
    assert "oracle" not in out
    for root, _, files in os.walk(os.path.join(ROOT, "ginger-lib_amd")):
        for f in files:
            if f.endswith((".h", ".hip", ".py", ".hpp", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle/" not in txt and "liboracle" not in txt and "import pyref" not in txt, f


def test_host_only_entry_points_work_without_gpu(gl):
    # gh_domain_supported mirrors EvaluationDomain::new's None condition (domain.rs:65-72)
    lib = gl.load_library()
    lg = ctypes.c_uint32()
    assert lib.gh_domain_supported(0, 1 << 20, ctypes.byref(lg)) == 1 and lg.value == 20
    assert lib.gh_domain_supported(0, (1 << 20) + 1, ctypes.byref(lg)) == 1 and lg.value == 21
    assert lib.gh_domain_supported(0, 1 << 30, ctypes.byref(lg)) == 0
    assert lib.gh_domain_supported(1, 1 << 14, ctypes.byref(lg)) == 1
    assert lib.gh_domain_supported(1, 1 << 15, ctypes.byref(lg)) == 0
    assert lib.gh_domain_supported(0, 0, ctypes.byref(lg)) == 1 and lg.value == 0


def test_key_cache_identifies_bases_by_content(gl):
    """gh_msm_cached (the drop-in for multi_scalar_mul, a pure function: variable_base.rs:85-90) keys its resident copies
    on a hash over ALL limbs: two different base sets passed through ONE buffer -- same address, same length, same first
    and last rows, only an interior limb differs -- are two keys; equal content at another address is the same key.
    Host-only part (the hash and the bookkeeping entry points need no device)."""
    rng = np.random.default_rng(5)
    for curve, deg in (("mnt4753_g1", 1), ("mnt6753_g2", 3)):
        n = 70000                                   # several hash chunks (2^16 words each), hashed on several threads
        buf = rng.integers(0, 1 << 63, size=(n, 24 * deg), dtype=np.uint64)
        addr = buf.ctypes.data
        h0 = gl.bases_content_hash(curve, buf)
        assert h0 == gl.bases_content_hash(curve, buf)                       # deterministic across calls / thread schedules
        copy = buf.copy()
        assert copy.ctypes.data != addr and gl.bases_content_hash(curve, copy) == h0
        seen = {h0}
        for row, col in ((n // 2, 7), (1, 0), (n - 2, 24 * deg - 1), (65536 // (24 * deg), 3)):
            old = buf[row, col]
            buf[row, col] ^= np.uint64(1)                                    # in-place mutation: same address, same ends
            assert buf.ctypes.data == addr
            h = gl.bases_content_hash(curve, buf)
            assert h not in seen and h[0] != h0[0] and h[1] != h0[1]
            seen.add(h)
            buf[row, col] = old
        assert gl.bases_content_hash(curve, buf) == h0
        # rows swapped: same multiset of limbs, another key; a shorter prefix: another key
        buf[[10, 11]] = buf[[11, 10]]
        assert gl.bases_content_hash(curve, buf) not in seen
        assert gl.bases_content_hash(curve, buf[: n - 1]) not in seen
        # infinity flags belong to the key; an all-zero flag array is the same key as no flag array
        inf = np.zeros(n, dtype=np.uint8)
        hz = gl.bases_content_hash(curve, buf)
        assert gl.bases_content_hash(curve, buf, inf) == hz
        inf[n // 3] = 1
        assert gl.bases_content_hash(curve, buf, inf) != hz
    assert gl.bases_content_hash("mnt4753_g1", np.zeros((4, 24), np.uint64)) != gl.bases_content_hash("mnt6753_g1", np.zeros((4, 24), np.uint64))
    st = gl.key_cache_stats()
    assert set(st) == {"entries", "bytes", "hits", "misses", "evictions", "tables_built", "collisions"}
    gl.key_cache_config(1 << 30, 0)
    gl.key_cache_clear()
    gl.key_cache_config()


def test_key_identity_has_verification_lanes_that_survive_a_forced_collision(gl):
    """A hit of gh_msm_cached needs all FOUR 64-bit lanes of the key identity to agree: two select the cache entry, an
    independent pair verifies it (round 3 trusted 128 unkeyed bits: a collision returned another key's sum with status 0).
    gh_test_hooks(1) makes the selection lanes of every key constant -- the forced collision; the verification lanes must
    still tell any two base vectors apart.  Host-only; the device-side consequence (a miss, the right sum, `collisions`
    counted) is tests/test_gpu_parity.py::test_msm_cached_survives_forced_hash_collisions_and_a_failed_table_build."""
    rng = np.random.default_rng(11)
    n = 5000
    a = rng.integers(0, 1 << 63, size=(n, 24), dtype=np.uint64)
    b = a.copy()
    b[n // 2, 5] ^= np.uint64(1 << 40)
    ida, idb = gl.bases_key_id("mnt4753_g1", a), gl.bases_key_id("mnt4753_g1", b)
    assert ida == gl.bases_key_id("mnt4753_g1", a.copy())
    assert gl.bases_content_hash("mnt4753_g1", a) == ida[:2]
    assert all(x != y for x, y in zip(ida, idb))                # every lane sees the flipped bit
    gl.set_test_hooks(1)
    try:
        fa, fb = gl.bases_key_id("mnt4753_g1", a), gl.bases_key_id("mnt4753_g1", b)
        assert fa[:2] == fb[:2]                                 # the forced collision of the selection lanes
        assert fa[2:] == ida[2:] and fb[2:] == idb[2:] and fa[2:] != fb[2:]
        fc = gl.bases_key_id("mnt6753_g1", a)
        assert fc[:2] == fa[:2] and fc[2:] != fa[2:]            # the curve is part of the verified identity
    finally:
        gl.set_test_hooks(0)
    assert gl.bases_key_id("mnt4753_g1", a) == ida


def test_compute_fails_loudly_without_gpu(gl):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gl.GingerHipError):
        gl.VariableBaseMSM.multi_scalar_mul("mnt4753_g1", np.zeros((1, 24), np.uint64), np.zeros((1, 12), np.uint64))
    with pytest.raises(gl.GingerHipError):
        gl.EvaluationDomain("mnt4753_fr", 4).fft(np.zeros((4, 12), np.uint64))
    with pytest.raises(gl.GingerHipError):
        gl.msm_cached("mnt4753_g1", np.zeros((1, 24), np.uint64), np.zeros((1, 12), np.uint64))


def test_long_branch_guard_rules():
    """tools/check_long_branches.py (the build-time guard against the hipcc relaxed-branch hang): its two
    rules on hand-made listings -- a leaf callee clobbering s[30:31] is flagged, a callee that saved s30 first and a kernel are not"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("clb", os.path.join(ROOT, "tools", "check_long_branches.py"))
    clb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(clb)
    dis = "\n".join([
        "0000000000001000 <leaf_fn>:", "\ts_getpc_b64 s[30:31]  // 1000", "\ts_setpc_b64 s[30:31]",
        "0000000000002000 <nonleaf_fn>:", "\tv_writelane_b32 v254, s30, 0", "\ts_getpc_b64 s[30:31]", "\ts_swappc_b64 s[30:31], s[30:31]",
        "0000000000003000 <other_pair>:", "\ts_getpc_b64 s[46:47]"])
    fns = dict(clb.functions(dis))
    assert set(fns) == {"leaf_fn", "nonleaf_fn", "other_pair"} and len(fns["leaf_fn"]) == 2
    assert len(clb.scan_callee("leaf_fn", fns["leaf_fn"], 64)) == 1
    assert clb.scan_callee("nonleaf_fn", fns["nonleaf_fn"], 300000) == []
    assert clb.scan_callee("other_pair", fns["other_pair"], 64) == []
    assert len(clb.scan_callee("other_pair", fns["other_pair"], 200000)) == 1          # a leaf above the relaxation size
    # and on the built library, when there is one: no finding
    lib = os.path.join(ROOT, "ginger-lib_amd", "libginger_hip.so")
    if os.path.exists(lib):
        found, n_funcs, n_kernels, _ = clb.check(lib)
        assert found == [] and n_kernels > 50 and n_funcs > n_kernels
    # a file without gfx950 code objects is "nothing checked" (status 2), not "0 findings" (status 0)
    import subprocess
    import sys
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".so") as f:
        f.write(b"\x7fELF not a fat binary")
        f.flush()
        assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "check_long_branches.py"), f.name], stdout=subprocess.DEVNULL) == 2
