"""The Rust side of the boundary is delivered as files (this image has no rustc / cargo, so they cannot be compiled
here): rust/algebra-hip-sys (the unsafe FFI crate) and rust/patches/algebra-gpu-feature.patch (the `gpu` feature of
ginger-lib's `algebra`: dispatch in variable_base.rs:85-90 and domain.rs:113-179).  What CAN be checked without a
toolchain is checked here: the extern block is exactly the two C headers (regenerated and compared), struct layouts
and constants agree with the headers, and the patch hooks every entry point the scope table names."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "rust", "algebra-hip-sys", "src", "lib.rs")
PATCH = os.path.join(ROOT, "rust", "patches", "algebra-gpu-feature.patch")


def strip_comments(src):
    return re.sub(r"//[^\n]*", "", re.sub(r"/\*.*?\*/", "", src, flags=re.S))


def header(name):
    return strip_comments(open(os.path.join(ROOT, "include", name)).read())


def test_extern_block_is_generated_from_the_headers():
    assert subprocess.call([sys.executable, os.path.join(ROOT, "tools", "gen_rust_ffi.py"), "--check"]) == 0


def test_every_c_function_is_declared_once_with_the_same_arity():
    lib = open(LIB).read()
    block = lib[lib.index("extern \"C\" {"):]
    block = block[:block.index("\n}\n")]
    rust = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (gh_\w+)\((.*?)\)", block)}
    n_total = 0
    for h in ("ginger_hip.h", "ginger_hip_dist.h"):
        for m in re.finditer(r"\b(gh_\w+)\s*\(([^;{}]*?)\)\s*;", header(h), flags=re.S):
            name, params = m.group(1), " ".join(m.group(2).split())
            if name == "gh_allgather_fn" or "(*" in m.group(0):
                continue
            n_c = 0 if params in ("", "void") else params.count(",") + 1
            assert name in rust, name
            n_r = 0 if not rust[name].strip() else rust[name].count(",") + 1
            assert n_c == n_r, (name, n_c, n_r)
            n_total += 1
    assert n_total == len(rust) == 74


def test_constants_and_timing_struct_match_the_header():
    lib = open(LIB).read()
    h = header("ginger_hip.h") + header("ginger_hip_dist.h")
    for name, rname in (("GH_E_BAD_ARG", "GH_E_BAD_ARG"), ("GH_E_UNSUPPORTED", "GH_E_UNSUPPORTED"), ("GH_E_NO_DEVICE", "GH_E_NO_DEVICE"),
                        ("GH_E_HIP", "GH_E_HIP"), ("GH_E_NOMEM", "GH_E_NOMEM"), ("GH_E_BAD_HANDLE", "GH_E_BAD_HANDLE"), ("GH_E_DIST", "GH_E_DIST")):
        c = int(re.search(r"#define %s \((-?\d+)\)" % name, h).group(1))
        r = int(re.search(r"pub const %s: c_int = (-?\d+);" % rname, lib).group(1))
        assert c == r, name
    for cname, rname in (("GH_MNT4753_G1", "MNT4753_G1"), ("GH_MNT4753_G2", "MNT4753_G2"), ("GH_MNT6753_G1", "MNT6753_G1"),
                         ("GH_MNT6753_G2", "MNT6753_G2"), ("GH_MNT4753_FR", "MNT4753_FR"), ("GH_MNT6753_FR", "MNT6753_FR")):
        c = int(re.search(r"%s = (\d+)" % cname, h).group(1))
        r = int(re.search(r"pub const %s: c_int = (\d+);" % rname, lib).group(1))
        assert c == r, cname
    assert re.search(r"#define GH_FFT_INVERSE 1u", h) and "pub const FFT_INVERSE: u32 = 1;" in lib
    assert re.search(r"#define GH_FFT_COSET 2u", h) and "pub const FFT_COSET: u32 = 2;" in lib
    # gh_msm_timing_t, field by field
    body = re.search(r"typedef struct \{([^{}]*)\} gh_msm_timing_t;", h, flags=re.S).group(1)
    c_fields = [(t.strip(), n) for t, n in re.findall(r"([\w ]+?)\s+(\w+);", body)]
    rbody = re.search(r"pub struct GhMsmTiming \{(.*?)\}", lib, flags=re.S).group(1)
    r_fields = re.findall(r"pub (\w+): (\w+),", rbody)
    cmap = {"float": "f32", "int": "c_int", "unsigned long long": "u64", "unsigned int": "u32"}
    assert [(n, cmap[t]) for t, n in c_fields] == r_fields
    # gh_key_cache_stats_t: all u64, same names in the same order
    body = re.search(r"typedef struct \{([^{}]*)\} gh_key_cache_stats_t;", h, flags=re.S).group(1)
    c_names = [n for decl in re.findall(r"uint64_t\s+([^;]+);", body) for n in re.split(r"\s*,\s*", decl.strip())]
    rbody = re.search(r"pub struct GhKeyCacheStats \{(.*?)\}", strip_comments(lib.replace("///", "//")), flags=re.S).group(1)
    assert re.findall(r"pub (\w+): (\w+),", rbody) == [(n, "u64") for n in c_names] and len(c_names) == 7


def test_patch_hooks_the_reference_entry_points():
    p = open(PATCH).read()
    for needle in ("algebra/src/msm/variable_base.rs", "super::gpu::multi_scalar_mul::<G>(bases, scalars)", "Self::msm_inner(bases, scalars)",
                   "algebra/src/fft/domain.rs", "fn fft_in_place", "fn ifft_in_place", "fn coset_fft_in_place", "fn coset_ifft_in_place",
                   "algebra/src/msm/gpu.rs", "algebra/src/fft/gpu.rs", 'gpu = ["algebra-hip-sys", "parallel", "fft"]',
                   "mnt4753::G1Affine", "mnt4753::G2Affine", "mnt6753::G1Affine", "mnt6753::G2Affine", "mnt4753::Fr", "mnt6753::Fr"):
        assert needle in p, needle
    # the algebra side stays free of unsafe code (algebra/src/lib.rs:34 forbids it)
    added = "\n".join(l[1:] for l in p.split("\n") if l.startswith("+") and not l.startswith("+++"))
    code = strip_comments(added)
    assert "unsafe" not in code and "transmute" not in code
