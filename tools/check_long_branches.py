"""Build-time guard against the hipcc long-branch hang.

hipcc (ROCm 7.2) relaxes a branch that spans more than 2^15 dwords into
    s_getpc_b64 s[N:N+1] ; s_add_u32 / s_addc_u32 ; s_setpc_b64 s[N:N+1]
and has been seen to pick s[30:31] as the scratch pair.  In a LEAF device function s[30:31] still holds the
return address, so the function returns into its own body and the wave never finishes.  Kernels are safe
(no return address); so are non-leaf callees, which save s[30:31] in a VGPR lane on entry (v_writelane_b32 vN, s30)
and restore it before s_setpc_b64 -- the G2 proj_*_call functions are 150-400 KB and fine for that reason.
Rules enforced (s_getpc_b64 is also the ordinary call sequence, so only the register pair matters):

    1. a device function that is not a kernel may hold `s_getpc_b64 s[30:31]` only after it saved s30;
    2. a LEAF device function (no s_swappc_b64) may not exceed MAX_LEAF_BYTES, the size at which relaxation starts.

Usage: python3 tools/check_long_branches.py [path/to/libginger_hip.so]
Exit status: 0 clean; 1 a finding (the build renames the library to .rejected); 2 the check could not run -- llvm-objdump /
llvm-readelf missing, or no gfx950 code object / no kernel found in the library (another bundle format): a guard that looked
at nothing must not report "0 findings", and a missing tool must not reject a good library.
Reads the gfx950 code objects out of the library's .hip_fatbin section (clang offload bundles) and disassembles
them with llvm-objdump; `__graft_entry__.build()` runs it after linking."""
import os
import re
import struct
import subprocess
import sys
import tempfile

def _llvm_dir():
    """llvm-objdump / llvm-readelf: $ROCM_PATH/lib/llvm/bin, what `hipcc --print-prog-name` names, /opt/rocm, then PATH"""
    import shutil
    cands = []
    if os.environ.get("ROCM_PATH"):
        cands.append(os.path.join(os.environ["ROCM_PATH"], "lib", "llvm", "bin"))
    try:
        out = subprocess.run(["hipcc", "--print-prog-name=llvm-objdump"], capture_output=True, text=True, timeout=30).stdout.strip()
        if out and os.path.isabs(out):
            cands.append(os.path.dirname(out))
    except Exception:       # noqa: hipcc absent or too old for the flag
        pass
    cands.append("/opt/rocm/lib/llvm/bin")
    for d in cands:
        if os.path.exists(os.path.join(d, "llvm-objdump")) and os.path.exists(os.path.join(d, "llvm-readelf")):
            return d
    w = shutil.which("llvm-objdump")
    if w and shutil.which("llvm-readelf"):
        return os.path.dirname(w)
    return None


class ToolUnavailable(RuntimeError):
    pass


LLVM = _llvm_dir()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
MAX_LEAF_BYTES = 100 * 1024        # 2^15 dwords = 128 KiB is where relaxation starts; stay well below


def code_objects(path, arch="gfx950"):
    data = open(path, "rb").read()
    pos = 0
    while True:
        at = data.find(MAGIC, pos)
        if at < 0:
            return
        pos = at + len(MAGIC)
        (count,) = struct.unpack_from("<Q", data, pos)
        q = pos + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if arch in triple and size:
                yield triple, data[at + off:at + off + size]


def functions(disasm):
    """yield (symbol, [instruction lines]) from llvm-objdump -d output"""
    name, body = None, []
    for line in disasm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            if name is not None:
                yield name, body
            name, body = m.group(1), []
        elif name is not None and line.startswith("\t"):
            body.append(line.strip())
    if name is not None:
        yield name, body


def scan_callee(name, body, size):
    """the two rules on one non-kernel device function"""
    out = []
    leaf = not any(i.startswith("s_swappc_b64") for i in body)
    if leaf and size > MAX_LEAF_BYTES:
        out.append("leaf device function %s is %d bytes (> %d): its branches get relaxed" % (name, size, MAX_LEAF_BYTES))
    saved = False
    for i in body:
        if re.match(r"v_writelane_b32 v\d+, s30\b", i):
            saved = True
        elif i.startswith("s_getpc_b64") and "s[30:31]" in i and not saved:
            out.append("%s: '%s' overwrites the live return address" % (name, i.split("//")[0].strip()))
    return out


def check(path):
    if LLVM is None:
        raise ToolUnavailable("llvm-objdump / llvm-readelf not found ($ROCM_PATH, hipcc --print-prog-name, /opt/rocm, PATH)")
    findings, n_funcs, n_kernels, n_relaxed = [], 0, 0, 0
    for idx, (triple, blob) in enumerate(code_objects(path)):
        with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
            f.write(blob)
            tmp = f.name
        try:
            syms = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", tmp], capture_output=True, text=True, check=True).stdout
            kernels, sizes = set(), {}
            for l in syms.split("\n"):
                p = l.split()
                if len(p) >= 8 and p[3] in ("FUNC", "OBJECT"):
                    if p[7].endswith(".kd"):
                        kernels.add(p[7][:-3])
                    elif p[3] == "FUNC":
                        sizes[p[7]] = int(p[2], 0) if not p[2].isdigit() else int(p[2])
            dis = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", tmp], capture_output=True, text=True, check=True).stdout
        finally:
            os.unlink(tmp)
        for name, body in functions(dis):
            n_funcs += 1
            getpc = [i for i in body if i.startswith("s_getpc_b64")]
            n_relaxed += len(getpc)
            if name in kernels:
                n_kernels += 1
                continue
            findings += ["code object %d: %s" % (idx, f) for f in scan_callee(name, body, sizes.get(name, 0))]
    return findings, n_funcs, n_kernels, n_relaxed


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "ginger-lib_amd", "libginger_hip.so")
    try:
        found, nf, nk, nr = check(lib)
    except (ToolUnavailable, subprocess.CalledProcessError, OSError) as e:
        print("[long-branch check] could not run: %s" % e)
        sys.exit(2)
    print("[long-branch check] %s: %d functions (%d kernels), %d s_getpc_b64, %d finding(s)" % (os.path.basename(lib), nf, nk, nr, len(found)))
    for f in found:
        print("  " + f)
    if found:
        sys.exit(1)
    if nf == 0 or nk == 0:
        print("  no gfx950 code object / kernel found in the library: nothing was checked")
        sys.exit(2)
    sys.exit(0)
