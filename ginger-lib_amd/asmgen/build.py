"""build.py -- generate the assembly kernels, assemble them for gfx950 and link one code object.

    python -m asmgen.build OUTDIR      (from ginger-lib_amd/)  ->  OUTDIR/gh_asm.s, gh_asm.o, gh_asm.hsaco, gh_asm.json

The code object is embedded into libginger_hip.so by __graft_entry__.build() (csrc/asm_blob.S .incbin) and loaded with
hipModuleLoadData (csrc/asm_kernels.h).  Nothing here needs a GPU.
"""
import json
import os
import subprocess
import sys

from . import g1_xyzz, g2_rounds, microbench, ntt_pass
from .isa import module_text

LLVM = os.environ.get("GH_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

P4 = 0x1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db117e776f218059db80f0da5cb537e38685acce9767254a4638810719ac425f0e39d54522cdd119f5e9063de245e8001
P6 = 0x1c4c62d92c41110229022eee2cdadb7f997505b8fafed5eb7e8f96c97d87307fdb925e8a0ed8d99d124d9a15af79db26c5c28c859a99b3eebca9429212636b9dff97634993aa4d6c381bc3f0057974ea099170fa13a4fd90776e240000001
R = 1 << 754


def programs():
    progs = [
        g1_xyzz.build("gh_asm_acc_g1_p4", P4, R % P4),
        g1_xyzz.build("gh_asm_acc_g1_p6", P6, R % P6),
    ]
    # the affine rounds of the G2 MSMs: forward / backward kernel of round 0 and of the later rounds, per tower
    c2 = g2_rounds.Cfg(2, 13, P4, R % P4)          # MNT4-753 G2: Fq2 = Fq[u] / (u^2 - 13)   (fields/mnt4753/fq2.rs:19)
    c3 = g2_rounds.Cfg(3, 11, P6, R % P6)          # MNT6-753 G2: Fq3 = Fq[u] / (u^3 - 11)   (fields/mnt6753/fq3.rs)
    c1a = g2_rounds.Cfg(1, 1, P4, R % P4)           # MNT4-753 G1 / MNT6-753 G1: the same rounds with one lane per element
    c1b = g2_rounds.Cfg(1, 1, P6, R % P6)
    for tag, cfg in (("f2", c2), ("f3", c3), ("f1p4", c1a), ("f1p6", c1b)):
        for fwd in (True, False):
            for r0 in (True, False):
                progs.append(g2_rounds.build("gh_asm_aff_%s_%s_%s" % (tag, "fwd" if fwd else "bwd", "r0" if r0 else "rn"), cfg, fwd, r0))
    # the NTT passes over the scalar fields (MNT4-753 Fr = P6, MNT6-753 Fr = P4), 6 to 8 stages per pass
    for tag, prime in (("p4", P4), ("p6", P6)):
        for k in (6, 7, 8):
            progs.append(ntt_pass.build("gh_asm_ntt_%s_k%d" % (tag, k), prime, k))
    progs.append(microbench.build("gh_asm_mb_mulpair", P4))
    if os.environ.get("GH_ASM_DEBUG"):          # stage markers for tools/asm_g2_check.py (not shipped)
        progs.append(g2_rounds.build("gh_asm_aff_f2_bwd_r0_dbg", c2, False, True, debug=True))
    if os.environ.get("GH_ASM_VARIANTS"):      # A/B variants of the gather for tools/asm_mb/acc_run.hip (not shipped)
        progs += [g1_xyzz.build("gh_asm_acc_g1_p4_s1", P4, R % P4, split=1),
                  g1_xyzz.build("gh_asm_acc_g1_p4_s2", P4, R % P4, split=2),
                  g1_xyzz.build("gh_asm_acc_g1_p4_s1pf", P4, R % P4, split=1, prefetch=True)]
    return progs


def build(outdir):
    os.makedirs(outdir, exist_ok=True)
    progs = programs()
    text, infos = module_text(progs)
    s_path = os.path.join(outdir, "gh_asm.s")
    with open(s_path, "w") as f:
        f.write(text)
    o_path = os.path.join(outdir, "gh_asm.o")
    co_path = os.path.join(outdir, "gh_asm.hsaco")
    subprocess.run([os.path.join(LLVM, "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950",
                    "-c", s_path, "-o", o_path], check=True)
    subprocess.run([os.path.join(LLVM, "ld.lld"), "-shared", o_path, "-o", co_path], check=True)
    for p in progs:
        infos[p.name]["instructions"] = p.count()
        infos[p.name]["lds_bytes"] = p.lds_bytes
        infos[p.name]["scratch_bytes_per_lane"] = 0
    with open(os.path.join(outdir, "gh_asm.json"), "w") as f:
        json.dump(infos, f, indent=1, sort_keys=True)
    return co_path, infos


if __name__ == "__main__":
    co, infos = build(sys.argv[1] if len(sys.argv) > 1 else "build")
    print(co, json.dumps(infos))
