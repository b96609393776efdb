"""stage markers of the debug build of a G2 round kernel (GH_ASM_DEBUG=1 python -m asmgen.build build/dbg): development only"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import asm_g2_check as K
T = K.T
K.module.__globals__["_mod"] = None
import ctypes
blob = open(os.path.join(ROOT, "build", "dbg", "gh_asm.hsaco"), "rb").read()
m = K.vp()
K.chk(K.hip.hipModuleLoadData(ctypes.byref(m), blob), "load")
K._mod = m
for cname, tag in (("mnt4753_g2", "f2"),):
    for fwd in (True, False):
        for r0 in (True, False):
            g, _ = T._prog(cname, fwd, r0)
            K.NAME[g.name] = "gh_asm_aff_%s_%s_%s" % (tag, "fwd" if fwd else "bwd", "r0" if r0 else "rn")
g, _ = T._prog("mnt4753_g2", False, True)
K.NAME[g.name] += "_dbg"

def runner(g, bufs, scalars, waves):
    K.hip_runner(g, bufs, scalars, waves)
    print(K.NAME[g.name], "flag words", [int(x) for x in bufs["flag"]])
orig = T._run_round
def patched(*a, **k):
    k["runner"] = runner
    return orig(*a, **k)
T._run_round = patched
try:
    T.test_g2_round_kernels_in_the_simulator("mnt4753_g2")
except AssertionError as e:
    print("assert", str(e)[:100])
