"""The generated G2 round kernels on the card against the CPU simulator, buffer by buffer: python3 tools/asm_g2_check.py
Loads build/gh_asm.hsaco with the HIP module API (ctypes on libamdhip64), runs the inputs of tests/test_asmgen_g2.py through
the forward and backward kernels on the GPU and in asmgen/sim.py, and reports the first buffer (staged inputs, prefix products,
running products, output list, flag) that differs.  A development tool: the product path is exercised by tests/test_gpu_*.py."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "ginger-lib_amd"))
import test_asmgen_g2 as T          # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
vp = ctypes.c_void_p


def chk(e, what):
    if e != 0:
        raise RuntimeError("%s failed: hip error %d" % (what, e))


_mod = None


def module():
    global _mod
    if _mod is None:
        blob = open(os.path.join(ROOT, "build", "gh_asm.hsaco"), "rb").read()
        m = vp()
        chk(hip.hipModuleLoadData(ctypes.byref(m), blob), "hipModuleLoadData")
        _mod = m
    return _mod


NAME = {}


def hip_runner(g, bufs, scalars, waves):
    name = NAME[g.name]
    fn = vp()
    chk(hip.hipModuleGetFunction(ctypes.byref(fn), module(), name.encode()), "hipModuleGetFunction " + name)
    dev = {}
    for k, v in bufs.items():
        d = vp()
        nbytes = max(v.nbytes, 4)
        chk(hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(nbytes)), "hipMalloc")
        chk(hip.hipMemcpy(d, v.ctypes.data_as(vp), ctypes.c_size_t(v.nbytes), 1), "hipMemcpy H2D")
        dev[k] = d
    karg = np.zeros(22, dtype=np.uint32)
    for j, k in enumerate(T.ARG_ORDER):
        a = dev[k].value
        karg[2 * j], karg[2 * j + 1] = a & 0xFFFFFFFF, a >> 32
    karg[18], karg[19], karg[20] = scalars
    size = ctypes.c_size_t(88)
    extra = (vp * 5)(vp(1), karg.ctypes.data_as(vp), vp(2), ctypes.cast(ctypes.pointer(size), vp), vp(3))
    blocks = (waves + 1 + 3) // 4
    chk(hip.hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, None, None, extra), "launch " + name)
    chk(hip.hipDeviceSynchronize(), "sync after " + name)
    for k in T.WRITABLE:
        chk(hip.hipMemcpy(bufs[k].ctypes.data_as(vp), dev[k], ctypes.c_size_t(bufs[k].nbytes), 2), "hipMemcpy D2H")
    for d in dev.values():
        hip.hipFree(d)


def both_runner(g, bufs, scalars, waves):
    """GPU and simulator on copies of the same buffers; the GPU's result goes on, differences are reported"""
    sim = {k: v.copy() for k, v in bufs.items()}
    T.sim_runner(g, sim, scalars, waves)
    hip_runner(g, bufs, scalars, waves)
    for k in T.WRITABLE:
        if not np.array_equal(sim[k], bufs[k]):
            d = np.argwhere(sim[k] != bufs[k])
            print("  %s: buffer %-7s differs in %d words; first at %s: sim %08x gpu %08x" %
                  (NAME[g.name], k, len(d), tuple(d[0]), sim[k][tuple(d[0])], bufs[k][tuple(d[0])]))
            if k in ("prefix", "accs", "out", "stage1", "stage2"):
                tiles = sorted(set(int(x[0]) for x in d))
                slots = sorted(set(int(x[2]) for x in d))
                chunks = sorted(set(int(x[1]) for x in d))
                print("     tiles", tiles[:12], "chunks", chunks, "slots", slots[:70])
        else:
            print("  %s: buffer %-7s equal" % (NAME[g.name], k))


def main():
    import pyref
    for cname, tag in (("mnt4753_g2", "f2"), ("mnt6753_g2", "f3")):
        for fwd in (True, False):
            for r0 in (True, False):
                g, _ = T._prog(cname, fwd, r0)
                NAME[g.name] = "gh_asm_aff_%s_%s_%s" % (tag, "fwd" if fwd else "bwd", "r0" if r0 else "rn")
    orig = T._run_round

    def patched(*a, **k):
        k["runner"] = both_runner
        return orig(*a, **k)
    T._run_round = patched
    ok = True
    for cname in ("mnt4753_g2", "mnt6753_g2"):
        print(cname)
        try:
            T.test_g2_round_kernels_in_the_simulator(cname)
            print("  outputs equal the group law")
        except AssertionError as e:
            ok = False
            print("  MISMATCH against the group law:", str(e)[:200])
    print("flag cases")
    try:
        T.test_g2_round_kernels_list_the_rare_cases("mnt4753_g2")
        print("  ok")
    except AssertionError as e:
        ok = False
        print("  MISMATCH", str(e)[:200])
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
