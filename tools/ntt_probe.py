"""ntt_probe.py LOG_N [REPS] -- device-resident transform times (ms per transform by kind) and a round-trip check.
GH_NTT_ASM=0 selects the hipcc pass kernel for A/B runs."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib

gl = importlib.import_module("ginger-lib_amd")
import pyref          # noqa: E402
import support as S   # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
field = sys.argv[3] if len(sys.argv) > 3 else "mnt4753_fr"
N = 1 << log_n
a = S.random_scalars_np(N, seed=7, below=S.FIELD_OF[field].p)
buf = gl.DeviceBuffer(N * 96).upload(a)
dom = gl.EvaluationDomain(field, N)
for fl in range(4):
    dom.fft_dev(buf, fl)
ms = {0: [], 1: [], 2: [], 3: []}
for i in range(4 * reps):
    dom.fft_dev(buf, i & 3)
    ms[i & 3].append(gl.fft_last_kernel_ms())
names = ["fft", "ifft", "coset_fft", "coset_ifft"]
print("asm" if os.environ.get("GH_NTT_ASM", "1") != "0" else "hipcc", field, "2^%d" % log_n,
      " ".join("%s %.3f" % (names[k], float(np.mean(v))) for k, v in ms.items()),
      "mean %.3f ms" % float(np.mean([np.mean(v) for v in ms.values()])), flush=True)
# (fft, ifft) and (coset_fft, coset_ifft) pairs return the input
ok = (buf.download().reshape(N, 12) == a).all()
print("round trips ok" if ok else "ROUND TRIP MISMATCH", flush=True)
buf.free()
sys.exit(0 if ok else 1)
