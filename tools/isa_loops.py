"""Instruction mix per basic block / loop of one kernel in a hipcc -S listing:
python3 tools/isa_loops.py build/g1.s aff_round_kernel"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(":") or (key in l and l.startswith("_ZN") and "; @" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur, name = [], [], "entry"
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); name, cur = m.group(1), []
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cur.append(t)
blocks.append((name, cur))
def kind(i):
    op = i.split()[0]
    if op.startswith("v_mad_u64_u32"): return "mad64"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    return "other"
tot = {}
for name, ins in blocks:
    c = {}
    for i in ins:
        k = kind(i); c[k] = c.get(k, 0) + 1; tot[k] = tot.get(k, 0) + 1
    if len(ins) >= 200:
        br = [i for i in ins if i.startswith("s_cbranch") or i.startswith("s_branch")]
        print("%-12s %6d instr  %s  %s" % (name, len(ins), " ".join("%s=%d" % kv for kv in sorted(c.items())), br[-1] if br else ""))
print("total", sum(tot.values()), tot)
