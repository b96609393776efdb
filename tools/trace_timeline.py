"""Timeline of the kernels of the LAST single MSM before a marker in a rocprofv3 kernel trace:
python3 tools/trace_timeline.py <run_kernel_trace.csv> [nth-from-last aff_round group]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
idx = [i for i, r in enumerate(rows) if 'digits' in r['Kernel_Name']]
j = idx[which]
end = idx[which + 1] if which + 1 < len(idx) else len(rows)
t0 = int(rows[j]['Start_Timestamp'])
for r in rows[j:end]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('void gh::', '').replace('gh::', '')[:34]
    if (e - s) < 20000 and 'aff' not in n: continue
    print("%-34s grid %9s  start %8.3f ms dur %8.3f ms scratch %s" % (n, r['Grid_Size_X'], (s - t0) / 1e6, (e - s) / 1e6, r['Scratch_Size']))
