"""Per-launch counter values of selected kernels from rocprofv3 counter_collection csv:
python3 tools/pmc_table.py <run_counter_collection.csv> <kernel substring> [...]"""
import csv, sys
from collections import OrderedDict
rows = list(csv.DictReader(open(sys.argv[1])))
keys = sys.argv[2:]
by = OrderedDict()
for r in rows:
    name = r['Kernel_Name']
    if not any(k in name for k in keys): continue
    d = by.setdefault(r['Dispatch_Id'], {'name': name.replace('void gh::', '').replace('gh::', '')[:40], 'grid': r.get('Grid_Size', '')})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, d in by.items():
    print(k, d['name'], d['grid'], " ".join("%s=%.4g" % (c, v) for c, v in d.items() if c not in ('name', 'grid')))
