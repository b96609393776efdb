#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) of
tools/acc_probe.py / tools/prof_run.py: HBM bytes of the accumulation launches of ONE MSM (the last in the run) and of
one NTT, stamped with the sha256 of the kernel sources (bench.py prints null for traffic measured on other sources).
usage: make_traffic_json.py out.json key:fetch.csv:write.csv:window_bits:mode [...]   (key e.g. mnt4753_g1_2p20, ntt_2p24)"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernels_sha

ACC = ("msm_accumulate_kernel", "msm_accumulate_xyzz_kernel", "msm_accumulate_split_kernel", "aff_round_kernel", "aff_desc_kernel", "msm_heavy_combine_kernel",
       "gh_asm_acc_g1", "gh_asm_aff", "aff_inv_kernel", "aff_fix_kernel", "msm_acc_tasks_kernel")


def per_dispatch(path, counter):
    by = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        d = by.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])
        d[1] += float(r["Counter_Value"])
    return [(k, v[0], v[1]) for k, v in sorted(by.items())]


def last_msm_bytes(path, counter):
    rows = per_dispatch(path, counter)
    starts = [i for i, r in enumerate(rows) if "msm_digits_kernel" in r[1]]
    seg = rows[starts[-1]:] if starts else rows
    # counters are in KB (rocprofv3 derived metric: 64-byte / 32-byte requests scaled to kilobytes)
    return sum(v for _, name, v in seg if any(k in name for k in ACC)) * 1024.0


def ntt_bytes(path, counter):
    rows = [r for r in per_dispatch(path, counter) if "ntt_pass_kernel" in r[1] or "gh_asm_ntt" in r[1]]
    # passes of the last transform: the launches after the last gap are identical in count per transform; take the last 3 (2^24) / all / n
    n_pass = 3
    return sum(v for _, _, v in rows[-n_pass:]) * 1024.0


def main():
    out_path = sys.argv[1]
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes); bytes = counter (KB) * 1024; "
                     "MI355X_MICROARCH.md: FETCH_SIZE reports half the bytes of wide coalesced streaming reads -- raw figures given",
           "kernels_sha256": kernels_sha(),
           "measured": "by the builder through gpurun on %s (UTC), MI355X box %s" % (__import__("time").strftime("%Y-%m-%d %H:%M", __import__("time").gmtime()),
                                                                                    __import__("socket").gethostname())}
    for spec in sys.argv[2:]:
        key, fetch, write, wbits, mode = spec.split(":", 4)
        if key.startswith("ntt"):
            out[key] = {"fetch_bytes_per_transform": ntt_bytes(fetch, "FETCH_SIZE"), "write_bytes_per_transform": ntt_bytes(write, "WRITE_SIZE")}
        else:
            out[key] = {"window_bits": int(wbits), "bucket_sums": mode, "fetch_bytes_per_msm": last_msm_bytes(fetch, "FETCH_SIZE"),
                        "write_bytes_per_msm": last_msm_bytes(write, "WRITE_SIZE")}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
