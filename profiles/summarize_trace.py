#!/usr/bin/env python3
"""Per-launch durations from a rocprofv3 --kernel-trace CSV, grouped by (kernel, grid size, pass) so that the main launch of a
POA class kernel and its mop-up launch (same kernel, same call, possibly the same grid) are not averaged together, which is
what rocprofv3's own --stats table does.  A kernel launched twice per call (calls = launches of the plan kernel) is split
into pass 0 (first launch of the call) and pass 1.   usage: summarize_trace.py trace_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


def short(name):
    if "poa_class_kernel" in name:
        return "poa_class_kernel<" + name.split("PoaCfg<")[1].split(">")[0].replace(" ", "").replace(",", ";") + ">"
    return name.split("(")[0].replace("void ", "")[:70]


def main(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    calls = sum(1 for r in rows if "poa_plan_scan_kernel" in r["Kernel_Name"])
    per_kernel = defaultdict(int)
    for r in rows:
        per_kernel[r["Kernel_Name"]] += 1
    seen = defaultdict(int)
    acc = defaultdict(list)
    for r in rows:
        kn = r["Kernel_Name"]
        two = calls and per_kernel[kn] == 2 * calls
        ps = seen[kn] % 2 if two else 0
        seen[kn] += 1
        acc[(short(kn), int(r["Grid_Size_X"]), ps, r["LDS_Block_Size"], r["VGPR_Count"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("kernel,grid,pass,lds_bytes,vgprs,launches,avg_us,min_us,max_us")
    for (k, g, ps, lds, vg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if "hypo" in k or "poa" in k or "scan" in k:
            print(f"{k},{g},{ps},{lds},{vg},{len(v)},{sum(v) / len(v) / 1e3:.1f},{min(v) / 1e3:.1f},{max(v) / 1e3:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
