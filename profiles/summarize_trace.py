#!/usr/bin/env python3
"""Per-launch durations from a rocprofv3 --kernel-trace CSV, grouped by (kernel, grid size) so that the main launch of a
POA class kernel and its few-wave mop-up launch (same kernel, same call) are not averaged together, which is what
rocprofv3's own --stats table does.   usage: summarize_trace.py trace_kernel_trace.csv"""
import csv
import sys
from collections import defaultdict


def short(name):
    if "poa_class_kernel" in name:
        return "poa_class_kernel<" + name.split("PoaCfg<")[1].split(">")[0].replace(" ", "").replace(",", ";") + ">"
    return name.split("(")[0].replace("void ", "")[:70]


def main(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]), r["LDS_Block_Size"], r["VGPR_Count"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("kernel,grid,lds_bytes,vgprs,launches,avg_us,min_us,max_us")
    for (k, g, lds, vg), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if "hypo" in k or "poa" in k or "scan" in k:
            print(f"{k},{g},{lds},{vg},{len(v)},{sum(v) / len(v) / 1e3:.1f},{min(v) / 1e3:.1f},{max(v) / 1e3:.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
