#!/usr/bin/env python3
"""Per-launch mean of rocprofv3 --pmc counters from *_counter_collection.csv files.  Launches are grouped by
(kernel, grid size, pass): the concurrent POA schedule runs the class-1 and class-2 kernels twice per call, the main launch
and a mop-up launch for windows re-queued meanwhile (pass 1: the second of two launches per call), and the two must not be averaged.
usage: summarize_pmc.py dir_with_pass_subdirs   (prints one table; FETCH/WRITE_SIZE in KiB per launch)"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name, _, grid = name.partition(" grid=")
    return _short(name) + " grid=" + grid


def _short(name):
    if "poa_class_kernel" in name:
        cfg = name.split("PoaCfg<")[1].split(">")[0].replace(" ", "")
        return "poa_class_kernel<" + cfg.replace(",", ";") + ">"
    return name.split("(")[0][:60]


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "*_counter_collection.csv")):
        per_dispatch = defaultdict(lambda: defaultdict(float))
        for row in csv.DictReader(open(f)):
            per_dispatch[(row["Dispatch_Id"], row["Kernel_Name"] + " grid=" + row["Grid_Size"])][row["Counter_Name"]] += float(row["Counter_Value"])
        calls = sum(1 for (did, kn) in per_dispatch if "poa_plan_scan_kernel" in kn)
        per_kernel = defaultdict(int)
        for (did, kn) in per_dispatch:
            per_kernel[kn] += 1
        seen = defaultdict(int)
        for (did, kn), cs in sorted(per_dispatch.items(), key=lambda kv: int(kv[0][0])):
            ps = seen[kn] % 2 if calls and per_kernel[kn] == 2 * calls else 0
            seen[kn] += 1
            for c, v in cs.items():
                acc[short(kn) + f" pass={ps}"][c].append(v)
    counters = sorted({c for k in acc for c in acc[k]})
    print("kernel," + ",".join(counters) + ",dispatches")
    for k in sorted(acc):
        if "hypo" not in k and "poa" not in k and "scan" not in k:
            continue
        n = max(len(v) for v in acc[k].values())
        print(k + "," + ",".join(f"{sum(acc[k][c]) / len(acc[k][c]):.1f}" if acc[k][c] else "" for c in counters) + f",{n}")


if __name__ == "__main__":
    main(sys.argv[1])
