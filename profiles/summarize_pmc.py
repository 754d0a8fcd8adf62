#!/usr/bin/env python3
"""Per-launch mean of rocprofv3 --pmc counters from *_counter_collection.csv files.  Launches are grouped by
(kernel, grid size): the concurrent POA schedule runs every LDS class kernel twice per call, the main launch (full
grid) and a mop-up launch for windows re-queued meanwhile (a handful of waves), and the two must not be averaged.
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
        for (did, kn), cs in per_dispatch.items():
            for c, v in cs.items():
                acc[short(kn)][c].append(v)
    counters = sorted({c for k in acc for c in acc[k]})
    print("kernel," + ",".join(counters) + ",dispatches")
    for k in sorted(acc):
        if "hypo" not in k and "poa" not in k and "scan" not in k:
            continue
        n = max(len(v) for v in acc[k].values())
        print(k + "," + ",".join(f"{sum(acc[k][c]) / len(acc[k][c]):.1f}" if acc[k][c] else "" for c in counters) + f",{n}")


if __name__ == "__main__":
    main(sys.argv[1])
