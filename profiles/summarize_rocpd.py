#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd sqlite result (`*_results.db`, the default output of ROCm 7.2's rocprofv3)
into the per-kernel statistics table `rocprofv3 --stats` prints (calls, total/avg/min/max ns, share).
usage: summarize_rocpd.py results.db > kernel_stats.csv"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = e - s
        a = agg.setdefault(name, [0, 0, None, 0])
        a[0] += 1
        a[1] += d
        a[2] = d if a[2] is None or d < a[2] else a[2]
        a[3] = d if d > a[3] else a[3]
    tot = sum(a[1] for a in agg.values()) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'"{name}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / tot:.4f},{a[2]},{a[3]}')


if __name__ == "__main__":
    main(sys.argv[1])
