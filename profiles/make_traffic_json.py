#!/usr/bin/env python3
"""Writes profiles/pmc_traffic.json (read by bench.py for roofline.traffic) from a PMC summary CSV made by
summarize_pmc.py.  FETCH_SIZE / WRITE_SIZE are KiB per dispatch; per MI355X_MICROARCH.md §HBM they come from
the L2's memory-side request counters.  The guide's x2 correction applies to wide (16 B/lane) coalesced streaming
reads only; this kernel's HBM reads are narrow (bytes / dwords of packed arms and descriptors), for which the
guide says the counter is uncalibrated — so the raw sum is reported and the doubled-read figure kept beside it.
usage: make_traffic_json.py summary.csv class:windows [class:windows ...]   (one entry per size-class kernel)"""
import csv
import json
import os
import sys

# class 0 runs as 32;2;47 (two groups per wave) or 16;4;47 (four), whichever poa_run picked for the profiled batch; class 1 is 64;2;79 since round 3 (32;4;79 before)
CLASS_CFG = {0: ("32;2;47", "16;4;47"), 1: ("64;2;79", "32;4;79"), 2: ("64;2;127;126",), 3: ("64;4;255",), 4: ("64;10;639",), 5: ("64;16;1023",)}


def entry(path, cls, n_windows):
    rows = [r for r in csv.DictReader(open(path)) if any(r["kernel"].startswith("poa_class_kernel<" + c) for c in CLASS_CFG[cls])
            and r["kernel"].endswith("pass=0")]                           # main launch (pass 1 = mop-up launch of the same kernel)
    if not rows:
        raise SystemExit(f"class {cls}: kernel not found")
    row = max(rows, key=lambda r: float(r.get("SQ_WAVE_CYCLES") or 0))      # the main launch (a mop-up launch with another grid size sorts first)
    f, w = float(row["FETCH_SIZE"]) * 1024, float(row["WRITE_SIZE"]) * 1024
    return {"kernel": f"poa_class_kernel<class {cls}>", "windows": n_windows,
            "hbm_bytes_per_launch": int(f + w), "fetch_bytes": int(f), "write_bytes": int(w),
            "fetch_bytes_if_wide_read_correction_applied": int(2 * f),
            "tcc_hit": float(row["TCC_HIT_sum"]), "tcc_miss": float(row["TCC_MISS_sum"]),
            # issue side of the same launch (bench.py: roofline.valu)
            "valu_insts": int(float(row.get("SQ_INSTS_VALU") or 0)), "salu_insts": int(float(row.get("SQ_INSTS_SALU") or 0)),
            "lds_insts": int(float(row.get("SQ_INSTS_LDS") or 0)),
            "wave_cycles_quad": int(float(row.get("SQ_WAVE_CYCLES") or 0)), "wait_any_quad": int(float(row.get("SQ_WAIT_ANY") or 0)),
            "active_inst_any_quad": int(float(row.get("SQ_ACTIVE_INST_ANY") or 0)),
            "source": os.path.basename(path)}


def main(path, spec):
    """spec: class:windows pairs, e.g. 0:59946 1:20232 2:16900 (the windows each class's main launch processed)"""
    out = {}
    for item in spec:
        cls, n = item.split(":")
        e = entry(path, int(cls), int(n))
        out[e["kernel"]] = e
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
