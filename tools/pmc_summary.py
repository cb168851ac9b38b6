#!/usr/bin/env python3
"""Reduce a tools/profile.sh output directory (rocprofv3 CSVs) to one JSON summary:
per-kernel time stats from --kernel-trace --stats, per-kernel PMC sums from the --pmc passes.
usage: pmc_summary.py <dir> <config>"""
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("hbk::", "")
    return name.split("(")[0][:120]


def main():
    root, cfg = sys.argv[1], sys.argv[2]
    out = {"config": cfg, "kernel_stats": [], "pmc": {}}
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            out["kernel_stats"].append({
                "kernel": short(r.get("Name", "")), "calls": int(r.get("Calls", 0)),
                "total_ns": int(float(r.get("TotalDurationNs", 0))), "avg_ns": float(r.get("AverageNs", 0)),
                "pct": float(r.get("Percentage", 0)), "min_ns": int(float(r.get("MinNs", 0))),
                "max_ns": int(float(r.get("MaxNs", 0)))})
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r.get("Kernel_Name", ""))
            c = r.get("Counter_Name", "")
            v = float(r.get("Counter_Value", 0))
            d = out["pmc"].setdefault(k, {}).setdefault(c, {"sum": 0.0, "dispatches": 0})
            d["sum"] += v
            d["dispatches"] += 1
    for k, cs in out["pmc"].items():
        for c, d in cs.items():
            d["per_dispatch"] = d["sum"] / max(d["dispatches"], 1)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
