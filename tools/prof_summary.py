#!/usr/bin/env python3
"""Condenses a rocprofv3 output directory into a small summary (committed under profiles/).

    python tools/prof_summary.py <rocprof_dir> <out.json> [kernel-name-substring ...]

Reads *_kernel_stats.csv (from --kernel-trace --stats) and *_counter_collection.csv (from
--pmc) if present; keeps the rows of kernels whose name contains one of the substrings."""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    d, out = sys.argv[1], sys.argv[2]
    keys = sys.argv[3:] or ["hnsw::"]
    res = {"source_dir": d, "kernel_stats": [], "counters": {}}
    for f in glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Name"] for k in keys):
                res["kernel_stats"].append({"name": r["Name"].split("(")[0], "calls": int(r["Calls"]),
                                            "total_ns": int(r["TotalDurationNs"]), "avg_ns": float(r["AverageNs"]),
                                            "min_ns": int(r["MinNs"]), "max_ns": int(r["MaxNs"]), "pct": float(r["Percentage"])})
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if any(k in name for k in keys):
                a = agg[name.split("(")[0]][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    for kname, cs in agg.items():
        res["counters"][kname] = {c: {"sum": v[0], "dispatches": v[1], "avg_per_dispatch": v[0] / max(1, v[1])} for c, v in cs.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res)[:2000])


if __name__ == "__main__":
    main()
