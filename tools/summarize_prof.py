#!/usr/bin/env python3
"""Reduce rocprofv3 CSV output to small per-kernel summaries (the raw traces exceed gpurun's 64-MiB return cap).

usage: summarize_prof.py <rocprof_out_dir> <summary.csv> [--delete-raw]
  *_kernel_stats.csv        -> copied as is (already a summary)
  *_counter_collection.csv  -> per kernel name and counter: dispatches, mean / min / max counter value
  *_kernel_trace.csv        -> per kernel name: dispatches, mean / min / max duration (ns)
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    delete = "--delete-raw" in sys.argv
    rows_out = []
    for f in glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows_out.append(["kernel_stats", r["Name"], "duration_ns", r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"],
                                 r["Percentage"]])
    for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: [0, 0.0, float("inf"), 0.0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = (r["Kernel_Name"], r["Counter_Name"])
                v = float(r["Counter_Value"])
                a = acc[k]
                a[0] += 1
                a[1] += v
                a[2] = min(a[2], v)
                a[3] = max(a[3], v)
        for (name, ctr), a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            rows_out.append(["pmc", name, ctr, a[0], a[1] / a[0], a[2], a[3], ""])
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        acc = defaultdict(lambda: [0, 0.0, float("inf"), 0.0])
        with open(f) as fh:
            for r in csv.DictReader(fh):
                d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                a = acc[r["Kernel_Name"]]
                a[0] += 1
                a[1] += d
                a[2] = min(a[2], d)
                a[3] = max(a[3], d)
        for name, a in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            rows_out.append(["kernel_trace", name, "duration_ns", a[0], a[1] / a[0], a[2], a[3], ""])
    with open(dst, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["source", "kernel", "quantity", "dispatches", "mean", "min", "max", "pct_time"])
        w.writerows(rows_out)
    if delete:
        for f in glob.glob(os.path.join(src, "**", "*.csv"), recursive=True):
            os.remove(f)
    print(f"wrote {dst}: {len(rows_out)} rows")


if __name__ == "__main__":
    main()
