#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 result: either the rocpd
SQLite database or the *_kernel_trace.csv it writes with --output-format csv.
    python tools/rocpd_stats.py gpurun_out/prof > profiles/rNN_kernel_stats.csv"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    c = sqlite3.connect(path)
    return [(n, s, e) for n, s, e in c.execute("select name, start, end from kernels")]


def from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out


def main():
    root = sys.argv[1]
    rows = []
    for p in glob.glob(os.path.join(root, "**", "*.db"), recursive=True):
        rows += from_db(p)
    if not rows:
        for p in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
            rows += from_csv(p)
    agg = defaultdict(list)
    for n, s, e in rows:
        agg[n].append(e - s)
    tot = sum(sum(v) for v in agg.values())
    w = csv.writer(sys.stdout)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage"])
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([n, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v), round(100.0 * sum(v) / tot, 2)])


if __name__ == "__main__":
    main()
