#!/usr/bin/env python3
"""Durations of the k_trace and k_shade launches of one render, in launch order, from a rocprofv3 kernel trace.
Usage: per_launch.py <dir with *kernel_trace.csv>"""
import csv, glob, os, sys
rows = []
for fn in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        n = r["Kernel_Name"]
        if "k_trace" in n or "k_shade" in n:
            rows.append((int(r["Start_Timestamp"]), "trace" if "k_trace" in n else "shade", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
t = [d for _, k, d in rows if k == "trace"]
s = [d for _, k, d in rows if k == "shade"]
print("k_trace ms per bounce:", " ".join(f"{x:.2f}" for x in t), " total", f"{sum(t):.2f}")
print("k_shade ms per bounce:", " ".join(f"{x:.2f}" for x in s), " total", f"{sum(s):.2f}")
