#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per kernel and counter."""
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(set)
for pat in sys.argv[1:]:
    for fn in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"]
            k = "k_" + k.split("k_")[1].split("(")[0].split("<")[0] if "k_" in k else k[:40]
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
for k, d in tot.items():
    print(k, "dispatches", len(calls[k]))
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v:18.0f}")
