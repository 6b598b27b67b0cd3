#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE in one run, WRITE_SIZE in another) of the same command
into HBM bytes per launch for every kernel.

    python tools/hbm_traffic.py <fetch_dir> <write_dir> "<command that was profiled>" > hbm_traffic.json

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies each 128-byte fabric read request
as 64 bytes, so it is doubled; WRITE_SIZE is exact."""
import collections, csv, glob, json, os, sys


def per_kernel(directory, counter):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for fn in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            k = "k_" + k.split("k_")[1].split("(")[0].split("<")[0] if "k_" in k else k[:48]
            tot[k] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in disp.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in fetch:
    if not k.startswith("k_") or k not in write:
        continue
    out[k] = {"FETCH_SIZE_KB_total": fetch[k], "FETCH_SIZE_dispatches": nf[k],
              "WRITE_SIZE_KB_total": write[k], "WRITE_SIZE_dispatches": nw[k],
              "hbm_bytes_per_launch": (2.0 * fetch[k] / nf[k] + write[k] / nw[k]) * 1024.0}
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, command: " + sys.argv[3] +
                ". hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / dispatches.")
json.dump(out, sys.stdout, indent=1)
