#!/usr/bin/env python3
"""Merge the rocprofv3 passes of tools/roofline_passes.sh into one JSON: per kernel, the dispatch count, the
summed duration (kernel trace of the same run) and the summed counters; for k_trace the derived per-segment
quantities and utilisations that bench.py turns into roofline ceilings.

FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies each 128-byte fabric read as 64 bytes,
so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections, csv, glob, json, os, sys

root, workload, spp = sys.argv[1], sys.argv[2], int(sys.argv[3])


def short(name):
    if "k_" in name:
        tail = name.split("k_", 1)[1]
        base = "k_" + tail.split("(")[0].split("<")[0]
        if base == "k_trace" and "<" in tail:
            base += "<" + tail.split("<", 1)[1].split(">")[0].replace(" ", "") + ">"
        return base
    return name[:40]


def load(pass_name):
    counters = collections.defaultdict(lambda: collections.defaultdict(float))
    dispatches = collections.defaultdict(set)
    for fn in glob.glob(os.path.join(root, pass_name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            k = short(r["Kernel_Name"])
            counters[k][r["Counter_Name"]] += float(r["Counter_Value"])
            dispatches[k].add(r["Dispatch_Id"])
    durations = collections.defaultdict(float)
    for fn in glob.glob(os.path.join(root, pass_name, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            durations[short(r["Kernel_Name"])] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-6
    return {k: {"dispatches": len(dispatches[k]), "ms": durations.get(k, 0.0), **counters[k]} for k in counters}


out = {"workload": workload, "spp": spp,
       "command": f"rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/profile_step.py {workload} {spp} (one pass per counter group)"}
step_file = os.path.join(root, "step.json")
step = json.load(open(step_file)) if os.path.exists(step_file) else {}
out["step"] = step
passes = {p: load(p) for p in ("sq", "tex", "fetch", "write", "peaks_sq", "peaks_tex")}
kernels = sorted({k for p in ("sq", "tex", "fetch", "write") for k in passes[p] if k.startswith("k_")})
out["kernels"] = {k: {p: passes[p].get(k, {}) for p in ("sq", "tex", "fetch", "write")} for k in kernels}
out["peak_kernels"] = {k: {p: passes[p].get(k, {}) for p in ("peaks_sq", "peaks_tex")}
                       for k in sorted(set(passes["peaks_sq"]) | set(passes["peaks_tex"]))}
peaks_file = os.path.join(root, "peaks.json")
if os.path.exists(peaks_file):
    out["peaks_unprofiled"] = json.load(open(peaks_file))

trace = [k for k in kernels if k.startswith("k_trace")]
if trace and step.get("segments"):
    k = trace[0]
    sq, tex, fe, wr = (passes[p].get(k, {}) for p in ("sq", "tex", "fetch", "write"))
    seg = float(step["segments"])
    d = {"kernel": k}
    if sq:
        d["valu_wave_instr_per_segment"] = sq["SQ_INSTS_VALU"] / seg
        d["active_lanes_per_valu_instr"] = sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_INSTS_VALU"]
        d["vmem_read_wave_instr_per_segment"] = sq["SQ_INSTS_VMEM_RD"] / seg
        d["ms_in_sq_pass"] = sq["ms"]
        d["valu_wave_instr_per_s_in_pass"] = sq["SQ_INSTS_VALU"] / (sq["ms"] * 1e-3)
    if tex:
        d["ta_busy_cycles_per_segment"] = tex["TA_TA_BUSY_sum"] / seg
        d["ta_busy_per_gui_cycle"] = tex["TA_TA_BUSY_sum"] / tex["GRBM_GUI_ACTIVE"]
        d["td_busy_per_gui_cycle"] = tex["TD_TD_BUSY_sum"] / tex["GRBM_GUI_ACTIVE"]
        d["l1_accesses_per_segment"] = tex["TCP_TOTAL_CACHE_ACCESSES_sum"] / seg
        d["l1_miss_rate"] = tex["TCP_TCC_READ_REQ_sum"] / tex["TCP_TOTAL_CACHE_ACCESSES_sum"]
        d["l2_hit_rate"] = tex["TCC_HIT_sum"] / max(tex["TCC_HIT_sum"] + tex["TCC_MISS_sum"], 1.0)
        d["ms_in_tex_pass"] = tex["ms"]
    if fe and wr:
        d["hbm_bytes_per_segment"] = (2.0 * fe["FETCH_SIZE"] + wr["WRITE_SIZE"]) * 1024.0 / seg
        d["hbm_fetch_bytes_per_segment"] = 2.0 * fe["FETCH_SIZE"] * 1024.0 / seg
        d["hbm_write_bytes_per_segment"] = wr["WRITE_SIZE"] * 1024.0 / seg
    out["k_trace_per_segment"] = d
    # ceilings from the calibration kernels, same counters, same tool
    pk = passes["peaks_sq"].get("k_valu_fma", {})
    if pk:
        out["peak_valu_wave_instr_per_s_profiled"] = pk["SQ_INSTS_VALU"] / (pk["ms"] * 1e-3)
        out["peak_valu_instr_per_busy_cu_cycle"] = pk["SQ_INSTS_VALU"] / pk["SQ_BUSY_CU_CYCLES"]
    pg = passes["peaks_tex"].get("k_gather16", {})
    if pg:
        out["peak_gather_ta_busy_per_gui_cycle"] = pg["TA_TA_BUSY_sum"] / pg["GRBM_GUI_ACTIVE"]
        out["peak_gather_td_busy_per_gui_cycle"] = pg["TD_TD_BUSY_sum"] / pg["GRBM_GUI_ACTIVE"]
json.dump(out, sys.stdout, indent=1)
