#!/bin/bash
# Counter passes behind the roofline ceilings of bench.py, run on the GPU box from the repository root:
#   bash tools/roofline_passes.sh <workload> <spp> <name>
# Every pass is its own rocprofv3 run of one production render (tools/profile_step.py) with the kernel trace
# beside the counters, so that counter sums and kernel durations come from the same dispatches; the same
# passes run over tools/ubench/peaks, whose kernels define the ceilings.  Summary: gpurun_out/<name>/roofline.json
# (copy to profiles/ to keep it).
set -e
export TMPDIR=/tmp
W=${1:-dragon}; SPP=${2:-8}; O=gpurun_out/${3:-roofline}
rm -rf $O && mkdir -p $O
SQ="SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_SALU GRBM_GUI_ACTIVE"
TEX="TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
pass() {   # pass <name> <counters> -- <program...>
    local name=$1; local counters=$2; shift 3
    timeout -s KILL 170 rocprofv3 --kernel-trace --pmc $counters --output-format csv -d $O/$name -o $name -- "$@" > $O/$name.log 2>&1
    echo "pass $name done"
}
pass sq "$SQ" -- python3 tools/profile_step.py $W $SPP $O/step.json
pass tex "$TEX" -- python3 tools/profile_step.py $W $SPP
pass fetch "FETCH_SIZE" -- python3 tools/profile_step.py $W $SPP
pass write "WRITE_SIZE" -- python3 tools/profile_step.py $W $SPP
pass peaks_sq "$SQ" -- tools/ubench/peaks
pass peaks_tex "$TEX" -- tools/ubench/peaks
timeout -k 10 60 tools/ubench/peaks > $O/peaks.json
python3 tools/roofline_summary.py $O $W $SPP > $O/roofline.json
rm -rf $O/sq $O/tex $O/fetch $O/write $O/peaks_sq $O/peaks_tex
cat $O/roofline.json
