#!/bin/bash
# Two rocprofv3 counter passes (SQ pipe use; texture path) of one short render.  Usage on the GPU box:
#   bash tools/pmc_passes.sh <workload> <spp> <outdir under gpurun_out>
set -e
export TMPDIR=/tmp
W=${1:-cornell}; SPP=${2:-8}; O=gpurun_out/${3:-pmc}
rm -rf $O && mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD --output-format csv -d $O/sq -o sq -- python3 tools/profile_step.py $W $SPP > $O/sq.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/tex -o tex -- python3 tools/profile_step.py $W $SPP > $O/tex.log 2>&1
python3 tools/pmc_summary.py "$O/sq/**/*counter_collection.csv" > $O/sq.txt
python3 tools/pmc_summary.py "$O/tex/**/*counter_collection.csv" > $O/tex.txt
rm -rf $O/sq $O/tex
grep -A9 k_trace $O/sq.txt; grep -A9 k_trace $O/tex.txt
