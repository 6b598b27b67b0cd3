#!/bin/bash
# A/B timing of two builds of the library on the GPU box: tools/ab.sh <tag> <other .so> [workloads...]
set -o pipefail
TAG=$1; OTHER=$2; shift 2
W=${@:-cornell dragon}
O=gpurun_out/$TAG
mkdir -p $O
RT_AMD_LIB=$OTHER timeout -k 10 300 python -u tools/bench_kernels.py $W 2>&1 | grep -v "triangles, preprocess" | sed "s/^/other /" | tee $O/kernels_other.txt || exit 1
timeout -k 10 300 python -u tools/bench_kernels.py $W 2>&1 | grep -v "triangles, preprocess" | sed "s/^/this  /" | tee $O/kernels_this.txt
