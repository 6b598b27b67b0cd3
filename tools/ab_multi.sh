#!/bin/bash
# A/B timing of several builds of the library on the GPU box: tools/ab_multi.sh <tag> <lib1.so> <lib2.so> ...
# (tools/bench_kernels.py on two scenes for each library, then for the default build; results in gpurun_out/<tag>/ab.txt)
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for L in "$@"; do
  RT_AMD_LIB=$L timeout -k 10 200 python -u tools/bench_kernels.py dragon cornell 2>&1 | grep "x16:" | sed "s|^|$(basename $L) |" | tee -a gpurun_out/$TAG/ab.txt || exit 1
done
timeout -k 10 200 python -u tools/bench_kernels.py dragon cornell 2>&1 | grep "x16:" | sed "s|^|default |" | tee -a gpurun_out/$TAG/ab.txt
