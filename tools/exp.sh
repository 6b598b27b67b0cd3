#!/bin/bash
# Development loop on the GPU box: the GPU suite, then per-kernel timings of the benchmark scenes.
# Usage: bash tools/exp.sh <tag> [workloads...]
set -o pipefail
TAG=${1:-exp}; shift
W=${@:-cornell dragon}
O=gpurun_out/$TAG
mkdir -p $O
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/pytest.log 2>&1
rc=$?
tail -2 $O/pytest.log
if [ $rc -ne 0 ]; then grep -E "Error|assert|FAILED" $O/pytest.log | head -20; exit $rc; fi
timeout -k 10 300 python -u tools/bench_kernels.py $W 2>&1 | tee $O/kernels.txt | grep -v "triangles, preprocess"
