#!/bin/bash
# Run on the GPU box from the repository root: counter passes behind the roofline ceilings, the default bench
# line, and the rocprofv3 kernel-trace statistics of that same bench command.  Results land in
# gpurun_out/refresh/; copy what should be kept into profiles/.
#   bash tools/refresh_profiles.sh [workload] [batch spp]
set -e
export TMPDIR=/tmp
W=${1:-dragon}; SPP=${2:-8}
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
bash tools/roofline_passes.sh $W $SPP refresh/roofline > $O/roofline.log 2>&1
mkdir -p profiles && cp $O/roofline/roofline.json profiles/roofline_$W.json     # bench.py reads it from there
timeout -k 10 400 python3 bench.py --workload $W > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $O/kt.json 2> $O/kt.log
find $O/kt -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
cat $O/bench.json
