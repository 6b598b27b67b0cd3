#!/bin/bash
# Run on the GPU box from the repository root: counter passes behind the roofline ceilings, the bench line, and
# the rocprofv3 kernel-trace statistics of that same bench command.  Results land in gpurun_out/refresh_<name>/;
# copy what should be kept into profiles/.
#   bash tools/refresh_profiles.sh [workload] [batch spp] [bsp|bvh]
set -e
export TMPDIR=/tmp
W=${1:-dragon}; SPP=${2:-8}; ACCEL=${3:-bsp}
SUFFIX=""; if [ "$ACCEL" = "bvh" ]; then SUFFIX="_bvh"; export RT_ACCEL=bvh; fi
O=gpurun_out/refresh_$W$SUFFIX
rm -rf $O && mkdir -p $O
bash tools/roofline_passes.sh $W $SPP refresh_$W$SUFFIX/roofline > $O/roofline.log 2>&1
mkdir -p profiles && cp $O/roofline/roofline.json profiles/roofline_$W$SUFFIX.json     # bench.py reads it from there
timeout -k 10 400 python3 bench.py --workload $W --accel $ACCEL > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --workload $W --accel $ACCEL --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $O/kt.json 2> $O/kt.log
find $O/kt -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
cat $O/bench.json
