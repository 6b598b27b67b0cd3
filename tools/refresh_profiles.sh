#!/bin/bash
# Run on the GPU box from the repository root: bench line, kernel-trace statistics and the two HBM-traffic
# counter passes of the default bench workload.  Results land in gpurun_out/refresh/; copy what should be
# kept into profiles/.
set -e
export TMPDIR=/tmp
O=gpurun_out/refresh
rm -rf $O && mkdir -p $O
CMD="bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -k 10 240 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o fetch -- python3 $CMD > $O/fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o write -- python3 $CMD > $O/write.log 2>&1
python3 tools/hbm_traffic.py $O/fetch $O/write "python3 $CMD (cornell 1920x1080x64spp; each pass also contains the untimed work-count step)" > $O/hbm_traffic.json
find $O -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/fetch $O/write
cat $O/bench.json
