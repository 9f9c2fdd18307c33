#!/bin/bash
# Run on the GPU box from the repo root: kernel statistics and HBM-traffic counters of the benchmark command.
# Outputs land in gpurun_out/prof_final/; copy the summaries into profiles/ afterwards.
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/${PROF_TAG:-prof_r02}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --loop-steps 0 > $OUT/stats_bench.json 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --loop-steps 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --loop-steps 0 > /dev/null 2>&1
cd $ROOT
python3 scripts/pmc_traffic.py $OUT/fetch/fetch_counter_collection.csv $OUT/write/write_counter_collection.csv 4 $OUT/pmc_traffic.json
rm -f $OUT/fetch/*kernel_trace.csv $OUT/write/*kernel_trace.csv $OUT/stats/*kernel_trace.csv $OUT/fetch/fetch_counter_collection.csv $OUT/write/write_counter_collection.csv
ls -la $OUT $OUT/stats
