#!/bin/bash
# Run on the GPU box from the repo root (gpurun): kernel statistics, HBM-traffic counters and the bench lines of one code state.
# Outputs land in gpurun_out/$PROF_TAG/; copy the summaries into profiles/ afterwards (scripts/collect_profiles.sh is the recipe the
# committed profiles/r03_* and r04_* files were made with; the default bench command now runs the fp32_auto headline).
set -e
ROOT=$(pwd)
TAG=${PROF_TAG:-prof_r04}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
# 1. the bench line itself (eager region + HIP-graph region + variants + CPU baseline) and the isolated per-shape launch table
python3 bench.py --steps 20 --warmup 5 --dump-shapes $OUT/launch_shapes.txt > $OUT/bench_n1.json 2> $OUT/bench_n1.err
echo "bench done"
# 2. rocprofv3 kernel statistics of the same command (no CPU baseline / variants: the GPU work is the same step)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --loop-steps 0 > $OUT/stats_bench.json 2> /dev/null
echo "stats done"
# 3. fabric traffic: separate PMC passes (never combined with other trace domains)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --loop-steps 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-variants --loop-steps 0 > /dev/null 2>&1
echo "pmc done"
cd $ROOT
# steps seen by the PMC passes: 1 warm-up + 3 eager + (2 graph warm-up + 1 capture + 3 replays) = 11
python3 scripts/pmc_traffic.py $(ls $OUT/fetch/*counter_collection.csv | head -1) $(ls $OUT/write/*counter_collection.csv | head -1) ${PMC_STEPS:-11} $OUT/pmc_traffic.json
cp $(ls $OUT/stats/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/fetch $OUT/write $OUT/stats
# 3b. the exact-fp32 arithmetic of rounds 1-3 on the same box (its own launch table)
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-variants --compute fp32 --dump-shapes $OUT/launch_shapes_fp32_exact.txt > $OUT/bench_fp32_exact.json 2> /dev/null || true
# 4. the other BASELINE configurations on the same harness
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-variants --opt "-dpm is -ispp True" > $OUT/bench_is.json 2> /dev/null || true
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-variants --opt "--compute_dtype bf16 --im_size 128" > $OUT/bench_bf16_128.json 2> /dev/null || true
# BASELINE configs[4] as built in round 3: bf16 STORAGE (csrc/igemm_bf16s.hip) — bench line + launch table, then the kernel statistics of the same command
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-variants --dump-shapes $OUT/launch_shapes_bf16s_128.txt --opt "--compute_dtype bf16 --storage_dtype bf16 --im_size 128" > $OUT/bench_bf16s_128.json 2> /dev/null || true
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats16 -o stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --loop-steps 0 --opt "--compute_dtype bf16 --storage_dtype bf16 --im_size 128" > $OUT/stats_bench_bf16s.json 2> /dev/null) || true
cp $(ls $OUT/stats16/*kernel_stats.csv | head -1) $OUT/kernel_stats_bf16s_128.csv || true
rm -rf $OUT/stats16
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-variants --opt "--im_size 128" > $OUT/bench_fp32_128.json 2> /dev/null || true
python3 scripts/mnist_step_time.py > $OUT/mnist_eager.txt 2> /dev/null || true
python3 scripts/mnist_step_time.py --graph > $OUT/mnist_graph.txt 2> /dev/null || true
ls -la $OUT
