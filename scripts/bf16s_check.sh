#!/bin/bash
# GPU box, repo root: the bf16-storage tests, then the configs[4] bench once per environment setting given (same-box A/B).
# usage: scripts/bf16s_check.sh <tag> [VAR=VAL ...]      (no settings: one run with the defaults)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_bf16s_gpu.py -q > $OUT/test.log 2>&1
grep -E "^(FAILED|ERROR)|passed|failed" $OUT/test.log | head -20
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --loop-steps 0 --no-variants"
O="--compute_dtype bf16 --storage_dtype bf16 --im_size 128"
run() {  # $1 = label, $2 = env assignment or ""
  env $2 timeout -k 10 200 $B --dump-shapes $OUT/shapes_$1.txt --opt "$O" > $OUT/bench_$1.json 2> $OUT/err_$1.txt
  python3 -c "
import json,sys
d=json.loads(open('$OUT/bench_$1.json').read().strip().splitlines()[-1]); print('$1 [$2]:', d['value'], 'images/s', d['ms_per_step'], 'ms', d['config']['launch'], d['graph_error'] or '')"
}
if [ $# -eq 0 ]; then run default ""; else i=0; for kv in "$@"; do i=$((i+1)); run s$i "$kv"; done; fi
