#!/bin/bash
# Same-box A/B/A/B of one environment switch (default CSLGAN_GP_STREAM; values 0 1): headline (HIP-graph) and eager step times.
# usage (GPU box): bash scripts/ab_env.sh CSLGAN_CLIP_STREAM
VAR=${1:-CSLGAN_GP_STREAM}
for rep in 1 2; do
  for v in 0 1; do
    env $VAR=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v %s %.3f ms  eager %s ms' % (d['config']['launch'], d['ms_per_step'], (d['variants'].get('eager') or {}).get('ms_per_step')))"
  done
done
