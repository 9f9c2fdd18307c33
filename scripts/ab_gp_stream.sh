#!/bin/bash
# Same-box A/B of the gradient-penalty side stream (CSLGAN_GP_STREAM=0/1): eager and HIP-graph step times.  Run on the GPU box.
for rep in 1 2; do
  for v in 0 1; do
    CSLGAN_GP_STREAM=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --loop-steps 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GP_STREAM=$v eager %.3f ms  graph %s ms  fp32_auto %s ms' % (d['ms_per_step'], d['variants']['hip_graph'].get('ms_per_step'), d['variants']['fp32_auto'].get('ms_per_step')))"
  done
done
