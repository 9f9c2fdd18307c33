#!/bin/bash
# Same-box A/B/A/B of the round-2 kernel switches: headline D-step time (bench.py, 30 steps) with each switch off / on.
# usage (GPU box, repo root): scripts/ab_switches.sh > gpurun_out/ab_switches.txt
run() { env "$@" python bench.py --no-variants --no-cpu-baseline --loop-steps 0 --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms  %.0f images/s' % (d['ms_per_step'], d['value']))"; }
echo "all switches at their defaults:"; run X=1; run X=1
for kv in CSLGAN_C3=0 CSLGAN_GHOST_MAX_PIX=16 CSLGAN_GRAM_CLS64=0 CSLGAN_SKINNY_ALL=0 CSLGAN_LINEAR_K1=0; do
  echo "$kv (off) / default (on), interleaved:"
  for i in 1 2; do echo -n "  off: "; run $kv; echo -n "  on:  "; run X=1; done
done
echo "everything off (round-2 kernels disabled):"; run CSLGAN_C3=0 CSLGAN_GHOST_MAX_PIX=16 CSLGAN_SKINNY_ALL=0 CSLGAN_LINEAR_K1=0; run CSLGAN_C3=0 CSLGAN_GHOST_MAX_PIX=16 CSLGAN_SKINNY_ALL=0 CSLGAN_LINEAR_K1=0
