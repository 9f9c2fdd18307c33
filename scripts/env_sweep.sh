#!/bin/bash
# Same-box A/B of kernel-selection thresholds (environment variables read by the library): each line is one bench run.
# usage (GPU box, repo root): bash scripts/env_sweep.sh OUTDIR "VAR=val VAR2=val" "VAR=val" ...
OUT=$1; shift
mkdir -p $OUT
i=0
for e in "BASE=1" "$@" "BASE=2"; do
  i=$((i+1))
  env $e timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --loop-steps 0 --dump-shapes $OUT/shapes_$i.txt > $OUT/bench_$i.json 2>/dev/null
  python - "$e" $OUT/bench_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print("%-50s %8.1f img/s  %7.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"]))
PY
done
