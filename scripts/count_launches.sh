#!/bin/bash
# GPU box, repo root: rocprofv3 kernel statistics of the default bench command -> dispatches per step, clip launches per step.
# usage: scripts/count_launches.sh <out dir under gpurun_out> [bench --opt string]
set -e
R=$(pwd); OUT=$R/gpurun_out/$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ -n "$2" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --loop-steps 0 --opt "$2" > $OUT/bench.json 2> $OUT/err.txt
else
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-variants --loop-steps 0 > $OUT/bench.json 2> $OUT/err.txt
fi
cd $R
cp $(ls $OUT/stats/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
rm -rf $OUT/stats
python3 - $OUT <<'PY'
import csv, json, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
n = int([r for r in rows if "adam_multi" in r["Name"]][0]["Calls"])
tot = sum(int(r["Calls"]) for r in rows)
own = sum(int(r["Calls"]) for r in rows if "cslgan" in r["Name"])
aten = sum(int(r["Calls"]) for r in rows if "at::" in r["Name"])
print("steps %d: %.1f dispatches/step (%.1f hand-written, %.1f ATen, %.1f other), %.3f ms of kernels per step" % (
    n, tot / n, own / n, aten / n, (tot - own - aten) / n, sum(float(r["TotalDurationNs"]) for r in rows) / n / 1e6))
for r in rows:
    if "clip_accum_noise" in r["Name"]:
        print("%s: %.1f launches/step, avg %.1f us" % (r["Name"][:60], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3))
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("bench under rocprofv3: %.1f images/s, %.3f ms/step (%s); roofline_hbm %s" % (d["value"], d["ms_per_step"], d["config"]["launch"], json.dumps({k: d["roofline_hbm"][k] for k in ("achieved", "frac", "launches_per_step", "MB_per_step")})))
PY
