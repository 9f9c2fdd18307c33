#!/bin/bash
# Hardware counters of one kernel: separate rocprofv3 --pmc passes (kernel-trace only) around a python command, summarised per
# counter for the kernels whose name contains $1.   usage (GPU box, repo root): scripts/pmc_kernel.sh <kernel substring> <tag> script.py [args]
set -e
PAT=$1; TAG=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o p$i -- python3 $ROOT/"$@" > $OUT/run$i.log 2>&1 || echo "pass $i ($set) failed"
done
cd $ROOT
python3 - "$PAT" $OUT <<'PY'
import csv, glob, sys, collections
pat, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(out + "/*counter_collection.csv")):
    seen = set()
    for r in csv.DictReader(open(f)):
        if pat not in r["Kernel_Name"]:
            continue
        k = r["Counter_Name"]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
        key = (f, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[f][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); dur[f][1] += 1
with open(out + "/summary.txt", "w") as fo:
    for f, (t, n) in dur.items():
        fo.write("%s: %d dispatches, mean %.1f us\n" % (f.split("/")[-1], n, t / max(n, 1) / 1e3))
    for k, (v, n) in sorted(agg.items()):
        fo.write("%-28s mean per dispatch %.4g  (n=%d)\n" % (k, v / max(n, 1), n))
print(open(out + "/summary.txt").read())
PY
rm -f $OUT/*kernel_trace.csv $OUT/*agent_info.csv
