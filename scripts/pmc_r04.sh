#!/bin/bash
# GPU box, repo root: hardware counters (separate rocprofv3 --pmc passes, python directly after --) of the six kernels that hold the
# fp32_auto step's time, one isolated launch shape each -> gpurun_out/pmc_r04.txt (copied to profiles/r04_pmc_x3_kernels.txt).
OUT=gpurun_out/pmc_r04.txt
: > $OUT
run() {   # tag, kernel substring, prof_one_conv args
  tag=$1; pat=$2; shift 2
  echo "==== $tag : kernels matching '$pat' : prof_one_conv.py $@" >> $OUT
  bash scripts/pmc_kernel.sh "$pat" "r04_$tag" scripts/prof_one_conv.py "$@" >> $OUT 2>&1
}
run g_b3_fwd        "igemm_x3h_kernel<128, 3, false, false>" 128 32 32 128 128 5 1 2 bf16x3 fwd
run g_b4_fwd        "igemm_x3h_kernel<64, 3, false, false>"  128 64 64 64 64 5 1 2 bf16x3 fwd
run d_conv3_fwd384  "igemm_x3h_kernel<128, 3, true, false>"  384 16 16 128 256 5 2 2 bf16x3 fwd
run d_conv2_dgrad384 "igemm_x3h_kernel<64, 3, true, false>"  384 32 32 64 128 5 2 2 bf16x3 dgrad
run d_conv2_wgrad384 "igemm_x3w_kernel<2, false>"     384 32 32 64 128 5 2 2 bf16x3 wgrad 1
run d_conv4_wgrad256 "igemm_x3w_kernel<2, true>"      256 8 8 256 512 5 2 2 bf16x3 wgrad 16
tail -5 $OUT
