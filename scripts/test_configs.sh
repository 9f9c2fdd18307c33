#!/bin/bash
# Synthetic-data counterpart of the reference's test_configs.sh: 8 smoke configurations
# ({MNIST, CelebA} x {gc, is} x {unconditional, conditional}), each a few iterations on one GPU.
set -u
DEV=${1:-cuda:0}
ITERS=${2:-6}
OUT=${3:-/tmp/cslgan_cfgs}
fail=0
for dataset in MNIST CelebA; do
  for pm in gc is; do
    for cond in "" "--conditional"; do
      name="$dataset-$pm-${cond:-uncond}"
      echo "==== $name ===="
      bs=32
      python -m csl_gan_amd.train $dataset -tss 1000 -dpm $pm -nms 1 --mean_sample_size 10 $cond -gd $DEV -dd $DEV \
          -bs $bs --synthetic --max_iters $ITERS --log_every $((bs*3)) -o $OUT/$name --manual_seed 7 > $OUT.$name.log 2>&1
      rc=$?
      tail -4 $OUT.$name.log | cut -c1-220
      if [ $rc -ne 0 ]; then echo "FAILED rc=$rc"; fail=1; fi
    done
  done
done
exit $fail
