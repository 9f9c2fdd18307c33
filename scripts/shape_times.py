#!/usr/bin/env python3
"""Per-shape conv launch times of the headline D-step (and of one G step with --g): HIP events per C-ABI launch."""
import os, sys, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from csl_gan_amd import ops

extra = sys.argv[sys.argv.index("--opt") + 1].split() if "--opt" in sys.argv else []
with contextlib.redirect_stdout(sys.stderr):
    opt, tr, img = bench.build_trainer(0, 1, 0, extra=extra)
B = img.shape[0]
g_mode = "--g" in sys.argv


def step():
    if g_mode:
        tr.train_G(tr.gen_z(B), None)
    else:
        tr.train_D(img, None, tr.gen_z(B), None, use_dp=True)
    tr.dev_stats.clear()


for _ in range(3):
    step()
torch.cuda.synchronize()
timer = ops.LaunchTimer(); ops.set_launch_timer(timer)
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
ops.set_launch_timer(None)
tot = 0.0
for k, v in sorted(timer.summary(by_shape=True).items(), key=lambda kv: -kv[1]["ms"]):
    tot += v["ms"] / N
    print("%-66s %7.3f ms/step %5.1f x  exec %6.1f TF" % (k, v["ms"] / N, v["n"] / N, v["exec_flop"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0))
print("sum %.3f ms/step" % tot)
