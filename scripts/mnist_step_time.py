#!/usr/bin/env python3
"""BASELINE configs[1]: MNIST conditional vanilla GAN, dp_mode=gc sigma=10 bs=600 — D-step time on one MI355X (synthetic data)."""
import os, sys, time, tempfile, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import init_util, options, ops
from csl_gan_amd.trainer import Trainer

B = 600
extra = sys.argv[sys.argv.index("--opt") + 1].split() if "--opt" in sys.argv else []
with contextlib.redirect_stdout(sys.stderr):
    opt = options.parse(["MNIST", "--model", "Vanilla", "--conditional", "-dpm", "gc", "--sigma", "10", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0",
                         "-o", tempfile.mkdtemp(prefix="cslgan_mnist_"), "--manual_seed", "1", "--synthetic"] + extra)
    G, D = init_util.init_models(opt)
    tr = Trainer(opt, G, D, log_to=os.path.join(opt.output_dir if hasattr(opt, "output_dir") else tempfile.mkdtemp(), "log.csv"))
    tr.setup_privacy_engine()
g = torch.Generator().manual_seed(3)
img = torch.rand(B, 1, 28, 28, generator=g).cuda()
lab = torch.randint(0, 10, (B,), generator=g).cuda()


GRAPH = "--graph" in sys.argv
if GRAPH:
    from csl_gan_amd.trainer import GraphedDStep
    gstep = GraphedDStep(tr)


def step():
    if GRAPH:
        gstep(img, lab)
        return
    tr.train_D(img, lab, tr.gen_z(B), lab, use_dp=True)
    tr.dev_stats.clear()


for _ in range(5):
    step()
torch.cuda.synchronize()
timer = ops.LaunchTimer()
if not GRAPH:
    ops.set_launch_timer(timer)
t0 = time.perf_counter()
N = 30
for _ in range(N):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
ops.set_launch_timer(None)
print("MNIST vanilla conditional gc bs=%d materialize=%s%s: %.3f ms/step, %.0f images/s" % (B, opt.materialize, " HIP-graph replay" if GRAPH else "", dt * 1e3, B / dt))
for k, v in sorted(timer.summary(by_shape=True).items(), key=lambda kv: -kv[1]["ms"])[:10]:
    print("  %-66s %7.3f ms/step %5.1f x" % (k, v["ms"] / N, v["n"] / N))
