#!/usr/bin/env python3
"""Wall time of the phases of one G step (train.py:502-517) at the headline config, plus per-entry kernel time."""
import os, sys, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from csl_gan_amd import ops, util

with contextlib.redirect_stdout(sys.stderr):
    opt, tr, img = bench.build_trainer(0, 1, 0)
B = img.shape[0]
D, G = tr.D, tr.G
acc = {}


class T:
    def __init__(self, name): self.name = name
    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True); self.e = torch.cuda.Event(enable_timing=True); self.s.record()
    def __exit__(self, *a):
        self.e.record(); acc.setdefault(self.name, []).append((self.s, self.e))


def step():
    util.zero_grad(G); util.freeze(D)
    z = tr.gen_z(B)
    with T("G forward (grad mode)"):
        fake = G(z, None)
    with T("D forward"):
        out, _ = D(fake, None)
        loss = G.loss(out, opt.d_device)
    with T("backward (D dgrad + G dgrad/wgrad/norm)"):
        loss.backward()
    util.unfreeze(D)
    with T("Adam (G)"):
        tr.g_optimizer.step()


for _ in range(3):
    step()
acc.clear()
torch.cuda.synchronize()
timer = ops.LaunchTimer(); ops.set_launch_timer(timer)
N = 5
for _ in range(N):
    step()
torch.cuda.synchronize()
ops.set_launch_timer(None)
tot = 0
for k, v in acc.items():
    ms = sum(s.elapsed_time(e) for s, e in v) / len(v)
    tot += ms
    print("%-46s %7.3f ms" % (k, ms))
print("%-46s %7.3f ms" % ("sum", tot))
for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1]["ms"]):
    print("  %-34s %7.3f ms/step  %5.1f launches  exec %6.1f TF" % (k, v["ms"] / N, v["n"] / N, v["exec_flop"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] else 0))
