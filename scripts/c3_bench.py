#!/usr/bin/env python3
"""First-layer (3 -> 64, 5x5, stride 2) forward and per-sample weight gradient: time per launch at the step's shapes.
CSLGAN_C3=0 in the environment selects the previous path (zero-padded 4th channel on the generic kernels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops

for N, HW in ((128, 64), (384, 64), (128, 128)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, HW, HW, 3, generator=g).cuda()
    w = (torch.randn(64, 5, 5, 3, generator=g) * 0.1).cuda()
    b = torch.randn(64, generator=g).cuda()
    gy = torch.randn(N, HW // 2, HW // 2, 64, generator=g).cuda()
    sq = torch.zeros(N, device="cuda")

    def run(f, n=20):
        """device time of the launches of one call (HIP events around every launch, host overhead excluded)"""
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        timer = ops.LaunchTimer()
        ops.set_launch_timer(timer)
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        ops.set_launch_timer(None)
        return sum(v["ms"] for v in timer.summary().values()) / n * 1e3
    t_f = run(lambda: ops.conv2d_fwd(x, w, b, stride=2, pad=2, act=ops.ACT_LRELU02))
    t_w = run(lambda: ops.conv2d_wgrad_grouped(gy, x, 5, 5, stride=2, pad=2, group=1, alpha=float(N), sq=sq))
    t_n = run(lambda: ops.conv2d_wgrad_grouped(gy, x, 5, 5, stride=2, pad=2, group=1, alpha=float(N), want_gw=False, sq=sq))
    t_d = run(lambda: ops.conv2d_wgrad_dense(gy, x, 5, 5, stride=2, pad=2, alpha=1.0))
    print("C3=%s N%d %dx%d: fwd %.1f us (%.0f GB/s out)  wgrad per-sample %.1f us  norms-only %.1f us  dense %.1f us" % (
        os.environ.get("CSLGAN_C3", "1"), N, HW, HW, t_f, gy.numel() * 4 / t_f / 1e3, t_w, t_n, t_d))
