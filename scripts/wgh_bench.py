#!/usr/bin/env python3
"""Per-sample weight gradient of the critic's conv2 / conv3 at bs=128: device time per launch (HIP events around the launch)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
for (N, HW, C, K) in ((128, 32, 64, 128), (128, 16, 128, 256)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, HW, HW, C, generator=g).cuda()
    gy = torch.randn(N, HW // 2, HW // 2, K, generator=g).cuda()
    sq = torch.zeros(N, device="cuda")
    out = torch.empty(N, K, 5, 5, C, device="cuda")
    f = lambda: ops.conv2d_wgrad_grouped(gy, x, 5, 5, stride=2, pad=2, group=1, alpha=float(N), sq=sq, out=out)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = ops.LaunchTimer(); ops.set_launch_timer(t)
    for _ in range(20):
        f()
    torch.cuda.synchronize(); ops.set_launch_timer(None)
    for k, v in t.summary(by_kernel=True).items():
        print("HALF=%s N%d %dx%d C%d K%d %s: %.1f us  %.1f TF" % (os.environ.get("CSLGAN_WGH_HALF", "0"), N, HW, HW, C, K, k, v["ms"] / v["n"] * 1e3, v["flop"] / v["ms"] / 1e9))
