#!/usr/bin/env python3
"""Data gradient of the critic's first conv (64 -> 3 channels out, 5x5 stride 2): device time per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
for (N, HW) in ((128, 64), (128, 128)):
    g = torch.Generator().manual_seed(1)
    w = (torch.randn(64, 5, 5, 3, generator=g) * 0.1).cuda()
    gy = torch.randn(N, HW // 2, HW // 2, 64, generator=g).cuda()
    f = lambda: ops.conv2d_dgrad(gy, w, (HW, HW), stride=2, pad=2)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = ops.LaunchTimer(); ops.set_launch_timer(t)
    for _ in range(20):
        f()
    torch.cuda.synchronize(); ops.set_launch_timer(None)
    for k, v in t.summary(by_kernel=True).items():
        print("ALL=%s N%d %dx%d %s: %.1f us" % (os.environ.get("CSLGAN_SKINNY_ALL", "1"), N, HW, HW, k, v["ms"] / v["n"] * 1e3))
