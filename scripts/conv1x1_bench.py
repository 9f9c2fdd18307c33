#!/usr/bin/env python3
"""The generator's shortcut convs (1x1 on the depth-to-space tensor) at bs=128: device time per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
for (HW, C, K) in ((64, 32, 64), (32, 64, 128), (16, 128, 256), (8, 128, 512)):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(128, HW, HW, C, generator=g).cuda()
    w = (torch.randn(K, 1, 1, C, generator=g) * 0.1).cuda()
    b = torch.randn(K, generator=g).cuda()
    f = lambda: ops.conv2d_fwd(x, w, b, stride=1, pad=0)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = ops.LaunchTimer(); ops.set_launch_timer(t)
    for _ in range(20):
        f()
    torch.cuda.synchronize(); ops.set_launch_timer(None)
    for k, v in t.summary(by_kernel=True).items():
        us = v["ms"] / v["n"] * 1e3
        print("CONV1X1=%s %dx%d C%d K%d %s: %.1f us  %.0f GB/s" % (os.environ.get("CSLGAN_CONV1X1", "1"), HW, HW, C, K, k, us, 4e-3 * (x.numel() + 128 * HW * HW * K) / us))
