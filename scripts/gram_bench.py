#!/usr/bin/env python3
"""Per-sample norms of the critic's conv3 weight gradient (16x16x128 -> 8x8x256, 5x5 stride 2) at bs=128: the Gram kernels
against the norm-only product kernel, and the clip-weighted dense sum (ghost clipping's second half).  Device time per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
N = 128
g = torch.Generator().manual_seed(1)
x = torch.randn(N, 16, 16, 128, generator=g).cuda()
gy = torch.randn(N, 8, 8, 256, generator=g).cuda()
sq = torch.zeros(N, device="cuda")
f = torch.rand(2 * N, generator=g).cuda()
x2, gy2 = torch.cat([x, x]), torch.cat([gy, gy])
cases = {
    "gram norms": lambda: ops.conv2d_wgrad_sqnorm_gram(gy, x, 5, 5, stride=2, pad=2, alpha=float(N), sq=sq),
    "product norms-only": lambda: ops.conv2d_wgrad_grouped(gy, x, 5, 5, stride=2, pad=2, group=1, alpha=float(N), want_gw=False, sq=sq),
    "product per-sample": lambda: ops.conv2d_wgrad_grouped(gy, x, 5, 5, stride=2, pad=2, group=1, alpha=float(N), sq=sq),
    "scaled dense, 256 rows": lambda: ops.conv2d_wgrad_dense(gy2, x2, 5, 5, stride=2, pad=2, alpha=1.0, row_scale=f),
    "dense, 128 rows": lambda: ops.conv2d_wgrad_dense(gy, x, 5, 5, stride=2, pad=2, alpha=1.0),
}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = ops.LaunchTimer(); ops.set_launch_timer(t)
    for _ in range(10):
        fn()
    torch.cuda.synchronize(); ops.set_launch_timer(None)
    tot = sum(v["ms"] for v in t.summary().values()) / 10 * 1e3
    print("%-24s %7.1f us   %s" % (name, tot, ", ".join("%s %.1f" % (k, v["ms"] / 10 * 1e3) for k, v in t.summary(by_kernel=True).items())))
