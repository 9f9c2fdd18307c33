#!/usr/bin/env python3
"""The fused pass's conv2 weight-gradient launch (384 rows = norms / dense / private blocks of 128) on igemm_wgh: device time."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
N, HW, C, K = 384, 32, 64, 128
g = torch.Generator().manual_seed(1)
x = torch.randn(N, HW, HW, C, generator=g).cuda()
gy = torch.randn(N, HW // 2, HW // 2, K, generator=g).cuda()
L = K * 25 * C
sq_a, sq_c = torch.zeros(128, device="cuda"), torch.zeros(128, device="cuda")
gw_b, gw_c = torch.empty(128, L, device="cuda"), torch.empty(128, L, device="cuda")
f = lambda: ops.conv2d_wgrad_blocks(gy, x, 5, 5, 2, 2, 128.0, [(128, None, sq_a), (128, gw_b, None), (128, gw_c, sq_c)])
for _ in range(3):
    f()
torch.cuda.synchronize()
t = ops.LaunchTimer(); ops.set_launch_timer(t)
for _ in range(10):
    f()
torch.cuda.synchronize(); ops.set_launch_timer(None)
for k, v in t.summary(by_kernel=True).items():
    print("%s: %.1f us  %.1f TF" % (k, v["ms"] / v["n"] * 1e3, v["flop"] / v["ms"] / 1e9))
