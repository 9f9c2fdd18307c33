#!/usr/bin/env python3
"""Launch time of the weight-gradient kernel on the shapes of the D-step (per-sample, dense) and the G-step (dense)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
from conv_microbench import timeit

B = 128
#        name, N, H, C, K, R, stride, pad, group, upsample
shapes = [("D c2 g2", B, 32, 64, 128, 5, 2, 2, 2, False), ("D c3 g8", B, 16, 128, 256, 5, 2, 2, 8, False),
          ("D c4 g16", B, 8, 256, 512, 5, 2, 2, 16, False), ("D c2 g1", B, 32, 64, 128, 5, 2, 2, 1, False),
          ("D c3 g1", B, 16, 128, 256, 5, 2, 2, 1, False),
          ("G b4c2 g2", B, 64, 64, 64, 5, 1, 2, 2, False), ("G b3c2 g4", B, 32, 128, 128, 5, 1, 2, 4, False),
          ("G b2c2 g16", B, 16, 256, 256, 5, 1, 2, 16, False), ("G b1c2 g16", B, 8, 512, 512, 5, 1, 2, 16, False),
          ("G b4c1 up g4", B, 32, 128, 64, 5, 1, 2, 4, True), ("G b3c1 up g8", B, 16, 256, 128, 5, 1, 2, 8, True)]
for name, N, H, C, K, R, s, p, g, up in shapes:
    x = torch.randn(N, H, H, C, device="cuda")
    P = ops.conv_out_size(H, R, s, p, up)
    gy = torch.randn(N, P, P, K, device="cuda")
    t = timeit(lambda: ops.conv2d_wgrad_grouped(gy, x, R, R, stride=s, pad=p, group=g, upsample=up), iters=10)
    fl = 2.0 * N * P * P * K * C * R * R * (9.0 / 25.0 if up else 1.0)
    print("%-14s %.3f ms %6.1f TF" % (name, t, fl / t / 1e9))
