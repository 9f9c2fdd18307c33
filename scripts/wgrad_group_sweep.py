#!/usr/bin/env python3
"""Dense weight gradient: launch time vs samples-per-slab (group) — block-count quantisation over 256 CUs x 3 workgroups."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
from conv_microbench import timeit

B = 128
shapes = [("D c2", B, 32, 64, 128, 5, 2, 2, False), ("D c3", B, 16, 128, 256, 5, 2, 2, False), ("D c4", B, 8, 256, 512, 5, 2, 2, False),
          ("G b4c2", B, 64, 64, 64, 5, 1, 2, False), ("G b3c2", B, 32, 128, 128, 5, 1, 2, False), ("G b2c2", B, 16, 256, 256, 5, 1, 2, False),
          ("G b1c2", B, 8, 512, 512, 5, 1, 2, False), ("G b4c1up", B, 32, 128, 64, 5, 1, 2, True), ("G b3c1up", B, 16, 256, 128, 5, 1, 2, True),
          ("G b2c1up", B, 8, 512, 256, 5, 1, 2, True), ("G b1c1up", B, 4, 512, 512, 5, 1, 2, True)]
for name, N, H, C, K, R, s, p, up in shapes:
    x = torch.randn(N, H, H, C, device="cuda")
    P = ops.conv_out_size(H, R, s, p, up)
    gy = torch.randn(N, P, P, K, device="cuda")
    res = []
    for g in (1, 2, 4, 8, 16, 32, 64, 128):
        def run():
            slabs = ops.conv2d_wgrad_grouped(gy, x, R, R, stride=s, pad=p, group=g, upsample=up)
            if slabs.shape[0] > 1:
                out = torch.empty(slabs[0].numel(), device="cuda")
                ops.clip_accum_noise([slabs.reshape(slabs.shape[0], -1)], [out])
        t = timeit(run, iters=6, warm=2)
        res.append("g%d %.3f" % (g, t))
    print("%-10s %s" % (name, "  ".join(res)))
