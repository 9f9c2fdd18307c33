#!/usr/bin/env python3
"""Launch time of the critic's stride-2 data gradients (B=128 and the fused 384 rows) under the current
CSLGAN_HALO_* / CSLGAN_KC_* environment; one line per shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
from conv_microbench import timeit

tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("CSLGAN_"))
out = []
for N in (128, 384):
    for H, C, K in ((32, 64, 128), (16, 128, 256), (8, 256, 512)):
        gy = torch.randn(N, H // 2, H // 2, K, device="cuda")
        w = torch.randn(K, 5, 5, C, device="cuda") * 0.05
        t = timeit(lambda: ops.conv2d_dgrad(gy, w, (H, H), stride=2, pad=2))
        fl = 2.0 * N * (H // 2) ** 2 * K * C * 25
        out.append("N%d %dx%d %.3f ms %.0f TF" % (N, H, H, t, fl / t / 1e9))
print("[%s] %s" % (tag or "default", " | ".join(out)))
