#!/usr/bin/env python3
"""Run ONE conv shape a few times (for rocprofv3 --pmc passes).
usage: prof_one_conv.py N H W C K R stride pad [compute_dtype] [kind: fwd | dgrad | wgrad] [group]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
N, H, W, C, K, R, s, p = [int(v) for v in sys.argv[1:9]]
ops.set_compute_dtype(sys.argv[9] if len(sys.argv) > 9 else "fp32")
kind = sys.argv[10] if len(sys.argv) > 10 else "fwd"
group = int(sys.argv[11]) if len(sys.argv) > 11 else 1
x = torch.randn(N, H, W, C, device="cuda"); w = torch.randn(K, R, R, C, device="cuda") * 0.05
P = (H + 2 * p - R) // s + 1
gy = torch.randn(N, P, P, K, device="cuda")
for _ in range(5):
    if kind == "fwd":
        ops.conv2d_fwd(x, w, None, stride=s, pad=p, act=1, wkey=("prof", 1))
    elif kind == "dgrad":
        ops.conv2d_dgrad(gy, w, (H, W), stride=s, pad=p, wkey=("prof", 1))
    else:
        ops.conv2d_wgrad_grouped(gy, x, R, R, stride=s, pad=p, group=group)
torch.cuda.synchronize()
