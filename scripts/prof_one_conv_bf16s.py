#!/usr/bin/env python3
"""Run ONE conv shape of the bf16 storage mode a few times (for rocprofv3 --pmc passes, scripts/pmc_kernel.sh).
usage: prof_one_conv_bf16s.py fwd|dgrad|wgrad N H W C K R stride pad [group]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
kind = sys.argv[1]
N, H, W, C, K, R, s, p = [int(v) for v in sys.argv[2:10]]
grp = int(sys.argv[10]) if len(sys.argv) > 10 else 1
P, Q = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
x = torch.randn(N, H, W, C, device="cuda").to(torch.bfloat16)
gy = torch.randn(N, P, Q, K, device="cuda").to(torch.bfloat16)
w = torch.randn(K, R, R, C, device="cuda") * 0.05
for _ in range(5):
    if kind == "fwd":
        ops.conv2d_fwd(x, w, None, stride=s, pad=p, act=1)
    elif kind == "dgrad":
        ops.conv2d_dgrad(gy, w, (H, W), stride=s, pad=p, mask=x)
    else:
        ops.conv2d_wgrad_grouped(gy, x, R, R, stride=s, pad=p, group=grp, sq=torch.zeros(N // grp, device="cuda"))
torch.cuda.synchronize()
