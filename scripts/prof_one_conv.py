#!/usr/bin/env python3
"""Run ONE conv shape a few times (for rocprofv3 --pmc passes).  usage: prof_one_conv.py N H W C K R stride pad [compute_dtype]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops
N, H, W, C, K, R, s, p = [int(v) for v in sys.argv[1:9]]
ops.set_compute_dtype(sys.argv[9] if len(sys.argv) > 9 else "fp32")
x = torch.randn(N, H, W, C, device="cuda"); w = torch.randn(K, R, R, C, device="cuda") * 0.05
for _ in range(5):
    ops.conv2d_fwd(x, w, None, stride=s, pad=p)
torch.cuda.synchronize()
