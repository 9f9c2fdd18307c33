#!/usr/bin/env python3
"""Can this PyTorch / RCCL build record an all-reduce into a HIP graph?  One rank (RCCL communicator of size 1) on one GPU:
capture all_reduce(SUM) between two elementwise kernels, replay three times, check the arithmetic."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.ones(1 << 20, device="cuda")
y = torch.zeros_like(x)
dist.all_reduce(x)                 # communicator set-up outside the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    x.mul_(2.0)
    dist.all_reduce(x)
    y.copy_(x).add_(1.0)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("captured all_reduce replayed: x=%g y=%g (expect 8, 9)" % (x[0].item(), y[0].item()))
dist.destroy_process_group()
