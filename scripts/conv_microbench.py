#!/usr/bin/env python3
"""Per-shape throughput of the conv kernels on the shapes the headline D-step executes (B=128).
Prints executed-FLOP TF/s per launch (HIP events, median of several runs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import ops  # noqa: E402

B = 128


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = "cuda"
    rows = []
    # (name, N, H, W, C, K, R, stride, pad, upsample)
    fwd = [
        ("G b1 convUp", B, 4, 4, 512, 512, 5, 1, 2, True), ("G b1 conv", B, 8, 8, 512, 512, 5, 1, 2, False),
        ("G b2 convUp", B, 8, 8, 512, 256, 5, 1, 2, True), ("G b2 conv", B, 16, 16, 256, 256, 5, 1, 2, False),
        ("G b3 convUp", B, 16, 16, 256, 128, 5, 1, 2, True), ("G b3 conv", B, 32, 32, 128, 128, 5, 1, 2, False),
        ("G b4 convUp", B, 32, 32, 128, 64, 5, 1, 2, True), ("G b4 conv", B, 64, 64, 64, 64, 5, 1, 2, False),
        ("G convOut", B, 64, 64, 64, 3, 3, 1, 1, False),
        ("D conv0", B, 64, 64, 3, 64, 5, 2, 2, False), ("D conv1", B, 32, 32, 64, 128, 5, 2, 2, False),
        ("D conv2", B, 16, 16, 128, 256, 5, 2, 2, False), ("D conv3", B, 8, 8, 256, 512, 5, 2, 2, False),
        ("D linOut", B, 1, 1, 8192, 1, 1, 1, 0, False),
    ]
    for name, N, H, W, C, K, R, s, p, up in fwd:
        x = torch.randn(N, H, W, C, device=dev)
        w = torch.randn(K, R, R, C, device=dev) * 0.05
        P = ops.conv_out_size(H, R, s, p, up)
        t = timeit(lambda: ops.conv2d_fwd(x, w, None, stride=s, pad=p, upsample=up))
        taps = R * R
        if up and R > 1:
            taps = (R // 2 + 1) ** 2 if R == 5 else taps
        exe = 2.0 * N * P * P * K * C * (9 if (up and R == 5) else R * R)
        alg = 2.0 * N * P * P * K * C * R * R
        rows.append(("fwd " + name, t, exe / t / 1e9, alg / t / 1e9))
    dg = [("D conv1", B, 32, 32, 64, 128), ("D conv2", B, 16, 16, 128, 256), ("D conv3", B, 8, 8, 256, 512), ("D conv0", B, 64, 64, 3, 64)]
    for name, N, H, W, C, K in dg:
        gy = torch.randn(N, H // 2, W // 2, K, device=dev)
        w = torch.randn(K, 5, 5, C, device=dev) * 0.05
        t = timeit(lambda: ops.conv2d_dgrad(gy, w, (H, W), stride=2, pad=2))
        fl = 2.0 * N * (H // 2) * (W // 2) * K * C * 25
        rows.append(("dgrad " + name, t, fl / t / 1e9, fl / t / 1e9))
    for name, N, H, W, C, K in [("D conv0", B, 64, 64, 3, 64)] + dg[:3]:
        gy = torch.randn(N, H // 2, W // 2, K, device=dev)
        x = torch.randn(N, H, W, C, device=dev)
        for group in (1, 8):
            out = torch.empty(N // group, K, 5, 5, C, device=dev)
            sq = torch.zeros(N // group, device=dev)
            t = timeit(lambda: ops.conv2d_wgrad_grouped(gy, x, 5, 5, stride=2, pad=2, group=group, alpha=1.0, out=out, sq=sq))
            fl = 2.0 * N * (H // 2) * (W // 2) * K * C * 25
            gb = out.numel() * 4 / t / 1e6
            rows.append(("wgrad g=%d %s (%.0f GB/s out)" % (group, name, gb), t, fl / t / 1e9, fl / t / 1e9))
    print("%-44s %9s %10s %10s" % ("launch", "ms", "TF exec", "TF algo"))
    for n, t, a, b in rows:
        print("%-44s %9.4f %10.1f %10.1f" % (n, t, a, b))


if __name__ == "__main__":
    main()
