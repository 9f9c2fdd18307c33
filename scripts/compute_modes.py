#!/usr/bin/env python3
"""Same-box comparison of the three arithmetic modes of the conv kernels (fp32 MFMA, bf16, bf16x3 = fp32 from three bf16
pieces): time per launch on the generator / critic shapes of the headline step, and error against an fp64 reference on a
small shape.  usage (GPU box): python scripts/compute_modes.py"""
import sys, time
sys.path.insert(0, ".")
import torch
import torch.nn.functional as F
from csl_gan_amd import ops


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


g = torch.Generator().manual_seed(0)
print("== accuracy vs fp64 (N4 16x16 C64 K96 5x5; max |err| / max |ref|, and RMS err / RMS ref)")
x = torch.randn(4, 64, 16, 16, generator=g); w = torch.randn(96, 64, 5, 5, generator=g) / 40.0
ref = F.conv2d(x.double(), w.double(), None, padding=2)
xd, wd = x.permute(0, 2, 3, 1).contiguous().cuda(), w.permute(0, 2, 3, 1).contiguous().cuda()
for mode in ("fp32", "bf16x3", "bf16"):
    with ops.compute_dtype(mode):
        y = ops.conv2d_fwd(xd, wd, None, stride=1, pad=2).permute(0, 3, 1, 2).cpu().double()
    e = y - ref
    print("  %-7s max %.3e   rms %.3e" % (mode, e.abs().max() / ref.abs().max(), e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()))
print("  torch-cpu fp32 conv: max %.3e" % ((F.conv2d(x, w, None, padding=2).double() - ref).abs().max() / ref.abs().max()))

print("== time per launch (ms) and logical TFLOP/s")
shapes = [("G b4 conv 64->64 @64", 128, 64, 64, 64, 64, 5, 1, 2), ("G b3 conv 128->128 @32", 128, 32, 32, 128, 128, 5, 1, 2),
          ("G b2 conv 256->256 @16", 128, 16, 16, 256, 256, 5, 1, 2), ("G b1 conv 512->512 @8", 128, 8, 8, 512, 512, 5, 1, 2),
          ("G b4 convUp 32->64 @64", 128, 64, 64, 32, 64, 5, 1, 2), ("D conv1 s2 384 rows", 384, 32, 32, 64, 128, 5, 2, 2),
          ("D conv2 s2 384 rows", 384, 16, 16, 128, 256, 5, 2, 2), ("D conv3 s2 384 rows", 384, 8, 8, 256, 512, 5, 2, 2),
          ("D conv2 s2 128 rows", 128, 16, 16, 128, 256, 5, 2, 2)]
for name, N, H, W, C, K, R, s, p in shapes:
    x = torch.randn(N, H, W, C, device="cuda"); w = torch.randn(K, R, R, C, device="cuda") / (C * R * R) ** 0.5
    P = (H + 2 * p - R) // s + 1
    flop = 2.0 * N * P * P * K * R * R * C
    gy = torch.randn(N, P, P, K, device="cuda")
    row = "%-26s" % name
    for mode in ("fp32", "bf16x3", "bf16"):
        with ops.compute_dtype(mode):
            t = timeit(lambda: ops.conv2d_fwd(x, w, None, stride=s, pad=p))
            td = timeit(lambda: ops.conv2d_dgrad(gy, w, (H, W), stride=s, pad=p))
        row += "  %s fwd %.3f (%5.0f TF) dgrad %.3f (%5.0f TF)" % (mode, t * 1e3, flop / t / 1e12, td * 1e3, flop / td / 1e12)
    print(row)
