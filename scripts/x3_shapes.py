#!/usr/bin/env python3
"""Same-box table of the headline step's conv launches on the exact-fp32 MFMA kernels and on the three-piece bf16 path (bf16x3):
forward and data gradient per shape, filters cached per weight (wkey) as in the step, HIP-event time over back-to-back launches.
usage (GPU box): python scripts/x3_shapes.py [--modes fp32,bf16x3] [--iters 20] [--only SUBSTR]"""
import argparse
import sys

sys.path.insert(0, ".")
import torch  # noqa: E402

from csl_gan_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="fp32,bf16x3")
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--only", default="")
a = ap.parse_args()


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


SHAPES = [
    # name, N, H, W, C, K, R, stride, pad, which ("f" forward, "d" data gradient)
    ("G b1 conv 512->512 @8", 128, 8, 8, 512, 512, 5, 1, 2, "f"),
    ("G b1 up   128->512 @8", 128, 8, 8, 128, 512, 5, 1, 2, "f"),
    ("G b2 conv 256->256 @16", 128, 16, 16, 256, 256, 5, 1, 2, "f"),
    ("G b2 up   128->256 @16", 128, 16, 16, 128, 256, 5, 1, 2, "f"),
    ("G b3 conv 128->128 @32", 128, 32, 32, 128, 128, 5, 1, 2, "f"),
    ("G b3 up   64->128 @32", 128, 32, 32, 64, 128, 5, 1, 2, "f"),
    ("G b4 conv 64->64 @64", 128, 64, 64, 64, 64, 5, 1, 2, "f"),
    ("G b4 up   32->64 @64", 128, 64, 64, 32, 64, 5, 1, 2, "f"),
    ("D conv2 s2 384 rows", 384, 32, 32, 64, 128, 5, 2, 2, "fd"),
    ("D conv3 s2 384 rows", 384, 16, 16, 128, 256, 5, 2, 2, "fd"),
    ("D conv4 s2 384 rows", 384, 8, 8, 256, 512, 5, 2, 2, "fd"),
    ("D conv2 s2 128 rows", 128, 32, 32, 64, 128, 5, 2, 2, "fd"),
    ("D conv3 s2 128 rows", 128, 16, 16, 128, 256, 5, 2, 2, "fd"),
    ("D conv4 s2 128 rows", 128, 8, 8, 256, 512, 5, 2, 2, "fd"),
]
modes = a.modes.split(",")
tot = {m: 0.0 for m in modes}
for name, N, H, W, C, K, R, s, p, which in SHAPES:
    if a.only and a.only not in name:
        continue
    x = torch.randn(N, H, W, C, device="cuda")
    w = torch.randn(K, R, R, C, device="cuda") / (C * R * R) ** 0.5
    P = (H + 2 * p - R) // s + 1
    flop = 2.0 * N * P * P * K * R * R * C
    gy = torch.randn(N, P, P, K, device="cuda")
    b = torch.randn(K, device="cuda")
    row = "%-26s" % name
    for mode in modes:
        with ops.compute_dtype(mode):
            if "f" in which:
                t = timeit(lambda: ops.conv2d_fwd(x, w, b, stride=s, pad=p, act=1, wkey=("x3s", name)), a.iters)
                kn = _lib.lib().cslgan_last_kernel().decode()
                row += "  %s fwd %.3f ms %5.0f TF [%s]" % (mode, t * 1e3, flop / t / 1e12, kn)
                tot[mode] += t
            if "d" in which:
                td = timeit(lambda: ops.conv2d_dgrad(gy, w, (H, W), stride=s, pad=p, wkey=("x3s", name)), a.iters)
                kn = _lib.lib().cslgan_last_kernel().decode()
                row += "  %s dgrad %.3f ms %5.0f TF [%s]" % (mode, td * 1e3, flop / td / 1e12, kn)
                tot[mode] += td
    print(row, flush=True)
WSHAPES = [
    # name, N, H, W, C, K, stride, group (0: the dense-sum policy of ops.dense_wgrad_group), clip-weighted
    ("D conv2 per-sample 128", 128, 32, 32, 64, 128, 2, 1, False),
    ("D conv2 per-sample 384", 384, 32, 32, 64, 128, 2, 1, False),
    ("D conv3 clip-weighted 256 g4", 256, 16, 16, 128, 256, 2, 4, True),
    ("D conv2 dense 128", 128, 32, 32, 64, 128, 2, 0, False),
    ("D conv3 dense 128", 128, 16, 16, 128, 256, 2, 0, False),
    ("D conv4 clip-weighted 256 g16", 256, 8, 8, 256, 512, 2, 16, True),
    ("D conv4 dense 128 g16", 128, 8, 8, 256, 512, 2, 16, False),
    ("G b3 conv dense 128", 128, 32, 32, 128, 128, 1, 0, False),
    ("G b4 conv dense 128", 128, 64, 64, 64, 64, 1, 0, False),
]
wtot = {m: 0.0 for m in modes}
for name, N, H, W, C, K, s, group, scaled in WSHAPES:
    if a.only and a.only not in name:
        continue
    R, p = 5, 2
    x = torch.randn(N, H, W, C, device="cuda")
    P = (H + 2 * p - R) // s + 1
    gy = torch.randn(N, P, P, K, device="cuda")
    f = torch.rand(N, device="cuda") if scaled else None
    flop = 2.0 * N * P * P * K * R * R * C
    row = "%-30s" % name
    for mode in modes:
        with ops.compute_dtype(mode):
            grp = group if group else ops.dense_wgrad_group(N, K, C, R, R, P * P, stride=s, out_hw=(P, P))
            out = torch.empty((N // grp, K, R, R, C), device="cuda")
            t = timeit(lambda: ops.conv2d_wgrad_grouped(gy, x, R, R, stride=s, pad=p, group=grp, row_scale=f, out=out), a.iters)
            kn = _lib.lib().cslgan_last_kernel().decode()
        row += "  %s g%d %.3f ms %5.0f TF [%s]" % (mode, grp, t * 1e3, flop / t / 1e12, kn)
        wtot[mode] += t
    print(row, flush=True)
print("weight gradients, sum: " + "  ".join("%s %.3f ms" % (m, wtot[m] * 1e3) for m in modes))
print("sum of the table: " + "  ".join("%s %.3f ms" % (m, tot[m] * 1e3) for m in modes))
