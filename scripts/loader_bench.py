#!/usr/bin/env python3
"""Input-pipeline throughput beside the step rate.  Generates CelebA-shaped JPEG files
(178 x 218) in a scratch directory and measures, on this box's host cores and GPU:
  (a) the reference-style path: CelebADataset (PIL decode + resize + crop + flip + normalise per image, datasets.py:20-63) behind a
      torch DataLoader with k workers, batches copied to the device synchronously (train.py:566-575);
  (b) building the preprocessed-tensor cache once (csl_gan_amd.pipeline.build_cache);
  (c) the cached pipeline alone: memmap gather -> pinned uint8 -> side-stream H2D -> conversion kernel (DevicePrefetcher);
  (d) the headline D-step fed by (c) — a NEW batch every step — against the same step on one resident batch.
usage (GPU box): python scripts/loader_bench.py [--n 4096] [--workers 8] [--steps 60] [--out gpurun_out/loader.txt]"""
import argparse
import contextlib
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--workers", type=int, default=8)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--out", type=str, default="")
a = ap.parse_args()

from PIL import Image  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

import bench  # noqa: E402
from csl_gan_amd import datasets as ds, pipeline as pl  # noqa: E402

lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


root = tempfile.mkdtemp(prefix="cslgan_loader_")
img_dir = os.path.join(root, "img")
os.makedirs(img_dir)
rng = np.random.default_rng(0)
t0 = time.perf_counter()
base = rng.integers(0, 256, (8, 218, 178, 3), dtype=np.uint8)
for i in range(a.n):          # smooth-ish content (random low-res pattern upsampled) so JPEG sizes resemble photographs
    lo = rng.integers(0, 256, (14, 12, 3), dtype=np.uint8)
    im = Image.fromarray(lo).resize((178, 218), Image.BILINEAR)
    im.save(os.path.join(img_dir, "%06d.jpg" % (i + 1)), quality=90)
say("generated %d JPEG files of 178x218 in %.1f s (%s)" % (a.n, time.perf_counter() - t0, img_dir))
B = 128
cpu_model = "unknown"
with open("/proc/cpuinfo") as f:
    for ln in f:
        if ln.lower().startswith("model name"):
            cpu_model = ln.split(":", 1)[1].strip()
            break
say("host: %s, %d logical CPUs visible; DataLoader workers %d; batch %d; 64x64 images" % (cpu_model, os.cpu_count(), a.workers, B))

# (a) reference-style path
data = ds.CelebADataset(img_dir, im_size=64, length=a.n)
dl = DataLoader(data, batch_size=B, shuffle=True, num_workers=a.workers, pin_memory=True, drop_last=True)
n_img, t0 = 0, None
for ep in range(2):
    for x, y in dl:
        if t0 is None:
            t0 = time.perf_counter()          # (the first batch pays worker start-up)
            continue
        x = x.cuda()
        n_img += x.shape[0]
torch.cuda.synchronize()
rate_ref = n_img / (time.perf_counter() - t0)
say("(a) reference-style loader (PIL per image, %d workers, synchronous H2D): %.0f images/s" % (a.workers, rate_ref))

# (b) cache build
t0 = time.perf_counter()
hdr = pl.build_cache(ds.CelebADataset(img_dir, im_size=64, length=a.n, flip=False), os.path.join(root, "cache", "celeba"))
t_build = time.perf_counter() - t0
say("(b) cache build, one pass, one process: %.1f s for %d images (%.0f images/s); cache %.1f MB uint8 NHWC" % (
    t_build, a.n, a.n / t_build, a.n * 64 * 64 * 3 / 1e6))
cache = pl.CachedImages(os.path.join(root, "cache", "celeba"))

# (c) cached pipeline alone
samp = pl.EpochSampler(len(cache), B, seed=1)
pf = pl.DevicePrefetcher(cache, samp, device="cuda:0", flip=True, depth=3)
n_img, t0 = 0, None
for ep in range(max(2, 40 * B * 8 // a.n)):
    samp.set_epoch(ep)
    for x, y in pf:
        if t0 is None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            continue
        n_img += x.shape[0]
torch.cuda.synchronize()
rate_pf = n_img / (time.perf_counter() - t0)
say("(c) cached pipeline alone (memmap gather -> pinned uint8 -> side-stream H2D -> u8_to_f32_nhwc): %.0f images/s" % rate_pf)

# (d) the headline step fed by the pipeline
with contextlib.redirect_stdout(sys.stderr):
    opt, tr, img = bench.build_trainer(0, 1, 0, extra=["--compute_dtype", "fp32_auto"])
gs = tr.graphed
for _ in range(gs.warmup + 3):
    gs(img, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    gs(img, None)
torch.cuda.synchronize()
rate_res = a.steps * B / (time.perf_counter() - t0)
done, t0 = 0, None
ep = 100
while done < a.steps:
    samp.set_epoch(ep)
    ep += 1
    for x, y in pf:
        if t0 is None:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        gs(x, None)
        done += 1
        if done >= a.steps:
            break
torch.cuda.synchronize()
rate_fed = a.steps * B / (time.perf_counter() - t0)
say("(d) headline D-step (fp32_auto, HIP graph): resident batch %.0f images/s; a NEW pipeline batch every step %.0f images/s (%.1f %% of resident)" % (
    rate_res, rate_fed, 100.0 * rate_fed / rate_res))
say("    -> the reference-style loader would cap the step at %.0f images/s (%.1f %% of the step rate); the cached pipeline has %.1fx headroom" % (
    rate_ref, 100.0 * rate_ref / rate_res, rate_pf / rate_res))
if a.out:
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")
