"""DP "mean samples" used as a public-data surrogate (reference mean_sampler.py:12-92).

Each mean sample is the mean of `mean_size` training images plus N(0, noise_std^2); sample() draws
a batch with per-image and per-pixel jitter.  The tensor of mean samples may live on the HIP device
so that the D-step needs no host->device copy (the reference samples on the host every step,
train.py:200-202, 214-216); the arithmetic is a gather plus two small Gaussian draws.
"""
import os

import numpy as np
import torch

from . import accountant, util

ALPHAS = accountant.DEFAULT_ALPHAS


class MeanSampler:
    def __init__(self, dataloader=None, path=None, transforms=None, noise_std=0.1, num_samples=32, mean_size=100,
                 dataset_size=180000, res=64, ch=3, save_path=None, default_batch_size=None, n_classes=1,
                 smallest_class_size=None, device="cpu", generator=None):
        self.dataloader, self.noise_std, self.num_samples, self.mean_size = dataloader, noise_std, num_samples, mean_size
        self.dataset_size, self.res, self.ch, self.default_batch_size = dataset_size, res, ch, default_batch_size
        denom = dataset_size if smallest_class_size is None else smallest_class_size
        self.sample_rate = mean_size / denom
        self.smallest_class_size, self.n_classes = smallest_class_size, n_classes
        self.device, self.generator = torch.device(device), generator
        self.mean_samples = None
        self._seed, self._draws = None, 0            # Philox stream of the device-side sample() (csrc/clip_kernels.hip)
        if path is not None:
            raise NotImplementedError("loading mean samples from image files needs PIL/torchvision transforms (out of scope)")
        if dataloader is not None:
            self.make_mean_samples(dataloader, save_path=save_path)

    def make_mean_samples(self, dataloader, save_path=None):
        per_class = [[] for _ in range(self.n_classes)]
        for _ in range(self.num_samples):
            samples, labels = next(iter(dataloader))
            for c in range(self.n_classes):
                pick = samples if self.n_classes == 1 else samples[labels == c][: self.mean_size]
                mean = pick.sum(dim=0) / self.mean_size          # divides by mean_size even if fewer were found
                per_class[c].append(mean + torch.empty(mean.shape).normal_(0, self.noise_std))
        self.mean_samples = torch.stack([torch.stack(v) for v in per_class]).to(self.device)
        if save_path is not None:
            save_path = util.add_slash(save_path)
            os.makedirs(save_path, exist_ok=True)
            np.save(save_path + "mean_samples.npy", self.mean_samples.cpu().numpy())

    def _nhwc_table(self):
        """The mean samples in the activation layout of the HIP critic (NHWC, csl_gan_amd DESIGN §3), made once per tensor: sample()
        then gathers rows that ARE channels-last images and the critic's first conv reads the batch without a layout pass."""
        ms = self.mean_samples
        key = (ms.data_ptr(), ms._version, tuple(ms.shape))
        if getattr(self, "_nhwc_key", None) != key:
            self._nhwc, self._nhwc_key = ms.permute(0, 1, 3, 4, 2).contiguous(), key
        return self._nhwc

    def sample(self, size, noise_std=0.01, noise_mean_std=0.01, requested_labels=None, out=None):
        """out (device path only): a channels-last dense [size, ch, res, res] tensor that receives the batch (a slice of the
        trainer's fused critic batch); the returned tensor is then `out`.  On the device the batch is ALWAYS channels-last
        (logical NCHW shape, NHWC memory)."""
        dev, gen = self.mean_samples.device, self.generator
        reps = (size - 1) // self.num_samples + 1
        if dev.type == "cuda":
            # ONE HIP kernel: the permutations (ranked Philox keys), the labels when none are requested, the gather, the jitter and
            # the noise — Philox keyed by a seed fixed at the first draw and a per-call counter (both checkpointed by the trainer)
            if self._seed is None:
                self._seed = int(gen.initial_seed() if gen is not None else torch.initial_seed()) ^ 0x6D65616E73616D70
            from . import ops
            self._draws += 1
            perms = None
            if self.num_samples > 1024:          # beyond the kernel's LDS ranking: permutations from one device sort
                perms = torch.rand(reps, self.num_samples, device=dev, generator=gen).argsort(dim=1).reshape(-1)[:size]
            labels = None if (requested_labels is None or self.n_classes == 1) else requested_labels.to(dev)
            tab = self._nhwc_table() if self.mean_samples.dim() == 5 else self.mean_samples
            dst = None
            if out is not None:
                dst = out.permute(0, 2, 3, 1) if out.dim() == 4 else out
                if not dst.is_contiguous():
                    raise RuntimeError("MeanSampler.sample: out must be a channels-last dense tensor")
            r, labels = ops.mean_sample(tab, labels, perms, noise_mean_std, noise_std, self._seed, self._draws, n=size,
                                        want_labels=True, out=dst)
            if self.mean_samples.dim() == 5:
                r = r.permute(0, 3, 1, 2)           # logical NCHW over NHWC memory (zero-copy)
            return (out if out is not None else r), (labels if self.n_classes > 1 else None)
        perms = torch.cat([torch.randperm(self.num_samples, device=dev, generator=gen) for _ in range(reps)])[:size]
        if requested_labels is None:
            requested_labels = torch.randint(0, self.n_classes, (size,), device=dev, generator=gen)
        requested_labels = requested_labels.to(dev)
        r = self.mean_samples[requested_labels, perms]
        if noise_mean_std is not None and noise_mean_std > 0:
            r = r + torch.empty(size, device=dev).normal_(0, noise_mean_std, generator=gen).view(-1, 1, 1, 1)
        if noise_std is not None and noise_std > 0:
            r = r + torch.empty(r.shape, device=dev).normal_(0, noise_std, generator=gen)
        return r, (requested_labels if self.n_classes > 1 else None)

    def state_dict(self):
        """The device-side draw stream (seed fixed at the first draw, per-call counter): saved next to the engine state so a
        resumed run continues the stream instead of repeating the first run's draws (the reference's host draws restart too)."""
        return {"seed": self._seed, "draws": self._draws}

    def load_state_dict(self, st):
        self._seed, self._draws = (None if st.get("seed") is None else int(st["seed"])), int(st.get("draws", 0))

    def get_privacy_cost(self, target_delta=1e-6, alphas=ALPHAS):
        pixel_sensitivity = 1 / self.mean_size / 2
        l2_sensitivity = np.sqrt(self.ch * self.res ** 2 * pixel_sensitivity ** 2)
        rdp = accountant.compute_rdp(self.sample_rate, self.noise_std / l2_sensitivity, self.num_samples * self.n_classes, alphas)
        return accountant.get_privacy_spent(alphas, rdp, target_delta)
