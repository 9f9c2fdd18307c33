"""Immediate-sensitivity DP engine (train.py:103-107 constructor, :457/:469 backward, :332-338
batch_sensitivity, :248-249 scaling_vec).  FORK-INFERRED, PARITY UNPINNED: the twosixlabs Opacus fork
that implements it is not available; this follows SURVEY.md §8 (a13) and oracle/dp_engine.py:

    g      = grad_theta L                               (kept as p.grad)
    s_b    = || d ||g||_2 / d x_b ||_2                  per sample b of the batch
    batch sensitivity = max_b s_b                        (one value, or one per parameter tensor with per_param)
    step:  p.grad <- g + sigma * sensitivity * N(0, I) / B

Everything runs through autograd over the HIP Functions (csl_gan_amd.functional): the parameter gradients
are taken with create_graph=True and each sensitivity is one more double-backward sweep to the input
batch; the per-sample L2 norms are cslgan_row_l2norm_f32.
"""
from __future__ import annotations

import types

import numpy as np
import torch

from . import accountant, functional as HF, ops
from .engine import _flat


class ISPrivacyEngine:
    def __init__(self, module, batch_size, sample_size, alphas, noise_multiplier, per_param=False, scaling_vec=None,
                 world_size=1, **_unused):
        self.module, self.batch_size, self.sample_size = module, batch_size, sample_size
        self.alphas, self.noise_multiplier = list(alphas), noise_multiplier
        self.per_param, self.scaling_vec = per_param, (None if scaling_vec is None else list(scaling_vec))
        self.world_size = world_size
        self.sample_rate = batch_size * world_size / sample_size
        self.params = list(module.parameters())
        if any(not p.is_cuda for p in self.params):
            raise RuntimeError("ISPrivacyEngine needs the discriminator on a HIP device")
        self.steps, self.seed, self._noise_calls = 0, 0, 0
        self._sens_dev = None
        self._sens_last, self._sens_host = None, None
        self._noise_ctr = None                  # device mirror of _noise_calls (the Philox call index of a replayed step)
        self.optimizer, self.grad_reducer = None, None
        self.host_noise = None

    @property
    def batch_sensitivity(self):
        """float (or ndarray with per_param) as train.py:332-338 reads it.  The value lives on the device; the host copy — a
        synchronisation — is made only when somebody reads this attribute (the trainer's logging accumulates on the device)."""
        if self._sens_host is None and self._sens_last is not None:
            host = self._sens_last.detach().cpu().numpy().astype(np.float64)
            self._sens_host = host if self.per_param else float(host[0])
        return self._sens_host

    @batch_sensitivity.setter
    def batch_sensitivity(self, v):
        self._sens_host, self._sens_last = v, None

    @property
    def batch_sensitivity_device(self):
        """The same values as a device tensor ([n_params] with per_param, else [1]); no synchronisation."""
        return self._sens_last

    # hooks are a gc-mode concept; train.py never calls them in is mode, kept for interface symmetry
    def enable_hooks(self):
        pass

    def disable_hooks(self):
        pass

    def set_scaling_vec(self, vec):
        self.scaling_vec = list(vec)

    def _set_seed(self, seed):
        self.seed = int(seed)
        self._set_noise_calls(0)

    def _set_noise_calls(self, n):
        self._noise_calls = int(n)
        if self._noise_ctr is not None:
            self._noise_ctr.fill_(int(n))

    def ensure_noise_counter(self):
        if self._noise_ctr is None:
            self._noise_ctr = torch.full((1,), self._noise_calls, device=self.params[0].device, dtype=torch.int64)
        return self._noise_ctr

    def backward(self, loss, inputs):
        """Parameter gradients plus immediate sensitivities (train.py:457)."""
        ps = self.params
        grads = torch.autograd.grad(loss, ps, create_graph=True, allow_unused=True)
        B = inputs.shape[0]

        def row_norms(gx):
            flat = gx.reshape(B, -1)
            return ops.row_l2norm(flat.contiguous()) if flat.is_cuda else flat.norm(2, dim=1)

        def sens_of(scalar):
            if not scalar.requires_grad:
                return torch.zeros((), device=inputs.device)
            gx, = torch.autograd.grad(scalar, inputs, retain_graph=True, allow_unused=True)
            if gx is None:
                return torch.zeros((), device=inputs.device)
            return row_norms(gx.detach()).max()

        def tensor_norm(g):
            return HF.RowL2Norm.apply(g.reshape(1, -1))[0]

        if self.per_param:
            sens = torch.stack([sens_of(tensor_norm(g)) if g is not None else torch.zeros((), device=inputs.device) for g in grads])
        else:
            sq = 0
            for i, g in enumerate(grads):
                if g is None:
                    continue
                n = tensor_norm(g)
                if self.scaling_vec is not None:
                    n = n / float(self.scaling_vec[i])
                sq = sq + n * n
            sens = sens_of(torch.sqrt(sq)).reshape(1)
        if self.world_size > 1:        # the batch maximum is over all ranks' samples
            from .distributed import average_across_ranks
            average_across_ranks(sens, use_max=True)
        self._sens_dev = sens
        self._sens_last, self._sens_host = sens.detach(), None
        for p, g in zip(ps, grads):
            p.grad = None if g is None else g.detach()

    def attach(self, optimizer):
        self.optimizer = optimizer
        engine, orig = self, optimizer.step

        def dp_step(self_opt, closure=None):
            engine._before_step()
            return orig()
        optimizer.privacy_engine = self
        optimizer.step = types.MethodType(dp_step, optimizer)

    def _before_step(self):
        if self._sens_dev is None:
            return                      # non-DP step
        ps = self.params
        R, B = self.world_size, self.batch_size
        dev = ps[0].device
        flat = torch.empty(sum(p.numel() for p in ps), device=dev, dtype=torch.float32)
        ins, outs, off = [], [], 0
        for p in ps:
            g = p.grad
            if g is None:
                g = torch.zeros_like(p, memory_format=torch.preserve_format)
            elif g.stride() != p.stride():
                g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
            ins.append(_flat(g).view(1, -1))
            outs.append(flat[off:off + p.numel()])
            p.grad = torch.as_strided(flat, p.size(), p.stride(), storage_offset=off)
            off += p.numel()
        sens = self._sens_dev if self.per_param else self._sens_dev.expand(len(ps))
        std = (sens * (self.noise_multiplier / (B * R ** 0.5))).contiguous()
        noises = None if self.host_noise is None else [z.to(dev) for z in self.host_noise]
        # out = (g + std*z) / R   — g is already the batch-mean gradient of this rank
        # Philox stream (seed, call, tensor, column): the call index is read from HBM, so a graph replay advances it too
        ctr = self.ensure_noise_counter()
        ops.clip_accum_noise(ins, outs, noise_std=std if self.noise_multiplier > 0 else None, noises=noises, seed=self.seed,
                             offset=0, call_counter=ctr, scale=1.0 / R)
        ctr.add_(1)
        if self.grad_reducer is not None:
            self.grad_reducer(flat)
        self._noise_calls += 1
        self.steps += 1
        self._sens_dev = None

    def state_dict(self):
        return {"steps": self.steps, "seed": self.seed, "noise_calls": self._noise_calls, "scaling_vec": self.scaling_vec,
                "noise_multiplier": self.noise_multiplier, "sample_rate": self.sample_rate}

    def load_state_dict(self, st):
        self.steps, self.seed = st["steps"], int(st["seed"])
        self._set_noise_calls(st["noise_calls"])
        self.scaling_vec = st.get("scaling_vec")

    def get_privacy_spent(self, target_delta=None):
        delta = 1e-6 if target_delta is None else target_delta
        rdp = accountant.compute_rdp(self.sample_rate, self.noise_multiplier, self.steps, self.alphas)
        return accountant.get_privacy_spent(self.alphas, rdp, delta)
