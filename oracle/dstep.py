"""Oracle D-step: CPU restatement of ``train_D`` (train.py:360-500).  TEST INFRASTRUCTURE.

Every random input of the step (z, mean-sample batches, penalty alpha, DP noise) is an explicit
argument so that the HIP path and the oracle can be driven with identical numbers.

Step structure followed (gc mode):
  train.py:361-362   zero grads, freeze G
  train.py:378-380   adaptive modes: update_adaptive_clipping_params (train.py:204-245)
  train.py:382-384   fake forward (G frozen, fake detached), real forward, d_loss
  train.py:387       backward with per-sample hooks (pass 0 = fake, pass 1 = real)
  train.py:397       update_grad_logging (train.py:310-329)
  train.py:399-402   clip(); accum_grads_across_passes()
  train.py:409-431   penalty on public/mean samples, param-grad * batch_size added to summed_grad
  train.py:484       noise + /B + Adam
  train.py:488-500   logger observables
and for is mode train.py:375, 453-460.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import dp_engine as E
from . import penalty as P


@dataclass
class StepConfig:
    dp_mode: Optional[str] = "gc"            # None | "gc" | "is"
    grad_clip_mode: str = "standard"          # standard | adaptive | constant-pl | adaptive-pl
    grad_clip_split: bool = True
    clipping_param: float = 200.0
    clipping_param_per_layer: Optional[List[float]] = None
    adaptive_scalar: float = 1.5
    adaptive_stat: str = "mean"
    sigma: float = 0.5
    penalty: Sequence[str] = ("WGAN-GP",)
    penalty_use_public_data: bool = True
    aux_penalty: bool = True
    use_aux_loss: bool = False
    d_fake_aux_loss: bool = True
    imm_sens_per_param: bool = True
    imm_sens_scaling_vec: Optional[List[float]] = None
    lr: float = 1e-4
    adam_b1: float = 0.0
    adam_b2: float = 0.9
    weight_decay: float = 0.0

    @property
    def per_layer(self):
        return self.grad_clip_mode not in ("standard", "adaptive")


class OracleDStep:
    def __init__(self, G, D, cfg: StepConfig):
        self.G, self.D, self.cfg = G, D, cfg
        self.hooks = E.HookPerSample(D) if cfg.dp_mode == "gc" else None
        if self.hooks:
            self.hooks.enabled = False
        self.adam_state = {}
        n = len(list(D.parameters()))
        if cfg.per_layer:
            self.max_grad_norm = list(cfg.clipping_param_per_layer or [1.0] * n)
        else:
            self.max_grad_norm = float(cfg.clipping_param)
        self.steps = 0

    # -- train.py:345-358 -------------------------------------------------
    def _fake(self, z, y):
        with torch.no_grad():
            fake = self.G(z, y)
        out, aux = self.D(fake, y, aux=self.cfg.d_fake_aux_loss)
        loss = self.D.fake_loss(out)
        aux_loss = self.D.aux_loss(aux, y, fake=True) if (self.cfg.use_aux_loss and self.cfg.d_fake_aux_loss) else 0
        return out, aux, loss, aux_loss, fake

    def _real(self, img, labels):
        out, aux = self.D(img, labels)
        loss = self.D.real_loss(out)
        aux_loss = self.D.aux_loss(aux, labels, fake=False) if self.cfg.use_aux_loss else 0
        return out, aux, loss, aux_loss

    # -- train.py:204-245 -------------------------------------------------
    def adaptive_update(self, ms_img, ms_labels, z=None):
        cfg = self.cfg
        self.hooks.reset()
        self.hooks.enabled = True
        if cfg.grad_clip_split:
            fl = fal = 0
        else:
            _, _, fl, fal, _ = self._fake(z, ms_labels)
        _, _, rl, ral = self._real(ms_img, ms_labels)
        (rl + fl + ral + fal).backward()
        B = ms_img.size(0)
        r = []
        for p in self.D.parameters():
            gn = E.row_norms(p.grad_sample[0].reshape(B, -1)).double()      # float64 reduction unless timed, see dp_engine.calc_sample_norms
            r.append(gn.mean().item() if cfg.adaptive_stat == "mean" else gn.max().item())
        if cfg.per_layer:
            self.max_grad_norm = [x * cfg.adaptive_scalar for x in r]
        else:
            self.max_grad_norm = float(torch.tensor(r).norm(2) * cfg.adaptive_scalar)
        self.hooks.reset()
        for p in self.D.parameters():
            p.grad = None
        return r

    # -- train.py:360-500 -------------------------------------------------
    def step(self, img, labels, z, y, *, use_dp=True, ms_adapt=None, ms_adapt_labels=None, z_adapt=None,
             pen_real=None, pen_labels=None, alpha=None, noise=None, noise_gen=None, apply_update=True):
        cfg, D = self.cfg, self.D
        obs = {}
        params = list(D.parameters())
        for p in params:
            p.grad = None
        B = img.size(0)
        gc = cfg.dp_mode == "gc" and use_dp
        im = cfg.dp_mode == "is" and use_dp

        if gc:
            if cfg.grad_clip_mode.startswith("adaptive"):
                obs["adaptive_stats"] = self.adaptive_update(ms_adapt, ms_adapt_labels, z_adapt)
            self.hooks.reset()
            self.hooks.enabled = True
        if im:
            img = img.detach().clone().requires_grad_(True)

        d_fake, d_fake_aux, fl, fal, fake = self._fake(z, y)
        d_real, d_real_aux, rl, ral = self._real(img, labels)
        d_loss = rl + fl + ral + fal
        obs.update(d_real_loss=float(rl.detach()), d_fake_loss=float(fl.detach()), d_real=d_real.detach().clone(),
                   d_fake=d_fake.detach().clone(), fake_img=fake)
        penalty = torch.tensor(0.0)

        if gc:
            d_loss.backward()
            self.hooks.enabled = False
            gsamp = [p.grad_sample for p in params]
            # update_grad_logging (train.py:310-329)
            all_norms = E.calc_sample_norms(gsamp, flat=not cfg.per_layer)
            col = 1 if cfg.grad_clip_split else 0
            nm = torch.stack(all_norms).numpy()[:, col]
            facs = E.clipping_factors(all_norms, self.max_grad_norm)
            obs.update(norms=torch.stack(all_norms).clone(), norm_means=nm.mean(1), norm_stds=nm.std(1),
                       norm_maxes=nm.max(1), clip_params=np.array(self.max_grad_norm),
                       grads_clipped=np.array([(f[col].reshape(-1).numpy() < 0.999).mean() for f in facs]),
                       clip_factors=torch.stack(facs).clone())
            summed = E.clip_and_sum(gsamp, self.max_grad_norm, accum_passes=not cfg.grad_clip_split,
                                    num_private_passes=1 if cfg.grad_clip_split else None)
            obs["summed_clipped"] = [s.clone() for s in summed]
            if len(cfg.penalty) > 0 and not cfg.penalty_use_public_data:
                # train.py:433-450, literally: the penalty is evaluated on the private batch per sample, one autograd call per
                # sample adds its parameter gradient to p.grad_sample[0, i] (pass index 0, as written there), then clip() again
                penalties = P.calc_penalty(D, list(cfg.penalty), img, labels, fake, alpha, per_sample=True, aux_penalty=cfg.aux_penalty)
                penalty = penalties.mean(dim=0)
                for i in range(len(penalties)):
                    pg = torch.autograd.grad(penalties[i], params, retain_graph=True, allow_unused=True)
                    with torch.no_grad():
                        for p, g in zip(params, pg):
                            if g is not None:
                                p.grad_sample[0, i] += g
                gsamp = [p.grad_sample for p in params]
                summed = E.clip_and_sum(gsamp, self.max_grad_norm, accum_passes=not cfg.grad_clip_split,
                                        num_private_passes=1 if cfg.grad_clip_split else None)
                obs["summed_clipped_with_penalty"] = [s.clone() for s in summed]
                obs["norms_with_penalty"] = torch.stack(E.calc_sample_norms(gsamp, flat=not cfg.per_layer)).clone()
            elif len(cfg.penalty) > 0:
                penalty = P.calc_penalty(D, list(cfg.penalty), pen_real, pen_labels, fake, alpha, aux_penalty=cfg.aux_penalty)
                pg = torch.autograd.grad(penalty, params, allow_unused=True)
                obs["penalty_grads"] = [None if g is None else g.clone() for g in pg]
                summed = [s + (0 if g is None else g * B) for s, g in zip(summed, pg)]
            obs["summed_grad"] = [s.clone() for s in summed]
            grads = E.noised_mean_grads(summed, self.max_grad_norm, cfg.sigma, B, generator=noise_gen, noise=noise)
            self.steps += 1
        elif im:
            if len(cfg.penalty) > 0:
                pr = pen_real if pen_real is not None else img
                penalty = P.calc_penalty(D, list(cfg.penalty), pr, pen_labels, fake, alpha, aux_penalty=cfg.aux_penalty)
                d_loss = d_loss + penalty
            g, sens = E.immediate_sensitivity(D, d_loss, img, cfg.imm_sens_per_param, cfg.imm_sens_scaling_vec)
            obs["batch_sensitivity"] = sens
            obs["is_param_grads"] = [t.clone() for t in g]
            sens_v = np.broadcast_to(np.asarray(sens, dtype=np.float64), (len(g),))
            grads = []
            for i, t in enumerate(g):
                if noise is not None:
                    zn = noise[i] * float(sens_v[i])      # caller passes unit-variance * sigma noise
                elif cfg.sigma > 0:
                    zn = torch.randn(t.shape, generator=noise_gen) * (cfg.sigma * float(sens_v[i]))
                else:
                    zn = torch.zeros_like(t)
                grads.append(t + zn / B)
            self.steps += 1
        else:
            if len(cfg.penalty) > 0:
                pr = pen_real if pen_real is not None else img
                penalty = P.calc_penalty(D, list(cfg.penalty), pr, pen_labels, fake, alpha, aux_penalty=cfg.aux_penalty)
                d_loss = d_loss + penalty
            gl = torch.autograd.grad(d_loss, params, allow_unused=True)
            grads = [torch.zeros_like(p) if g is None else g for g, p in zip(gl, params)]

        obs["penalty"] = float(penalty.detach())
        obs["grads"] = [g.clone() for g in grads]
        if apply_update:
            with torch.no_grad():
                E.adam_step(params, grads, self.adam_state, cfg.lr, cfg.adam_b1, cfg.adam_b2, weight_decay=cfg.weight_decay)
        obs["d_real_acc"] = 100 * float((d_real.detach() > 0).float().mean())
        obs["d_fake_acc"] = 100 * float((d_fake.detach() < 0).float().mean())
        if cfg.use_aux_loss:
            obs["d_real_aux_loss"] = float(ral.detach())
        return obs
