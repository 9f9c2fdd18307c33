"""Oracle DP engine: per-sample gradients, norms, clip, accumulate, noise, immediate sensitivity.

TEST INFRASTRUCTURE — see oracle/__init__.py.

The reference delegates all of this to ``opacus`` @ git+git://github.com/twosixlabs/opacus
(requirements.txt:9, no commit pin, not in this container) — PARITY UNPINNED.  What is
restated here is the specification SURVEY.md §8 (a7)-(a13) derives from the reference's
call sites:

  train.py:233,388,447   p.grad_sample has shape [n_passes, B, *p.shape]
  train.py:311-315       calc_sample_norms(named_params, flat) -> list of [n_passes, B]
                         (one entry per parameter tensor, or a single entry when flat)
  train.py:324-328       calc_clipping_factors(norms) -> iterable of [n_passes, B]
  train.py:399-402       clip(); accum_grads_across_passes()
  train.py:417,431       accumulate_batch(); p.summed_grad is a SUM over samples
  train.py:484           wrapped optimizer.step(): grad = (summed + N(0,(sigma*C)^2)) / B
  train.py:457,332-338   ISPrivacyEngine.backward(loss, inputs); .batch_sensitivity

Per-sample gradient of sample b is defined as the gradient of that sample's own loss
term (micro-batch-of-one autograd); with a mean-reduced batch loss this equals
B * d(batch loss restricted to b)/d theta, which is what the hook route produces.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import nn

CLIP_EPS = 1e-6


# ---------------------------------------------------------------------------
# per-sample gradients — two routes that must agree (definition and hook route)
# ---------------------------------------------------------------------------
def per_sample_grads_microbatch(D: nn.Module, per_sample_loss: Callable, x, y=None) -> List[torch.Tensor]:
    """Definition: one autograd call per sample.  per_sample_loss(D, x_b, y_b) -> scalar loss of sample b."""
    params = [p for p in D.parameters()]
    out = [torch.zeros((x.size(0),) + tuple(p.shape), dtype=p.dtype) for p in params]
    for b in range(x.size(0)):
        lb = per_sample_loss(D, x[b:b + 1], None if y is None else y[b:b + 1])
        gs = torch.autograd.grad(lb, params, allow_unused=True)
        for o, g in zip(out, gs):
            if g is not None:
                o[b] = g
    return out


class HookPerSample:
    """Hook route (the algorithm family Opacus uses, and what bench.py times as CPU baseline).

    Forward hooks stash each Conv2d/Linear input; full-backward hooks turn (input, grad_output)
    into per-sample weight/bias gradients:  conv: unfold + einsum('bot,bkt->bok');
    linear: einsum('bo,bi->boi').  Results are scaled by B (mean-reduced loss) and appended
    along a leading ``pass`` dimension in forward-call order.
    """

    def __init__(self, module: nn.Module, batch_scale: bool = True):
        self.module, self.enabled, self.batch_scale = module, True, batch_scale
        self._handles = []
        self.layers = [m for m in module.modules() if isinstance(m, (nn.Conv2d, nn.Linear))]
        for m in self.layers:
            self._handles.append(m.register_forward_hook(self._fwd))
        self.reset()

    def reset(self):
        self._n_fwd = {id(m): 0 for m in self.layers}
        for m in self.layers:
            for p in m.parameters(recurse=False):
                if hasattr(p, "grad_sample"):
                    del p.grad_sample
        self._store = {}

    def remove(self):
        for h in self._handles:
            h.remove()

    def _fwd(self, m, inp, out):
        if not self.enabled or not torch.is_grad_enabled() or not out.requires_grad:
            return
        # the pass index is fixed at forward time; autograd may run independent passes in any order
        idx = self._n_fwd[id(m)]
        self._n_fwd[id(m)] += 1
        a = inp[0].detach()
        out.register_hook(lambda g, m=m, a=a, idx=idx: self._on_grad(m, a, idx, g))

    def _on_grad(self, m, a, pass_idx, g):
        if not self.enabled:
            return
        g = g.detach()
        B = g.size(0)
        scale = float(B) if self.batch_scale else 1.0
        if isinstance(m, nn.Conv2d):
            pad = m.padding if not isinstance(m.padding, str) else tuple(k // 2 for k in m.kernel_size)
            cols = F.unfold(a, m.kernel_size, dilation=m.dilation, padding=pad, stride=m.stride)
            gw = torch.einsum("bot,bkt->bok", g.reshape(B, g.size(1), -1), cols).reshape((B,) + tuple(m.weight.shape))
            gb = g.reshape(B, g.size(1), -1).sum(-1) if m.bias is not None else None
        else:
            gw = torch.einsum("bo,bi->boi", g, a)
            gb = g if m.bias is not None else None
        self._put(m.weight, pass_idx, gw * scale)
        if gb is not None:
            self._put(m.bias, pass_idx, gb * scale)

    def _put(self, p, pass_idx, val):
        d = self._store.setdefault(id(p), {})
        d[pass_idx] = val
        n = max(d) + 1
        if all(i in d for i in range(n)):
            p.grad_sample = torch.stack([d[i] for i in range(n)])


# ---------------------------------------------------------------------------
# norms / clip / accumulate / noise
# ---------------------------------------------------------------------------
# Element type the per-sample squared sums are REDUCED in.  float64 is the checker's setting (see calc_sample_norms); bench.py's
# cpu_baseline leg switches to float32 — the reduction the reference's dependency performs (``grad_sample.norm(2)`` on fp32
# tensors) — so that the timed CPU step is the hook-based fp32 unfold+einsum algorithm and nothing else (a float64 copy of the
# 2.2 GB of per-sample gradients per pass is checker hygiene, not part of the algorithm that is being timed).
_NORM_DTYPE = torch.float64


def set_norm_dtype(dtype):
    global _NORM_DTYPE
    assert dtype in (torch.float32, torch.float64)
    _NORM_DTYPE = dtype


def row_norms(t2d: torch.Tensor) -> torch.Tensor:
    """L2 norm of every row of a [rows, len] tensor, reduced in the configured element type."""
    return t2d.to(_NORM_DTYPE).norm(2, dim=-1)


def calc_sample_norms(grad_samples: Sequence[torch.Tensor], flat: bool) -> List[torch.Tensor]:
    """grad_samples: per-parameter tensors [n_passes, B, ...] -> list of [n_passes, B] norms.

    flat=True returns a one-element list holding the L2 norm over all parameters (train.py:311-315).
    The squares are summed in float64: torch's CPU float32 ``norm`` over a 3.3 M-element row (D's last conv) is 1.3e-4 low
    (sequential accumulation), which is an artefact of the host reduction, not part of the definition (the device reduces
    by trees; tests/golden/dstep_*.npz hold float64 reductions of the reference classes' float32 gradients).
    """
    per = [row_norms(g.reshape(g.size(0), g.size(1), -1)).to(g.dtype) for g in grad_samples]
    if flat:
        return [torch.stack(per, dim=0).norm(2, dim=0)]
    return per


def clipping_factors(norms: Sequence[torch.Tensor], max_grad_norm, eps: float = CLIP_EPS) -> List[torch.Tensor]:
    """f = min(1, C / (norm + eps)); one C for flat, one per tensor for per-layer."""
    if isinstance(max_grad_norm, (list, tuple)):
        assert len(max_grad_norm) == len(norms)
        return [(c / (n + eps)).clamp(max=1.0) for n, c in zip(norms, max_grad_norm)]
    return [(float(max_grad_norm) / (n + eps)).clamp(max=1.0) for n in norms]


def clip_and_sum(grad_samples: Sequence[torch.Tensor], max_grad_norm, *, accum_passes: bool,
                 num_private_passes: Optional[int]) -> List[torch.Tensor]:
    """clip() + accum_grads_across_passes(): returns per-parameter summed_grad (a SUM over samples).

    accum_passes=True  : per-sample grads of all passes are added per sample, then clipped once.
    accum_passes=False : only the last ``num_private_passes`` passes are clipped; earlier passes
                         (generated data) are summed unclipped (train.py:112-113, 401-402).
    """
    per_layer = isinstance(max_grad_norm, (list, tuple))
    gs = list(grad_samples)
    if accum_passes:
        gs = [g.sum(dim=0, keepdim=True) for g in gs]
        n_private = 1
    else:
        n_private = gs[0].size(0) if num_private_passes is None else num_private_passes
    norms = calc_sample_norms(gs, flat=not per_layer)
    fac = clipping_factors(norms, max_grad_norm)
    out = []
    n_pass = gs[0].size(0)
    for i, g in enumerate(gs):
        f = fac[i] if per_layer else fac[0]
        f = f.clone()
        f[: n_pass - n_private] = 1.0
        shape = f.shape + (1,) * (g.dim() - 2)
        out.append((g * f.reshape(shape)).sum(dim=(0, 1)))
    return out


def noise_stds(max_grad_norm, sigma: float, n_tensors: int) -> List[float]:
    if isinstance(max_grad_norm, (list, tuple)):
        return [sigma * float(c) for c in max_grad_norm]
    return [sigma * float(max_grad_norm)] * n_tensors


def noised_mean_grads(summed: Sequence[torch.Tensor], max_grad_norm, sigma: float, batch_size: int,
                      generator: Optional[torch.Generator] = None, noise: Optional[Sequence[torch.Tensor]] = None):
    """grad = (summed + N(0, (sigma*C)^2)) / B  (train.py:484 via the wrapped step)."""
    stds = noise_stds(max_grad_norm, sigma, len(summed))
    out = []
    for i, (s, sd) in enumerate(zip(summed, stds)):
        if noise is not None:
            z = noise[i]
        elif sd > 0:
            z = torch.randn(s.shape, generator=generator, dtype=s.dtype) * sd
        else:
            z = torch.zeros_like(s)
        out.append((s + z) / batch_size)
    return out


def adam_step(params, grads, state, lr, b1, b2, eps=1e-8, weight_decay=0.0):
    """torch.optim.Adam semantics (train.py:76) restated for the oracle step."""
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    for i, (p, g) in enumerate(zip(params, grads)):
        if weight_decay:
            g = g + weight_decay * p
        m = state.setdefault(("m", i), torch.zeros_like(p))
        v = state.setdefault(("v", i), torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.data.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


# ---------------------------------------------------------------------------
# backprop_clip.py:18-22
# ---------------------------------------------------------------------------
def l2_clip(t: torch.Tensor, C: float) -> torch.Tensor:
    flat = t.reshape(t.size(0), -1)
    n = flat.norm(2, dim=1, keepdim=True)
    return torch.where(n > C, C * (flat / n), flat).reshape(t.shape)


# ---------------------------------------------------------------------------
# immediate sensitivity  (train.py:103-107, 457; FORK-INFERRED, parity unpinned)
# ---------------------------------------------------------------------------
def immediate_sensitivity(D: nn.Module, loss: torch.Tensor, inputs: torch.Tensor, per_param: bool,
                          scaling_vec: Optional[Sequence[float]] = None):
    """Returns (param_grads, batch_sensitivity).

    s_b = || d ||grad_theta L||_2 / d x_b ||_2 ;  batch sensitivity = max_b s_b.
    per_param=True computes one sensitivity per parameter tensor.  With a scaling vector the
    single norm is taken over the per-tensor gradients divided by their scale.
    """
    params = list(D.parameters())
    grads = torch.autograd.grad(loss, params, create_graph=True, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(p, requires_grad=True) for g, p in zip(grads, params)]
    B = inputs.size(0)

    def sens_of(norm):
        gx, = torch.autograd.grad(norm, inputs, retain_graph=True, allow_unused=True)
        if gx is None:
            return 0.0
        return gx.reshape(B, -1).norm(2, dim=1).max().item()

    if per_param:
        sens = [sens_of(g.reshape(-1).norm(2)) if g.requires_grad and g.grad_fn is not None else 0.0 for g in grads]
        import numpy as np
        sens = np.asarray(sens)
    else:
        if scaling_vec is None:
            total = torch.sqrt(sum((g ** 2).sum() for g in grads))
        else:
            total = torch.sqrt(sum(((g / s) ** 2).sum() for g, s in zip(grads, scaling_vec)))
        sens = sens_of(total)
    return [g.detach() for g in grads], sens
