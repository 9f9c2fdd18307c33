"""Oracle WGAN gradient penalty (CPU).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows gradient_penalty.py:4-18 (calc_penalty), :31-41 (calc_WGAN_GP_penalty),
:43-65 (calc_lipschitz_penalty_WRT).  DRAGAN (:20-29) is not restated: it raises in the
reference.

The interpolation weights ``alpha`` are drawn by the caller (the reference draws
``torch.rand(B,1)`` on the CPU generator, gradient_penalty.py:33) so that the GPU path and
the oracle can be fed the same numbers.
"""
from __future__ import annotations

import torch


def lipschitz_penalty(D, inputs, labels=None, *, per_sample=False, one_sided=False, aux_penalty=True):
    x = inputs.detach().requires_grad_(True)
    out, aux = D(x, None if labels is None else labels.detach())
    g, = torch.autograd.grad(out, x, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True)
    n = g.reshape(g.size(0), -1).norm(2, dim=1)
    pen = (n - 1).clamp(min=0) ** 2 if one_sided else (n - 1) ** 2
    if aux_penalty and aux is not None:
        for i in range(aux.size(1)):
            ga, = torch.autograd.grad(aux[:, i], x, grad_outputs=torch.ones_like(aux[:, i]),
                                      create_graph=True, retain_graph=True)
            na = ga.reshape(ga.size(0), -1).norm(2, dim=1)
            pen = pen + ((na - 1).clamp(min=0) ** 2 if one_sided else (na - 1) ** 2)
    return pen if per_sample else pen.mean()


def wgan_gp(D, real, real_labels, fake, alpha, *, per_sample=False, one_sided=False, weight=10.0, aux_penalty=False):
    """alpha: [B] in [0,1).  x_hat = alpha*real + (1-alpha)*fake."""
    a = alpha.reshape(-1, *([1] * (real.dim() - 1))).to(real.dtype)
    x_hat = a * real + (1 - a) * fake
    return weight * lipschitz_penalty(D, x_hat, real_labels, per_sample=per_sample, one_sided=one_sided,
                                      aux_penalty=aux_penalty)


def calc_penalty(D, penalty_types, real, real_labels, fake, alpha, *, per_sample=False, weights=None, aux_penalty=False):
    weights = [1.0 / len(penalty_types)] * len(penalty_types) if weights is None else weights
    total = 0
    for w, name in zip(weights, penalty_types):
        if not name.startswith("WGAN-GP"):
            raise Exception("Unknown / unsupported penalty type: " + name)
        total = total + w * wgan_gp(D, real, real_labels, fake, alpha, per_sample=per_sample,
                                    one_sided=name.endswith("1"), aux_penalty=aux_penalty)
    return total
