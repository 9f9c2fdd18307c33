"""WGAN gradient penalty (reference gradient_penalty.py:4-65) with the same call signature.

The double backward runs on the HIP conv Functions (csl_gan_amd.functional): autograd.grad with
create_graph=True records Dgrad nodes, and the later parameter-gradient call differentiates them
through Conv / Wgrad.  The per-sample input-gradient norm is the cslgan_row_l2norm_f32 kernel.
DRAGAN is rejected: it raises inside the reference too.
"""
import torch
from torch import autograd

from . import functional as HF
from . import ops


def calc_penalty(model, penalty_types, real_data, real_labels, fake_data, fake_labels, device="cpu", per_sample=False,
                 weights=None, aux_penalty=False, alpha=None):
    """alpha (optional, [B]): interpolation weights; drawn with torch.rand on the CPU generator when
    omitted, exactly where the reference draws them (gradient_penalty.py:33)."""
    if weights is None:
        weights = [1 / len(penalty_types)] * len(penalty_types)
    total = None
    for w, kind in zip(weights, penalty_types):
        if kind.startswith("WGAN-GP"):
            p = calc_WGAN_GP_penalty(model, real_data, real_labels, fake_data, fake_labels, device=device,
                                     per_sample=per_sample, one_sided=kind.endswith("1"), aux_penalty=aux_penalty, alpha=alpha)
        elif kind.startswith("DRAGAN"):
            raise Exception("DRAGAN penalty is not supported (it raises in the reference as well)")
        else:
            raise Exception("Unknown penalty type: " + kind)
        p = p if w == 1 else w * p          # (one penalty type: weight 1 — no "0 + 1 * p" launches and their backward)
        total = p if total is None else total + p
    return 0 if total is None else total


def calc_WGAN_GP_penalty(model, real_data, real_labels, fake_data, fake_labels, device="cpu", per_sample=False,
                         one_sided=False, weight=10.0, aux_penalty=False, alpha=None):
    B = real_data.size(0)
    if alpha is None:
        alpha = torch.rand(B, 1)
    fake_data = fake_data.to(real_data.device)
    if real_data.is_cuda and real_data.dtype == torch.float32 and fake_data.shape == real_data.shape:
        from . import ops
        # one launch on the layout the critic's first conv reads (NHWC memory for images): no separate layout copy afterwards
        fmt = torch.channels_last if real_data.dim() == 4 else torch.contiguous_format
        interpolates = ops.lerp_rows(real_data.contiguous(memory_format=fmt), fake_data.contiguous(memory_format=fmt),
                                     alpha.reshape(B).to(device=real_data.device, dtype=torch.float32).contiguous())
    else:
        a = alpha.reshape(B, *([1] * (real_data.dim() - 1))).to(device=real_data.device, dtype=real_data.dtype)
        interpolates = a * real_data + (1 - a) * fake_data
    return calc_lipschitz_penalty_WRT(model, interpolates, real_labels, device=device, per_sample=per_sample,
                                      one_sided=one_sided, aux_penalty=aux_penalty, weight=weight)


def _row_norms(g):
    flat = g.reshape(g.size(0), -1)
    if flat.is_cuda:
        return HF.RowL2Norm.apply(flat)
    return flat.norm(2, dim=1)


def calc_lipschitz_penalty_WRT(model, inputs, input_labels=None, device="cpu", per_sample=False, one_sided=False,
                               aux_penalty=True, weight=1.0):
    """weight: the factor the caller multiplies the result by (gradient_penalty.py:41: 10.0), folded into the fused device
    kernel's coefficient together with the batch mean."""
    x = inputs.detach().requires_grad_(True)
    labels = None if input_labels is None else input_labels.detach()
    with HF.input_grads_only():
        out, aux_out = model(x, labels)

    B = x.size(0)

    def term(scalar_outputs):
        ones = ops.ones_like_const(scalar_outputs) if scalar_outputs.is_cuda else torch.ones_like(scalar_outputs)
        g, = autograd.grad(outputs=scalar_outputs, inputs=x, grad_outputs=ones,
                           create_graph=True, retain_graph=True, only_inputs=True)
        flat = g.reshape(B, -1)
        if flat.is_cuda and flat.dtype == torch.float32:
            return HF.LipschitzTerm.apply(flat, one_sided, weight if per_sample else weight / B, per_sample)
        d = flat.norm(2, dim=1) - 1
        t = d.clamp(min=0) ** 2 if one_sided else d ** 2
        return weight * (t if per_sample else t.mean())

    pen = term(out)
    if aux_penalty and aux_out is not None:
        for i in range(aux_out.size(1)):
            pen = pen + term(aux_out[:, i])
    return pen
