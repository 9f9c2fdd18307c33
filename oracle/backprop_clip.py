"""TEST INFRASTRUCTURE — CPU restatement of the reference's backprop clipping (backprop_clip.py), used only by tests/.

  * `bounds(layers, back, fwd, aas, awgs)`  — the parameter bookkeeping of PGCWrapper.__init__ (backprop_clip.py:49-96);
  * `attach(D, input_clips, back_clips, state)` — the data path of PGCWrapper.forward / backward_hook (backprop_clip.py:98-103)
    as plain forward-pre hooks (clip the layer input per sample) and tensor hooks on the layer output (clip the arriving gradient
    per sample while state["on"]), on an unmodified torch module tree.

Pinned by tests/golden/bpc_*.npz (made by executing the reference's own classes; tests/test_backprop_clip.py)."""
import numpy as np
import torch
from torch import nn


def l2_clip(t, C):
    """backprop_clip.py:18-22."""
    norm = t.flatten(1).norm(2, dim=1).reshape([-1] + [1] * (t.dim() - 1))
    return torch.where(norm > C, C * (t / norm), t)


def bounds(layers, back=None, fwd=None, aas=0.5, awgs=1e-4):
    """layers: [(kind, weight_numel, has_bias, in_numel, out_spatial_numel)] in module order.
    Returns (grad_l2_bounds per parameter, back_clip per layer, input_clip per layer)."""
    auto = back is None or fwd is None
    gb, bc, ic = [], [], []
    for li, (kind, wn, has_bias, in_n, out_sp) in enumerate(layers):
        if auto:
            i_c = np.sqrt(in_n * aas ** 2)
            wb = np.sqrt(wn * awgs ** 2)
            b_c = wb / i_c if kind == "linear" else np.sqrt(out_sp) * wb / i_c
        else:
            i_c, b_c = fwd[li], back[li]
            wb = i_c * b_c if kind == "linear" else i_c * np.sqrt(out_sp) * b_c
        gb.append(wb)
        if has_bias:
            gb.append(b_c if kind == "linear" else b_c * out_sp)
        bc.append(b_c)
        ic.append(i_c)
    return gb, bc, ic


def attach(D, input_clips, back_clips, state):
    leaves = [m for m in D.modules() if len(list(m.children())) < 1 and any(p.requires_grad for p in m.parameters())]
    assert len(leaves) == len(input_clips) == len(back_clips)
    for m, ic, bc in zip(leaves, input_clips, back_clips):
        m.register_forward_pre_hook(lambda mod, inp, ic=ic: (l2_clip(inp[0], ic),))

        def fwd_hook(mod, inp, out, bc=bc):
            if out.requires_grad:
                out.register_hook(lambda g, bc=bc: l2_clip(g, bc) if state["on"] else g)
        m.register_forward_hook(fwd_hook)
    return leaves
