"""The reference's experimental "backprop clipping" (backprop_clip.py): every parameterised leaf layer of D clips its INPUT
activations per sample on the way forward (l2_clip, backprop_clip.py:18-22,103) and the gradient arriving at its OUTPUT per
sample on the way back (the full-backward hook of PGCWrapper's dummy layer, backprop_clip.py:98-100), which bounds every
per-sample parameter-gradient norm analytically (backprop_clip.py:63-93); train.py:84-92 turns the bounds into the per-layer
clip norms of the DP engine.

Same constructor, attributes (grad_l2_bounds, back_clip_params, input_clip_params, hooks_enabled, enable_hooks / disable_hooks)
and numbers as the reference's BackpropClipper.  Differences, both forced by what is in the image:
  * layer shapes come from one forward of a zero batch with forward hooks instead of torchinfo.summary (absent); like the
    reference the probe is a 1x1x28x28 image (backprop_clip.py:123), so only the MNIST models work;
  * layers are not re-parented under wrapper modules (state_dict keys stay `blocks.0.weight`, not `blocks.0.module.weight`):
    the clip parameters hang on the layer (`layer._bpc`), the HIP layers clip their input with cslgan_l2_clip_rows_f32 and the
    pre-activation gradient inside their backward — the activation is fused into the conv kernel's epilogue here, and the
    reference's hook sits between the conv and F.leaky_relu (DCResNet_models.py:132), i.e. on exactly that gradient.
As in the reference, scalar clip parameters (the non "-pl" modes, train.py:86) fail when indexed (backprop_clip.py:80-81).
Pinned by tests/golden/bpc_*.npz, produced by executing the reference's own PGCWrapper / BackpropClipper.convert."""
import numpy as np
import torch
from torch import nn

from . import ops


def prod(t):
    out = 1
    for v in t:
        out *= v
    return out


def l2_size(n, activation_scale):
    """l2 norm of an n-element tensor whose elements all equal activation_scale (backprop_clip.py:14-16)."""
    return np.sqrt(n * activation_scale ** 2)


def l2_to_l1(l2, n):
    return np.sqrt(n) * l2


def l2_clip(t, C):
    """Rows (samples) whose L2 norm over all non-batch dims exceeds C are rescaled to norm C (backprop_clip.py:18-22)."""
    if t.is_cuda:
        if t.dim() == 4 and not t.is_contiguous() and t.is_contiguous(memory_format=torch.channels_last):
            v = t.permute(0, 2, 3, 1)
            return ops.l2_clip_rows(v, C).permute(0, 3, 1, 2)
        return ops.l2_clip_rows(t.contiguous(), C)
    dims = tuple(range(1, t.dim()))
    norm = t.norm(2, dim=dims, keepdim=True)
    return torch.where(norm > C, C * (t / norm), t)


class L2Clip(torch.autograd.Function):
    """Differentiable l2_clip: rows above C become C * t / ||t|| (the gradient also flows through the norm, as in the reference's
    torch.where expression).  Forward on cslgan_l2_clip_rows_f32 for device tensors."""

    @staticmethod
    def forward(ctx, t, C):
        flat = t.reshape(t.size(0), -1)
        n = flat.norm(2, dim=1) if not t.is_cuda else ops.row_l2norm(flat.contiguous())
        ctx.save_for_backward(t, n)
        ctx.C = float(C)
        return l2_clip(t, C)

    @staticmethod
    def backward(ctx, g):
        t, n = ctx.saved_tensors
        C, B = ctx.C, t.size(0)
        tf, gf = t.reshape(B, -1), g.reshape(B, -1)
        clipped = (n > C).unsqueeze(1)
        s = (C / n.clamp_min(1e-30)).unsqueeze(1)
        dot = (tf * gf).sum(dim=1, keepdim=True) / (n * n).clamp_min(1e-30).unsqueeze(1)
        return torch.where(clipped, s * (gf - tf * dot), gf).reshape(t.shape), None


class ClipGrad(torch.autograd.Function):
    """Identity whose backward clips the gradient per sample when the clipper's hooks are on (PGCWrapper.backward_hook)."""

    @staticmethod
    def forward(ctx, y, layer_clip):
        ctx.lc = layer_clip
        return y.view_as(y)

    @staticmethod
    def backward(ctx, g):
        return ctx.lc.clip_grad(g), None


class LayerClip:
    """What PGCWrapper holds for one layer: the forward (input) and backward (output-gradient) clip parameters."""

    def __init__(self, pgc, input_clip_param, back_clip_param):
        self.pgc, self.input_clip_param, self.back_clip_param = pgc, float(input_clip_param), float(back_clip_param)

    def clip_input(self, x):
        return L2Clip.apply(x, self.input_clip_param) if (torch.is_grad_enabled() and x.requires_grad) else l2_clip(x, self.input_clip_param)

    def clip_grad(self, g):
        return l2_clip(g.contiguous(), self.back_clip_param) if self.pgc.hooks_enabled else g


def module_requires_grad(module):
    return any(p.requires_grad for p in module.parameters())


class BackpropClipper:
    def __init__(self, model, back_clip_params=None, input_clip_params=None, auto_activation_scale=0.5, auto_weight_grad_scale=1e-4,
                 device="cpu", input_size=(1, 1, 28, 28)):
        self.hooks_enabled = True
        self.parameter_ind = 0
        self.back_clip_params = [] if back_clip_params is None else back_clip_params
        self.layer_ind = 0
        self.input_clip_params = [] if input_clip_params is None else input_clip_params
        self.device = device
        self.auto_activation_scale = auto_activation_scale
        self.auto_weight_grad_scale = auto_weight_grad_scale
        self.grad_l2_bounds = []
        self._store_shapes(model, input_size)
        self.convert(model, auto_params=(back_clip_params is None or input_clip_params is None))
        print("L2 Bounds:", self.grad_l2_bounds)
        print("Backprop Clipping Params:", self.back_clip_params)
        print("Forward Clipping Params:", self.input_clip_params)

    def _store_shapes(self, model, input_size):
        """in_shape / out_shape of every leaf layer (what torchinfo.summary's layer_info gives, backprop_clip.py:123-128)."""
        handles = []

        def hook(m, inp, out):
            m.in_shape, m.out_shape = tuple(inp[0].shape[1:]), tuple(out.shape[1:])
        for m in model.modules():
            if len(list(m.children())) < 1:
                handles.append(m.register_forward_hook(hook))
        p = next(model.parameters())
        was = model.training
        # the reference's probe has no label, so its conditional discriminators fail inside torchinfo; a zero label is passed here
        y = torch.zeros(input_size[0], dtype=torch.long, device=p.device) if getattr(model, "n_classes", 0) > 1 else None
        from . import nn as hip_nn
        log = {}
        hip_nn.set_shape_log(log)          # the HIP convs chain NHWC tensors without going through Module.__call__
        try:
            with torch.no_grad():
                model(torch.zeros(input_size, device=p.device, dtype=p.dtype), y)
        finally:
            hip_nn.set_shape_log(None)
            for h in handles:
                h.remove()
        model.train(was)
        for m, (i, o) in log.items():
            m.in_shape, m.out_shape = i, o

    def enable_hooks(self):
        self.hooks_enabled = True

    def disable_hooks(self):
        self.hooks_enabled = False

    def convert(self, module, auto_params=False):
        for name, m in module.named_modules():
            if module_requires_grad(m) and len(list(m.children())) < 1:
                m._bpc = self._layer_clip(m, auto_params)

    def _layer_clip(self, m, auto_params):
        """The parameter bookkeeping of PGCWrapper.__init__ (backprop_clip.py:49-96), line for line in its order."""
        p = list(m.parameters())
        n_p = len(p)
        if auto_params:
            input_clip = l2_size(prod(m.in_shape), self.auto_activation_scale)
            self.input_clip_params.append(input_clip)
            if isinstance(m, nn.Linear):
                self.grad_l2_bounds.append(l2_size(p[0].numel(), self.auto_weight_grad_scale))
                back_clip = self.grad_l2_bounds[self.parameter_ind] / input_clip
                self.back_clip_params.append(back_clip)
                if n_p > 1:
                    self.grad_l2_bounds.append(back_clip)
            elif isinstance(m, nn.Conv2d):
                self.grad_l2_bounds.append(l2_size(p[0].numel(), self.auto_weight_grad_scale))
                back_clip = l2_to_l1(self.grad_l2_bounds[self.parameter_ind], prod(m.out_shape[1:])) / input_clip
                self.back_clip_params.append(back_clip)
                if n_p > 1:
                    self.grad_l2_bounds.append(back_clip * prod(m.out_shape[1:]))
            else:
                raise NotImplementedError("backprop clipping knows Linear and Conv2d layers (as the reference)")
        else:
            input_clip = self.input_clip_params[self.layer_ind]
            back_clip = self.back_clip_params[self.layer_ind]
            if isinstance(m, nn.Linear):
                self.grad_l2_bounds.append(input_clip * back_clip)
                if n_p > 1:
                    self.grad_l2_bounds.append(back_clip)
            elif isinstance(m, nn.Conv2d):
                self.grad_l2_bounds.append(input_clip * l2_to_l1(back_clip, prod(m.out_shape[1:])))
                if n_p > 1:
                    self.grad_l2_bounds.append(back_clip * prod(m.out_shape[1:]))
        self.layer_ind += 1
        self.parameter_ind += n_p
        return LayerClip(self, input_clip, back_clip)
