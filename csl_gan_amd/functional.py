"""Autograd glue over the HIP kernels.  All tensors here are NHWC-contiguous device tensors.

The three conv Functions form a closed set under differentiation, so the WGAN-GP double backward
(gradient_penalty.py:48-54 then train.py:427) runs entirely on the hand-written kernels:

    Conv   (x, w, b) -> y          d/dx = Dgrad(gz, w)       d/dw = Wgrad(gz, x)      d/db = sum(gz)
    Dgrad  (gy, w)   -> gx         d/dgy = Conv(ggx, w)      d/dw = Wgrad(gy, ggx)
    Wgrad  (gy, x)   -> gw         d/dgy = Conv(x, ggw)      d/dx = Dgrad(gy, ggw)

LeakyReLU / ReLU are fused into the forward kernel's epilogue; their backward multiplies by a mask
recovered from the OUTPUT (slope > 0 keeps the sign), so no pre-activation tensor is stored.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops

_SLOPE = {ops.ACT_LRELU02: 0.2, ops.ACT_RELU: 0.0}

_INPUT_GRADS_ONLY = False


class input_grads_only:
    """Context for forwards whose FIRST-order backward is only ever asked for input gradients — the
    gradient-penalty forward (gradient_penalty.py:46-50): parameters enter the penalty solely through the
    second-order Dgrad nodes, so the first-order weight / bias gradient kernels would be wasted work
    (autograd cannot tell a custom Function which of its outputs the caller requested)."""

    def __enter__(self):
        global _INPUT_GRADS_ONLY
        self._prev, _INPUT_GRADS_ONLY = _INPUT_GRADS_ONLY, True

    def __exit__(self, *a):
        global _INPUT_GRADS_ONLY
        _INPUT_GRADS_ONLY = self._prev


_PS_SINK = None


class per_sample_param_grads:
    """Context for the SECOND-order sweep of a per-sample gradient penalty (train.py:433-450, penalties computed on private
    data): the reference differentiates penalties[i] for every sample i separately (B autograd calls over the whole batch
    graph).  penalties[i] depends on row i alone, so the per-sample parameter gradients are the per-sample (group = 1) weight
    gradients of ONE sweep of sum_i penalties[i]: inside this context every second-order weight-gradient node hands its
    per-sample rows [B, numel(param)] (parameter memory order) to sink(param, rows) and returns no dense gradient."""

    def __init__(self, sink, params):
        self.sink, self.by_ptr = sink, {p.data_ptr(): p for p in params}

    def __enter__(self):
        global _PS_SINK
        self._prev, _PS_SINK = _PS_SINK, self

    def __exit__(self, *a):
        global _PS_SINK
        _PS_SINK = self._prev


def nhwc(x: torch.Tensor) -> torch.Tensor:
    """logical NCHW -> NHWC-contiguous view (copy only if x is not already channels-last)."""
    return x.permute(0, 2, 3, 1).contiguous()


def nchw_view(y: torch.Tensor) -> torch.Tensor:
    """NHWC-contiguous -> logical NCHW view (channels-last strides, no copy)."""
    return y.permute(0, 3, 1, 2)


class ActBwd(Function):
    """out = g * (y > 0 ? 1 : slope).  Linear in g; the mask has zero derivative."""

    @staticmethod
    def forward(ctx, g, y, slope):
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(y)
        ctx.slope = slope
        return ops.act_bwd(g.contiguous(), y, slope)

    @staticmethod
    def backward(ctx, gg):
        if gg is None:
            return None, None, None
        (y,) = ctx.saved_tensors
        return ActBwd.apply(gg, y, ctx.slope), None, None


class SegmentMeans(Function):
    """vec[s] = scale[s] * sum(x[row block s]), total = sum(vec): the critic's real_loss / fake_loss (DCResNet_models.py:149-153,
    -mean / +mean) over the row blocks of ONE fused pass and their sum, in one launch; the backward writes the (piecewise constant)
    cotangent in one launch (torch: three means, negations, adds, and their mirror image in the backward — ~17 launches)."""

    @staticmethod
    def forward(ctx, x, sizes, scale):
        ctx.set_materialize_grads(False)
        ctx.meta = (tuple(int(n) for n in sizes), tuple(float(c) for c in scale), x.shape)
        return ops.segment_means(x.contiguous().reshape(-1), ctx.meta[0], ctx.meta[1])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_vec, g_total):
        if g_vec is None and g_total is None:
            return None, None, None
        sizes, scale, shape = ctx.meta
        dev = (g_vec if g_vec is not None else g_total).device
        g_vec = None if g_vec is None else g_vec.contiguous()
        return ops.segment_means_bwd(g_total, g_vec, sizes, scale, shape, dev), None, None


class LipschitzTerm(Function):
    """coef * phi(||t_b||), phi(n) = (n-1)^2 or max(n-1,0)^2 (gradient_penalty.py:52-54), summed over the batch unless per_sample —
    one launch forward, one backward (torch: norm, sub, clamp, pow, mean, mul, mul and their backward)."""

    @staticmethod
    def forward(ctx, t, one_sided, coef, per_sample):
        t = t.contiguous()
        norm, per, total = ops.lipschitz_term(t, one_sided, coef)
        ctx.cfg = (bool(one_sided), float(coef), bool(per_sample))
        ctx.save_for_backward(t, norm)
        return per if per_sample else total

    @staticmethod
    def backward(ctx, g):
        t, norm = ctx.saved_tensors
        one_sided, coef, per_sample = ctx.cfg
        if torch.is_grad_enabled() and (g.requires_grad or t.requires_grad):
            # differentiable form (immediate sensitivity differentiates the parameter gradients once more): plain torch ops
            n = t.norm(2, dim=1)
            d = (n - 1).clamp(min=0) if one_sided else n - 1
            gb = g if per_sample else g.expand(t.shape[0])
            return (gb * coef * 2 * d / n).unsqueeze(1) * t, None, None, None
        g = g.contiguous()
        return ops.lipschitz_term_bwd(t, norm, None if per_sample else g, g if per_sample else None, one_sided, coef), None, None, None


class BiasGrad(Function):
    """gb[k] = sum over all pixels/samples of gy[...,k]."""

    @staticmethod
    def forward(ctx, gy):
        ctx.shape, ctx.dtype = gy.shape, gy.dtype
        N = gy.shape[0]
        group = _dense_group(N)
        part = ops.bias_grad_grouped(gy.contiguous(), group=group)
        if part.shape[0] == 1:
            return part[0]
        out = torch.empty(part.shape[1], device=gy.device, dtype=torch.float32)
        ops.sum_rows(part, out)
        return out

    @staticmethod
    def backward(ctx, ggb):
        return ggb.reshape((1,) * (len(ctx.shape) - 1) + (-1,)).expand(ctx.shape).to(ctx.dtype)


def _dense_group(N: int, tiles: int = 1) -> int:
    """Samples per slab for dense (summed) gradients: the largest group that still leaves >= 512
    workgroups (256 CUs x 2), so small layers keep per-sample slabs and big layers write few."""
    for g in (16, 8, 4, 2):
        if N % g == 0 and (N // g) * tiles >= 512:
            return g
    return 1


class Conv(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, residual, wkey=None, alg_scale=1.0, wversion=None, bpc=None, in_mask=False,
                out_masked=False, out_dtype=None):
        """out_dtype: torch.bfloat16 stores y as bfloat16 (ops.set_storage_dtype); None follows x's element type."""
        y = ops.conv2d_fwd(x, w, b, stride=stride, pad=pad, residual=residual, act=act, wkey=wkey, alg_scale=alg_scale, wversion=wversion,
                           out_dtype=out_dtype)
        ctx.set_materialize_grads(False)      # an output nobody differentiates reaches backward as None, not as a zero tensor
        ctx.in_mask, ctx.out_masked = in_mask, out_masked
        ctx.bpc = bpc              # backprop clipping (csl_gan_amd.backprop_clip): per-sample clip of the pre-activation gradient
        # the owning layer's cache token lets ops reuse repacked filters while the parameter is unchanged; a derived filter
        # (wversion given) must not be cached by the backward, whose caches key on w's own version counter
        ctx.wkey = wkey if wversion is None else None
        ctx.cfg = (stride, pad, act)
        ctx.has_res = residual is not None
        ctx.input_only = _INPUT_GRADS_ONLY
        ctx.save_for_backward(x, w, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:
            return (None,) * 14
        x, w, y = ctx.saved_tensors
        stride, pad, act = ctx.cfg
        gz = gy
        if ctx.out_masked:
            pass            # the consumer's data-gradient epilogue already applied this layer's LeakyReLU slope pattern
        elif act in _SLOPE:
            # y.detach(): the slope pattern has zero derivative, so the double backward must not see an edge from this node back
            # into the forward graph.  With the edge, autograd.grad(penalty, params) (train.py:427) walked the whole first-order
            # forward again with zero-filled gradients — a wasted data-gradient chain per step (0.53 ms of 13.6 at bs=128).
            gz = ActBwd.apply(gy, y.detach(), _SLOPE[act])
        elif act == ops.ACT_TANH:
            gz = gy * (1 - y * y)
        gz = gz.contiguous()
        if ctx.bpc is not None:
            gz = ctx.bpc.clip_grad(gz)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = Dgrad.apply(gz, w, x.shape[1], x.shape[2], stride, pad, ctx.wkey, x.detach() if ctx.in_mask else None, x.dtype)
        if ctx.needs_input_grad[1] and not ctx.input_only:
            gw = Wgrad.apply(gz, x, w.shape[1], w.shape[2], stride, pad, ctx.in_mask)
        if ctx.needs_input_grad[2] and not ctx.input_only:
            gb = BiasGrad.apply(gz)
        # the residual is added before the activation (ResBlockUp's "o + s", DCResNet_models.py:38): its gradient is gz
        gres = gz if (ctx.has_res and ctx.needs_input_grad[6]) else None
        return gx, gw, gb, None, None, None, gres, None, None, None, None, None, None, None


class DepthToSpace(Function):
    """UpsampleConv's cat([x]*4, 1) + pixel_shuffle(2) (DCResNet_models.py:13-15) reduced to its C/4 distinct channels:
    [N,H,W,C] -> [N,2H,2W,C/4] (ops.depth_to_space).  A permutation: the backward is the inverse map."""

    @staticmethod
    def forward(ctx, x):
        return ops.depth_to_space(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return SpaceToDepth.apply(g)


class SpaceToDepth(Function):
    @staticmethod
    def forward(ctx, g):
        return ops.depth_to_space(g.contiguous(), inverse=True)

    @staticmethod
    def backward(ctx, gg):
        return DepthToSpace.apply(gg)


class FoldChannels4(Function):
    """wf[k,r,s,c'] = sum_q w[k,r,s,c'+q*C/4]: the filter of UpsampleConv's conv as seen by the depth-to-space tensor,
    whose four channel groups are identical.  Linear; the backward copies the folded gradient to the four groups."""

    @staticmethod
    def forward(ctx, w):
        return ops.fold_channels4(w.contiguous())      # never the cached tensor: autograd owns this output

    @staticmethod
    def backward(ctx, gwf):
        return UnfoldChannels4.apply(gwf)


class UnfoldChannels4(Function):
    @staticmethod
    def forward(ctx, gwf):
        return ops.unfold_channels4(gwf.contiguous())

    @staticmethod
    def backward(ctx, ggw):
        return FoldChannels4.apply(ggw)


class NormAct(Function):
    """GroupNorm / BatchNorm (+ReLU) on NHWC data with the HIP forward and backward kernels.
    groups == 0 selects BatchNorm (batch statistics per channel, running stats updated in place)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu, running_mean, running_var, momentum):
        if groups > 0:
            y, stats = ops.groupnorm_act(x, gamma, beta, groups, eps=eps, relu=relu, return_stats=True)
            ctx.rows_per_stat, ctx.groups = x.shape[1] * x.shape[2], groups
        else:
            y, stats = ops.batchnorm_act(x, gamma, beta, running_mean, running_var, momentum=momentum, eps=eps, relu=relu,
                                         return_stats=True)
            ctx.rows_per_stat, ctx.groups = x.numel() // x.shape[-1], x.shape[-1]
        ctx.eps, ctx.relu = eps, relu
        ctx.save_for_backward(x, y, gamma, stats)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gy):
        x, y, gamma, stats = ctx.saved_tensors
        dx, dgamma, dbeta = ops.norm_act_bwd(x, gy.contiguous(), y, gamma, stats, ctx.rows_per_stat, ctx.groups, ctx.eps, ctx.relu)
        return dx, dgamma, dbeta, None, None, None, None, None, None


class Dgrad(Function):
    @staticmethod
    def forward(ctx, gy, w, H, W, stride, pad, wkey=None, mask=None, out_dtype=None):
        """mask (optional, the conv's own input when that is a LeakyReLU(0.2) output): gx *= lrelu'(mask) in the kernel's epilogue.
        out_dtype: element type of gx = that of the conv's input x (bf16-stored activations get bf16 gradients)."""
        ctx.set_materialize_grads(False)
        ctx.cfg = (H, W, stride, pad)
        ctx.wkey = wkey
        ctx.save_for_backward(gy, w, mask)
        return ops.conv2d_dgrad(gy, w, (H, W), stride=stride, pad=pad, wkey=wkey, mask=mask, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, ggx):
        if ggx is None:
            return (None,) * 9
        gy, w, mask = ctx.saved_tensors
        H, W, stride, pad = ctx.cfg
        ggx = ggx.contiguous()
        if mask is not None:        # the epilogue's pointwise factor, applied to the incoming cotangent first (it is linear)
            ggx = ActBwd.apply(ggx, mask, 0.2)
        g_gy = g_w = None
        if ctx.needs_input_grad[0]:
            g_gy = Conv.apply(ggx, w, None, stride, pad, ops.ACT_NONE, None, ctx.wkey, 1.0, None, None, False, False, gy.dtype)
        if ctx.needs_input_grad[1]:
            ps = _PS_SINK
            param = None if ps is None else ps.by_ptr.get(w.data_ptr())
            if param is not None:
                with torch.no_grad():
                    K, R, S, Cc = w.shape
                    rows = ops.conv2d_wgrad_grouped(gy.contiguous(), ggx, R, S, stride=stride, pad=pad, group=1, alpha=1.0)
                    ps.sink(param, rows.reshape(rows.shape[0], -1))
            else:
                g_w = Wgrad.apply(gy, ggx, w.shape[1], w.shape[2], stride, pad)
        return g_gy, g_w, None, None, None, None, None, None, None


class Wgrad(Function):
    """Dense weight gradient: grouped slabs on the MFMA kernel, then a column sum."""

    @staticmethod
    def forward(ctx, gy, x, R, S, stride, pad, in_mask=False):
        """in_mask: x is a LeakyReLU(0.2) output whose producer relies on its consumers to apply the slope pattern
        (fused_act_masks): the gradient this node sends to x (immediate sensitivity differentiates weight gradients with respect
        to the inputs) must carry it too."""
        ctx.set_materialize_grads(False)
        ctx.in_mask = in_mask
        ctx.cfg = (R, S, stride, pad)
        ctx.save_for_backward(gy, x)
        return ops.conv2d_wgrad_dense(gy, x, R, S, stride=stride, pad=pad)

    @staticmethod
    def backward(ctx, ggw):
        if ggw is None:
            return (None,) * 7
        gy, x = ctx.saved_tensors
        R, S, stride, pad = ctx.cfg
        ggw = ggw.contiguous()
        g_gy = g_x = None
        if ctx.needs_input_grad[0]:
            g_gy = Conv.apply(x, ggw, None, stride, pad, ops.ACT_NONE, None, None, 1.0, None, None, False, False, gy.dtype)
        if ctx.needs_input_grad[1]:
            g_x = Dgrad.apply(gy, ggw, x.shape[1], x.shape[2], stride, pad, None, x.detach() if ctx.in_mask else None, x.dtype)
        return g_gy, g_x, None, None, None, None, None


class ConvPerSample(Function):
    """Forward identical to Conv; backward writes per-sample weight/bias gradients into the engine's
    store (the Opacus-hook replacement, train.py:373,387) and returns only the data gradient."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, act, sink, pass_idx, wkey=None, bpc=None, in_mask=False, out_masked=False, out_dtype=None):
        stored = x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16       # the bf16 filter copy is cached per parameter version
        y = ops.conv2d_fwd(x, w, b, stride=stride, pad=pad, act=act, wkey=wkey if stored else None, out_dtype=out_dtype)
        ctx.wkey, ctx.bpc = wkey, bpc
        ctx.in_mask, ctx.out_masked = in_mask, out_masked
        ctx.cfg = (stride, pad, act)
        ctx.sink, ctx.pass_idx = sink, pass_idx
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act != ops.ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        stride, pad, act = ctx.cfg
        with torch.no_grad():
            gz = gy.contiguous()
            if ctx.out_masked:
                pass        # already multiplied by this layer's slope pattern in the consumer's data-gradient epilogue
            elif act in _SLOPE:
                gz = ops.act_bwd(gz, y, _SLOPE[act])
            elif act == ops.ACT_TANH:
                gz = gz * (1 - y * y)
            if ctx.bpc is not None:
                gz = ctx.bpc.clip_grad(gz)
            side = ctx.sink.side_stream()
            if side is None:
                ctx.sink.collect(ctx.pass_idx, gz, x, w.shape[1], w.shape[2], stride, pad, ctx.has_bias)
            else:
                # the per-sample / norm / dense weight-gradient launches do not feed the backward chain: they go to a side
                # stream and fill the CUs the (small) data-gradient launches leave idle; the engine joins before clip()
                cur = torch.cuda.current_stream()
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ctx.sink.collect(ctx.pass_idx, gz, x, w.shape[1], w.shape[2], stride, pad, ctx.has_bias)
                gz.record_stream(side)
                x.record_stream(side)
            gx = None
            if ctx.needs_input_grad[0]:
                gx = ops.conv2d_dgrad(gz, w, (x.shape[1], x.shape[2]), stride=stride, pad=pad, wkey=ctx.wkey,
                                      mask=x if ctx.in_mask else None, out_dtype=x.dtype)
        return gx, None, None, None, None, None, None, None, None, None, None, None, None


class RowL2Norm(Function):
    """norms[b] = ||t[b,:]||_2 (gradient_penalty.py:52-53) with a differentiable backward."""

    @staticmethod
    def forward(ctx, t):
        t = t.contiguous()
        n = ops.row_l2norm(t)
        ctx.save_for_backward(t, n)
        return n

    @staticmethod
    def backward(ctx, gn):
        t, n = ctx.saved_tensors
        if torch.is_grad_enabled() and (gn.requires_grad or t.requires_grad):
            # differentiable form for higher-order use (never hit by the D-step: the norm's backward is
            # the last first-order node before the parameter gradients)
            return gn.unsqueeze(1) * t / n.unsqueeze(1)
        return ops.row_l2norm_bwd(t, n, gn.contiguous())
