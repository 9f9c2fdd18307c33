"""Layer modules that keep torch.nn's parameter names / shapes / default initialisation (so
state_dict keys and weights_seed-initialised values match the reference's nn.Conv2d / nn.Linear /
nn.GroupNorm layers) but run their device math on the HIP kernels.

Device dispatch is explicit, never a fallback: a module whose parameters live on a HIP device always
calls the C-ABI library (and raises if it is missing); a module on the CPU runs the stock torch op —
that mode exists only for BASELINE.json configs[0] ("-gd cpu -dd cpu", plumbing without a GPU).
"""
from __future__ import annotations

import itertools

import torch
import torch.nn.functional as F
from torch import nn

from . import functional as HF
from . import ops


# Cache keys for repacked filters (ops.repack_cache).  id(parameter) is NOT safe: CPython reuses ids and the caching
# allocator reuses addresses, so a model built later in the same process (a test suite, a resumed run) could collide with
# a dead one's entry — same id, same data_ptr, same shape, same version counter, different weights.  A token is never reused.
_tokens = itertools.count(1)


class ActivationMaskRecorder:
    """Parity-test hook: records the sign mask (output > 0) of every fused ReLU / LeakyReLU this package applies, per module
    and in call order, as logical-NCHW bool tensors on the host.  The CPU oracle replays them (oracle.nets.MaskPlayer) so
    gradient tensors can be compared entry by entry without the discontinuity of a unit that sits at zero.  Off (None) in
    every product run; while installed, the fused depth-to-space normalisation takes its two-launch form."""

    def __init__(self, **nets):
        self.names = {id(m): "%s.%s" % (tag, n) for tag, net in nets.items() for n, m in net.named_modules()}
        self.masks = {}

    def record(self, module, y, nhwc=True):
        m = y.detach() > 0
        if nhwc and m.dim() == 4:
            m = m.permute(0, 3, 1, 2)
        self.masks.setdefault(self.names[id(module)], []).append(m.contiguous().cpu())


_mask_recorder = None


def set_mask_recorder(r):
    global _mask_recorder
    _mask_recorder = r


def _record_mask(module, y, act=None, nhwc=True):
    if _mask_recorder is not None and (act is None or act in (ops.ACT_LRELU02, ops.ACT_RELU)):
        _mask_recorder.record(module, y, nhwc)
    return y


class VersionRef:
    """The version counter of a parameter, readable again later: the cache key of a workspace DERIVED from a parameter (the pieces of a
    folded filter) — ops.repack_cache.refresh_pinned() re-reads it to decide whether a pinned workspace must be rebuilt."""

    def __init__(self, p):
        self.p = p

    version = property(lambda self: self.p._version)


class PerSampleSink:
    """Interface the DP engine implements to receive per-sample gradients from a layer's backward."""

    enabled = False

    def next_pass(self, layer) -> int:
        raise NotImplementedError

    def collector(self, layer):
        raise NotImplementedError


class _PerSampleMixin:
    _sink = None  # set by the engine (csl_gan_amd.engine.PrivacyEngine)
    _bpc = None   # set by csl_gan_amd.backprop_clip.BackpropClipper: per-sample input / output-gradient clip of this layer

    def _per_sample_active(self):
        s = self._sink
        return s is not None and s.enabled and torch.is_grad_enabled()


class HipConv2d(nn.Conv2d, _PerSampleMixin):
    """nn.Conv2d (DCResNet_models.py:118, :11, :26, :85) with a fused activation epilogue.

    act: ops.ACT_* applied in the conv kernel's epilogue (D: LeakyReLU 0.2, DCResNet_models.py:132).
    """

    def __init__(self, cin, cout, k, stride=1, padding=0, bias=True, act=ops.ACT_NONE):
        if padding == "same":
            padding = k // 2
        super().__init__(cin, cout, k, stride=stride, padding=padding, bias=bias)
        self.act = act
        self._wtoken = next(_tokens)

    def forward_nhwc(self, x, residual=None, in_mask=False, out_masked=False, out_dtype=None, in_affine=None, out=None):
        """x: NHWC-contiguous device tensor -> NHWC output (+ residual before the activation).
        out_dtype: torch.bfloat16 stores the output as bfloat16 (ops.set_storage_dtype; the model that owns the chain decides);
        None follows x's element type.

        Activation-backward fusion, decided by the MODEL that owns the chain (DCResNetDiscriminator.forward):
          in_mask     x is the LeakyReLU(0.2) output of a layer that was called with out_masked: this layer multiplies the data
                      gradient it sends to x by that slope pattern (recovered from the sign of x) in its kernel's epilogue;
          out_masked  every consumer of this layer's output does so, hence this layer skips its own activation-backward pass."""
        w = self.weight.permute(0, 2, 3, 1).contiguous()
        if in_affine is not None:
            # frozen forward only (ops.in_affine_ok): the producing layer's GroupNorm + ReLU rides in this conv's staging as a
            # per-(image, channel) affine map — x is the RAW conv output and the normalised activation never reaches HBM
            if torch.is_grad_enabled() or self._bpc is not None:
                raise RuntimeError("in_affine is an inference-only fusion")
            return ops.conv2d_fwd(x, w.detach(), None if self.bias is None else self.bias.detach(), stride=self.stride[0], pad=self.padding[0],
                                  residual=residual, act=self.act, wkey=self._wkey(w), in_affine=in_affine, out=out)
        bpc = self._bpc
        if bpc is not None:
            x = bpc.clip_input(x)              # backprop_clip.py:103 (PGCWrapper.forward)
        im, om = bool(in_mask), bool(out_masked) and self.act == ops.ACT_LRELU02
        if (im or om) and bpc is not None:
            raise RuntimeError("activation-backward fusion and backprop clipping cannot be combined on one layer")
        if self._per_sample_active() and residual is None:
            sink = self._sink
            y = HF.ConvPerSample.apply(x, w, self.bias, self.stride[0], self.padding[0], self.act, sink.collector(self), sink.next_pass(self),
                                       self._wkey(w), bpc, im, om, out_dtype)
        else:
            y = HF.Conv.apply(x, w, self.bias, self.stride[0], self.padding[0], self.act, residual, self._wkey(w), 1.0, None, bpc, im, om,
                              out_dtype)
        if _shape_log is not None:
            _shape_log[self] = ((x.shape[3], x.shape[1], x.shape[2]), (y.shape[3], y.shape[1], y.shape[2]))
        return _record_mask(self, y, self.act)

    def forward_shuffled(self, x_ps):
        """UpsampleConv's conv (DCResNet_models.py:16) on the depth-to-space tensor x_ps[N,2H,2W,C/4]: the reference convolves
        four identical channel groups, i.e. x_ps with the filter summed over the groups (a quarter of the MACs)."""
        w = self.weight.permute(0, 2, 3, 1).contiguous()
        if torch.is_grad_enabled() and w.requires_grad:
            wf = HF.FoldChannels4.apply(w)
            return HF.Conv.apply(x_ps, wf, self.bias, 1, self.padding[0], self.act, None, None, 4.0)
        wk = self._wkey(w)
        wf = ops.fold_channels4(w.detach(), wkey=wk)
        # wf is derived from the parameter (its own version counter is always 0, and a later fold may reuse its address): caches
        # downstream of it are keyed on the PARAMETER's version under a token of their own
        return HF.Conv.apply(x_ps, wf, self.bias, 1, self.padding[0], self.act, None, None if wk is None else (wk, "fold4"), 4.0,
                             VersionRef(self.weight))

    def _wkey(self, w):
        # only a zero-copy view of the parameter shares its version counter; a re-laid-out copy must not be cached
        return self._wtoken if w.data_ptr() == self.weight.data_ptr() else None

    def forward(self, x):
        if not x.is_cuda:
            return _forward_cpu(self, super().forward, x)
        return HF.nchw_view(self.forward_nhwc(HF.nhwc(x)))


_shape_log = None      # {layer: (in_shape, out_shape)} while BackpropClipper probes the model (the NHWC fast path skips module hooks)


def set_shape_log(d):
    global _shape_log
    _shape_log = d


def _forward_cpu(layer, plain_forward, x):
    """CPU plumbing path of a layer: optional backprop-clip of the input and of the pre-activation gradient around the stock op."""
    bpc = layer._bpc
    if bpc is None:
        return _act_cpu(plain_forward(x), layer.act)
    from .backprop_clip import ClipGrad
    return _act_cpu(ClipGrad.apply(plain_forward(bpc.clip_input(x)), bpc), layer.act)


class HipLinear(nn.Linear, _PerSampleMixin):
    """nn.Linear (DCResNet_models.py:77,124,126; MNIST_models.py:14-15,36-39) as a 1x1 conv on the
    MFMA kernel: x[B,in] is the NHWC tensor [B,1,1,in]."""

    def __init__(self, fin, fout, bias=True, act=ops.ACT_NONE):
        super().__init__(fin, fout, bias=bias)
        self.act = act
        self._wtoken = next(_tokens)

    def forward(self, x, in_mask=False):
        """in_mask: see HipConv2d.forward_nhwc."""
        if not x.is_cuda:
            return _forward_cpu(self, super().forward, x)
        B = x.shape[0]
        bpc = self._bpc
        if bpc is not None:
            x = bpc.clip_input(x)
        x4 = x.contiguous().reshape(B, 1, 1, self.in_features)
        w4 = self.weight.reshape(self.out_features, 1, 1, self.in_features)
        wkey = self._wtoken if w4.data_ptr() == self.weight.data_ptr() else None
        im, om = bool(in_mask) and bpc is None, False
        if self._per_sample_active():
            sink = self._sink
            y = HF.ConvPerSample.apply(x4, w4, self.bias, 1, 0, self.act, sink.collector(self), sink.next_pass(self), wkey, bpc, im, om)
        else:
            y = HF.Conv.apply(x4, w4, self.bias, 1, 0, self.act, None, wkey, 1.0, None, bpc, im, om)
        return _record_mask(self, y.reshape(B, self.out_features), self.act)


class HipGroupNormAct(nn.GroupNorm):
    """nn.GroupNorm(32, C) followed by ReLU (DCResNet_models.py:55-57, 63-67, 84, 101-102) as one HIP op on
    NHWC data, forward and backward (cslgan_groupnorm_act_f32 / cslgan_norm_act_bwd_f32)."""

    def __init__(self, groups, channels, relu=True):
        super().__init__(groups, channels)
        self.relu = relu

    def forward_nhwc(self, x, part=None):
        """part: (partials, slots per image) left by the conv that produced x (ops.gn_partials) — inference only."""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            y = HF.NormAct.apply(x, self.weight, self.bias, self.num_groups, self.eps, self.relu, None, None, 0.0)
        elif part is not None and not ops.storage_bf16() and x.dtype == torch.float32:
            y = ops.groupnorm_act(x, self.weight.detach(), self.bias.detach(), self.num_groups, eps=self.eps, relu=self.relu, part=part)
        else:
            # --storage_dtype bf16: the frozen generator of a D-step keeps its activations in HBM as bfloat16 from here on (the convs
            # answer bf16 inputs in kind); a differentiated forward (train_G) stays fp32
            y = ops.groupnorm_act(x, self.weight.detach(), self.bias.detach(), self.num_groups, eps=self.eps, relu=self.relu,
                                  out_dtype=torch.bfloat16 if ops.storage_bf16() else None)
        return _record_mask(self, y) if self.relu else y

    def forward_shuffled(self, x, part=None):
        """(depth_to_space(act(norm(x))), depth_to_space(x)) — the inputs of ResBlockUp's convUp and shortcut."""
        if (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)) or _mask_recorder is not None:
            raw = x
            if ops.storage_bf16() and x.dtype == torch.float32 and not (torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad)):
                raw = ops.cast_bf16(x)        # what the fused kernel writes for the shortcut in the bf16 storage mode
            return HF.DepthToSpace.apply(self.forward_nhwc(x)), HF.DepthToSpace.apply(raw)
        if part is not None and not ops.storage_bf16() and x.dtype == torch.float32:
            return ops.groupnorm_act(x, self.weight.detach(), self.bias.detach(), self.num_groups, eps=self.eps, relu=self.relu,
                                     d2s=True, want_raw=True, part=part)
        return ops.groupnorm_act(x, self.weight.detach(), self.bias.detach(), self.num_groups, eps=self.eps, relu=self.relu,
                                 d2s=True, want_raw=True, out_dtype=torch.bfloat16 if ops.storage_bf16() else None)

    def forward(self, x):
        if not x.is_cuda:
            y = super().forward(x)
            return F.relu(y) if self.relu else y
        return HF.nchw_view(self.forward_nhwc(HF.nhwc(x)))


def _act_cpu(y, act):
    if act == ops.ACT_LRELU02:
        return F.leaky_relu(y, 0.2)
    if act == ops.ACT_RELU:
        return F.relu(y)
    if act == ops.ACT_TANH:
        return torch.tanh(y)
    return y


def to_device_layout(module: nn.Module) -> nn.Module:
    """After .to(device): keep 4-D conv filters channels-last in HBM (KRSC), so the kernels read them
    in place and per-sample gradients / Adam state share that layout."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d) and m.weight.is_cuda:
            m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
    return module
