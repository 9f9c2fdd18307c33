"""Deep-conv ResNet GAN stacks (reference DCResNet_models.py) on the HIP kernels.

Same class names, constructor arguments, sub-module names (state_dict keys) and construction
order (weights_seed parity) as the reference; device tensors run NHWC through
csl_gan_amd.nn.HipConv2d / HipLinear / HipGroupNormAct.

MI355X-first differences, all exact re-associations of the reference arithmetic:
  * UpsampleConv (DCResNet_models.py:8-17) is cat([x]*4, 1) + pixel_shuffle(2) + conv.  pixel_shuffle is channel-major, so the
    shuffled tensor is up[c,2h+i,2w+j] = x[(4c+2i+j) mod C,h,w]: its C channels are the C/4 channels of the plain depth-to-space
    tensor ps[c',2h+i,2w+j] = x[4c'+2i+j,h,w] repeated four times.  The device path therefore never builds `up`: the
    normalisation kernel writes ps directly and the conv runs on ps with its filter summed over the four channel groups
    (ops.fold_channels4) — a quarter of the reference's multiply-adds for these layers;
  * the 1x1 shortcut's result is added in the epilogue of the block's last conv;
  * bias, LeakyReLU(0.2) / tanh are conv-kernel epilogues; GroupNorm/BatchNorm+ReLU is one op.
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import functional as HF
from . import ops
from .models import Discriminator, Generator
from .nn import HipConv2d, HipGroupNormAct, HipLinear


class UpsampleConv(nn.Module):
    """cat([x]*4, dim=1) -> pixel_shuffle(2) -> 'same' conv (DCResNet_models.py:8-17)."""

    def __init__(self, in_ch, out_ch, filter_size, bias=True):
        super().__init__()
        self.conv = HipConv2d(in_ch, out_ch, filter_size, padding="same", bias=bias)

    def forward_shuffled(self, x_ps):
        """x_ps: the depth-to-space tensor [N,2H,2W,C/4] (NHWC) of this layer's input."""
        return self.conv.forward_shuffled(x_ps)

    def forward(self, x):
        if x.is_cuda:
            if x.shape[1] % 4:
                raise NotImplementedError("UpsampleConv on HIP needs in_ch %% 4 == 0 (got %d)" % x.shape[1])
            return HF.nchw_view(self.forward_shuffled(HF.DepthToSpace.apply(HF.nhwc(x))))
        return self.conv(F.pixel_shuffle(torch.cat([x, x, x, x], 1), 2))


class _BatchNormAct(nn.BatchNorm2d):
    """BatchNorm2d + ReLU for the bn=True generator of the non-per-sample modes (init_util.py:46): batch statistics and the
    running-stat update run on cslgan_batchnorm_act_f32, the backward on cslgan_norm_act_bwd_f32; eval mode (running
    statistics: the sampling path train.py:298-308, gensamples.py:26-41) on cslgan_batchnorm_eval_act_f32."""

    def _train_stats(self):
        return self.training or self.running_mean is None

    def forward_nhwc(self, x, d2s=False):
        from . import nn as hnn
        if hnn._mask_recorder is not None:          # parity-test hook (csl_gan_amd.nn.ActivationMaskRecorder): two-launch form
            y = hnn._record_mask(self, self._forward_nhwc(x, False))
            return (HF.DepthToSpace.apply(y), HF.DepthToSpace.apply(x)) if d2s else y
        return self._forward_nhwc(x, d2s)

    def _forward_nhwc(self, x, d2s):
        g, b = self.weight, self.bias
        if not self._train_stats():
            if torch.is_grad_enabled() and (x.requires_grad or g.requires_grad):
                raise NotImplementedError("eval-mode BatchNorm on HIP is inference-only (no backward)")
            return ops.batchnorm_eval_act(x, g.detach(), b.detach(), self.running_mean, self.running_var, eps=self.eps, relu=True,
                                          d2s=d2s, want_raw=d2s)
        if self.num_batches_tracked is not None:
            self.num_batches_tracked += 1
        if torch.is_grad_enabled() and (x.requires_grad or g.requires_grad):
            y = HF.NormAct.apply(x, g, b, 0, self.eps, True, self.running_mean, self.running_var, self.momentum)
            return (HF.DepthToSpace.apply(y), HF.DepthToSpace.apply(x)) if d2s else y
        return ops.batchnorm_act(x, g.detach(), b.detach(), self.running_mean, self.running_var, momentum=self.momentum,
                                 eps=self.eps, relu=True, d2s=d2s, want_raw=d2s)

    def forward_shuffled(self, x):
        return self.forward_nhwc(x, d2s=True)

    def forward(self, x):
        if x.is_cuda:
            return HF.nchw_view(self.forward_nhwc(HF.nhwc(x)))
        return F.relu(super().forward(x))


def _gn_fusable(x):
    """GroupNorm statistics may come from the producing conv's epilogue: a frozen fp32 device forward with no recorder attached."""
    from . import nn as hnn
    return (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32 and not ops.storage_bf16() and hnn._mask_recorder is None


def _norm_as_affine(norm, x, part, conv):
    """GroupNorm(+ReLU) `norm` of x folded into the staging of `conv` (the only consumer): (scale, shift, relu) tables from the
    statistics x's producer left (ops.groupnorm_affine), or None when the statistics or the conv's route are not there."""
    if part is None or not ops.in_affine_ok(x, conv.weight.permute(0, 2, 3, 1), conv.stride[0], conv.padding[0]):
        return None
    N, H, W, C = x.shape
    sc, sh = ops.groupnorm_affine(part, norm.weight.detach(), norm.bias.detach(), norm.num_groups, norm.eps, N, H * W, C)
    return sc, sh, bool(norm.relu)


def _norm_act(bn, ch):
    return _BatchNormAct(ch) if bn else HipGroupNormAct(32, ch, relu=True)


class ResBlockUp(nn.Module):
    """DCResNet_models.py:19-38.  Sub-module creation order is part of the weights_seed contract."""

    def __init__(self, in_ch, out_ch, filter_size, bn=True):
        super().__init__()
        self.shortcut = UpsampleConv(in_ch, out_ch, 1)
        self.bn1 = _norm_act(bn, in_ch)
        self.convUp = UpsampleConv(in_ch, out_ch, filter_size, bias=False)
        self.bn2 = _norm_act(bn, out_ch)
        self.conv = HipConv2d(out_ch, out_ch, filter_size, padding="same")

    def forward_nhwc(self, x):
        return self.forward_nhwc_parts(x)[0]

    def forward_nhwc_parts(self, x, x_part=None, next_norm=None):
        """(block output, GroupNorm partial statistics of it for `next_norm` or None).  In a frozen forward (the generator inside a
        D-step) every GroupNorm of the stack reads its statistics from the epilogue of the conv that produced its input
        (ops.gn_partials, cslgan_conv_t.gn_part): x_part are the ones the previous block left for this block's bn1."""
        if x.shape[-1] % 4:
            raise NotImplementedError("ResBlockUp on HIP needs in_ch %% 4 == 0 (got %d)" % x.shape[-1])
        fuse = _gn_fusable(x)
        a_ps, x_ps = self.bn1.forward_shuffled(x, part=x_part) if x_part is not None else self.bn1.forward_shuffled(x)
        s = self.shortcut.forward_shuffled(x_ps)        # 1x1 conv over the C/4 shuffled channels, full resolution
        affine = None
        if fuse and isinstance(self.bn2, HipGroupNormAct):
            with ops.gn_partials(self.bn2.num_groups) as cell:
                o = self.convUp.forward_shuffled(a_ps)
            affine = _norm_as_affine(self.bn2, o, cell.part, self.conv)
            h = o if affine is not None else self.bn2.forward_nhwc(o, part=cell.part)
        else:
            h = self.bn2.forward_nhwc(self.convUp.forward_shuffled(a_ps))
        if fuse and isinstance(next_norm, HipGroupNormAct):
            with ops.gn_partials(next_norm.num_groups) as cell:
                out = self.conv.forward_nhwc(h, residual=s, in_affine=affine)
            return out, cell.part
        return self.conv.forward_nhwc(h, residual=s, in_affine=affine), None

    def forward(self, x):
        if x.is_cuda:
            return HF.nchw_view(self.forward_nhwc(HF.nhwc(x)))
        return self.conv(self.bn2(self.convUp(self.bn1(x)))) + self.shortcut(x)


class DCResNetGenerator(Generator):
    def __init__(self, channels, first_filter_size, **kwargs):
        super().__init__(**kwargs)
        self.first_filter_size = first_filter_size
        extra = self.n_classes if self.emb_mode == "concat" else 0
        self.linIn = HipLinear(self.z_dim + extra, first_filter_size ** 2 * channels[0])
        self.blocks = nn.ModuleList([ResBlockUp(a, b, 5, bn=self.bn) for a, b in zip(channels[:-1], channels[1:])])
        self.bn = _norm_act(self.bn, channels[-1])
        self.convOut = HipConv2d(channels[-1], self.out_ch, 3, padding="same", act=ops.ACT_TANH)

    def forward(self, z, y=None, out=None):
        """out (frozen device forward only): an NHWC-contiguous [B, H, W, out_ch] fp32 tensor the output conv writes the images
        into — the trainer hands in the generated rows' slice of the fused critic batch, so no copy follows."""
        f = self.first_filter_size
        x = self.linIn(self._condition(z, y)).reshape(z.size(0), -1, f, f)
        if not x.is_cuda:
            for blk in self.blocks:
                x = blk(x)
            return self.convOut(self.bn(x))
        x = HF.nhwc(x)
        part = None
        for i, blk in enumerate(self.blocks):           # each block leaves the statistics the NEXT normalisation needs (frozen forward)
            nxt = self.blocks[i + 1].bn1 if i + 1 < len(self.blocks) else self.bn
            x, part = blk.forward_nhwc_parts(x, part, nxt)
        affine = _norm_as_affine(self.bn, x, part, self.convOut) if part is not None else None
        if affine is not None:                          # the last normalisation rides in the output conv's staging
            return HF.nchw_view(self.convOut.forward_nhwc(x, in_affine=affine, out=out))
        h = self.bn.forward_nhwc(x, part=part) if part is not None else self.bn.forward_nhwc(x)
        return HF.nchw_view(self.convOut.forward_nhwc(h))

    def loss(self, d_output, device):
        return -torch.mean(d_output)


class DCResNetDiscriminator(Discriminator):
    linear_critic_losses = True     # real_loss = -mean(out), fake_loss = +mean(out): the trainer may evaluate them with one fused launch

    def __init__(self, channels, last_filter_size, **kwargs):
        super().__init__(**kwargs)
        channels = list(channels)       # the reference mutates its default list (DCResNet_models.py:115); we copy
        if self.emb_mode == "concat" and self.n_classes > 1:
            channels[0] += self.n_classes
        self.blocks = nn.ModuleList([HipConv2d(a, b, 5, stride=2, padding=2, act=ops.ACT_LRELU02)
                                     for a, b in zip(channels[:-1], channels[1:])])
        size = channels[-1] * last_filter_size ** 2
        if self.n_classes < 2 or self.conditional_arch != "WCGAN":
            self.linOut = HipLinear(size, 1, bias=False)
        if self.n_classes > 1 and self.conditional_arch in ("ACGAN", "WCGAN"):
            self.linOutAux = HipLinear(size, self.n_classes, bias=True)

    def forward(self, x, y=None, aux=True):
        B = x.size(0)
        o = x
        if self.emb_mode == "concat" and self.n_classes > 1:
            planes = F.one_hot(y, self.n_classes).to(x.dtype).view(B, -1, 1, 1).expand(-1, -1, x.size(2), x.size(3))
            o = torch.cat((x, planes), dim=1)
        fuse = False
        if o.is_cuda:
            # Activation-backward fusion: every LeakyReLU output of this stack is consumed ONLY by the next conv or by the linear
            # heads, so each consumer applies the producer's slope pattern in the epilogue of its data-gradient kernel and the
            # producer skips its own activation-backward pass (a read + a write of every activation gradient less).  Never with
            # backprop clipping: its input clip sits between the activation and the next conv.
            fuse = all(getattr(m, "_bpc", None) is None for m in self.modules())
            o = HF.nhwc(o)
            prev_lrelu = False
            # --storage_dtype bf16: the conv layers' outputs (and with them their gradients) live in HBM as bfloat16; the heads
            # answer in fp32 (csl_gan_amd.ops.set_storage_dtype)
            od = torch.bfloat16 if ops.storage_bf16() else None
            for blk in self.blocks:
                o = blk.forward_nhwc(o, in_mask=fuse and prev_lrelu, out_masked=fuse, out_dtype=od)   # conv + bias + LeakyReLU(0.2) in one kernel
                prev_lrelu = blk.act == ops.ACT_LRELU02
            fuse = fuse and prev_lrelu
            o = HF.nchw_view(o)
        else:
            for blk in self.blocks:
                o = blk(o)
        o = o.reshape(B, -1)                        # (c,h,w) feature order, as the reference's linOut expects
        head = (lambda lin: lin(o, in_mask=True)) if fuse else (lambda lin: lin(o))
        out_aux = head(self.linOutAux) if aux and hasattr(self, "linOutAux") else None
        if out_aux is not None and self.conditional_arch == "WCGAN":
            out = (out_aux * F.one_hot(y, self.n_classes)).sum(dim=1)
        else:
            out = head(self.linOut)
        return out, out_aux

    def real_loss(self, output, device):
        return -torch.mean(output)

    def fake_loss(self, output, device):
        return torch.mean(output)
