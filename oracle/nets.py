"""Oracle restatement of the reference generator / discriminator stacks (CPU, torch.nn).

TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows (reference file:line):
  models.py:7-67            Generator / Discriminator bases, aux_loss
  DCResNet_models.py:8-17   UpsampleConv  (cat x4 on channels + pixel_shuffle(2), then "same" conv).  NOT a nearest
                            up-sample: pixel_shuffle is channel-major, so out[c,2h+i,2w+j] = x[(4c+2i+j) mod C,h,w]
  DCResNet_models.py:19-38  ResBlockUp
  DCResNet_models.py:72-107 DCResNetGenerator
  DCResNet_models.py:109-153 DCResNetDiscriminator
  MNIST_models.py:9-60      vanilla MLP G/D and the MNIST DCRN sizes
  CelebA_models.py:10-24    CelebA DCRN sizes
  init_util.py:44-71        construction order + seeding (G first, then D, one RNG stream)

Layers are created in the same order as the reference so that a given
``weights_seed`` yields the same initial weights (torch.nn default inits draw
weight first, bias second, from the global CPU generator).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

# ---------------------------------------------------------------------------
# size tables (CelebA_models.py:10-24, MNIST_models.py:54-60)
# ---------------------------------------------------------------------------
G_SPECS = {
    ("CelebA", 64): dict(z_dim=128, channels=(512, 512, 256, 128, 64), first=4, out_ch=3),
    ("CelebA", 48): dict(z_dim=128, channels=(512, 512, 256, 128), first=6, out_ch=3),
    ("MNIST", 28): dict(z_dim=128, channels=(128, 128, 64), first=7, out_ch=1),
    # build extension (BASELINE config 5): 128x128, one more up block
    ("CelebA", 128): dict(z_dim=128, channels=(512, 512, 256, 128, 64, 64), first=4, out_ch=3),
}
D_SPECS = {
    ("CelebA", 64): dict(channels=(3, 64, 128, 256, 512), last=4),
    ("CelebA", 48): dict(channels=(3, 128, 256, 512), last=6),
    ("MNIST", 28): dict(channels=(1, 64, 128), last=7),
    ("CelebA", 128): dict(channels=(3, 64, 128, 256, 512), last=8),
}


def one_hot(y, n):
    return F.one_hot(y, n)


# ---------------------------------------------------------------------------
# Activation-mask playback (parity tests only).  ReLU / LeakyReLU make gradient TENSORS discontinuous in the
# pre-activations: a unit within fp32 rounding of zero may take the other slope on another device.  To compare
# gradients entry by entry, a test records the sign masks the device path used (csl_gan_amd.nn.ActivationMaskRecorder)
# and installs a MaskPlayer here: every piecewise-linear activation of the oracle then multiplies by the recorded
# slope pattern instead of deciding from its own pre-activation.  With identical masks both sides are the same smooth
# function, so any remaining difference is a wiring / arithmetic error, not a flipped unit.
# ---------------------------------------------------------------------------
class MaskPlayer:
    def __init__(self, masks, **nets):
        """masks: {"<tag>.<module name>": [bool tensors in call order]}; nets: tag -> oracle module (names as named_modules())."""
        self.masks = {k: list(v) for k, v in masks.items()}
        self.names = {id(m): "%s.%s" % (tag, n) for tag, net in nets.items() for n, m in net.named_modules()}

    def act(self, module, pre, slope):
        q = self.masks[self.names[id(module)]]
        m, n = q[0], pre.size(0)
        if m.size(0) == n:
            q.pop(0)
        else:                       # the device ran several logical passes as one concatenated batch: consume its rows in order
            assert m.size(0) > n, "recorded mask has %d rows, forward has %d" % (m.size(0), n)
            q[0], m = m[n:], m[:n]
        assert m.shape == pre.shape, (self.names[id(module)], tuple(m.shape), tuple(pre.shape))
        return pre * torch.where(m, 1.0, float(slope)).to(pre.dtype)

    def exhausted(self):
        return all(len(q) == 0 for q in self.masks.values())


_player = None


def set_mask_player(p):
    global _player
    _player = p


def _act(module, pre, slope):
    """LeakyReLU(slope) / ReLU (slope 0) of `pre`, the output of `module` — or the recorded mask when a player is installed."""
    if _player is not None:
        return _player.act(module, pre, slope)
    return F.leaky_relu(pre, slope) if slope else F.relu(pre)


class _UpConv(nn.Module):
    """cat([x]*4, dim=1) -> pixel_shuffle(2) -> 'same' conv (DCResNet_models.py:8-17), literally."""

    def __init__(self, cin, cout, k, bias=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding="same", bias=bias)

    def forward(self, x):
        # a channel-interleaving depth-to-space: up[c, 2h+i, 2w+j] = x[(4c + 2i + j) mod C, h, w]
        up = F.pixel_shuffle(torch.cat([x, x, x, x], 1), 2)
        return self.conv(up)


class _ResUp(nn.Module):
    def __init__(self, cin, cout, k, bn):
        super().__init__()
        norm = (lambda c: nn.BatchNorm2d(c)) if bn else (lambda c: nn.GroupNorm(32, c))
        # creation order matters for RNG parity: shortcut, bn1, convUp, bn2, conv
        self.shortcut = _UpConv(cin, cout, 1)
        self.bn1 = norm(cin)
        self.convUp = _UpConv(cin, cout, k, bias=False)
        self.bn2 = norm(cout)
        self.conv = nn.Conv2d(cout, cout, k, padding="same")

    def forward(self, x):
        s = self.shortcut(x)
        o = self.convUp(_act(self.bn1, self.bn1(x), 0.0))
        o = self.conv(_act(self.bn2, self.bn2(o), 0.0))
        return o + s


class OracleDCRNGenerator(nn.Module):
    def __init__(self, spec, z_dim=None, bn=True, n_classes=0, emb_mode="concat"):
        super().__init__()
        self.z_dim = z_dim if z_dim is not None else spec["z_dim"]
        self.n_classes, self.emb_mode, self.first = n_classes, emb_mode, spec["first"]
        ch = spec["channels"]
        self.emb = nn.Embedding(n_classes, self.z_dim) if (n_classes > 1 and emb_mode == "embed") else None
        self.linIn = nn.Linear(self.z_dim + (n_classes if emb_mode == "concat" else 0), self.first ** 2 * ch[0])
        self.blocks = nn.ModuleList([_ResUp(ch[i - 1], ch[i], 5, bn) for i in range(1, len(ch))])
        self.bn = nn.BatchNorm2d(ch[-1]) if bn else nn.GroupNorm(32, ch[-1])
        self.convOut = nn.Conv2d(ch[-1], spec["out_ch"], 3, padding="same")

    def forward(self, z, y=None):
        x = z
        if y is not None:
            if self.emb_mode == "embed":
                x = z * self.emb(y)
            elif self.emb_mode == "concat":
                x = torch.cat((z, one_hot(y, self.n_classes)), dim=1)
        x = self.linIn(x).reshape(z.size(0), -1, self.first, self.first)
        for blk in self.blocks:
            x = blk(x)
        return torch.tanh(self.convOut(_act(self.bn, self.bn(x), 0.0)))

    def loss(self, d_out, device=None):  # DCResNet_models.py:106-107
        return -d_out.mean()


class _DiscBase(nn.Module):
    """models.py:23-67."""

    def __init__(self, n_classes=0, emb_mode="concat", conditional_arch="CGAN",
                 aux_loss_type="wasserstein", aux_loss_scalar=1):
        super().__init__()
        self.n_classes, self.emb_mode = n_classes, emb_mode
        self.conditional_arch, self.aux_loss_type, self.aux_loss_scalar = conditional_arch, aux_loss_type, aux_loss_scalar
        if n_classes > 1:
            if emb_mode == "embed":
                raise Exception("Embed for D not implemented")
            if conditional_arch == "ACGAN":
                self.emb_mode = None  # models.py:37 — ACGAN critic does not see the label

    def aux_loss(self, output, labels, device=None, fake=False):
        # models.py:51-67
        if self.conditional_arch == "ACGAN":
            if self.aux_loss_type == "wasserstein":
                oh = one_hot(labels, self.n_classes)
                sign = oh * (-2) + 1
                per_class = oh.sum(dim=0)[labels].unsqueeze(1).expand_as(output)
                return self.aux_loss_scalar * torch.sum(sign * torch.sigmoid(output) / per_class)
            return self.aux_loss_scalar * F.cross_entropy(output, labels)
        if self.conditional_arch == "WCGAN":
            return torch.zeros(1, device=output.device if output is not None else device)
        return None  # reference falls off the end for CGAN (models.py:51-67)


class OracleDCRNDiscriminator(_DiscBase):
    def __init__(self, spec, **kw):
        super().__init__(**kw)
        ch = list(spec["channels"])
        if self.emb_mode == "concat" and self.n_classes > 1:
            ch[0] += self.n_classes
        self.blocks = nn.ModuleList([nn.Conv2d(ch[i - 1], ch[i], 5, stride=2, padding=2) for i in range(1, len(ch))])
        feat = ch[-1] * spec["last"] ** 2
        if self.n_classes < 2 or self.conditional_arch != "WCGAN":
            self.linOut = nn.Linear(feat, 1, bias=False)
        if self.n_classes > 1 and self.conditional_arch in ("ACGAN", "WCGAN"):
            self.linOutAux = nn.Linear(feat, self.n_classes, bias=True)

    def forward(self, x, y=None, aux=True):
        o = x
        if self.emb_mode == "concat" and self.n_classes > 1:
            planes = one_hot(y, self.n_classes).view(x.size(0), -1, 1, 1).expand(-1, -1, x.size(2), x.size(3))
            o = torch.cat((x, planes.to(x.dtype)), dim=1)
        for conv in self.blocks:
            o = _act(conv, conv(o), 0.2)
        o = o.reshape(x.size(0), -1)
        out_aux = self.linOutAux(o) if (aux and hasattr(self, "linOutAux")) else None
        if out_aux is not None and self.conditional_arch == "WCGAN":
            out = (out_aux * one_hot(y, self.n_classes)).sum(dim=1)
        else:
            out = self.linOut(o)
        return out, out_aux

    def real_loss(self, out, device=None):  # DCResNet_models.py:149-150
        return -out.mean()

    def fake_loss(self, out, device=None):  # DCResNet_models.py:152-153
        return out.mean()


class OracleVanillaG(nn.Module):
    """MNIST_models.py:9-26."""

    def __init__(self, z_dim=100, n_classes=0, **_):
        super().__init__()
        self.z_dim, self.n_classes = z_dim, n_classes
        self.lin1 = nn.Linear(z_dim + n_classes, 128)
        self.lin2 = nn.Linear(128, 784)

    def forward(self, z, y=None):
        x = z if y is None else torch.cat([z, one_hot(y, self.n_classes)], dim=1)
        return torch.sigmoid(self.lin2(_act(self.lin1, self.lin1(x), 0.0))).reshape(z.size(0), 1, 28, 28)

    def loss(self, d_out, device=None):
        return F.binary_cross_entropy_with_logits(d_out, torch.ones_like(d_out))


class OracleVanillaD(_DiscBase):
    """MNIST_models.py:28-52."""

    def __init__(self, **kw):
        super().__init__(**kw)
        if self.n_classes > 1 and self.aux_loss_type != "cross_entropy":
            raise Exception("Cross entropy loss is the only aux loss supported for vanilla architecture.")
        self.lin1 = nn.Linear(784 + self.n_classes, 128)
        self.lin2 = nn.Linear(128, 1)
        if self.n_classes > 1:
            self.linOutAux = nn.Linear(128, self.n_classes) if self.conditional_arch == "ACGAN" else None

    def forward(self, x, y=None, aux=True):
        o = x.reshape(x.size(0), -1)
        if y is not None:
            o = torch.cat([o, one_hot(y, self.n_classes).to(o.dtype)], dim=1)
        h = _act(self.lin1, self.lin1(o), 0.0)
        use_aux = aux and self.conditional_arch == "ACGAN" and self.n_classes > 1
        return self.lin2(h), (self.linOutAux(h) if use_aux else None)

    def real_loss(self, out, device=None):
        return F.binary_cross_entropy_with_logits(out, torch.ones_like(out))

    def fake_loss(self, out, device=None):
        return F.binary_cross_entropy_with_logits(out, torch.zeros_like(out))


def build_models(dataset="CelebA", model="DeepConvResNet", im_size=64, *, weights_seed=42, manual_seed=1,
                 conditional=False, n_classes=2, per_sample_grad=True, g_latent_dim=None,
                 g_label_emb_mode="concat", d_label_emb_mode="concat", conditional_arch="ACGAN",
                 aux_loss_type="wasserstein", aux_loss_scalar=1, init_G=True, init_D=True, dtype=torch.float32):
    """init_util.py:44-71 — seed, build G then D from one RNG stream, reseed."""
    ncls = n_classes if conditional else 0
    bn = not per_sample_grad
    torch.manual_seed(weights_seed)
    G = D = None
    dkw = dict(n_classes=ncls, emb_mode=d_label_emb_mode, conditional_arch=conditional_arch,
               aux_loss_type=aux_loss_type, aux_loss_scalar=aux_loss_scalar)
    if model == "Vanilla":
        if dataset != "MNIST":
            raise Exception("No vanilla architecture for CelebA.")
        if init_G:
            G = OracleVanillaG(z_dim=g_latent_dim or 100, n_classes=ncls)
        if init_D:
            D = OracleVanillaD(**dkw)
    else:
        key = (dataset, 28 if dataset == "MNIST" else im_size)
        if init_G:
            G = OracleDCRNGenerator(G_SPECS[key], z_dim=g_latent_dim, bn=bn, n_classes=ncls, emb_mode=g_label_emb_mode)
        if init_D:
            D = OracleDCRNDiscriminator(D_SPECS[key], **dkw)
    torch.manual_seed(manual_seed)
    if dtype != torch.float32:
        G = G.to(dtype) if G is not None else None
        D = D.to(dtype) if D is not None else None
    return G, D
