"""MNIST model family on the HIP layers: the vanilla 2-layer MLP GAN and the DCResNet sizes.

Behavioural contract (reference MNIST_models.py:9-60): class names, constructor keywords, sub-module
names lin1 / lin2 / linOutAux (state_dict keys), creation order (weights_seed parity), the (out, aux)
return convention of the critic and BCE-with-logits losses against all-ones / all-zeros targets.
The hidden ReLU is fused into the first linear layer's kernel epilogue.
"""
import torch
import torch.nn.functional as F

from . import ops
from .DCResNet_models import DCResNetDiscriminator, DCResNetGenerator
from .models import Discriminator, Generator
from .nn import HipLinear

_HIDDEN, _PIXELS = 128, 28 * 28


def _with_labels(t, y, n_classes):
    """Concatenate one-hot labels to a [B, F] tensor (no-op for unconditional models)."""
    if y is None:
        return t
    return torch.cat((t, F.one_hot(y, num_classes=n_classes).to(t.dtype)), dim=1)


def _bce_against(logits, target_value):
    return F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, target_value))


class MNISTVanillaG(Generator):
    """z (+ one-hot label) -> 128 -> 784 -> sigmoid image."""

    def __init__(self, **kwargs):
        kwargs["out_ch"] = 1
        super().__init__(**kwargs)
        self.lin1 = HipLinear(self.z_dim + self.n_classes, _HIDDEN, act=ops.ACT_RELU)
        self.lin2 = HipLinear(_HIDDEN, _PIXELS * self.out_ch)

    def forward(self, z, y=None):
        hidden = self.lin1(_with_labels(z, y, self.n_classes))
        return torch.sigmoid(self.lin2(hidden)).reshape(z.size(0), self.out_ch, 28, 28)

    def loss(self, d_output, device):
        return _bce_against(d_output, 1.0)


class MNISTVanillaD(Discriminator):
    """image (+ one-hot label) -> 128 -> real/fake logit, plus an ACGAN class head on the hidden layer."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        conditional = self.n_classes > 1
        if conditional and self.aux_loss_type != "cross_entropy":
            raise Exception("Cross entropy loss is the only aux loss supported for vanilla architecture.")
        self.lin1 = HipLinear(_PIXELS + self.n_classes, _HIDDEN, act=ops.ACT_RELU)
        self.lin2 = HipLinear(_HIDDEN, 1)
        if conditional:
            self.linOutAux = HipLinear(_HIDDEN, self.n_classes, bias=True) if self.conditional_arch == "ACGAN" else None

    def forward(self, x, y=None, aux=True):
        hidden = self.lin1(_with_labels(x.reshape(x.size(0), -1), y, self.n_classes))
        has_head = self.conditional_arch == "ACGAN" and self.n_classes > 1
        return self.lin2(hidden), (self.linOutAux(hidden) if (aux and has_head) else None)

    def real_loss(self, output, device):
        return _bce_against(output, 1.0)

    def fake_loss(self, output, device):
        return _bce_against(output, 0.0)


def _dcrn(base, fixed, **defaults):
    """A DCResNet subclass whose constructor carries MNIST's default sizes (MNIST_models.py:54-60)."""
    class _Sized(base):
        def __init__(self, **kwargs):
            merged = dict(defaults, **kwargs)
            merged["channels"] = list(merged["channels"])
            super().__init__(**merged, **fixed)
    return _Sized


MNIST_DCRN_G = _dcrn(DCResNetGenerator, dict(out_ch=1), z_dim=128, channels=(128, 128, 64), first_filter_size=7, bn=True, n_classes=10)
MNIST_DCRN_D = _dcrn(DCResNetDiscriminator, {}, channels=(1, 64, 128), last_filter_size=7, n_classes=10)
MNIST_DCRN_G.__name__ = MNIST_DCRN_G.__qualname__ = "MNIST_DCRN_G"
MNIST_DCRN_D.__name__ = MNIST_DCRN_D.__qualname__ = "MNIST_DCRN_D"
