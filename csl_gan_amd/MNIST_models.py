"""MNIST model family (reference MNIST_models.py:9-60): vanilla MLP GAN and the DCResNet sizes."""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .DCResNet_models import DCResNetDiscriminator, DCResNetGenerator
from .models import Discriminator, Generator
from .nn import HipLinear


class MNISTVanillaG(Generator):
    def __init__(self, **kwargs):
        super().__init__(**kwargs, out_ch=1)
        self.criterion = nn.BCEWithLogitsLoss()
        self.lin1 = HipLinear(self.z_dim + self.n_classes, 128, act=ops.ACT_RELU)
        self.lin2 = HipLinear(128, 784 * self.out_ch)

    def forward(self, z, y=None):
        x = z if y is None else torch.cat([z, F.one_hot(y, num_classes=self.n_classes).to(z.dtype)], dim=1)
        return torch.sigmoid(self.lin2(self.lin1(x))).reshape(z.size(0), self.out_ch, 28, 28)

    def loss(self, d_output, device):
        return self.criterion(d_output, torch.ones(d_output.shape, device=device))


class MNISTVanillaD(Discriminator):
    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        self.criterion = nn.BCEWithLogitsLoss()
        if self.n_classes > 1 and self.aux_loss_type != "cross_entropy":
            raise Exception("Cross entropy loss is the only aux loss supported for vanilla architecture.")
        self.lin1 = HipLinear(784 + self.n_classes, 128, act=ops.ACT_RELU)
        self.lin2 = HipLinear(128, 1)
        if self.n_classes > 1:
            self.linOutAux = HipLinear(128, self.n_classes, bias=True) if self.conditional_arch == "ACGAN" else None

    def forward(self, x, y=None, aux=True):
        o = x.reshape(x.size(0), -1)
        if y is not None:
            o = torch.cat([o, F.one_hot(y, num_classes=self.n_classes).to(o.dtype)], dim=1)
        h = self.lin1(o)
        want_aux = aux and self.conditional_arch == "ACGAN" and self.n_classes > 1
        return self.lin2(h), (self.linOutAux(h) if want_aux else None)

    def real_loss(self, output, device):
        return self.criterion(output, torch.ones(output.shape, device=device))

    def fake_loss(self, output, device):
        return self.criterion(output, torch.zeros(output.shape, device=device))


class MNIST_DCRN_G(DCResNetGenerator):
    def __init__(self, z_dim=128, channels=(128, 128, 64), first_filter_size=7, bn=True, n_classes=10, **kwargs):
        super().__init__(z_dim=z_dim, channels=list(channels), first_filter_size=first_filter_size, bn=bn, out_ch=1,
                         n_classes=n_classes, **kwargs)


class MNIST_DCRN_D(DCResNetDiscriminator):
    def __init__(self, channels=(1, 64, 128), last_filter_size=7, n_classes=10, **kwargs):
        super().__init__(channels=list(channels), last_filter_size=last_filter_size, n_classes=n_classes, **kwargs)
