"""Generator / Discriminator bases (reference models.py:7-67), same constructor kwargs and methods."""
import torch
import torch.nn.functional as F
from torch import nn


class Generator(nn.Module):
    def __init__(self, z_dim=100, out_ch=3, n_classes=1, emb_mode="concat", bn=True):
        super().__init__()
        self.z_dim, self.out_ch, self.n_classes, self.emb_mode, self.bn = z_dim, out_ch, n_classes, emb_mode, bn
        use_emb = n_classes > 1 and emb_mode == "embed"
        self.emb = nn.Embedding(n_classes, z_dim) if use_emb else None

    def forward(self, z, y=None):
        raise NotImplementedError("Abstract method")

    def loss(self, d_output, device):
        raise NotImplementedError("Abstract method")

    def _condition(self, z, y):
        """Label conditioning of the latent (DCResNet_models.py:88-93)."""
        if y is None:
            return z
        if self.emb_mode == "embed":
            return z * self.emb(y)
        if self.emb_mode == "concat":
            return torch.cat((z, F.one_hot(y, self.n_classes).to(z.dtype)), dim=1)
        return z


class Discriminator(nn.Module):
    def __init__(self, n_classes=0, emb_mode="concat", conditional_arch="CGAN", aux_loss_type="wasserstein",
                 aux_loss_scalar=1):
        super().__init__()
        self.n_classes, self.emb_mode = n_classes, emb_mode
        self.conditional_arch, self.aux_loss_type, self.aux_loss_scalar = conditional_arch, aux_loss_type, aux_loss_scalar
        if n_classes > 1:
            if emb_mode == "embed":
                raise Exception("Embed for D not implemented")
            if conditional_arch == "ACGAN":
                self.emb_mode = None          # the ACGAN critic never sees the label (models.py:36-37)
                if aux_loss_type == "cross_entropy":
                    self.aux_criterion = nn.CrossEntropyLoss()

    def forward(self, x, y=None, aux=True):
        raise NotImplementedError("Abstract method")

    def real_loss(self, output, device):
        raise NotImplementedError("Abstract method")

    def fake_loss(self, output, device):
        raise NotImplementedError("Abstract method")

    def aux_loss(self, output, labels, device, fake=False):
        """models.py:51-67.  Small [B, n_classes] tensors: plain torch ops."""
        if self.conditional_arch == "ACGAN":
            if self.aux_loss_type != "wasserstein":
                return self.aux_loss_scalar * self.aux_criterion(output, labels)
            hot = F.one_hot(labels, self.n_classes)
            sign = 1 - 2 * hot                      # -1 on the true class, +1 elsewhere
            count_of_own_class = hot.sum(dim=0)[labels].unsqueeze(1)
            return self.aux_loss_scalar * (sign * torch.sigmoid(output) / count_of_own_class).sum()
        if self.conditional_arch == "WCGAN":
            return torch.tensor([0], device=device)
        return None
