"""Model factory (reference init_util.py:44-71).  init_data (init_util.py:13-42) is dataset I/O and
out of scope: csl_gan_amd.data provides the synthetic loaders the CLI uses."""
import torch

from . import CelebA_models as CM
from . import MNIST_models as MM
from .nn import to_device_layout


def model_classes(opt):
    if opt.dataset == "MNIST":
        if opt.model == "DeepConvResNet":
            return MM.MNIST_DCRN_G, MM.MNIST_DCRN_D
        if opt.model == "Vanilla":
            return MM.MNISTVanillaG, MM.MNISTVanillaD
    elif opt.dataset == "CelebA":
        if opt.model == "Vanilla":
            raise Exception("No vanilla architecture for CelebA.")
        if opt.model == "DeepConvResNet":
            size = getattr(opt, "im_size", 64)
            return {48: (CM.CelebA_DCRN_G48, CM.CelebA_DCRN_D48), 128: (CM.CelebA_DCRN_G128, CM.CelebA_DCRN_D128)}.get(
                size, (CM.CelebA_DCRN_G64, CM.CelebA_DCRN_D64))
    raise Exception("Unknown dataset/model: %s/%s" % (opt.dataset, opt.model))


def init_models(opt, init_G=True, init_D=True):
    """Seed with weights_seed, build G then D from one RNG stream (this order fixes D's initial
    weights), move to the devices, reseed with manual_seed."""
    n_classes = opt.n_classes if opt.conditional else 0
    bn = not opt.per_sample_grad            # GroupNorm(32) in G when per-sample gradients are on
    GObj, DObj = model_classes(opt)
    torch.manual_seed(opt.weights_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(opt.weights_seed)
    G = D = None
    if init_G:
        G = GObj(z_dim=opt.g_latent_dim, bn=bn, n_classes=n_classes, emb_mode=opt.g_label_emb_mode).to(opt.g_device)
        to_device_layout(G)
    if init_D:
        D = DObj(n_classes=n_classes, emb_mode=opt.d_label_emb_mode, conditional_arch=opt.conditional_arch,
                 aux_loss_type=opt.aux_loss_type, aux_loss_scalar=opt.aux_loss_scalar).to(opt.d_device)
        to_device_layout(D)
    torch.manual_seed(opt.manual_seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(opt.manual_seed)
    return G, D
