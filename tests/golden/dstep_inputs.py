"""Seeded inputs of the D-step fixtures (tests/golden/dstep_*.npz).

The fixtures hold seeds + checksums instead of the image tensors: make_golden.py (build container, reference classes) and the
tests (CPU here, HIP on the GPU box) regenerate the same batches from the CPU generator and the tests verify the checksums the
fixture carries.  Test infrastructure only.
"""
import numpy as np
import torch


def dstep_inputs(seed, B, ch, im, latent, ncls=0, unit_range=False):
    """One D-step's explicit inputs.  unit_range: images in [0, 1] (MNIST vanilla: the generator ends in a sigmoid)."""
    g = torch.Generator().manual_seed(int(seed))

    def image(std):
        t = torch.randn(B, ch, im, im, generator=g) * std
        return (t + 0.5).clamp(0, 1) if unit_range else t.clamp(-1, 1)

    d = dict(img=image(0.5), ms_adapt=image(0.3), ms_pen=image(0.3), z=torch.randn(B, latent, generator=g),
             z_adapt=torch.randn(B, latent, generator=g), alpha=torch.rand(B, generator=g))
    if ncls:
        for k in ("labels", "y", "ms_adapt_labels", "ms_pen_labels"):
            v = torch.randint(0, ncls, (B,), generator=g)
            v[:min(ncls, B)] = torch.arange(min(ncls, B))      # every class present (aux_loss divides by class counts)
            d[k] = v
    else:
        d.update(labels=None, y=None, ms_adapt_labels=None, ms_pen_labels=None)
    return d


def checksums(d):
    """[sum, sum of squares] in float64 of every float input, in a fixed key order."""
    keys = ("img", "ms_adapt", "ms_pen", "z", "z_adapt", "alpha")
    return np.array([[d[k].double().sum().item(), (d[k].double() ** 2).sum().item()] for k in keys])


def sample_idx(numel, n=2048):
    """Indices (memory order of the logical-shape tensor) of the strided entry sample stored per gradient tensor."""
    step = max(1, numel // n)
    return np.arange(0, numel, step)[:n]


# name -> (dataset, model, im_size, latent, conditional kwargs for oracle.nets.build_models, extra CLI flags of csl_gan_amd)
DSTEP_CASES = {
    "dstep_celeba64_b8": ("CelebA", "DeepConvResNet", 64, 128, {}, []),
    "dstep_celeba64_cond_acgan_b8": ("CelebA", "DeepConvResNet", 64, 128, dict(conditional=True, n_classes=2),
                                     ["--conditional", "-cpl"] + ["1"] * 11),     # the default per-layer table has 9 entries; ACGAN's D has 11 tensors
    "dstep_mnist_vanilla_cond_b16": ("MNIST", "Vanilla", 28, 100, dict(conditional=True, n_classes=10, aux_loss_type="cross_entropy"),
                                     ["--model", "Vanilla", "--conditional"]),
    "dstep_mnist_vanilla_b16": ("MNIST", "Vanilla", 28, 100, {}, ["--model", "Vanilla"]),
    "dstep_mnist_dcrn_b6": ("MNIST", "DeepConvResNet", 28, 16, {}, ["--model", "DeepConvResNet", "--penalty", "WGAN-GP"]),
    "dstep_celeba128_b4": ("CelebA", "DeepConvResNet", 128, 128, {}, ["--im_size", "128"]),
}


def load_case(golden_dir, name):
    """(fixture, regenerated inputs) with the input checksums verified."""
    import os
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    B, im, seed, latent, ncls, ch = (int(v) for v in z["meta"])
    inp = dstep_inputs(seed, B, ch, im, latent, ncls, unit_range=(DSTEP_CASES[name][1] == "Vanilla"))
    np.testing.assert_allclose(checksums(inp), z["input_checksums"], rtol=1e-9, err_msg="regenerated inputs differ from the fixture's")
    return z, inp


def sampled(t):
    """The fixture's strided entry sample of a gradient tensor given in LOGICAL shape (any strides)."""
    f = torch.as_tensor(t).detach().cpu().contiguous().reshape(-1)
    return f[torch.from_numpy(sample_idx(f.numel()))].double().numpy()
