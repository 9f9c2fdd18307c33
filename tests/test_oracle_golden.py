"""Pin the CPU oracle against vectors produced by the reference's own importable modules
(tests/golden/make_golden.py: gradient_penalty.py, models.py, logger.py)."""
import io
import os
import contextlib

import numpy as np
import pytest
import torch

from oracle import penalty as OP
from oracle.nets import build_models


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


CASES = [
    ("gp_mnist_dcrn_b6", "MNIST", 28, False, False),
    ("gp_mnist_dcrn_b6_onesided", "MNIST", 28, True, False),
    ("gp_celeba64_b4", "CelebA", 64, False, False),
    ("gp_celeba64_cond_aux_b3", "CelebA", 64, False, True),
]


@pytest.mark.parametrize("name,dataset,im,one_sided,cond", CASES)
def test_penalty_matches_reference(golden_dir, name, dataset, im, one_sided, cond):
    z = _load(golden_dir, name)
    _, D = build_models(dataset=dataset, model="DeepConvResNet", im_size=im, weights_seed=42, manual_seed=1,
                        init_G=False, conditional=cond, n_classes=10 if dataset == "MNIST" else 2)
    # weight-init parity with the build that produced the fixture
    np.testing.assert_allclose([p.norm().item() for p in D.parameters()], z["weight_norms"], rtol=1e-6)
    real, fake = torch.from_numpy(z["real"]), torch.from_numpy(z["fake"])
    labels = torch.from_numpy(z["labels"]) if cond else None
    alpha = torch.from_numpy(z["alpha"])
    with torch.no_grad():
        out, aux = D(real, labels)
    np.testing.assert_allclose(out.numpy(), z["d_out_real"], rtol=1e-5, atol=1e-6)
    ptype = ["WGAN-GP1" if one_sided else "WGAN-GP"]
    pen = OP.calc_penalty(D, ptype, real, labels, fake, alpha, aux_penalty=bool(z["meta"][5]))
    assert pen.item() == pytest.approx(float(z["penalty"]), rel=1e-5, abs=1e-7)
    grads = torch.autograd.grad(pen, list(D.parameters()), allow_unused=True)
    norms = np.array([0.0 if g is None else g.norm().item() for g in grads])
    np.testing.assert_allclose(norms, z["grad_norms"], rtol=1e-4, atol=1e-7)
    heads = np.stack([np.zeros(8, np.float32) if g is None else g.reshape(-1)[:8].numpy() for g in grads])
    np.testing.assert_allclose(heads, z["grad_heads"], rtol=1e-3, atol=1e-6)
    per = OP.calc_penalty(D, ptype, real, labels, fake, alpha, per_sample=True, aux_penalty=bool(z["meta"][5]))
    np.testing.assert_allclose(per.detach().numpy(), z["penalty_per_sample"], rtol=1e-4, atol=1e-6)


def test_aux_loss_matches_reference(golden_dir):
    from oracle.nets import _DiscBase
    z = np.load(os.path.join(golden_dir, "aux_loss.npz"))
    for ncls in (2, 10):
        logits, labels = torch.from_numpy(z[f"logits_{ncls}"]), torch.from_numpy(z[f"labels_{ncls}"])
        for typ in ("wasserstein", "cross_entropy"):
            d = _DiscBase(n_classes=ncls, conditional_arch="ACGAN", aux_loss_type=typ, aux_loss_scalar=0.5)
            assert d.aux_loss(logits, labels).item() == pytest.approx(float(z[f"{typ}_{ncls}"]), rel=1e-6)
