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
    np.testing.assert_allclose([p.detach().double().norm().item() for p in D.parameters()], z["weight_norms"], rtol=1e-6)
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
    norms = np.array([0.0 if g is None else g.double().norm().item() for g in grads])
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


# ---- model stacks: vectors computed by the reference's OWN classes (make_golden.py: reference_model_classes) -------------
MODEL_CASES = [
    # name, dataset, model, im, kwargs for oracle.nets.build_models
    ("model_celeba64_gn_b2", "CelebA", "DeepConvResNet", 64, {}),
    ("model_celeba64_bn_b3", "CelebA", "DeepConvResNet", 64, dict(per_sample_grad=False)),
    ("model_celeba48_gn_b2", "CelebA", "DeepConvResNet", 48, {}),
    ("model_celeba64_cond_acgan_b4", "CelebA", "DeepConvResNet", 64, dict(conditional=True, n_classes=2)),
    ("model_mnist_dcrn_gn_b4", "MNIST", "DeepConvResNet", 28, {}),
    ("model_mnist_dcrn_cond_cgan_bn_b4", "MNIST", "DeepConvResNet", 28,
     dict(conditional=True, n_classes=10, conditional_arch="CGAN", per_sample_grad=False)),
    ("model_mnist_vanilla_b8", "MNIST", "Vanilla", 28, {}),
    ("model_mnist_vanilla_cond_b8", "MNIST", "Vanilla", 28, dict(conditional=True, n_classes=10, aux_loss_type="cross_entropy")),
]


def test_upsample_conv_is_the_reference_op(golden_dir):
    """UpsampleConv (DCResNet_models.py:8-17) is a channel-interleaving depth-to-space, not a nearest up-sample:
    the oracle layer and the closed form out[c,2h+i,2w+j] = x[(4c+2i+j) mod C] both reproduce the reference's output."""
    from oracle.nets import _UpConv
    z = np.load(os.path.join(golden_dir, "upsample_conv.npz"))
    for tag in ("c8_k6_f5", "c16_k4_f1", "c4_k3_f3"):
        x, w, y = torch.from_numpy(z["x_" + tag]), torch.from_numpy(z["w_" + tag]), z["y_" + tag]
        K, C, k, _ = w.shape
        m = _UpConv(C, K, k, bias=("b_" + tag) in z)
        with torch.no_grad():
            m.conv.weight.copy_(w)
            if m.conv.bias is not None:
                m.conv.bias.copy_(torch.from_numpy(z["b_" + tag]))
            got = m(x)
            N, _, H, W = x.shape
            up = torch.empty(N, C, 2 * H, 2 * W)
            for c in range(C):
                for i in range(2):
                    for j in range(2):
                        up[:, c, i::2, j::2] = x[:, (4 * c + 2 * i + j) % C]
            closed = torch.nn.functional.conv2d(up, w, m.conv.bias, padding=k // 2)
        np.testing.assert_allclose(got.numpy(), y, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(closed.numpy(), y, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name,dataset,model,im,kw", MODEL_CASES)
def test_model_stacks_match_reference_classes(golden_dir, name, dataset, model, im, kw):
    z = _load(golden_dir, name)
    B, latent, ncls = int(z["meta"][0]), int(z["meta"][3]), int(z["meta"][4])
    G, D = build_models(dataset=dataset, model=model, im_size=im, weights_seed=42, manual_seed=1, g_latent_dim=latent, **kw)
    # init order / seeding (init_util.py:63-69): every parameter tensor of G and D
    assert [n for n, _ in G.named_parameters()] == list(z["g_param_names"])
    np.testing.assert_allclose([p.detach().double().norm().item() for p in G.parameters()], z["g_weight_norms"], rtol=1e-6)
    np.testing.assert_allclose([p.detach().double().norm().item() for p in D.parameters()], z["d_weight_norms"], rtol=1e-6)
    zz, real = torch.from_numpy(z["z"]), torch.from_numpy(z["real"])
    y = torch.from_numpy(z["labels"]) if ncls else None
    fake = G(zz, y)
    np.testing.assert_allclose(fake.detach().numpy(), z["fake"], rtol=0, atol=1e-6)
    d_fake, d_fake_aux = D(fake, y)
    scale = float(np.abs(z["d_real"]).max() + np.abs(z["d_fake"]).max())
    np.testing.assert_allclose(d_fake.detach().numpy(), z["d_fake"], rtol=0, atol=1e-6 * max(scale, 1.0))
    g_loss = G.loss(d_fake)
    assert g_loss.item() == pytest.approx(float(z["g_loss"]), abs=1e-6)
    total = g_loss
    if d_fake_aux is not None and D.conditional_arch == "ACGAN":
        np.testing.assert_allclose(d_fake_aux.detach().numpy(), z["d_fake_aux"], rtol=0, atol=1e-6)
        total = total + D.aux_loss(d_fake_aux, y)
    assert total.item() == pytest.approx(float(z["g_total_loss"]), abs=1e-6)
    grads = torch.autograd.grad(total, list(G.parameters()), allow_unused=True)
    norms = np.array([0.0 if g is None else g.double().norm().item() for g in grads])
    np.testing.assert_allclose(norms, z["g_grad_norms"], rtol=1e-4, atol=1e-9)
    for g, head, nrm in zip(grads, z["g_grad_heads"], z["g_grad_norms"]):
        if g is not None:
            v = g.reshape(-1)[:8].numpy()
            np.testing.assert_allclose(v, head[:v.size], rtol=1e-3, atol=1e-5 * max(nrm, 1e-12))
    with torch.no_grad():
        d_real, _ = D(real, y)
    np.testing.assert_allclose(d_real.numpy(), z["d_real"], rtol=0, atol=1e-6 * max(scale, 1.0))
    assert D.real_loss(d_real).item() == pytest.approx(float(z["d_real_loss"]), abs=1e-6)
    assert D.fake_loss(d_fake).item() == pytest.approx(float(z["d_fake_loss"]), abs=1e-6)
    if "fake_eval" in z.files:
        np.testing.assert_allclose(G.blocks[0].bn1.running_mean.numpy(), z["bn_running_mean0"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(G.blocks[0].bn1.running_var.numpy(), z["bn_running_var0"], rtol=1e-5, atol=1e-7)
        G.eval()
        with torch.no_grad():
            np.testing.assert_allclose(G(zz, y).numpy(), z["fake_eval"], rtol=0, atol=1e-6)
