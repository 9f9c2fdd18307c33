"""csl_gan_amd's own model classes against vectors computed by the REFERENCE's classes (tests/golden/model_*.npz,
gp_*.npz; generator: tests/golden/make_golden.py, which executes the reference's class bodies in the build container).

  * not gpu : the CPU plumbing path of the package (BASELINE configs[0] runs on it) — init order, G forward, D forward,
              G.loss, train_G gradients;
  * gpu     : the same checks through the HIP kernels (depth-to-space + channel-folded UpsampleConv, GroupNorm /
              BatchNorm kernels, MFMA convs and their backward), and csl_gan_amd.gradient_penalty against the
              reference's gradient_penalty.py vectors.

Tolerance on the device: 1e-3 of each tensor's scale (the north-star bar).  Generator gradients cross ReLU / LeakyReLU
units; the few whose pre-activation sits within fp32 rounding of zero may take the other slope, so gradient TENSORS
are held to 5e-3 in relative L2 there (their norms to 1e-3) — the masked-parity tests in test_dstep_gpu.py pin the
same wiring per entry.
"""
import os

import numpy as np
import pytest
import torch

CASES = [
    # name, argv tail (dataset first), latent
    ("model_celeba64_gn_b2", ["CelebA", "-dpm", "gc"], 128),
    ("model_celeba64_bn_b3", ["CelebA", "-dpm", "is"], 128),
    ("model_celeba48_gn_b2", ["CelebA", "-dpm", "gc", "--im_size", "48"], 128),
    ("model_celeba64_cond_acgan_b4", ["CelebA", "-dpm", "gc", "--conditional"], 128),
    ("model_mnist_dcrn_gn_b4", ["MNIST", "-dpm", "gc", "--model", "DeepConvResNet"], 16),
    ("model_mnist_dcrn_cond_cgan_bn_b4", ["MNIST", "-dpm", "is", "--model", "DeepConvResNet", "--conditional",
                                          "--conditional_arch", "CGAN"], 16),
    ("model_mnist_vanilla_b8", ["MNIST", "-dpm", "gc", "--model", "Vanilla"], 100),
    ("model_mnist_vanilla_cond_b8", ["MNIST", "-dpm", "gc", "--model", "Vanilla", "--conditional", "--aux_loss_type",
                                     "cross_entropy"], 100),
]


def _build(tmp_path, argv, latent, device, init_G=True):
    from csl_gan_amd import init_util, options
    opt = options.parse(argv + ["-nms", "4", "-bs", "4", "-gd", device, "-dd", device, "-o", str(tmp_path), "--manual_seed", "1",
                                "--g_latent_dim", str(latent)])
    G, D = init_util.init_models(opt, init_G=init_G)
    return opt, G, D


def _rel(got, exp, scale=None):
    got = torch.as_tensor(got).detach().cpu().double().reshape(-1)
    exp = torch.as_tensor(exp).detach().cpu().double().reshape(-1)
    assert got.shape == exp.shape, (got.shape, exp.shape)
    s = (exp.abs().max().item() if scale is None else scale) + 1e-30
    return (got - exp).abs().max().item() / s


def _check_models(tmp_path, golden_dir, name, argv, latent, device, tol, grad_l2):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    opt, G, D = _build(tmp_path, argv, latent, device)
    ncls = int(z["meta"][4])
    assert bool(z["meta"][5]) == bool(opt.per_sample_grad)
    # init_util.py:63-69: same parameter names, order and values as the reference classes under weights_seed
    assert [n for n, _ in G.named_parameters()] == list(z["g_param_names"])
    np.testing.assert_allclose([p.detach().cpu().double().norm().item() for p in G.parameters()], z["g_weight_norms"], rtol=1e-5)
    np.testing.assert_allclose([p.detach().cpu().double().norm().item() for p in D.parameters()], z["d_weight_norms"], rtol=1e-5)
    zz = torch.from_numpy(z["z"]).to(device)
    real = torch.from_numpy(z["real"]).to(device)
    y = torch.from_numpy(z["labels"]).to(device) if ncls else None
    G.train(); D.train()
    fake = G(zz, y)
    assert _rel(fake, z["fake"], scale=1.0) <= tol, "G(z): %.3e" % _rel(fake, z["fake"], scale=1.0)     # tanh/sigmoid output: scale 1
    d_fake, d_fake_aux = D(fake, y)
    dscale = float(max(np.abs(z["d_real"]).max(), np.abs(z["d_fake"]).max()))
    assert _rel(d_fake, z["d_fake"], scale=dscale) <= tol
    g_loss = G.loss(d_fake, device)
    assert abs(g_loss.item() - float(z["g_loss"])) <= tol * max(dscale, abs(float(z["g_loss"])))
    total = g_loss
    if d_fake_aux is not None and D.conditional_arch == "ACGAN":
        assert _rel(d_fake_aux, z["d_fake_aux"]) <= tol
        total = total + D.aux_loss(d_fake_aux, y, device)
        assert abs(total.item() - float(z["g_total_loss"])) <= tol * max(1.0, abs(float(z["g_total_loss"])))
    grads = torch.autograd.grad(total, list(G.parameters()), allow_unused=True)
    gmax = float(z["g_grad_norms"].max())
    for (n, p), g, nrm, head in zip(G.named_parameters(), grads, z["g_grad_norms"], z["g_grad_heads"]):
        if g is None:
            assert nrm == 0.0, n
            continue
        got = g.detach().cpu().double().norm().item()
        # (a bias that feeds a BatchNorm has an exactly-zero gradient: both sides hold rounding noise there)
        assert abs(got - nrm) <= grad_l2 * nrm + 1e-6 * gmax, "%s: grad norm %.6e vs %.6e" % (n, got, nrm)
        # leading entries in the parameter's LOGICAL order (the fixture flattens NCHW-contiguous reference gradients)
        v = g.detach().cpu().contiguous().reshape(-1)[:8].double().numpy()
        err = np.abs(v - head[:v.size]).max()
        assert err <= 10 * grad_l2 * max(np.abs(head).max(), nrm / max(np.sqrt(g.numel()), 1.0)) + 1e-6 * gmax, (n, v, head)
    with torch.no_grad():
        d_real, d_real_aux = D(real, y)
    assert _rel(d_real, z["d_real"], scale=dscale) <= tol
    assert abs(D.real_loss(d_real, device).item() - float(z["d_real_loss"])) <= tol * max(dscale, 1e-6)
    assert abs(D.fake_loss(d_fake, device).item() - float(z["d_fake_loss"])) <= tol * max(dscale, 1e-6)
    if "fake_eval" in z.files:
        # BatchNorm running statistics after the one training-mode forward, then the eval-mode (sampling) forward
        bn1 = G.blocks[0].bn1
        assert _rel(bn1.running_mean, z["bn_running_mean0"]) <= tol and _rel(bn1.running_var, z["bn_running_var0"]) <= tol
        G.eval()
        with torch.no_grad():
            assert _rel(G(zz, y), z["fake_eval"], scale=1.0) <= tol
        G.train()


@pytest.mark.parametrize("name,argv,latent", CASES)
def test_package_models_on_cpu_match_reference_classes(tmp_path, golden_dir, name, argv, latent):
    _check_models(tmp_path, golden_dir, name, argv, latent, "cpu", tol=2e-6, grad_l2=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name,argv,latent", CASES)
def test_hip_models_match_reference_classes(tmp_path, golden_dir, name, argv, latent):
    _check_models(tmp_path, golden_dir, name, argv, latent, "cuda:0", tol=1e-3, grad_l2=5e-3)


GP_CASES = [
    ("gp_mnist_dcrn_b6", ["MNIST", "-dpm", "gc", "--model", "DeepConvResNet"], False),
    ("gp_mnist_dcrn_b6_onesided", ["MNIST", "-dpm", "gc", "--model", "DeepConvResNet"], True),
    ("gp_celeba64_b4", ["CelebA", "-dpm", "gc"], False),
    ("gp_celeba64_cond_aux_b3", ["CelebA", "-dpm", "gc", "--conditional"], False),
    # every parameter of D scaled so ||dD/dx|| is ~1.3 (two-sided) / straddles 1 (one-sided): the penalty value, the clamp
    # (gradient_penalty.py:54) and the parameter gradients are all sensitive to the input-gradient norm
    ("gp_mnist_dcrn_b6_scaled", ["MNIST", "-dpm", "gc", "--model", "DeepConvResNet"], False),
    ("gp_mnist_dcrn_b6_onesided_scaled", ["MNIST", "-dpm", "gc", "--model", "DeepConvResNet"], True),
    ("gp_celeba64_b4_scaled", ["CelebA", "-dpm", "gc"], False),
    ("gp_celeba64_b4_onesided_scaled", ["CelebA", "-dpm", "gc"], True),
    ("gp_celeba64_cond_aux_b3_scaled", ["CelebA", "-dpm", "gc", "--conditional"], False),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,argv,one_sided", GP_CASES)
def test_hip_gradient_penalty_matches_reference_vectors(tmp_path, golden_dir, name, argv, one_sided):
    """csl_gan_amd.gradient_penalty (double backward on the HIP conv Functions) directly against the vectors the
    reference's gradient_penalty.py produced on the reference's own D (tests/golden/gp_*.npz)."""
    from csl_gan_amd.gradient_penalty import calc_penalty
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    opt, _, D = _build(tmp_path, argv, 128, "cuda:0", init_G=False)      # the fixture's D was built alone (make_golden.gp_case)
    k = float(z["weight_scale"])
    if k != 1.0:
        with torch.no_grad():
            for p in D.parameters():
                p.mul_(k)
    np.testing.assert_allclose([p.detach().cpu().double().norm().item() for p in D.parameters()], z["weight_norms"], rtol=1e-5)
    real, fake = torch.from_numpy(z["real"]).cuda(), torch.from_numpy(z["fake"]).cuda()
    labels = torch.from_numpy(z["labels"]).cuda() if "labels" in z.files else None
    alpha = torch.from_numpy(z["alpha"])
    aux = bool(z["meta"][5])
    ptype = ["WGAN-GP1" if one_sided else "WGAN-GP"]
    with torch.no_grad():
        d_out, _ = D(real, labels)
    assert _rel(d_out, z["d_out_real"]) <= 1e-3
    # per-sample ||dD(x_hat)/dx_hat|| (gradient_penalty.py:48-52) through the HIP data-gradient kernels, 1e-3
    from csl_gan_amd import functional as HF
    a4 = alpha.view(-1, 1, 1, 1).cuda()
    xh = (a4 * real + (1 - a4) * fake).detach().requires_grad_(True)
    with HF.input_grads_only():
        o_, _ = D(xh, labels)
    gx, = torch.autograd.grad(o_, xh, torch.ones_like(o_))
    n_in = gx.reshape(gx.size(0), -1).double().norm(2, dim=1).cpu().numpy()
    np.testing.assert_allclose(n_in, z["input_grad_norms"], rtol=1e-3)
    # the penalty is weight * mean_b phi(n_b), phi = (n-1)^2 (clamped when one-sided; plus the aux-logit terms): a 1e-3 relative
    # error of a norm moves phi by 2|n-1| * 1e-3 n, so the penalty is held to 1e-3 of itself PLUS that propagated bound
    n_ref = z["input_grad_norms"]
    prop = 10.0 * float(np.mean(2 * np.abs(n_ref - 1) * 1e-3 * n_ref + (1e-3 * n_ref) ** 2)) * (1 + (int(z["d_aux_real"].shape[1]) if aux else 0))
    pen = calc_penalty(D, ptype, real, labels, fake, labels, device="cuda:0", aux_penalty=aux, alpha=alpha)
    exp = float(z["penalty"])
    assert abs(pen.item() - exp) <= 1e-3 * abs(exp) + prop, (pen.item(), exp, prop)
    per = calc_penalty(D, ptype, real, labels, fake, labels, device="cuda:0", per_sample=True, aux_penalty=aux, alpha=alpha)
    B = real.size(0)
    assert np.abs(per.detach().cpu().double().numpy() - z["penalty_per_sample"]).max() <= 1e-3 * float(np.abs(z["penalty_per_sample"]).max()) + prop * 2, name
    if name.endswith("_scaled"):
        assert exp > 1e-4 and float(z["grad_norms"].max()) > 1e-2            # the fixture bites
    grads = torch.autograd.grad(pen, list(D.parameters()), allow_unused=True)
    gmax = float(z["grad_norms"].max())
    for (n, p), g, nrm, head in zip(D.named_parameters(), grads, z["grad_norms"], z["grad_heads"]):
        got = 0.0 if g is None else g.detach().cpu().double().norm().item()
        # bias gradients of the penalty are exactly zero in the reference
        # gradient of the penalty = 20/B sum_b (n_b - 1) dn_b/dtheta: when the norms sit near 1 (scaled fixtures) the factor
        # (n_b - 1) amplifies a 1e-3 error of n_b by n/(n-1)
        amp = float(np.max(n_ref / np.maximum(np.abs(n_ref - 1), 0.05))) if name.endswith("_scaled") else 1.0
        assert abs(got - nrm) <= (5e-3 + (1e-3 * amp if amp > 1 else 0)) * max(nrm, 1e-5 * gmax), "%s: %.6e vs %.6e" % (n, got, nrm)
        if g is not None and nrm > 0:
            v = g.detach().cpu().contiguous().reshape(-1)[:8].double().numpy()
            assert np.abs(v - head).max() <= 5e-2 * max(np.abs(head).max(), nrm / np.sqrt(g.numel())), (n, v, head)
