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
    # weights scaled so ||dD/dx|| is ~1.3 (two-sided) or straddles 1 (one-sided): penalty, clamp and gradients all bite
    ("gp_mnist_dcrn_b6_scaled", "MNIST", 28, False, False),
    ("gp_mnist_dcrn_b6_onesided_scaled", "MNIST", 28, True, False),
    ("gp_celeba64_b4_scaled", "CelebA", 64, False, False),
    ("gp_celeba64_b4_onesided_scaled", "CelebA", 64, True, False),
    ("gp_celeba64_cond_aux_b3_scaled", "CelebA", 64, False, True),
]


@pytest.mark.parametrize("name,dataset,im,one_sided,cond", CASES)
def test_penalty_matches_reference(golden_dir, name, dataset, im, one_sided, cond):
    z = _load(golden_dir, name)
    _, D = build_models(dataset=dataset, model="DeepConvResNet", im_size=im, weights_seed=42, manual_seed=1,
                        init_G=False, conditional=cond, n_classes=10 if dataset == "MNIST" else 2)
    k = float(z["weight_scale"])
    if k != 1.0:
        with torch.no_grad():
            for p in D.parameters():
                p.mul_(k)
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
    a4 = alpha.view(-1, 1, 1, 1)
    xh = (a4 * real + (1 - a4) * fake).requires_grad_(True)
    gx, = torch.autograd.grad(D(xh, labels)[0].sum(), xh)
    np.testing.assert_allclose(gx.reshape(gx.size(0), -1).double().norm(2, dim=1).numpy(), z["input_grad_norms"], rtol=1e-5)
    if name.endswith("_scaled"):        # the scaled fixtures bite: non-zero penalty and gradients, norms on both sides of 1 when one-sided
        assert float(z["penalty"]) > 1e-4 and float(z["grad_norms"].max()) > 1e-2
        if one_sided:
            assert (z["input_grad_norms"] < 1).any() and (z["input_grad_norms"] > 1).any()
            assert (z["penalty_per_sample"] == 0).any() and (z["penalty_per_sample"] > 0).any()


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


# ---- D-step observables: per-sample norms, clip factors, clipped sums, adaptive statistics, penalty gradients -------------
# tests/golden/dstep_*.npz are computed through the reference's OWN G / D classes and loss methods (make_golden.dstep_case:
# one autograd call per sample, clip rule written out in float64) — the oracle's hook engine and D-step must reproduce them.
from tests.golden.dstep_inputs import DSTEP_CASES, load_case, sampled      # noqa: E402


def _rel_close(got, exp, what, tol=1e-6, scale=None):
    got, exp = np.asarray(got, dtype=np.float64), np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    s = (np.abs(exp).max() if scale is None else scale) + 1e-30
    err = np.abs(got - exp).max()
    assert err <= tol * s, "%s: max abs err %.3e at scale %.3e (rel %.3e)" % (what, err, s, err / s)


def _check_grads(z, key, tensors, what, tol):
    _rel_close([t.double().norm().item() for t in tensors], z[key + "_norms"], what + " norms", tol=tol, scale=float(z[key + "_norms"].max()))
    for i, t in enumerate(tensors):
        amax = float(z[key + "_absmax"][i])             # the tensor's error scale (for sums over samples: of their terms)
        if amax <= 1e-7 * float(z[key + "_absmax"].max()):
            assert t.abs().max().item() <= 1e-5 * float(z[key + "_absmax"].max()), (what, i)
        else:
            _rel_close(sampled(t), z["%s_s%d" % (key, i)], "%s[%d] entries" % (what, i), tol=tol, scale=amax)


def _oracle_for(name, z, **cfg_kw):
    from oracle.dstep import OracleDStep, StepConfig
    dataset, model, im, latent, kw, _ = DSTEP_CASES[name]
    G, D = build_models(dataset=dataset, model=model, im_size=im, weights_seed=42, manual_seed=1, g_latent_dim=latent, **kw)
    np.testing.assert_allclose([p.detach().double().norm().item() for p in D.parameters()], z["d_weight_norms"], rtol=1e-6)
    np.testing.assert_allclose([p.detach().double().norm().item() for p in G.parameters()], z["g_weight_norms"], rtol=1e-6)
    assert [n for n, _ in D.named_parameters()] == list(z["d_param_names"])
    cond = bool(kw.get("conditional"))
    has_pen = "penalty" in z.files
    cfg = StepConfig(dp_mode="gc", sigma=0.0, penalty=("WGAN-GP",) if has_pen else (), aux_penalty=True, use_aux_loss=cond,
                     d_fake_aux_loss=True, adaptive_scalar=float(z["adaptive_scalar"]), **cfg_kw)
    return OracleDStep(G, D, cfg), D


def _step(oracle, inp, has_pen):
    return oracle.step(inp["img"], inp["labels"], inp["z"], inp["y"], ms_adapt=inp["ms_adapt"], ms_adapt_labels=inp["ms_adapt_labels"],
                       z_adapt=inp["z_adapt"], pen_real=inp["ms_pen"] if has_pen else None, pen_labels=inp["ms_pen_labels"],
                       alpha=inp["alpha"], apply_update=False)


@pytest.mark.parametrize("name", sorted(DSTEP_CASES))
def test_dstep_observables_match_reference_classes(golden_dir, name):
    z, inp = load_case(golden_dir, name)
    has_pen = "penalty" in z.files
    L = len(z["d_weight_norms"])
    tol = 2e-6
    # (1) adaptive per-layer clipping (BASELINE configs[2] mode) + penalty on the public batch
    oracle, D = _oracle_for(name, z, grad_clip_mode="adaptive-pl", grad_clip_split=True)
    obs = _step(oracle, inp, has_pen)
    fake = obs["fake_img"]
    _rel_close(fake.reshape(-1)[::max(1, fake.numel() // 4096)][:4096].numpy(), z["fake_sample"], "G(z)", tol=tol)
    _rel_close(obs["d_real"].numpy(), z["d_real"], "d_real", tol=tol, scale=float(np.abs(z["d_real"]).max() + np.abs(z["d_fake"]).max()))
    _rel_close(obs["d_fake"].numpy(), z["d_fake"], "d_fake", tol=tol, scale=float(np.abs(z["d_real"]).max() + np.abs(z["d_fake"]).max()))
    assert obs["d_real_loss"] == pytest.approx(float(z["d_real_loss"]), rel=1e-5, abs=1e-7)
    assert obs["d_fake_loss"] == pytest.approx(float(z["d_fake_loss"]), rel=1e-5, abs=1e-7)
    _rel_close(obs["adaptive_stats"], z["adaptive_mean"], "adaptive statistics (train.py:204-245)", tol=tol)
    _rel_close(obs["clip_params"], z["c_adaptive_pl"], "adaptive per-layer C", tol=tol)
    _rel_close(obs["norms"].numpy(), z["layer_norms"], "per-layer per-sample norms [L, pass, B]", tol=tol)
    _rel_close(obs["clip_factors"].numpy()[:, 1], z["factors_pl"], "per-layer clip factors (private pass)", tol=tol)
    _check_grads(z, "sum_pl_split", obs["summed_clipped"], "clipped sum (per-layer C, split passes)", tol)
    if has_pen:
        assert obs["penalty"] == pytest.approx(float(z["penalty"]), rel=1e-5)
        pg = [torch.zeros_like(p) if g is None else g for g, p in zip(obs["penalty_grads"], D.parameters())]
        _check_grads(z, "pen_grad", pg, "penalty parameter gradients", 1e-5)
        _check_grads(z, "summed_grad_pl", obs["summed_grad"], "summed_grad = clipped sum + B * penalty gradient (train.py:431)", 1e-5)
    # (2) one flat constant C
    oracle, D = _oracle_for(name, z, grad_clip_mode="standard", grad_clip_split=True, clipping_param=float(z["c_flat"]))
    obs = _step(oracle, inp, has_pen)
    _rel_close(obs["norms"].numpy()[0], z["flat_norms"], "flat per-sample norms [pass, B]", tol=tol)
    _rel_close(obs["clip_factors"].numpy()[0, 1], z["factors_flat"], "flat clip factors", tol=tol)
    _check_grads(z, "sum_flat_split", obs["summed_clipped"], "clipped sum (flat C, split passes)", tol)
    if has_pen:
        _check_grads(z, "summed_grad_flat", obs["summed_grad"], "summed_grad (flat C)", 1e-5)
    # (3) adaptive flat C
    oracle, D = _oracle_for(name, z, grad_clip_mode="adaptive", grad_clip_split=True)
    obs = _step(oracle, inp, has_pen)
    _rel_close(obs["clip_params"], float(z["c_adaptive_flat"]), "adaptive flat C", tol=tol)
    _rel_close(obs["clip_factors"].numpy()[0, 1], z["factors_adaptive_flat"], "adaptive flat clip factors", tol=tol)
    # (4) accumulated passes (-gcs False): per-sample sum over both passes, one flat C
    oracle, D = _oracle_for(name, z, grad_clip_mode="standard", grad_clip_split=False, clipping_param=float(z["c_accum"]))
    obs = _step(oracle, inp, has_pen)
    _check_grads(z, "sum_flat_accum", obs["summed_clipped"], "clipped sum (accumulated passes)", tol)


def test_survey_probe_vector(golden_dir):
    """SURVEY.md §8c's probe: per-layer means of the per-sample gradient norms of the reference's D64 (built alone under seed 42),
    x = randn(16,3,64,64).clamp(-1,1) under manual_seed 1, real loss."""
    from oracle import dp_engine as E
    z = np.load(os.path.join(golden_dir, "dstep_survey_probe.npz"))
    _, D = build_models(dataset="CelebA", model="DeepConvResNet", im_size=64, weights_seed=42, manual_seed=1, init_G=False)
    torch.manual_seed(1)
    x = torch.randn(16, 3, 64, 64).clamp(-1, 1)
    np.testing.assert_allclose([x.double().sum().item(), (x.double() ** 2).sum().item()], z["x_checksum"], rtol=1e-9)
    gs = E.per_sample_grads_microbatch(D, lambda D_, xb, yb: D_.real_loss(D_(xb)[0]), x)
    n = torch.stack([g.reshape(16, -1).double().norm(2, dim=1) for g in gs]).numpy()
    _rel_close(n, z["layer_norms"], "probe per-sample norms", tol=2e-6)
    assert np.abs(n.mean(axis=1) - z["survey_quote"]).max() < 1.5e-3


@pytest.mark.parametrize("name", ["dstep_celeba64_b8", "dstep_mnist_dcrn_b6"])
def test_per_sample_private_penalty_matches_reference_loop(golden_dir, name):
    """train.py:433-450 (--penalty_use_public_data False): vectors from the reference's own loop — calc_penalty(per_sample=True) and
    one autograd.grad per sample on the reference's D — against the oracle's restatement of that branch."""
    z, inp = load_case(golden_dir, name)
    oracle, D = _oracle_for(name, z, grad_clip_mode="standard", grad_clip_split=True, clipping_param=float(z["c_flat"]))
    oracle.cfg.penalty_use_public_data = False
    obs = oracle.step(inp["img"], inp["labels"], inp["z"], inp["y"], alpha=inp["alpha"], apply_update=False)
    assert obs["penalty"] == pytest.approx(float(z["private_penalty_mean"]), rel=1e-5)
    _check_grads(z, "sum_flat_split", obs["summed_clipped"], "first clip", 2e-6)
    _check_grads(z, "sum_flat_split_private_pen", obs["summed_clipped_with_penalty"], "second clip, penalty gradients in p.grad_sample[0]", 1e-5)
