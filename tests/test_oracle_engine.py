"""Oracle self-consistency: the hook route (unfold+einsum) must equal the micro-batch definition,
and clip / accumulate must satisfy their defining properties.  PARITY UNPINNED vs the opacus fork
(absent); these tests pin the oracle to the mathematical definition."""
import numpy as np
import pytest
import torch

from oracle import dp_engine as E
from oracle.dstep import OracleDStep, StepConfig
from oracle.nets import build_models


def _loss_real(D, xb, yb):
    return D.real_loss(D(xb, yb)[0])


@pytest.mark.parametrize("dataset,model,im", [("MNIST", "Vanilla", 28), ("MNIST", "DeepConvResNet", 28)])
def test_hook_equals_microbatch(dataset, model, im):
    _, D = build_models(dataset=dataset, model=model, im_size=im, init_G=False)
    B = 5
    x = torch.rand(B, 1, 28, 28, generator=torch.Generator().manual_seed(3))
    ref = E.per_sample_grads_microbatch(D, _loss_real, x)
    h = E.HookPerSample(D)
    out, _ = D(x)
    D.real_loss(out).backward()
    for p, g in zip(D.parameters(), ref):
        assert p.grad_sample.shape[:2] == (1, B)
        torch.testing.assert_close(p.grad_sample[0], g, rtol=1e-4, atol=1e-6)
    # sum of per-sample grads / B == dense grad
    for p in D.parameters():
        torch.testing.assert_close(p.grad_sample[0].sum(0) / B, p.grad, rtol=1e-4, atol=1e-6)
    h.remove()


def test_two_passes_are_indexed_in_forward_order():
    _, D = build_models(dataset="MNIST", model="Vanilla", init_G=False)
    g = torch.Generator().manual_seed(4)
    xf, xr = torch.rand(4, 1, 28, 28, generator=g), torch.rand(4, 1, 28, 28, generator=g)
    ref_f = E.per_sample_grads_microbatch(D, lambda D, xb, yb: D.fake_loss(D(xb)[0]), xf)
    ref_r = E.per_sample_grads_microbatch(D, _loss_real, xr)
    h = E.HookPerSample(D)
    lf = D.fake_loss(D(xf)[0])
    lr = D.real_loss(D(xr)[0])
    (lf + lr).backward()
    h.enabled = False
    for p, a, b in zip(D.parameters(), ref_f, ref_r):
        assert p.grad_sample.shape[0] == 2
        torch.testing.assert_close(p.grad_sample[0], a, rtol=1e-4, atol=1e-6)
        torch.testing.assert_close(p.grad_sample[1], b, rtol=1e-4, atol=1e-6)
    h.remove()


def test_clip_properties():
    g = torch.Generator().manual_seed(0)
    gs = [torch.randn(2, 6, 3, 4, generator=g) * 3, torch.randn(2, 6, 5, generator=g)]
    # flat, split: pass 0 untouched, pass 1 clipped to <= C
    C = 2.0
    s = E.clip_and_sum(gs, C, accum_passes=False, num_private_passes=1)
    norms = E.calc_sample_norms(gs, flat=True)[0]
    f = (C / (norms + 1e-6)).clamp(max=1.0)
    exp0 = gs[0][0].sum(0) + (gs[0][1] * f[1].view(-1, 1, 1)).sum(0)
    torch.testing.assert_close(s[0], exp0)
    clipped = [g_[1] * f[1].view(-1, *[1] * (g_.dim() - 2)) for g_ in gs]
    tot = torch.sqrt(sum(c.reshape(6, -1).pow(2).sum(1) for c in clipped))
    assert (tot <= C + 1e-4).all()
    # per-layer
    s2 = E.clip_and_sum(gs, [1.0, 100.0], accum_passes=False, num_private_passes=None)
    n1 = gs[1].reshape(2, 6, -1).norm(dim=2)
    assert (n1 < 100).all()
    torch.testing.assert_close(s2[1], gs[1].sum((0, 1)))
    # accum_passes: passes summed per sample first
    s3 = E.clip_and_sum(gs, 1e9, accum_passes=True, num_private_passes=None)
    torch.testing.assert_close(s3[0], gs[0].sum((0, 1)))


def test_l2_clip_definition():
    t = torch.randn(7, 3, 4, 4, generator=torch.Generator().manual_seed(1)) * 2
    out = E.l2_clip(t, 3.0)
    n_in = t.reshape(7, -1).norm(dim=1)
    n_out = out.reshape(7, -1).norm(dim=1)
    assert torch.allclose(n_out, n_in.clamp(max=3.0), rtol=1e-5)
    keep = n_in <= 3.0
    assert torch.equal(out[keep], t[keep])


def test_oracle_dstep_runs_all_gc_modes():
    for mode in ("standard", "adaptive", "constant-pl", "adaptive-pl"):
        G, D = build_models(dataset="MNIST", model="DeepConvResNet", im_size=28, g_latent_dim=16)
        n = len(list(D.parameters()))
        cfg = StepConfig(grad_clip_mode=mode, clipping_param=5.0, clipping_param_per_layer=[1.0] * n, sigma=0.0)
        st = OracleDStep(G, D, cfg)
        g = torch.Generator().manual_seed(2)
        B = 4
        img = torch.rand(B, 1, 28, 28, generator=g)
        ms = torch.rand(B, 1, 28, 28, generator=g)
        obs = st.step(img, None, torch.randn(B, 16, generator=g), None, ms_adapt=ms, pen_real=ms,
                      alpha=torch.rand(B, generator=g), z_adapt=torch.randn(B, 16, generator=g))
        assert np.isfinite(obs["penalty"]) and len(obs["grads"]) == n
        assert obs["norms"].shape[1:] == (2, B)
