"""Full BASELINE size (CelebA D64, bs=128) checks through size-independent properties — the CPU oracle
would need minutes and ~10 GB here, so instead of a reference comparison these assert identities that
must hold at any size: linearity (per-sample gradients sum to the dense gradient), the contract norm
kernel vs the wgrad-epilogue norms, clip invariants, C=inf clipping == plain sum, noise statistics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B = 128


@pytest.fixture(scope="module")
def setup(tmp_path_factory):
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    out = tmp_path_factory.mktemp("full")
    opt = options.parse(["CelebA", "-dpm", "gc", "-gcm", "constant-pl", "-nms", "4", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0",
                         "-o", str(out), "--manual_seed", "1", "--materialize", "all"])
    G, D = init_util.init_models(opt)
    tr = Trainer(opt, G, D, log_to=str(out / "log.csv"))
    pe = tr.setup_privacy_engine()
    g = torch.Generator().manual_seed(5)
    img = (torch.randn(B, 3, 64, 64, generator=g) * 0.5).clamp(-1, 1).cuda()
    return opt, tr, pe, D, img


def _backward_real(pe, D, img):
    pe.zero_grad()
    pe.enable_hooks()
    out, _ = D(img)
    D.real_loss(out, "cuda:0").backward()
    pe.disable_hooks()


def test_per_sample_grads_sum_to_dense_gradient(setup):
    opt, tr, pe, D, img = setup
    _backward_real(pe, D, img)
    gs = [p.grad_sample for p in D.parameters()]
    assert gs[6].shape == (1, B, 512, 256, 5, 5) and gs[8].shape == (1, B, 1, 8192)
    total_bytes = sum(g.numel() * 4 for g in gs)
    assert abs(total_bytes - B * 4314752 * 4) == 0                     # 2.21 GB materialised
    summed = [g[0].sum(0) / B for g in gs]                              # grad_sample carries the xB loss scaling
    for p in D.parameters():
        p.grad = None
    out, _ = D(img)                                                     # hooks off: dense path
    D.real_loss(out, "cuda:0").backward()
    for i, (p, s) in enumerate(zip(D.parameters(), summed)):
        scale = p.grad.abs().max().item() + 1e-12
        assert (p.grad - s).abs().max().item() <= 2e-4 * scale, "param %d" % i


def test_contract_norm_kernel_matches_epilogue_and_clip_invariants(setup):
    from csl_gan_amd import ops
    opt, tr, pe, D, img = setup
    _backward_real(pe, D, img)
    a, b = pe.sample_sqnorms(recompute=False), pe.sample_sqnorms(recompute=True)
    assert a.shape == (9, B)
    assert ((a - b).abs() <= 1e-4 * b.abs().max()).all()
    # per-layer clip norms at the median norm of each layer: about half the samples are clipped
    C = a.sqrt().median(dim=1).values
    pe.set_max_grad_norm_device(C)
    pe.clip()
    f = pe.last_factors
    assert f.shape == (9, B) and (f <= 1).all() and (f > 0).all()
    clipped_norm = f * a.sqrt()
    assert (clipped_norm <= C[:, None] * (1 + 1e-5)).all()
    frac = (f < 0.999).float().mean(dim=1)
    assert ((frac > 0.3) & (frac < 0.7)).all()
    got = [p.summed_grad.clone() for p in D.parameters()]
    # definition: sum_b f_b g_b  (torch ops as the independent check, layer by layer to bound memory)
    for i, p in enumerate(D.parameters()):
        ref = (p.grad_sample[0] * f[i].view(B, *([1] * (p.dim())))).sum(0)
        scale = ref.abs().max().item() + 1e-12
        assert (got[i] - ref).abs().max().item() <= 2e-4 * scale, "param %d" % i
    # C = inf: clipping is the identity -> plain sum
    pe.set_max_grad_norm_device(torch.full((9,), 1e30, device="cuda"))
    pe.clip()
    for i, p in enumerate(D.parameters()):
        ref = p.grad_sample[0].sum(0)
        assert (p.summed_grad - ref).abs().max().item() <= 2e-4 * (ref.abs().max().item() + 1e-12)


def test_noise_and_step_at_full_size(setup):
    opt, tr, pe, D, img = setup
    _backward_real(pe, D, img)
    C = torch.full((9,), 2.0, device="cuda")
    pe.set_max_grad_norm_device(C)
    pe.clip()
    clean = [p.summed_grad.clone() for p in D.parameters()]
    pe.noise_multiplier = 0.5
    before = [p.detach().clone() for p in D.parameters()]
    steps0 = pe.steps
    tr.d_optimizer.step()                                                # noise + 1/B + Adam on the HIP kernels
    assert pe.steps == steps0 + 1
    w = list(D.parameters())[6]                                          # 3.27M-element layer
    z = (w.grad * B - clean[6]) / (0.5 * 2.0)                            # recovered unit normals
    assert abs(z.mean().item()) < 3e-3 and abs(z.std().item() - 1) < 3e-3
    assert abs((z ** 4).mean().item() - 3.0) < 0.05
    moved = (w.detach() - before[6]).abs()
    assert 0 < moved.max().item() <= 1.01 * opt.d_lr                     # Adam, b1=0: |step| <= lr
    assert not hasattr(w, "grad_sample")                                 # per-sample state is dropped after the step


def test_edge_batches(tmp_path):
    """B=1 and a ragged B=5 go through the whole D-step; an empty batch is refused loudly."""
    from csl_gan_amd import init_util, options, ops
    from csl_gan_amd.trainer import Trainer
    for Bs in (1, 5):
        opt = options.parse(["MNIST", "--model", "DeepConvResNet", "-dpm", "gc", "-nms", "2", "-bs", str(Bs), "-gd", "cuda:0", "-dd", "cuda:0",
                             "-o", str(tmp_path / str(Bs)), "--manual_seed", "1", "--g_latent_dim", "8", "--penalty", "WGAN-GP"])
        G, D = init_util.init_models(opt)
        tr = Trainer(opt, G, D, log_to=str(tmp_path / ("log%d.csv" % Bs)))
        tr.setup_privacy_engine()
        tr.explicit = dict(pen_real=torch.rand(Bs, 1, 28, 28), ms_adapt=torch.rand(Bs, 1, 28, 28))
        tr.train_D(torch.rand(Bs, 1, 28, 28, device="cuda"), None, tr.gen_z(Bs), None, use_dp=True)
        assert all(torch.isfinite(p).all() for p in D.parameters())
    with pytest.raises(RuntimeError, match="null argument|non-positive dimension"):
        ops.conv2d_fwd(torch.zeros(0, 8, 8, 4, device="cuda"), torch.zeros(4, 3, 3, 4, device="cuda"), pad=1)
