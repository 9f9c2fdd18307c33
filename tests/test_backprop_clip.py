"""--backprop_clip: csl_gan_amd.backprop_clip.BackpropClipper against vectors made by the REFERENCE's own
PGCWrapper / BackpropClipper.convert (tests/golden/bpc_*.npz, tests/golden/make_golden.py:bpc_case) — the analytic bounds, the
forward output with clipped layer inputs, and the parameter / input gradients with the per-sample output-gradient clip on and off.

  * not gpu : the package's CPU plumbing path and the oracle restatement (oracle/backprop_clip.py);
  * gpu     : the HIP layers (cslgan_l2_clip_rows_f32 on the layer input and on the pre-activation gradient inside the conv
              backward), plus one DP D-step with the clipper driving the engine's per-layer clip norms (train.py:84-92)."""
import os

import numpy as np
import pytest
import torch

CASES = [
    ("bpc_mnist_dcrn_auto_b6", ["MNIST", "--model", "DeepConvResNet"]),
    ("bpc_mnist_vanilla_cond_auto_b8", ["MNIST", "--model", "Vanilla", "--conditional", "--conditional_arch", "CGAN", "--aux_loss_type", "cross_entropy"]),
    ("bpc_mnist_dcrn_explicit_b5", ["MNIST", "--model", "DeepConvResNet"]),
]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sample(t, n=4096):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)][:n].cpu().numpy()


def _D(tmp_path, argv, seed, device):
    from csl_gan_amd import init_util, options
    opt = options.parse(argv + ["-dpm", "gc", "-nms", "4", "-bs", "4", "-gd", device, "-dd", device, "-o", str(tmp_path), "--manual_seed", "1",
                                "--weights_seed", str(seed)])
    _, D = init_util.init_models(opt, init_G=False)
    return opt, D


def _run(D, z, device, clipper_state):
    B = int(z["B"])
    x = torch.from_numpy(z["x"]).to(device)
    y = torch.from_numpy(z["y"]).to(device) if "y" in z.files else None
    w = torch.from_numpy(z["w"]).to(device)
    res = {}
    for tag, on in (("on", True), ("off", False)):
        clipper_state(on)
        for p in D.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        out, _ = D(xi, y)
        ((out.reshape(B) * w).sum() / B).backward()
        res[tag] = (out.detach(), xi.grad, [p.grad for p in D.parameters()])
    return res


def _compare(res, z, tol):
    for tag in ("on", "off"):
        out, gx, grads = res[tag]
        oscale = np.abs(z["out"]).max()
        assert np.abs(out.cpu().numpy().reshape(-1) - z["out"].reshape(-1)).max() <= tol * oscale
        exp = z["gx_" + tag]
        assert np.abs(gx.cpu().numpy() - exp).max() <= tol * np.abs(exp).max(), "gx " + tag
        for i, g in enumerate(grads):
            e = z["g%d_%s" % (i, tag)]
            got = _sample(g)
            assert np.abs(got - e).max() <= tol * max(np.abs(e).max(), 1e-30), "param %d %s: %.3e" % (i, tag, np.abs(got - e).max() / np.abs(e).max())
            assert abs(g.double().cpu().norm().item() - z["gnorm_" + tag][i]) <= tol * z["gnorm_" + tag][i]
    # the clip does something in these cases
    assert (z["gnorm_on"] < 0.9 * z["gnorm_off"]).all()


def _check_package(tmp_path, name, argv, device, tol):
    from csl_gan_amd.backprop_clip import BackpropClipper
    z = np.load(os.path.join(GOLD, name + ".npz"))
    opt, D = _D(tmp_path, argv, int(z["seed"]), device)
    np.testing.assert_allclose([p.detach().cpu().double().norm().item() for p in D.parameters()], z["wnorms"], rtol=1e-5)
    back, fwd = (list(z["back_in"]), list(z["fwd_in"])) if "back_in" in z.files else (None, None)
    names = [n for n, _ in D.named_parameters()]
    c = BackpropClipper(D, back, fwd, float(z["aas"]), float(z["awgs"]), device=device)
    assert [n for n, _ in D.named_parameters()] == names           # layers are not re-parented: checkpoints keep their keys
    np.testing.assert_allclose(c.grad_l2_bounds, z["grad_l2_bounds"], rtol=1e-12)
    np.testing.assert_allclose(c.back_clip_params, z["back_clip_params"], rtol=1e-12)
    np.testing.assert_allclose(c.input_clip_params, z["input_clip_params"], rtol=1e-12)
    _compare(_run(D, z, device, lambda on: c.enable_hooks() if on else c.disable_hooks()), z, tol)


@pytest.mark.parametrize("name,argv", CASES, ids=[c[0] for c in CASES])
def test_package_cpu_path_matches_reference_pgcwrapper(tmp_path, name, argv):
    _check_package(tmp_path, name, argv, "cpu", 1e-5)


@pytest.mark.parametrize("name,argv", CASES, ids=[c[0] for c in CASES])
def test_oracle_restatement_matches_reference_pgcwrapper(tmp_path, name, argv):
    from oracle import backprop_clip as OB
    from oracle import nets as ON
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cond = "--conditional" in argv
    _, D = ON.build_models("MNIST", "Vanilla" if "Vanilla" in argv else "DeepConvResNet", 28, weights_seed=int(z["seed"]), conditional=cond,
                           n_classes=10, conditional_arch="CGAN", aux_loss_type="cross_entropy", init_G=False)
    np.testing.assert_allclose([p.detach().double().norm().item() for p in D.parameters()], z["wnorms"], rtol=1e-5)
    layers = _layer_table(D)
    back, fwd = (list(z["back_in"]), list(z["fwd_in"])) if "back_in" in z.files else (None, None)
    gb, bc, ic = OB.bounds(layers, back, fwd, float(z["aas"]), float(z["awgs"]))
    np.testing.assert_allclose(gb, z["grad_l2_bounds"], rtol=1e-12)
    np.testing.assert_allclose(bc, z["back_clip_params"], rtol=1e-12)
    state = {"on": True}
    OB.attach(D, ic, bc, state)
    _compare(_run(D, z, "cpu", lambda on: state.__setitem__("on", on)), z, 1e-5)


def _layer_table(D):
    """(kind, weight numel, bias?, input numel, output spatial numel) of D's parameterised leaves on a 28x28 image."""
    from torch import nn
    rows, hs = [], []
    leaves = [m for m in D.modules() if len(list(m.children())) < 1 and any(True for _ in m.parameters())]
    for m in leaves:
        hs.append(m.register_forward_hook(lambda m, i, o: setattr(m, "_io", (i[0].shape[1:], o.shape[1:]))))
    ncls = getattr(D, "n_classes", 0)
    with torch.no_grad():
        D(torch.zeros(1, 1, 28, 28), torch.zeros(1, dtype=torch.long) if ncls > 1 else None)
    for h in hs:
        h.remove()
    for m in leaves:
        i, o = m._io
        rows.append(("linear" if isinstance(m, nn.Linear) else "conv", m.weight.numel(), m.bias is not None, int(np.prod(i)),
                     int(np.prod(o[1:]))))
    return rows


def test_scalar_clip_parameters_fail_as_in_the_reference(tmp_path):
    """train.py:86 hands scalars in the non '-pl' modes; backprop_clip.py:80 indexes them -> TypeError in the reference too."""
    from csl_gan_amd.backprop_clip import BackpropClipper
    _, D = _D(tmp_path, CASES[0][1], 3, "cpu")
    with pytest.raises(TypeError):
        BackpropClipper(D, 0.01, 20.0)


def test_trainer_turns_the_bounds_into_per_layer_clip_norms(tmp_path):
    """train.py:84-92: clipping_param_per_layer = bound * batch_size, clipping_param = their 2-norm."""
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    opt = options.parse(["MNIST", "--model", "DeepConvResNet", "-dpm", "gc", "-gcm", "constant-pl", "-bpc", "True", "-nms", "4", "-bs", "8",
                         "-gd", "cpu", "-dd", "cpu", "-o", str(tmp_path), "--manual_seed", "1"])
    G, D = init_util.init_models(opt)
    tr = Trainer(opt, G, D, log_to=os.path.join(str(tmp_path), "log.csv"))
    c = tr.setup_backprop_clip()
    assert len(c.grad_l2_bounds) == len(list(D.parameters()))
    np.testing.assert_allclose(opt.clipping_param_per_layer, [8 * b for b in c.grad_l2_bounds])
    assert opt.clipping_param == pytest.approx(np.linalg.norm(opt.clipping_param_per_layer))


@pytest.mark.gpu
@pytest.mark.parametrize("name,argv", CASES, ids=[c[0] for c in CASES])
def test_hip_layers_match_reference_pgcwrapper(tmp_path, name, argv):
    _check_package(tmp_path, name, argv, "cuda:0", 1e-3)


@pytest.mark.gpu
def test_dp_step_with_backprop_clip_feeds_the_engine_the_clipped_per_sample_gradients(tmp_path):
    """One DP D-step (gc, constant-pl) with the clipper on (train.py:370-393): the per-sample gradient norms the engine measured
    equal those of the oracle's hook restatement, sample by sample (autograd on the CPU, one sample at a time), the engine clipped
    with bound x batch_size, and the hooks are back on after the step (--bpc_during_g_train, train.py:481-482)."""
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    from oracle import backprop_clip as OB
    from oracle import nets as ON
    B = 8
    opt = options.parse(["MNIST", "--model", "DeepConvResNet", "-dpm", "gc", "-gcm", "constant-pl", "-bpc", "True", "-nms", "4", "-bs", str(B),
                         "-gd", "cuda:0", "-dd", "cuda:0", "-o", str(tmp_path), "--manual_seed", "1", "--g_latent_dim", "16", "--sigma", "0.5",
                         "--materialize", "all", "--synthetic", "-bpcaas", "0.05"])
    G, D = init_util.init_models(opt)
    tr = Trainer(opt, G, D, log_to=os.path.join(str(tmp_path), "log.csv"))
    pe = tr.setup_privacy_engine()
    c = tr.prop_grad_clipper
    assert c is not None and not tr._can_fuse(True)
    np.testing.assert_allclose(np.asarray(pe.max_grad_norm, dtype=np.float64), [B * b for b in c.grad_l2_bounds], rtol=1e-6)
    _, Do = ON.build_models("MNIST", "DeepConvResNet", 28, init_G=False)
    with torch.no_grad():
        for po, p in zip(Do.parameters(), D.parameters()):
            po.copy_(p.detach().cpu())
    state = {"on": True}
    OB.attach(Do, c.input_clip_params, c.back_clip_params, state)
    g = torch.Generator().manual_seed(5)
    img = (torch.rand(B, 1, 28, 28, generator=g) * torch.logspace(-1.5, 0, B).view(B, 1, 1, 1)).cuda()
    tr.train_D(img, None, tr.gen_z(B), None, use_dp=True)
    torch.cuda.synchronize()
    assert c.hooks_enabled
    got = pe.last_sq.sqrt().cpu().double()               # [params, 2 passes x B]: generated rows, then real rows
    rows = torch.cat([tr.last["fake_img"].cpu(), img.cpu()])
    exp = torch.zeros_like(got)
    for r in range(2 * B):
        out, _ = Do(rows[r:r + 1])
        # the hooks see the gradient of the batch-MEAN loss (1/B per sample); the engine rescales per-sample gradients by B
        gr = torch.autograd.grad(out.sum() / B, list(Do.parameters()))
        exp[:, r] = torch.stack([x.double().norm() * B for x in gr])
    for i, (n, _) in enumerate(D.named_parameters()):
        err = (got[i] - exp[i]).abs().max().item() / exp[i].abs().max().item()
        assert err <= 1e-3, "%s: %.3e" % (n, err)
    assert all(torch.isfinite(p.grad).all() for p in D.parameters())
