"""HIP ``Trainer.train_D`` directly against vectors computed through the REFERENCE's own classes (-m gpu).

tests/golden/dstep_*.npz (make_golden.dstep_case) hold the D-step observables SURVEY.md §8(c) lists — per-layer per-sample
gradient norms ``[L, pass, B]``, flat norms, clip factors, the pre-noise clipped sums, the adaptive statistics of
train.py:204-245, the WGAN-GP penalty with its parameter gradients and ``summed_grad`` (train.py:431) — each computed
with one autograd call per sample through the reference's G / D classes and loss methods and the clip rule written out in
float64.  Nothing of oracle/ is involved here: parity of the device path does not rest on the oracle chain.

Tolerance: 1e-3 (the north-star bar) — of each vector's scale for norms / factors / statistics / losses, of each gradient
tensor's term scale (``*_absmax``: for a sum over samples the largest sum of |terms|) for the sampled entries and 1e-3 relative
for tensor norms.  Both sides decide their own LeakyReLU / ReLU masks (a fixture cannot replay the device's); measured
(CSLGAN_GOLDEN_REPORT, 411 comparisons): everything agrees to 1e-6 .. 6e-4 except one flipped unit in the 128x128 case, which
`_close(flips=True)` admits for gradient-tensor entries only (the masked-oracle tests in test_dstep_gpu.py cover the general case).
"""
import os

import numpy as np
import pytest
import torch

from tests.golden.dstep_inputs import DSTEP_CASES, load_case, sampled

pytestmark = pytest.mark.gpu
TOL = 1e-3


_REPORT = os.environ.get("CSLGAN_GOLDEN_REPORT")      # optional: append every comparison's relative error to this file


def _close(got, exp, what, tol=TOL, scale=None, flips=False):
    got = np.asarray(torch.as_tensor(got).detach().cpu().double().numpy() if torch.is_tensor(got) else got, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    s = (np.abs(exp).max() if scale is None else scale) + 1e-30
    d = np.abs(got - exp)
    err = d.max()
    if _REPORT:
        with open(_REPORT, "a") as f:
            f.write("%-70s rel %.3e  frac>tol %.4f  n %d  [%s]\n" % (what, err / s, float((np.abs(got - exp) > tol * s).mean()), got.size,
                                                                 os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1]))
    if flips and err > tol * s:
        # One LeakyReLU unit of one sample within fp32 rounding of zero takes the other slope on the device: the handful of gradient
        # entries that unit feeds move by a few 1e-3 while every other entry agrees to ~1e-6 (measured on dstep_celeba128_b4, whose
        # critic evaluates 2e6 units per forward: 4 of 2048 sampled entries of conv1's penalty gradient at 1.5e-3, the remaining
        # 2044 and every entry of the 64x64 cases <= 3e-6; another box: one of conv2's 128 bias-gradient entries at 1.08e-3).
        # Allowed: <= 0.5 % of a tensor's entries (at least two), none beyond 5e-3.
        assert (d > tol * s).sum() <= max(2, 5e-3 * d.size) and err <= 5 * tol * s, "%s: %.4f of the entries beyond %.0e, max rel %.3e" % (
            what, (d > tol * s).mean(), tol, err / s)
        return
    assert err <= tol * s, "%s: max abs err %.3e at scale %.3e (rel %.3e)" % (what, err, s, err / s)


def _check_grads(z, key, tensors, what, tol=TOL, flip_probe=None):
    """flip_probe (every arithmetic; for exact fp32 it compares the round-3 and round-4 summation orders): a callable returning how many LeakyReLU / ReLU units take a
    DIFFERENT slope in this run than in an exact-fp32 run of the same step on the device.  Both arithmetics are fp32-accurate but round
    differently (~2e-6 of a layer's scale), so a unit whose pre-activation is that close to zero can land on the other side — and a
    flipped unit deep in the critic shifts EVERY upstream gradient entry of its sample a little (B = 4..16 here: 1e-3..3e-3 of a summed
    tensor; measured on dstep_celeba64_cond_acgan_b8: one unit of the third conv's output, 9 % of the first conv's entries moved by up
    to 2.5e-3 while every launch agrees with its exact-fp32 twin to 3e-6).  A fixture cannot replay masks, so when — and only when —
    the strict per-entry check fails AND the probe proves flips between the two device arithmetics, the tensor is held to the
    free-running bound instead: 1e-2 in relative L2 and on its norm, the bar tests/test_dstep_gpu.py::_close_grad applies to small
    batches (measured here with 3..8 flipped units at B = 8: 4.6e-3 on the first conv's filter gradient, 5.3e-3 on its bias — a sum
    over pixels that cancels).  Per entry at 1e-3 with shared masks is tests/test_dstep_gpu.py's and tests/test_fullsize_gpu.py's
    job, in these modes too."""
    top = float(z[key + "_absmax"].max())
    for i, t in enumerate(tensors):
        amax, nrm = float(z[key + "_absmax"][i]), float(z[key + "_norms"][i])
        got = torch.zeros(1) if t is None else t
        if amax <= 1e-7 * top:          # exactly-zero gradients (penalty bias gradients)
            assert got.abs().max().item() <= 1e-5 * top, (what, i)
            continue
        gn = got.detach().double().norm().item()
        try:
            assert abs(gn - nrm) <= tol * max(nrm, 1e-3 * float(z[key + "_norms"].max())), "%s[%d] norm %.6e vs %.6e" % (what, i, gn, nrm)
            _close(sampled(got), z["%s_s%d" % (key, i)], "%s[%d] entries" % (what, i), tol=tol, scale=amax, flips=True)
        except AssertionError:
            n_flips = flip_probe() if flip_probe is not None else 0
            if n_flips == 0:
                raise
            exp = np.asarray(z["%s_s%d" % (key, i)], dtype=np.float64)
            l2 = np.linalg.norm(np.asarray(sampled(got), dtype=np.float64) - exp) / (np.linalg.norm(exp) + 1e-30)
            print("%s[%d]: %d unit(s) flipped against the exact-fp32 device run; relative L2 %.3e" % (what, i, n_flips, l2))
            assert l2 <= 1e-2 and abs(gn - nrm) <= 1e-2 * max(nrm, 1e-3 * float(z[key + "_norms"].max())), (what, i, l2, gn, nrm)


def _flip_probe(tmp_path, name, z, inp, mode_flags, materialize, compute, has_pen):
    """Number of activation units whose sign differs between the `compute` run and an exact-fp32 run of the same step on the round-3
    kernels (device masks recorded by csl_gan_amd.nn.ActivationMaskRecorder); evaluated at most once.  For compute == "fp32" the two
    runs are the same exact-fp32 products summed in two orders: the round-3 kernels and the round-4 halo kernel."""
    cache = []

    def probe():
        if not cache:
            from csl_gan_amd import nn as hnn, ops
            masks = []
            for k, mode in enumerate(("fp32", compute)):
                opt, tr = _trainer(tmp_path / ("probe%d" % k), name, z, mode_flags, materialize, mode)
                rec = hnn.ActivationMaskRecorder(G=tr.G, D=tr.D)
                hnn.set_mask_recorder(rec)
                prev = ops.set_f32_halo(False) if k == 0 else None
                try:
                    _run(tr, inp, has_pen)
                finally:
                    hnn.set_mask_recorder(None)
                    if k == 0:
                        ops.set_f32_halo(prev)
                        ops.repack_cache.clear()
                masks.append(rec.masks)
            assert masks[0].keys() == masks[1].keys()
            cache.append(sum(int((a != b).sum()) for k in masks[0] for a, b in zip(masks[0][k], masks[1][k])))
        return cache[0]
    return probe


def _trainer(tmp_path, name, z, mode_flags, materialize, compute="fp32"):
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    dataset, _, _, latent, _, extra = DSTEP_CASES[name]
    B = int(z["meta"][0])
    argv = [dataset, "-dpm", "gc", "-nms", "4", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", str(tmp_path), "--manual_seed", "1",
            "--g_latent_dim", str(latent), "--sigma", "0.5", "--materialize", materialize, "-as", repr(float(z["adaptive_scalar"])),
            "--compute_dtype", compute]
    opt = options.parse(argv + extra + mode_flags)
    G, D = init_util.init_models(opt)
    np.testing.assert_allclose([p.detach().cpu().double().norm().item() for p in D.parameters()], z["d_weight_norms"], rtol=1e-5)
    np.testing.assert_allclose([p.detach().cpu().double().norm().item() for p in G.parameters()], z["g_weight_norms"], rtol=1e-5)
    assert [n for n, _ in D.named_parameters()] == list(z["d_param_names"])
    tr = Trainer(opt, G, D, log_to=str(tmp_path / "log.csv"))
    pe = tr.setup_privacy_engine()
    pe.noise_multiplier = 0.0
    return opt, tr


def _run(tr, inp, has_pen):
    cu = lambda t: None if t is None else t.cuda()
    tr.explicit = dict(ms_adapt=inp["ms_adapt"], ms_adapt_labels=inp["ms_adapt_labels"], alpha=inp["alpha"], z_adapt=inp["z_adapt"].cuda(),
                       keep=True)
    if has_pen:
        tr.explicit["pen_real"] = inp["ms_pen"]
    tr.train_D(inp["img"].cuda(), cu(inp["labels"]), inp["z"].cuda(), cu(inp["y"]), use_dp=True)
    torch.cuda.synchronize()
    return tr.last


def _private_cols(t, B):
    """norms / factors of the private (real) pass: the last B columns ([L, 2B] when every pass is materialised, [L, B] otherwise)."""
    t = t.reshape(t.shape[0], -1) if t.dim() > 1 else t.reshape(1, -1)
    return t[:, -B:]


CASES = [(n, m) for n in sorted(DSTEP_CASES) for m in ("all", "ghost")]
# The fp32-accurate arithmetic on the bf16 matrix cores (csrc/igemm_bf16.hip, csrc/igemm_x3.hip) is held to the SAME 1e-3 against
# the SAME reference vectors.  "bf16x3" sends every MFMA launch of the step through the three-piece kernels (a superset of what
# "fp32_auto" — bench.py's headline arithmetic — selects at any size: fp32_auto's rule is by launch size, and these fixtures are
# B = 4..16); "fp32_auto" itself runs too, so the routing code is the one the headline uses.
COMPUTES = ("fp32", "bf16x3", "fp32_auto")


@pytest.mark.parametrize("compute", COMPUTES)
@pytest.mark.parametrize("name,materialize", CASES)
def test_train_D_adaptive_pl_matches_reference_vectors(tmp_path, golden_dir, name, materialize, compute):
    """BASELINE configs[2] mode: -gcm adaptive-pl (+ WGAN-GP on the public batch where the model has a penalty), on the fork's
    layout (--materialize all) and on the benchmarked route (ghost clipping, fused passes)."""
    z, inp = load_case(golden_dir, name)
    has_pen = "penalty" in z.files
    B = int(z["meta"][0])
    opt, tr = _trainer(tmp_path, name, z, ["-gcm", "adaptive-pl"], materialize, compute)
    assert opt.compute_dtype == compute
    probe = _flip_probe(tmp_path, name, z, inp, ["-gcm", "adaptive-pl"], materialize, compute, has_pen)
    last = _run(tr, inp, has_pen)
    fake = last["fake_img"].detach().cpu().contiguous()
    _close(fake.reshape(-1)[::max(1, fake.numel() // 4096)][:4096], z["fake_sample"], "G(z)")
    dscale = float(np.abs(z["d_real"]).max() + np.abs(z["d_fake"]).max())
    _close(last["d_real"], z["d_real"], "d_real", scale=dscale)
    _close(last["d_fake"], z["d_fake"], "d_fake", scale=dscale)
    _close(last["d_real_loss"].reshape(1), [float(z["d_real_loss"])], "d_real_loss", scale=max(dscale, abs(float(z["d_real_loss"]))))
    _close(last["d_fake_loss"].reshape(1), [float(z["d_fake_loss"])], "d_fake_loss", scale=max(dscale, abs(float(z["d_fake_loss"]))))
    _close(last["adaptive_stats"], z["adaptive_mean"], "adaptive statistics (train.py:204-245)")
    _close(last["clip_params"], z["c_adaptive_pl"], "adaptive per-layer C")
    _close(_private_cols(last["norms"], B), z["layer_norms"][:, 1], "per-layer per-sample norms, private pass")
    _close(_private_cols(last["clip_factors"], B), z["factors_pl"], "per-layer clip factors, private pass")
    if materialize == "all":            # the fork's layout: pass 0 (generated batch) is materialised too
        _close(last["norms"].reshape(len(z["layer_norms"]), -1)[:, :B], z["layer_norms"][:, 0], "per-layer per-sample norms, generated pass")
    _check_grads(z, "sum_pl_split", last["summed_clipped"], "clipped sum (per-layer C, split passes)", flip_probe=probe)
    if has_pen:
        assert abs(last["penalty"].item() - float(z["penalty"])) <= TOL * float(z["penalty"])
        _check_grads(z, "pen_grad", last["penalty_grads"], "penalty parameter gradients", flip_probe=probe)
        _check_grads(z, "summed_grad_pl", last["summed_grad"], "summed_grad (train.py:431)", flip_probe=probe)


@pytest.mark.parametrize("compute", ("fp32", "bf16x3"))
@pytest.mark.parametrize("name", sorted(DSTEP_CASES))
def test_train_D_flat_clip_matches_reference_vectors(tmp_path, golden_dir, name, compute):
    """One constant flat C chosen to clip about half of the private samples; adaptive flat C; accumulated passes (-gcs False)."""
    z, inp = load_case(golden_dir, name)
    has_pen = "penalty" in z.files
    B = int(z["meta"][0])
    opt, tr = _trainer(tmp_path / "flat", name, z, ["-c", repr(float(z["c_flat"]))], "all", compute)
    mk = lambda flags, mat: _flip_probe(tmp_path / ("p" + mat + str(len(flags))), name, z, inp, flags, mat, compute, has_pen)
    last = _run(tr, inp, has_pen)
    n = last["norms"].reshape(1, -1)
    _close(n[:, :B], z["flat_norms"][0:1], "flat per-sample norms, generated pass")
    _close(n[:, B:], z["flat_norms"][1:2], "flat per-sample norms, private pass")
    _close(_private_cols(last["clip_factors"], B), z["factors_flat"].reshape(1, -1), "flat clip factors")
    assert ((z["factors_flat"] < 0.999).any() and (z["factors_flat"] > 0.999).any())
    pr = mk(["-c", repr(float(z["c_flat"]))], "all")
    _check_grads(z, "sum_flat_split", last["summed_clipped"], "clipped sum (flat C, split passes)", flip_probe=pr)
    if has_pen:
        _check_grads(z, "summed_grad_flat", last["summed_grad"], "summed_grad (flat C)", flip_probe=pr)
    # the default (ghost) route with the same flat C
    opt, tr = _trainer(tmp_path / "ghost", name, z, ["-c", repr(float(z["c_flat"]))], "ghost", compute)
    last = _run(tr, inp, has_pen)
    _close(_private_cols(last["norms"], B), z["flat_norms"][1:2], "flat per-sample norms (ghost route)")
    _check_grads(z, "sum_flat_split", last["summed_clipped"], "clipped sum (flat C, ghost route)", flip_probe=mk(["-c", repr(float(z["c_flat"]))], "ghost"))
    # adaptive flat C = adaptive_scalar * ||r||_2 (train.py:243)
    opt, tr = _trainer(tmp_path / "aflat", name, z, ["-gcm", "adaptive"], "all", compute)
    last = _run(tr, inp, has_pen)
    _close(last["clip_params"].reshape(1), [float(z["c_adaptive_flat"])], "adaptive flat C")
    _close(_private_cols(last["clip_factors"], B), z["factors_adaptive_flat"].reshape(1, -1), "adaptive flat clip factors")
    # accumulated passes: per-sample sum over the generated and the private pass, clipped together
    opt, tr = _trainer(tmp_path / "accum", name, z, ["-gcs", "False", "-c", repr(float(z["c_accum"]))], "all", compute)
    last = _run(tr, inp, has_pen)
    _check_grads(z, "sum_flat_accum", last["summed_clipped"], "clipped sum (accumulated passes)",
                 flip_probe=mk(["-gcs", "False", "-c", repr(float(z["c_accum"]))], "all"))


def test_survey_probe_vector_on_hip(tmp_path, golden_dir):
    """SURVEY.md §8c's probe (reference D64 built alone under seed 42, x = randn(16,3,64,64).clamp(-1,1) under manual_seed 1,
    real loss): per-layer per-sample norms through the HIP per-sample-gradient engine."""
    import os
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    z = np.load(os.path.join(golden_dir, "dstep_survey_probe.npz"))
    opt = options.parse(["CelebA", "-dpm", "gc", "-nms", "4", "-bs", "16", "-gd", "cuda:0", "-dd", "cuda:0", "-o", str(tmp_path),
                         "--manual_seed", "1", "--materialize", "all"])
    _, D = init_util.init_models(opt, init_G=False)
    np.testing.assert_allclose([p.detach().cpu().double().norm().item() for p in D.parameters()], z["d_weight_norms"], rtol=1e-5)
    G, _ = init_util.init_models(opt, init_D=False)
    tr = Trainer(opt, G, D, log_to=str(tmp_path / "log.csv"))
    pe = tr.setup_privacy_engine()
    torch.manual_seed(1)
    x = torch.randn(16, 3, 64, 64).clamp(-1, 1)
    np.testing.assert_allclose([x.double().sum().item(), (x.double() ** 2).sum().item()], z["x_checksum"], rtol=1e-9)
    pe.zero_grad(); pe.enable_hooks()
    out, _ = D(x.cuda())
    D.real_loss(out, "cuda:0").backward()
    pe.disable_hooks()
    n = pe.sample_sqnorms(recompute=False)[:, :16].sqrt()
    _close(out, z["d_real"], "D(x)")
    _close(n, z["layer_norms"], "probe per-sample norms")
    assert np.abs(n.mean(dim=1).cpu().numpy() - z["survey_quote"]).max() < 2e-3


@pytest.mark.parametrize("name", ["dstep_celeba64_b8", "dstep_mnist_dcrn_b6"])
def test_train_D_private_penalty_matches_reference_loop(tmp_path, golden_dir, name):
    """--penalty_use_public_data False (train.py:433-450) against vectors from the reference's own per-sample loop
    (make_golden.dstep_case, private_penalty): the device's ONE second-order sweep with per-sample weight-gradient kernels must give
    the sum the reference's B separate autograd calls give."""
    z, inp = load_case(golden_dir, name)
    opt, tr = _trainer(tmp_path, name, z, ["-c", repr(float(z["c_flat"])), "--penalty_use_public_data", "False", "-nms", "0"], "all")
    assert opt.materialize == "all"
    tr.explicit = dict(alpha=inp["alpha"], keep=True)
    tr.train_D(inp["img"].cuda(), None, inp["z"].cuda(), None, use_dp=True)
    torch.cuda.synchronize()
    last = tr.last
    assert abs(last["penalty"].item() - float(z["private_penalty_mean"])) <= TOL * float(z["private_penalty_mean"])
    _check_grads(z, "sum_flat_split", last["summed_clipped"], "first clip")
    _check_grads(z, "sum_flat_split_private_pen", last["summed_clipped_with_penalty"], "second clip with the per-sample penalty gradients")


def test_train_D_bf16_storage_128x128_against_reference_vectors(tmp_path, golden_dir):
    """BASELINE configs[4]'s arithmetic (`--compute_dtype bf16 --storage_dtype bf16`, 3x128x128) against the vectors the REFERENCE's
    own classes gave for `dstep_celeba128_b4` in fp32 — no oracle, no second route of the product in the loop.

    Tolerances follow tests/test_bf16s_gpu.py's error model (each bfloat16 rounding a relative +-2^-9, 16..32 rounding stages on the
    path of a critic weight gradient): observables that are continuous in the activations (generated image, critic outputs, losses,
    per-sample norms, adaptive statistics, clip norms, clip factors) are held to 4e-2 of scale per entry (3 sigma of the model for a
    maximum over ~10^5 entries).  Gradient TENSORS are free-running here — a fixture cannot replay the device's activation masks —
    and under ANY bf16 arithmetic ~0.3 % of the LeakyReLU / ReLU units sit closer to zero than their own rounding error and take the
    other slope: sqrt(0.003) x 0.8 = 4e-2 of a tensor in relative L2 on top of the 2e-2 rounding bound (measured free-running against
    the oracle: 3.5e-2..8e-2 per tensor).  They are therefore held to 1e-1 in relative L2 over the fixture's 2048 sampled entries
    (2e-1 for bias gradients — sums over pixels that cancel — and for the penalty's second-order gradients) and 6e-2 on the tensor
    norm; the per-tensor 2e-2 bound with shared masks is tests/test_bf16s_gpu.py's."""
    name = "dstep_celeba128_b4"
    z, inp = load_case(golden_dir, name)
    B = int(z["meta"][0])
    from csl_gan_amd import init_util, options, ops
    from csl_gan_amd.trainer import Trainer
    dataset, _, _, latent, _, extra = DSTEP_CASES[name]
    argv = [dataset, "-dpm", "gc", "-nms", "4", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", str(tmp_path), "--manual_seed", "1",
            "--g_latent_dim", str(latent), "--sigma", "0.5", "--materialize", "ghost", "-as", repr(float(z["adaptive_scalar"])),
            "--compute_dtype", "bf16", "--storage_dtype", "bf16", "-gcm", "adaptive-pl"]
    opt = options.parse(argv + extra)
    G, D = init_util.init_models(opt)
    tr = Trainer(opt, G, D, log_to=str(tmp_path / "log.csv"))
    assert (ops.get_compute_dtype(), ops.get_storage_dtype()) == ("bf16", "bf16") and opt.materialize == "ghost"
    tr.setup_privacy_engine().noise_multiplier = 0.0
    last = _run(tr, inp, True)
    T = 4e-2
    fake = last["fake_img"].float().detach().cpu().contiguous()
    _close(fake.reshape(-1)[::max(1, fake.numel() // 4096)][:4096], z["fake_sample"], "G(z), bf16-stored generator", tol=T)
    dscale = float(np.abs(z["d_real"]).max() + np.abs(z["d_fake"]).max())
    _close(last["d_real"].float(), z["d_real"], "d_real", tol=T, scale=dscale)
    _close(last["d_fake"].float(), z["d_fake"], "d_fake", tol=T, scale=dscale)
    _close(last["adaptive_stats"], z["adaptive_mean"], "adaptive statistics", tol=T)
    _close(last["clip_params"], z["c_adaptive_pl"], "adaptive per-layer C", tol=T)
    _close(_private_cols(last["norms"], B), z["layer_norms"][:, 1], "per-layer per-sample norms, private pass", tol=T)
    _close(_private_cols(last["clip_factors"], B), z["factors_pl"], "per-layer clip factors", tol=T)
    assert abs(last["penalty"].item() - float(z["penalty"])) <= T * float(z["penalty"])
    for key, tensors in (("sum_pl_split", last["summed_clipped"]), ("pen_grad", last["penalty_grads"]), ("summed_grad_pl", last["summed_grad"])):
        top = float(z[key + "_absmax"].max())
        for i, t in enumerate(tensors):
            amax, nrm = float(z[key + "_absmax"][i]), float(z[key + "_norms"][i])
            if amax <= 1e-7 * top:
                assert t is None or t.abs().max().item() <= 1e-4 * top, (key, i)
                continue
            gn = t.detach().double().norm().item()
            assert abs(gn - nrm) <= 6e-2 * max(nrm, 1e-3 * float(z[key + "_norms"].max())), "%s[%d] norm %.5e vs %.5e" % (key, i, gn, nrm)
            got = np.asarray(sampled(t.float()), dtype=np.float64)
            exp = np.asarray(z["%s_s%d" % (key, i)], dtype=np.float64)
            l2 = np.linalg.norm(got - exp) / (np.linalg.norm(exp) + 1e-30)
            # bias gradients are sums over pixels that cancel (between two bf16 runs they already differ by 1.3e-2..2.2e-2;
            # measured against the fp32 fixture: 1.15e-1 on the first conv's bias): twice the filter bound
            # the penalty's parameter gradients come out of a SECOND-order sweep (forward, data gradient to the image, and both again):
            # twice the rounding stages and twice the activation masks on the path (measured 1.0e-1 on the last conv's filter)
            lim = 2e-1 if (t.dim() == 1 or key == "pen_grad") else 1e-1
            assert l2 <= lim, "%s[%d]: relative L2 %.3e over the sampled entries (free-running masks)" % (key, i, l2)
