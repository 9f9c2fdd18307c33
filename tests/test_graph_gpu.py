"""HIP-graph replay of the D-step (-m gpu): csl_gan_amd.trainer.GraphedDStep records one Trainer.train_D and replays it; the
replayed steps must leave the critic exactly where the same sequence run eagerly leaves it (same inputs, same device RNG
draws, same Philox noise: the call counter and Adam's step count are read from HBM, so replays advance them)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(tmp_path, tag, argv, B, use_graph, n_steps, img_shape, conditional, smooth=False):
    """smooth: every ReLU / LeakyReLU of G and D switched off (the step is then a smooth function of its inputs: float-atomic
    reordering cannot flip a unit, so eager and replayed gradients agree to rounding).  lr = 0 during the two eager warm-up steps: the critic is bit-identical in both runs when the third step is captured /
    run eagerly, so its noised gradients can be compared before Adam's normalisation amplifies float-atomic reordering."""
    from csl_gan_amd import init_util, options
    from csl_gan_amd.mean_sampler import MeanSampler
    from csl_gan_amd.trainer import GraphedDStep, Trainer
    out = tmp_path / tag
    opt = options.parse(argv + ["-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", str(out), "--manual_seed", "1"])
    G, D = init_util.init_models(opt)
    if smooth:
        from csl_gan_amd import ops
        from csl_gan_amd.nn import HipConv2d, HipGroupNormAct, HipLinear
        for m in list(G.modules()) + list(D.modules()):
            if isinstance(m, HipGroupNormAct):
                m.relu = False
            elif isinstance(m, (HipConv2d, HipLinear)):
                m.in_lrelu = False              # no producer has an activation any more (functional.fused_act_masks)
                if m.act in (ops.ACT_LRELU02, ops.ACT_RELU):
                    m.act = ops.ACT_NONE
    ms = None
    if opt.num_mean_samples > 0:
        ms = MeanSampler(num_samples=opt.num_mean_samples, mean_size=10, device="cuda:0", n_classes=opt.n_classes if conditional else 1,
                         res=img_shape[-1], ch=img_shape[0])
        g = torch.Generator().manual_seed(9)
        ms.mean_samples = (torch.randn((ms.n_classes, opt.num_mean_samples) + img_shape, generator=g) * 0.2).cuda()
    tr = Trainer(opt, G, D, mean_sampler=ms, log_to=str(out / "log.csv"))
    pe = tr.setup_privacy_engine()
    step = GraphedDStep(tr, use_graph=use_graph, warmup=2)
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(123)
    torch.cuda.manual_seed(123)
    lr = tr.d_optimizer.param_groups[0]["lr"]
    grads = []
    for i in range(n_steps):
        tr.d_optimizer.param_groups[0]["lr"] = 0.0 if i < 2 else lr
        img = (torch.rand((B,) + img_shape, generator=g) * 2 - 1).cuda()
        lab = torch.randint(0, opt.n_classes, (B,), generator=g).cuda() if conditional else None
        step(img, lab)
        if i in (2, 3):         # the recorded step's first replay, and the replay after one Adam update
            torch.cuda.synchronize()
            grads.append([p.grad.detach().cpu().clone() for p in D.parameters()])
    torch.cuda.synchronize()
    tr.flush_stats()
    return ([p.detach().cpu().clone() for p in D.parameters()], pe.steps, pe._noise_calls, int(pe._noise_ctr.item()),
            [st["step"] for st in tr.d_optimizer.state.values()], step.graph is not None, dict(tr.logger.stats), grads)


@pytest.mark.parametrize("name,argv,B,shape,cond,smooth", [
    # BASELINE configs[1]: MNIST conditional vanilla GAN, dp_mode=gc, sigma=10 (bs=600 in the benchmark; 64 here)
    ("mnist_cond", ["MNIST", "--model", "Vanilla", "--conditional", "-dpm", "gc", "--sigma", "10"], 64, (1, 28, 28), True, True),
    ("mnist_cond_all", ["MNIST", "--model", "Vanilla", "--conditional", "-dpm", "gc", "--sigma", "10", "--materialize", "all"], 64, (1, 28, 28), True, False),
    # BASELINE configs[2] (headline): adaptive per-layer clipping, ghost + fused passes, WGAN-GP on mean samples
    ("celeba_smooth", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "8", "--sigma", "0.5"], 8, (3, 64, 64), False, True),
    ("celeba", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "8", "--sigma", "0.5"], 8, (3, 64, 64), False, False),
    # larger batches: image-sized tensors leave the allocator's small-block pool at B >= 24, the full benchmark size is 128
    ("celeba_smooth_b32", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "8", "--sigma", "0.5"], 32, (3, 64, 64), False, True),
    ("celeba_smooth_b128", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "32", "--sigma", "0.5"], 128, (3, 64, 64), False, True),
    # BASELINE configs[3]: immediate sensitivity (one value, and one per parameter tensor).  B = 32: the batch whose replays were
    # wrong while the library zeroed its norm accumulators with hipMemsetAsync (csrc/common.h zero_floats)
    ("celeba_is_smooth_b32", ["CelebA", "-dpm", "is", "-nms", "8", "--sigma", "0.5"], 32, (3, 64, 64), False, True),
    ("celeba_ispp_smooth_b32", ["CelebA", "-dpm", "is", "-ispp", "True", "-nms", "8", "--sigma", "0.5"], 32, (3, 64, 64), False, True),
    ("celeba_ispp_b32", ["CelebA", "-dpm", "is", "-ispp", "True", "-nms", "8", "--sigma", "0.5"], 32, (3, 64, 64), False, False),
    ("mnist_is_cond", ["MNIST", "--model", "Vanilla", "--conditional", "-dpm", "is", "--sigma", "1"], 64, (1, 28, 28), True, True),
    # BASELINE configs[3] at its FULL per-GPU size (bs = 128, -ispp True, 32 mean samples): nine double-backward sweeps per step
    ("celeba_ispp_smooth_b128", ["CelebA", "-dpm", "is", "-ispp", "True", "-nms", "32", "--sigma", "0.5"], 128, (3, 64, 64), False, True),
    # the headline arithmetic (fp32 from three bfloat16 pieces wherever a launch is large enough) at the headline size
    ("celeba_auto_smooth_b128", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "32", "--sigma", "0.5", "--compute_dtype", "fp32_auto"],
     128, (3, 64, 64), False, True),
])
def test_graph_replay_equals_eager_steps(tmp_path, name, argv, B, shape, cond, smooth):
    n = 6
    eager = _run(tmp_path, name + "_eager", argv, B, False, n, shape, cond, smooth)
    graph = _run(tmp_path, name + "_graph", argv, B, True, n, shape, cond, smooth)
    assert graph[5] and not eager[5], "the graph was not recorded"
    assert eager[1:5] == graph[1:5] == (n, n, n, [n] * len(eager[4])), (eager[1:5], graph[1:5])
    # PRIMARY: the noised gradients of the recorded step's first replay — identical weights, inputs, RNG draws and Philox counter on
    # both sides, BEFORE Adam: only float atomics reorder.  Smooth networks: 1e-5 of each tensor's scale.  With the activations on,
    # reordering can flip a LeakyReLU unit that sits within rounding of zero (measured on the CelebA case: 5.7e-4 on conv1's
    # gradient, 2.8e-3 on its cancelling bias gradient, eager against eager alike): 1e-2 there.
    tol1 = 3e-5 if smooth else 1e-2          # 1.03e-5 / 1.05e-5 measured at B = 32 / 128 on a bias gradient that is rounding residue (atomic order)
    if not smooth and "is" in argv:
        tol1 = 3e-2             # the noise scale is itself a max over samples of a double-backward norm: 1.3e-2 measured (unit flips)
    # tensors whose gradient is analytically zero (a conv bias in front of a GroupNorm; is mode adds almost no noise to them, so
    # what is left is rounding residue of terms of the step's overall gradient size) are held to 1e-2 of that overall size
    floor = 1e-2 * max(a.abs().max().item() for a in eager[7][0])
    for i, (a, b) in enumerate(zip(eager[7][0], graph[7][0])):
        scale = max(a.abs().max().item(), floor) + 1e-30
        assert (a - b).abs().max().item() <= tol1 * scale, "noised gradient %d of the first replayed step: rel %.3e" % (
            i, (a - b).abs().max().item() / scale)
    # the next replay runs on weights one Adam step later (every entry moved by ~lr * sign(g)): the gradients still agree closely,
    # which they would not if the replay had kept the capture-time weights, Adam step count or noise offset
    for i, (a, b) in enumerate(zip(eager[7][1], graph[7][1])):
        scale = max(a.abs().max().item(), floor) + 1e-30
        assert (a - b).abs().max().item() <= (1e-3 if smooth else max(1e-2, tol1)) * scale, "noised gradient %d of the second replayed step: rel %.3e" % (
            i, (a - b).abs().max().item() / scale)
    assert any((a - b).abs().max().item() > 1e-3 * a.abs().max().item() for a, b in zip(eager[7][0], eager[7][1])), "steps 3 and 4 must differ"
    lr = 1e-4 if name.startswith("celeba") else 2e-4       # d_lr defaults (options.py)
    for i, (a, b) in enumerate(zip(eager[0], graph[0])):
        # SECONDARY (post-Adam weights after all steps): Adam normalises every entry to a step of ~lr whatever the gradient's size, so
        # atomic reordering shows as a few per cent of lr on cancelling sums and whole steps where the gradient is ~0
        err = (a - b).abs()
        assert err.max().item() <= 2.1 * lr * n, "parameter %d differs between eager and replayed steps: %.3e" % (i, err.max().item())
        if eager[7][0][i].abs().max().item() < floor:
            continue            # gradient of rounding residue only: Adam turns its sign pattern into steps of +-lr
        assert err.mean().item() <= 0.1 * lr, "parameter %d: mean difference %.3e lr" % (i, err.mean().item() / lr)
    for k, v in eager[6].items():
        w = graph[6][k]
        if "Acc" in k:          # a COUNT of critic outputs on one side of zero (100 / B per sample and step): one borderline sample may differ
            assert abs(float(v) - float(w)) <= 2 * 100.0 / B + 1e-9, (k, v, w)
            continue
        assert torch.allclose(torch.as_tensor(v, dtype=torch.float64), torch.as_tensor(w, dtype=torch.float64),
                              rtol=2e-3 if smooth or "is" not in argv else 1e-2, atol=5e-4 if "is" in argv else 1e-4), (k, v, w)
        # is mode: 2.6e-3 measured on the sensitivity maxima with the activations on; 1.7e-4 absolute on a mean critic output of
        # -0.033 (the zero-gradient biases above take +-lr Adam steps whose signs are rounding residue, on either side)


@pytest.mark.parametrize("name,argv,B,shape,cond,latent", [
    ("mnist_nd1", ["MNIST", "--model", "Vanilla", "-dpm", "gc", "--sigma", "10"], 32, (1, 28, 28), False, 100),      # n_d_steps = 1
    ("celeba_nd2", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "8", "--sigma", "0.5", "--n_d_steps", "2"], 8, (3, 64, 64), False, 128),
    ("celeba_auto", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "8", "--sigma", "0.5", "--n_d_steps", "2", "--compute_dtype",
                     "fp32_auto"], 8, (3, 64, 64), False, 128),
    # bf16 storage (BASELINE configs[4]): the bf16 filter copies of G and D are re-rounded after every Adam step, train_G differentiates
    # through the critic's bf16-stored activations; tolerances at the bf16 level (tests/test_bf16s_gpu.py states the error model)
    ("celeba_bf16s", ["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "8", "--sigma", "0.5", "--n_d_steps", "2", "--compute_dtype",
                      "bf16", "--storage_dtype", "bf16"], 8, (3, 64, 64), False, 128),
])
def test_graph_replay_interleaved_with_generator_steps(tmp_path, name, argv, B, shape, cond, latent):
    """train() interleaves eager generator steps with replayed D-steps.  After such a sequence (a) the generator
    INSIDE the graph is the current one (its folded / pre-split filters are re-packed by the replay, not frozen at capture), (b) eager
    code sees the critic the replays left (re-packed data-gradient / stride-2 filters follow the replayed Adam updates), (c) nothing
    reads a freed re-pack buffer.  Checked against torch on the CPU with the device's current weights."""
    from csl_gan_amd import init_util, options
    from csl_gan_amd.mean_sampler import MeanSampler
    from csl_gan_amd.trainer import GraphedDStep, Trainer
    from oracle.nets import build_models
    # learning rates 20x the defaults: after eight steps the weights have moved by about their own size, so a replay or an eager
    # launch that used capture-time filters would be off by O(1), far above the activation-flip noise of a gradient comparison
    opt = options.parse(argv + ["-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", str(tmp_path), "--manual_seed", "1", "--d_lr", "0.002",
                                "--g_lr", "0.002"])
    G, D = init_util.init_models(opt)
    ms = None
    if opt.num_mean_samples > 0:
        ms = MeanSampler(num_samples=opt.num_mean_samples, mean_size=10, device="cuda:0", res=shape[-1], ch=shape[0])
        ms.mean_samples = (torch.randn((1, opt.num_mean_samples) + shape, generator=torch.Generator().manual_seed(9)) * 0.2).cuda()
    tr = Trainer(opt, G, D, mean_sampler=ms, log_to=str(tmp_path / "log.csv"))
    tr.setup_privacy_engine()
    step = GraphedDStep(tr, use_graph=True, warmup=1)
    g = torch.Generator().manual_seed(5)
    torch.manual_seed(3); torch.cuda.manual_seed(3)
    tr.train_G(tr.gen_z(B), None)              # an eager generator step BEFORE the capture: caches D's re-packs at the current version
    for i in range(7):
        step((torch.rand((B,) + shape, generator=g) * 2 - 1).cuda(), None)
        if (i + 1) % opt.n_d_steps == 0:
            tr.train_G(tr.gen_z(B), None)
    step((torch.rand((B,) + shape, generator=g) * 2 - 1).cuda(), None)         # a replay right after a generator step
    torch.cuda.synchronize()
    assert step.graph is not None
    Go, Do = build_models(dataset=opt.dataset, model=opt.model, im_size=opt.im_size, weights_seed=opt.weights_seed, manual_seed=1,
                          per_sample_grad=True, g_latent_dim=latent)
    Go.load_state_dict({k: v.detach().cpu() for k, v in G.state_dict().items()})
    # (a) the fake batch the LAST replay produced = the current generator on the z it was given (D moved since, G did not)
    z = step.bufs["z"].detach().cpu()
    with torch.no_grad():
        want = Go(z)
    got = tr.last["fake_img"].detach().cpu()
    tol = 4e-2 if name.endswith("bf16s") else 1e-3
    assert (got - want).abs().max().item() <= tol * want.abs().max().item(), "the replay ran a stale generator: %.3e" % (got - want).abs().max().item()
    # (b) eager forward + data gradient of the critic after the replays
    Do.load_state_dict({k: v.detach().cpu() for k, v in D.state_dict().items()})
    x = (torch.rand((B,) + shape, generator=g) * 2 - 1)
    xo = x.clone().requires_grad_(True)
    oo, _ = Do(xo)
    go, = torch.autograd.grad(oo.sum(), xo)
    xd = x.cuda().requires_grad_(True)
    od, _ = D(xd)
    gd, = torch.autograd.grad(od.sum(), xd)
    assert (od.detach().cpu() - oo.detach()).abs().max().item() <= tol * oo.detach().abs().max().item(), "eager D forward uses stale re-packed filters"
    # a gradient through LeakyReLU units: a unit within rounding of zero may take the other slope on the device and moves that sample's
    # gradient by a per cent or two — relative L2 (measured 1e-3 .. 2e-2 of max |g| per entry in the flip case), decisive against O(1)
    l2 = ((gd.cpu() - go).norm() / go.norm()).item()
    assert l2 <= (1.5e-1 if name.endswith("bf16s") else 3e-2), "eager D data gradient uses stale re-packed filters: relative L2 error %.3e" % l2


def test_capturable_adam_checkpoint_roundtrip():
    """--hip_graph + resume: the device step counter is not part of optimizer.state (torch would cast it to the
    parameter's dtype on load and the Adam layout would differ from exp_avg / exp_avg_sq / step); a capturable optimizer loaded from a
    checkpoint continues exactly like the one that wrote it, and so does a multi-tensor step against torch.optim.Adam."""
    from csl_gan_amd.engine import HipAdam
    g = torch.Generator().manual_seed(0)
    shapes = [(64, 3, 5, 5), (64,), (10, 128), (7,)]           # the last two are not 16-byte multiples: the scalar path
    ws = [torch.randn(s, generator=g) for s in shapes]
    grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(4)]

    def make(capturable):
        ps = [torch.nn.Parameter(w.clone().cuda()) for w in ws]
        o = HipAdam(ps, lr=1e-2, betas=(0.5, 0.9), weight_decay=0.01)
        o.capturable = capturable
        return ps, o

    def run(ps, o, steps):
        for k in steps:
            for p, gr in zip(ps, grads[k]):
                p.grad = gr.cuda()
            o.step()

    pa, oa = make(True)
    run(pa, oa, [0, 1])
    import copy
    sd = copy.deepcopy(oa.state_dict())          # load_state_dict keeps tensors that need no cast: without the copy both optimizers share moments
    assert all(set(st) == {"step", "exp_avg", "exp_avg_sq"} for st in sd["state"].values()), [set(st) for st in sd["state"].values()]
    pb, ob = make(True)
    with torch.no_grad():
        for p, q in zip(pb, pa):
            p.copy_(q)
    ob.load_state_dict(sd)
    run(pa, oa, [2, 3])
    run(pb, ob, [2, 3])
    for p, q in zip(pa, pb):
        assert torch.equal(p.detach().cpu(), q.detach().cpu())
    # and the multi-tensor kernel against torch.optim.Adam on the CPU
    pc = [torch.nn.Parameter(w.clone()) for w in ws]
    oc = torch.optim.Adam(pc, lr=1e-2, betas=(0.5, 0.9), weight_decay=0.01)
    for k in range(4):
        for p, gr in zip(pc, grads[k]):
            p.grad = gr.clone()
        oc.step()
    for p, q in zip(pa, pc):
        assert (p.detach().cpu() - q.detach()).abs().max().item() <= 1e-5 * q.detach().abs().max().item()
