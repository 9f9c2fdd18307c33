import sys, os, torch, tempfile
sys.path.insert(0, "/root/repo")
from csl_gan_amd import options, init_util, ops
from csl_gan_amd.trainer import Trainer
B = int(os.environ.get("DBG_B", "32"))
def build():
    opt = options.parse(["CelebA", "-tss", "1000", "-dpm", "is", "-nms", "1", "--mean_sample_size", "10", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0",
                         "-o", tempfile.mkdtemp(), "--synthetic", "--manual_seed", "7", "--penalty", "--hip_graph", "False"])
    G, D = init_util.init_models(opt)
    fixed = torch.tanh(torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(3))).cuda()
    if os.environ.get("DBG_CL", "1") == "1":
        fixed = fixed.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    if not os.environ.get("DBG_TRAING"):
        G.forward = lambda z, y=None: fixed
    tr = Trainer(opt, G, D, log_to=opt.output_dir + "/log.csv")
    pe = tr.setup_privacy_engine()
    pe.noise_multiplier = 0.0
    if os.environ.get("DBG_NOLOG", "1") == "1":
        tr.update_is_logging = lambda: None
        ops.dstep_stats = lambda *a, **k: None
    return tr, pe, D
g = torch.Generator().manual_seed(1)
imgs = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda() for _ in range(6)]
def run(use_graph):
    tr, pe, D = build()
    static = imgs[0].clone()
    z = torch.zeros(B, 128, device="cuda")
    res, gr = [], None
    def f():
        tr.train_D(static, None, z, None, use_dp=True)
    gd = None
    if os.environ.get("DBG_GD") and use_graph:
        from csl_gan_amd.trainer import GraphedDStep
        gd = GraphedDStep(tr, use_graph=True, warmup=2)
    for k in range(6):
        if gd is not None:
            if k == 2:
                pe._dbg_rows = []
            src = imgs[k].cpu().cuda() if os.environ.get("DBG_FRESH") else imgs[k]          # a fresh image-sized eager allocation per step
            gd(src, None)
            del src
            if k == 0 and os.environ.get("DBG_TRAING"):
                tr.train_G(torch.zeros(B, 128, device="cuda"), None)
            torch.cuda.synchronize()
            res.append(torch.cat([pe._sens_last.detach().cpu()[::2], torch.zeros(4)]))
            if k >= 2 and use_graph:
                rows = [r.detach().cpu() for r in pe._dbg_rows[:5]]
                for i, r in enumerate(rows):
                    badi = [j for j in range(B) if not (r[j] < 1e3)]
                    if badi:
                        print("  step", k, "sweep", i, "bad rows", badi, "vals", ["%.2g" % r[j] for j in badi[:6]])
            continue
        with torch.no_grad():
            static.copy_(imgs[k])
        if not use_graph or k < 2:
            f()
            if k == 0 and os.environ.get("DBG_TRAING"):
                tr.train_G(torch.zeros(B, 128, device="cuda"), None)
        elif gr is None:
            tr.d_optimizer.prepare_capture(); pe.ensure_noise_counter(); ops.repack_cache.clear()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                f()
            ops.repack_cache.clear()
            gr.replay()
        else:
            gr.replay()
        if gr is not None and os.environ.get("DBG_BUMP"):
            tr.d_optimizer.bump_versions()
        torch.cuda.synchronize()
        res.append(torch.cat([pe._sens_last.detach().cpu()[::2], torch.tensor([sum(p.detach().abs().sum().item() for p in D.parameters()) - 30600.0,
                                                                        float(tr.last["d_real_loss"]), float(tr.last["d_fake_loss"]), tr.last["fake_img"].abs().sum().item() / 1e4])]))
    return res
e, gph = run(False), run(True)
for k, (a, b) in enumerate(zip(e, gph)):
    print(k, "eager", " ".join("%.4g" % v for v in a.tolist()), "| graph", " ".join("%.4g" % v for v in b.tolist()))
