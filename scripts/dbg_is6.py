"""Which op of a captured immediate-sensitivity step first produces garbage on a bad replay: every csl_gan_amd.ops call made
during the capture has its outputs kept alive, and after each replay they are scanned in launch order."""
import sys, os, torch, tempfile, types
sys.path.insert(0, "/root/repo")
from csl_gan_amd import options, init_util, ops
from csl_gan_amd.trainer import Trainer, GraphedDStep
B = int(os.environ.get("DBG_B", "32"))
LOG = []
def wrap(name, fn):
    def w(*a, **k):
        r = fn(*a, **k)
        if torch.cuda.is_current_stream_capturing():
            outs = r if isinstance(r, (tuple, list)) else (r,)
            ins = [t for t in a if torch.is_tensor(t)]
            LOG.append((name, [t for t in outs if torch.is_tensor(t)], [tuple(t.shape) for t in ins], ins if name == "row_l2norm" else None))
        return r
    return w
for n, f in list(vars(ops).items()):
    if isinstance(f, types.FunctionType) and not n.startswith("_") and f.__module__ == ops.__name__:
        setattr(ops, n, wrap(n, f))
opt = options.parse(["CelebA", "-tss", "1000", "-dpm", "is", "-nms", "1", "--mean_sample_size", "10", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0",
                     "-o", tempfile.mkdtemp(), "--synthetic", "--manual_seed", "7", "--penalty", "--hip_graph", "False"])
G, D = init_util.init_models(opt)
fixed = torch.tanh(torch.randn(B, 3, 64, 64, generator=torch.Generator().manual_seed(3))).cuda()
fixed = fixed.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
G.forward = lambda z, y=None: fixed
tr = Trainer(opt, G, D, log_to=opt.output_dir + "/log.csv")
pe = tr.setup_privacy_engine()
pe.noise_multiplier = 0.0
g = torch.Generator().manual_seed(1)
imgs = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda() for _ in range(6)]
gd = GraphedDStep(tr, use_graph=True, warmup=2)
for k in range(6):
    src = imgs[k].cpu().cuda() if os.environ.get("DBG_FRESH", "1") == "1" else imgs[k]
    gd(src, None)
    del src
    torch.cuda.synchronize()
    print("step", k, "sens", " ".join("%.3g" % v for v in pe._sens_last.detach().cpu()[:6].tolist()), "ops logged", len(LOG), flush=True)
    if k < 2:
        continue
    shown = 0
    for i, (name, outs, inshapes, kept) in enumerate(LOG):
        for t in outs:
            if not t.is_floating_point() or t.numel() == 0:
                continue
            f = t.detach().float()
            bad = ~(f.abs() < 1e6)
            if bad.any():
                lead = bad.reshape(bad.shape[0], -1).any(1).nonzero().flatten().tolist() if bad.dim() > 1 else []
                print("   op #%d %s out %s ins %s: %d bad of %d, first-dim rows %s, max %.3g" % (
                    i, name, tuple(t.shape), inshapes, int(bad.sum()), bad.numel(), lead[:12], float(torch.nan_to_num(f, 0, 0, 0).abs().max())))
                shown += 1
                if kept:
                    ref = kept[0].detach().double().norm(2, dim=1).float()
                    print("      input rows finite:", bool(torch.isfinite(kept[0]).all()), "input absmax %.3g" % float(kept[0].abs().max()),
                          "torch norms of the kept input at bad rows", ["%.3g" % v for v in ref[bad].tolist()[:4]], "kernel", ["%.3g" % v for v in f[bad].tolist()[:4]],
                          "ptr %x out ptr %x" % (kept[0].data_ptr(), t.data_ptr()))
        if shown >= 6:
            break
