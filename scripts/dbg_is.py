import sys, os, torch
sys.path.insert(0, "/root/repo")
from csl_gan_amd import train as T, options, init_util
from csl_gan_amd.trainer import Trainer, GraphedDStep
from csl_gan_amd.mean_sampler import MeanSampler
import tempfile
out = tempfile.mkdtemp()
B = int(os.environ.get('DBG_B', '32'))
opt = options.parse(["CelebA", "-tss", "1000", "-dpm", "is", "-nms", "1", "--mean_sample_size", "10", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", out, "--synthetic", "--manual_seed", "7"] + sys.argv[1:])
G, D = init_util.init_models(opt)
ms = MeanSampler(num_samples=1, mean_size=10, device="cuda:0", res=64, ch=3)
ms.mean_samples = (torch.randn(1, 1, 3, 64, 64) * 0.2).cuda()
tr = Trainer(opt, G, D, mean_sampler=ms, log_to=out + "/log.csv")
tr.setup_privacy_engine()
if os.environ.get("DBG_GRAPH", "1") == "0":
    tr.graphed = GraphedDStep(tr, use_graph=False)
if os.environ.get("DBG_SIGMA0"):
    tr.privacy_engine.noise_multiplier = 0.0
from csl_gan_amd import ops as _ops
if os.environ.get("DBG_NOLOG"):
    tr.update_is_logging = lambda: None
if os.environ.get("DBG_NOSTATS"):
    _ops.dstep_stats = lambda *a, **k: None
if os.environ.get("DBG_GEVAL"):
    G.eval(); G.train = lambda *a, **k: G
if os.environ.get("DBG_NOFAKE"):
    import types
    fixed = torch.tanh(torch.randn(B, 3, 64, 64)).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    G.forward = lambda z, y=None: fixed
print("graphed", tr.graphed is not None, "n_d_steps", opt.n_d_steps, "threshold", opt.train_d_until_threshold)
g = torch.Generator().manual_seed(1)
def bad():
    return [n for n, p in list(D.named_parameters()) + list(G.named_parameters()) if not torch.isfinite(p).all()]
for it in range(6):
    img = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1)
    lab = torch.zeros(B, dtype=torch.long)
    if os.environ.get("DBG_DIRECT"):
        tr.graphed(img.cuda(), None)
        if it == 0 and os.environ.get("DBG_DIRECT") == "2":
            tr.train_G(tr.gen_z(B), None)
    else:
        tr.train(0, it, img, lab, use_dp=True)
    torch.cuda.synchronize()
    import numpy as np
    tr.privacy_engine._sens_host = None
    print(it, "graph" if (tr.graphed is not None and tr.graphed.graph is not None) else "eager", "sens", " ".join("%.3g" % v for v in np.atleast_1d(tr.privacy_engine.batch_sensitivity)),
          
          "pen %.5f" % float(tr.last["penalty"]), "dreal %.6f" % float(tr.last["d_real_loss"]), "|w| %.6f" % sum(p.detach().abs().sum().item() for p in D.parameters()),
          "|g| %.6e" % sum(p.grad.abs().sum().item() for p in D.parameters()))
