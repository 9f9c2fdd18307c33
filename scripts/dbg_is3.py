import sys, os, torch, tempfile
sys.path.insert(0, "/root/repo")
from csl_gan_amd import options, init_util, ops
from csl_gan_amd.is_engine import ISPrivacyEngine
from csl_gan_amd.engine import HipAdam
B = int(os.environ.get("DBG_B", "32"))
PARTS = os.environ.get("DBG_PARTS", "real,adam").split(",")
def build():
    opt = options.parse(["CelebA", "-dpm", "is", "-nms", "1", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", tempfile.mkdtemp(), "--synthetic", "--manual_seed", "7"])
    _, D = init_util.init_models(opt, init_G=False)
    pe = ISPrivacyEngine(D, batch_size=B, sample_size=1000, alphas=[2.0], noise_multiplier=0.0, per_param=True)
    opt_ = HipAdam(D.parameters(), lr=1e-4, betas=(0.0, 0.9))
    pe.attach(opt_)
    return D, pe, opt_
g = torch.Generator().manual_seed(1)
imgs = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda() for _ in range(6)]
fakes = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda() for _ in range(6)]
def run(use_graph):
    D, pe, opt_ = build()
    static = imgs[0].clone().requires_grad_(True)
    fake = fakes[0].clone()
    def f():
        out, _ = D(static)
        loss = D.real_loss(out, "cuda:0")
        if "fake" in PARTS:
            loss = loss + D.fake_loss(D(fake)[0], "cuda:0")
        pe.backward(loss, static)
        s = pe._sens_last.clone()
        if "adam" in PARTS:
            opt_.step()
        return s
    res = []
    gr = None
    for k in range(6):
        with torch.no_grad():
            static.copy_(imgs[k]); fake.copy_(fakes[k])
        if not use_graph or k < 2:
            s = f()
        elif gr is None:
            opt_.prepare_capture(); pe.ensure_noise_counter(); ops.repack_cache.clear()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                sout = f()
            ops.repack_cache.clear()
            gr.replay(); opt_.bump_versions(); s = sout
        else:
            gr.replay(); opt_.bump_versions(); s = sout
            for st in opt_.state.values(): st["step"] += 1
        torch.cuda.synchronize()
        res.append(s.detach().cpu().clone())
    return res
e, gph = run(False), run(True)
for k, (a, b) in enumerate(zip(e, gph)):
    print(k, "eager", " ".join("%.3g" % v for v in a.tolist()[::2]), "| graph", " ".join("%.3g" % v for v in b.tolist()[::2]))
