import sys, os, torch, tempfile
sys.path.insert(0, "/root/repo")
from csl_gan_amd import options, init_util
from csl_gan_amd.is_engine import ISPrivacyEngine
B = int(os.environ.get("DBG_B", "32"))
opt = options.parse(["CelebA", "-dpm", "is", "-nms", "1", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", tempfile.mkdtemp(), "--synthetic", "--manual_seed", "7"])
_, D = init_util.init_models(opt, init_G=False)
pe = ISPrivacyEngine(D, batch_size=B, sample_size=1000, alphas=[2.0], noise_multiplier=0.5, per_param=True)
g = torch.Generator().manual_seed(1)
imgs = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda() for _ in range(4)]
static = imgs[0].clone().requires_grad_(True)
MODE = os.environ.get("DBG_MODE", "full")
def f():
    out, _ = D(static)
    loss = D.real_loss(out, "cuda:0")
    if MODE == "full":
        pe.backward(loss, static)
        return pe._sens_last.clone(), [p.grad.clone() for p in D.parameters()]
    ps = list(D.parameters())
    grads = torch.autograd.grad(loss, ps, create_graph=True, allow_unused=True)
    if MODE == "grads":
        return torch.zeros(1, device="cuda"), [g_.detach().clone() for g_ in grads]
    from csl_gan_amd import functional as HF
    i = int(MODE)          # one sweep: parameter i
    n = HF.RowL2Norm.apply(grads[i].reshape(1, -1))[0]
    gx, = torch.autograd.grad(n, static, retain_graph=True)
    return gx.detach().reshape(B, -1).norm(dim=1).max().reshape(1).clone(), [gx.detach().clone()]
for _ in range(2):
    f()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    res = f()
for k, im in enumerate(imgs):
    with torch.no_grad():
        static.copy_(im)
    gr.replay(); torch.cuda.synchronize()
    got_s, got_g = res[0].clone(), [t.clone() for t in res[1]]
    want_s, want_g = f()
    torch.cuda.synchronize()
    errs = [((a - b).abs().max() / (b.abs().max() + 1e-30)).item() for a, b in zip(got_g, want_g)]
    print("replay", k, "sens err %.2e" % ((got_s - want_s).abs().max() / (want_s.abs().max() + 1e-30)).item(), "grad errs", ["%.1e" % e for e in errs])
