import sys, os, torch, tempfile, traceback
sys.path.insert(0, "/root/repo")
from csl_gan_amd import options, init_util, ops
from csl_gan_amd.trainer import Trainer, GraphedDStep
B = 32
opt = options.parse(["CelebA", "-tss", "1000", "-dpm", "is", "-nms", "1", "--mean_sample_size", "10", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0",
                     "-o", tempfile.mkdtemp(), "--synthetic", "--manual_seed", "7", "--penalty", "--hip_graph", "False"])
G, D = init_util.init_models(opt)
fixed = torch.tanh(torch.randn(B, 3, 64, 64)).cuda().permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
G.forward = lambda z, y=None: fixed
tr = Trainer(opt, G, D, log_to=opt.output_dir + "/log.csv")
pe = tr.setup_privacy_engine(); pe.noise_multiplier = 0.0
gd = GraphedDStep(tr, use_graph=True, warmup=2)
g = torch.Generator().manual_seed(1)
imgs = [(torch.rand(B, 3, 64, 64, generator=g) * 2 - 1) for _ in range(4)]
log = []
orig_p = ops._p
def rec_p(t):
    if t is not None and gd.graph is None and gd.warmup == 0:
        fr = traceback.extract_stack(limit=4)
        log.append((t.data_ptr(), t.numel() * t.element_size(), "%s:%d < %s:%d" % (os.path.basename(fr[-2].filename), fr[-2].lineno, os.path.basename(fr[-3].filename), fr[-3].lineno)))
    return orig_p(t)
ops._p = rec_p
for k in range(3):
    gd(imgs[k].cuda(), None)
torch.cuda.synchronize()
ops._p = orig_p
snap = torch.cuda.memory_snapshot()
segs = sorted((s["address"], s["address"] + s["total_size"], s.get("segment_pool_id", None)) for s in snap)
def pool_of(ptr):
    for a, b, pid in segs:
        if a <= ptr < b:
            return pid
    return "?"
known = {p.data_ptr() for p in D.parameters()} | {t.data_ptr() for t in gd.bufs.values() if t is not None}
seen = set()
for ptr, nbytes, where in log:
    pid = pool_of(ptr)
    if tuple(pid) == (0, 0) and nbytes >= 100000 and (ptr, where) not in seen:
        seen.add((ptr, where))
        tag = "param" if ptr in {p.data_ptr() for p in D.parameters()} else ("static" if ptr in known else "")
        print("default-pool pointer %x  %8.2f MB  %s  %s" % (ptr, nbytes / 1e6, tag, where))
print("logged", len(log), "pointers; segments", len(segs))
