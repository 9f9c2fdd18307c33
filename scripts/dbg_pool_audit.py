"""Audit of one captured D-step: every device pointer handed to a C-ABI launch during the capture must lie in the graph's private pool
or in memory that was allocated before the capture and is still alive after it.  usage: python scripts/dbg_pool_audit.py gc|is [B]"""
import sys, os, torch, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csl_gan_amd import options, init_util, ops
from csl_gan_amd.trainer import Trainer, GraphedDStep
from csl_gan_amd.mean_sampler import MeanSampler


def audit(mode, B):
    extra = ["-gcm", "adaptive-pl"] if mode == "gc" else []
    opt = options.parse(["CelebA", "-tss", "1000", "-dpm", mode, "-nms", "4", "--mean_sample_size", "10", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0",
                         "-o", tempfile.mkdtemp(), "--synthetic", "--manual_seed", "7", "--hip_graph", "False"] + extra)
    G, D = init_util.init_models(opt)
    ms = MeanSampler(num_samples=4, mean_size=10, device="cuda:0", res=64, ch=3)
    ms.mean_samples = (torch.randn(1, 4, 3, 64, 64) * 0.2).cuda()
    tr = Trainer(opt, G, D, mean_sampler=ms, log_to=opt.output_dir + "/log.csv")
    tr.setup_privacy_engine()
    gd = GraphedDStep(tr, use_graph=True, warmup=2)
    g = torch.Generator().manual_seed(1)
    log, orig_p = [], ops._p

    def rec_p(t):
        if t is not None and torch.cuda.is_current_stream_capturing():
            log.append((t.data_ptr(), t.numel() * t.element_size()))
        return orig_p(t)
    from torch.utils._python_dispatch import TorchDispatchMode
    from torch.utils._pytree import tree_flatten
    aten_log = []

    class Rec(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            if torch.cuda.is_current_stream_capturing():
                for t in tree_flatten((args, kwargs or {}, out))[0]:
                    if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 0:
                        aten_log.append((t.data_ptr(), t.numel() * t.element_size(), str(func)))
            return out
    pre = None
    for k in range(3):
        if k == 2:
            torch.cuda.synchronize()
            ops._p = rec_p
            pre = [(b_["address"], b_["address"] + b_["size"]) for s in torch.cuda.memory_snapshot() for b_ in s["blocks"] if b_["state"] == "active_allocated"]
        x = (torch.rand(B, 3, 64, 64, generator=g) * 2 - 1).cuda()
        if k == 2 and os.environ.get("AUDIT_ATEN", "1") == "1":
            with Rec():
                gd(x, None)
        else:
            gd(x, None)
    torch.cuda.synchronize()
    ops._p = orig_p
    snap = torch.cuda.memory_snapshot()
    segs = [(s["address"], s["address"] + s["total_size"], tuple(s.get("segment_pool_id", (0, 0)))) for s in snap]
    post = [(b_["address"], b_["address"] + b_["size"]) for s in snap for b_ in s["blocks"] if b_["state"] == "active_allocated"]
    inside = lambda ptr, rs: any(a <= ptr < b for a, b in rs)
    stray = []
    for ptr, nbytes in log:
        pool = next((pid for a, b, pid in segs if a <= ptr < b), None)
        if pool == (0, 0) and not (inside(ptr, pre) and inside(ptr, post)):
            stray.append((ptr, nbytes))
    stray_aten = {}
    for ptr, nbytes, fn in aten_log:
        pool = next((pid for a, b, pid in segs if a <= ptr < b), None)
        if pool == (0, 0) and not (inside(ptr, pre) and inside(ptr, post)):
            stray_aten.setdefault(fn, [0, 0])
            stray_aten[fn][0] += 1
            stray_aten[fn][1] = max(stray_aten[fn][1], nbytes)
    print("aten tensors seen during capture:", len(aten_log), "stray by op:", stray_aten)
    return len(log), stray


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "gc"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    n, stray = audit(mode, B)
    print("%s B=%d: %d pointers logged during capture, %d outside the graph pool and not persistent (%.1f MB)" % (
        mode, B, n, len(stray), sum(b for _, b in stray) / 1e6))
