#!/usr/bin/env python3
"""One headline D-step under torch.profiler: every device kernel in launch order with its duration and the Python frame of
csl_gan_amd that launched it (for hunting small pointwise launches).  usage: python scripts/step_trace.py [out.txt] [--opt "..."]"""
import collections
import contextlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/step_trace.txt"
extra = sys.argv[sys.argv.index("--opt") + 1].split() if "--opt" in sys.argv else []
with contextlib.redirect_stdout(sys.stderr):
    opt, tr, img = bench.build_trainer(0, 1, 0, extra=extra)
B = img.shape[0]


def step():
    tr.train_D(img, None, tr.gen_z(B), None, use_dp=True)
    tr.dev_stats.clear()


for _ in range(4):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.events()
# kernel events carry no stack; the launching CPU op (same correlation id) does
by_corr = {}
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and getattr(e, "stack", None):
        by_corr.setdefault(e.id, e)
rows = []
for e in ev:
    if e.device_type != torch.autograd.DeviceType.CUDA:
        continue
    rows.append((e.time_range.start, e.name, e.time_range.elapsed_us()))
rows.sort()
# CPU-side ops with stacks, summarised by (op, first csl_gan_amd frame)
agg = collections.defaultdict(lambda: [0, 0.0])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.self_device_time_total > 0:
        frame = next((f for f in (e.stack or []) if "csl_gan_amd" in f or "bench.py" in f), "?")
        k = (e.name, frame.strip())
        agg[k][0] += 1
        agg[k][1] += e.self_device_time_total
with open(out, "w") as f:
    tot = sum(r[2] for r in rows)
    f.write("# %d kernels, %.3f ms of kernel time\n" % (len(rows), tot / 1e3))
    f.write("# ---- aten ops with device time, by launching frame ----\n")
    for (name, frame), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write("%8.1f us %4d x  %-40s %s\n" % (us, n, name[:40], frame))
    f.write("# ---- kernels in launch order ----\n")
    for t, name, us in rows:
        f.write("%8.1f us  %s\n" % (us, name[:150]))

# ---- second pass: which csl_gan_amd line issues which torch call (one step under a TorchFunctionMode) ----
import traceback  # noqa: E402
from torch.overrides import TorchFunctionMode, resolve_name  # noqa: E402

sites = collections.Counter()


class Tap(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        name = resolve_name(func) or getattr(func, "__name__", str(func))
        fr = next((f for f in reversed(traceback.extract_stack(limit=12)) if "csl_gan_amd" in f.filename or f.filename.endswith("bench.py")), None)
        if fr is not None:
            sites[(name, "%s:%d" % (os.path.basename(fr.filename), fr.lineno))] += 1
        return func(*args, **(kwargs or {}))


with Tap():
    step()
torch.cuda.synchronize()
skip = ("size", "shape", "dim", "is_cuda", "device", "dtype", "data_ptr", "numel", "stride", "is_contiguous", "requires_grad", "__get__",
        "_version", "grad.", "view", "reshape", "permute", "detach", "unsqueeze", "__getitem__", "grad_fn", "is_leaf", "storage_offset")
with open(out, "a") as f:
    f.write("# ---- torch calls of one step by call site (views / attribute reads dropped) ----\n")
    for (name, site), n in sorted(sites.items(), key=lambda kv: (-kv[1], kv[0])):
        if any(k in name for k in skip):
            continue
        f.write("%4d x  %-45s %s\n" % (n, name[:45], site))
print("wrote", out)
