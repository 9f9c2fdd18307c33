#!/usr/bin/env python3
"""Which ATen operators the RECORDED D-step still launches, by call site: one step of bench.py's GraphedDStep run eagerly (the exact
python a capture records) under a TorchDispatchMode; views, allocations and metadata ops dropped.  The hunt list for the launch tail
(DESIGN §4.16).  usage (GPU box): python scripts/graph_step_ops.py [out.txt] [--opt "..."]"""
import collections
import contextlib
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from csl_gan_amd.trainer import GraphedDStep  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "gpurun_out/graph_step_ops.txt"
extra = sys.argv[sys.argv.index("--opt") + 1].split() if "--opt" in sys.argv else []
with contextlib.redirect_stdout(sys.stderr):
    opt, tr, img = bench.build_trainer(0, 1, 0, extra=extra)
gs = GraphedDStep(tr, use_graph=False)
for _ in range(3):
    gs(img)
    tr.dev_stats.clear()
torch.cuda.synchronize()

NO_KERNEL = ("view", "as_strided", "slice", "select", "permute", "expand", "detach", "alias", "transpose", "unsqueeze", "squeeze", "split",
             "empty", "t.default", "reshape", "_local_scalar_dense", "is_", "size", "stride", "numel", "unbind", "narrow", "_to_copy.default_meta",
             "lift_fresh", "resize_", "set_", "record_stream", "unfold", "chunk", "contiguous", "_reshape_alias", "is_pinned", "_has_compatible")
sites = collections.Counter()
phase = ["fill"]


class Tap(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(k in name for k in NO_KERNEL):
            fr = next((f for f in reversed(traceback.extract_stack(limit=16)) if "csl_gan_amd" in f.filename and "graph_step_ops" not in f.filename), None)
            site = "%s:%d" % (os.path.basename(fr.filename), fr.lineno) if fr is not None else "(autograd engine)"
            shp = next((tuple(a.shape) for a in args if torch.is_tensor(a)), ())
            sites[(phase[0], name, site, shp)] += 1
        return func(*args, **(kwargs or {}))


orig_eager = gs._eager


def eager_marked():
    phase[0] = "recorded"
    try:
        return orig_eager()
    finally:
        phase[0] = "after"


gs._eager = eager_marked
with Tap():
    gs(img)
torch.cuda.synchronize()
with open(out, "w") as f:
    for ph in ("fill", "recorded", "after"):
        rows = [(k, n) for k, n in sites.items() if k[0] == ph]
        f.write("# ---- %s: %d ATen operator calls that launch device work ----\n" % (ph, sum(n for _, n in rows)))
        for (p, name, site, shp), n in sorted(rows, key=lambda kv: (kv[0][2], kv[0][1])):
            f.write("%3d x  %-42s %-28s %s\n" % (n, name[:42], site, shp))
print("wrote", out)
