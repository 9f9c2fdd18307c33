#!/usr/bin/env python3
"""Shadow check of the fp32_auto / bf16x3 arithmetic inside a real D-step: every ops.conv2d_fwd / conv2d_dgrad call of one golden-fixture
step is recomputed on the exact-fp32 kernels and compared.  usage (GPU box): python scripts/dbg_x3_shadow.py [case] [mode]"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from tests.golden.dstep_inputs import DSTEP_CASES, load_case
from csl_gan_amd import init_util, options, ops, _lib
from csl_gan_amd.trainer import Trainer

name = sys.argv[1] if len(sys.argv) > 1 else "dstep_celeba64_cond_acgan_b8"
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32_auto"
z, inp = load_case("tests/golden", name)
dataset, _, _, latent, _, extra = DSTEP_CASES[name]
B = int(z["meta"][0])
opt = options.parse([dataset, "-dpm", "gc", "-nms", "4", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0", "-o", "/tmp/dbg_x3", "--manual_seed", "1",
                     "--g_latent_dim", str(latent), "--sigma", "0.5", "--materialize", "ghost", "-as", repr(float(z["adaptive_scalar"])),
                     "--compute_dtype", mode, "-gcm", "adaptive-pl", "--hip_graph", "False"] + extra)
G, D = init_util.init_models(opt)
tr = Trainer(opt, G, D, log_to="/tmp/dbg_x3/log.csv")
tr.setup_privacy_engine().noise_multiplier = 0.0
orig_fwd, orig_dgrad = ops.conv2d_fwd, ops.conv2d_dgrad


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-300)).item()


def fwd(x, w, bias=None, **kw):
    y = orig_fwd(x, w, bias, **kw)
    kn = _lib.lib().cslgan_last_kernel().decode()
    if y.dtype == torch.float32 and kw.get("out") is None:
        with ops.compute_dtype("fp32"):
            y0 = orig_fwd(x, w, bias, **{k: v for k, v in kw.items() if k != "wkey"})
        e = rel(y, y0)
        print("fwd   x%s w%s s%d -> %-34s rel %.2e%s" % (tuple(x.shape), tuple(w.shape), kw.get("stride", 1), kn, e, "   <<<<" if e > 2e-5 else ""))
    return y


def dgrad(gy, w, in_hw, **kw):
    gx = orig_dgrad(gy, w, in_hw, **kw)
    kn = _lib.lib().cslgan_last_kernel().decode()
    if gx.dtype == torch.float32:
        with ops.compute_dtype("fp32"):
            g0 = orig_dgrad(gy, w, in_hw, **{k: v for k, v in kw.items() if k != "wkey"})
        e = rel(gx, g0)
        print("dgrad gy%s w%s s%d mask=%s -> %-34s rel %.2e%s" % (tuple(gy.shape), tuple(w.shape), kw.get("stride", 1), kw.get("mask") is not None, kn, e,
                                                                  "   <<<<" if e > 2e-5 else ""))
    return gx


ops.conv2d_fwd, ops.conv2d_dgrad = fwd, dgrad
cu = lambda t: None if t is None else t.cuda()
tr.explicit = dict(ms_adapt=inp["ms_adapt"], ms_adapt_labels=inp["ms_adapt_labels"], alpha=inp["alpha"], z_adapt=inp["z_adapt"].cuda(), keep=True)
if "penalty" in z.files:
    tr.explicit["pen_real"] = inp["ms_pen"]
tr.train_D(inp["img"].cuda(), cu(inp["labels"]), inp["z"].cuda(), cu(inp["y"]), use_dp=True)
torch.cuda.synchronize()
n = tr.last["norms"]
print("norms rel err per layer:", [("%.1e" % rel(a, torch.as_tensor(b))) for a, b in zip(n.reshape(n.shape[0], -1)[:, -B:].cpu(), z["layer_norms"][:, 1])])
