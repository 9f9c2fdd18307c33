#!/usr/bin/env python3
"""Wall time of each phase of the headline D-step (HIP events around the phases; 10 steps, mean)."""
import os, sys, contextlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from csl_gan_amd import util
from csl_gan_amd.gradient_penalty import calc_penalty
from torch import autograd

with contextlib.redirect_stdout(sys.stderr):
    opt, tr, img = bench.build_trainer(0, 1, 0)
B = img.shape[0]
pe, D, G = tr.privacy_engine, tr.D, tr.G
acc = {}


class T:
    def __init__(self, name): self.name = name
    def __enter__(self):
        self.s = torch.cuda.Event(enable_timing=True); self.e = torch.cuda.Event(enable_timing=True); self.s.record()
    def __exit__(self, *a):
        self.e.record(); acc.setdefault(self.name, []).append((self.s, self.e))


def step():
    util.zero_grad(D); util.freeze(G); pe.zero_grad()
    z = tr.gen_z(B)
    with T("G forward"):
        with torch.no_grad():
            fake = G(z, None)
    with T("mean-sample draws"):
        xa, _ = tr.mean_sampler.sample(B)
    with T("D fused fwd (384 rows)"):
        pe.enable_hooks(); pe.row_roles = [("norms", B), ("dense", B), ("private", B)]
        out, _ = D(torch.cat([xa, fake, img], 0), None)
        oa, of, orl = torch.split(out, [B, B, B])
        loss = D.real_loss(orl, 0) + D.fake_loss(of, 0) + D.real_loss(oa, 0)
    with T("D fused bwd + per-sample/dense/norm wgrad"):
        loss.backward(); pe.disable_hooks()
    with T("adaptive C + clip"):
        norms = pe.norms_rows_sqnorms().sqrt()
        pe.set_max_grad_norm_device(norms.mean(dim=1) * 1.5); pe.row_roles = None
        pe.clip()
    with T("penalty fwd + 1st-order bwd"):
        pr, _ = tr.mean_sampler.sample(B)
        pen = calc_penalty(D, ["WGAN-GP"], pr, None, fake, None, device=opt.d_device, aux_penalty=True)
    with T("penalty 2nd-order bwd"):
        pg = autograd.grad(pen, list(D.parameters()), allow_unused=True)
        with torch.no_grad():
            for p, g in zip(D.parameters(), pg):
                if g is not None:
                    p.summed_grad.add_(g, alpha=B)
    with T("noise + Adam"):
        tr.d_optimizer.step()
    util.unfreeze(G)


for _ in range(3):
    step()
acc.clear()
torch.cuda.synchronize()
for _ in range(10):
    step()
torch.cuda.synchronize()
tot = 0
for k, v in acc.items():
    ms = sum(s.elapsed_time(e) for s, e in v) / len(v)
    tot += ms
    print("%-46s %7.3f ms" % (k, ms))
print("%-46s %7.3f ms" % ("sum", tot))
