#!/usr/bin/env python3
"""Generate golden vectors by running the reference's own importable modules.

Runs ONLY in the build container (needs /root/reference); the GPU box uses the committed
``*.npz`` / ``*.txt`` outputs.  Imported from the reference, unmodified and without any
stand-in modules:  ``gradient_penalty`` , ``models`` , ``logger``  — the three files whose
imports resolve here.  (DCResNet_models / MNIST_models / util / options / backprop_clip /
mean_sampler import torchvision, torchinfo or the opacus fork, which are absent; they are
NOT imported and no stand-ins are fabricated — see DESIGN.md "Oracle".)

The discriminator handed to the reference's ``calc_penalty`` is the oracle restatement
(oracle/nets.py), whose weights are regenerated from seeds, so fixtures hold only inputs'
seeds, small inputs and expected outputs.

usage:  python tests/golden/make_golden.py
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(1, "/root/reference")

import gradient_penalty as ref_gp      # noqa: E402  (reference, direct import)
import models as ref_models            # noqa: E402
import logger as ref_logger            # noqa: E402

from oracle.nets import build_models   # noqa: E402


def gp_case(name, dataset, im_size, B, seed, one_sided=False, conditional=False, aux_penalty=False,
            conditional_arch="ACGAN"):
    _, D = build_models(dataset=dataset, model="DeepConvResNet", im_size=im_size, weights_seed=42, manual_seed=1,
                        init_G=False, conditional=conditional, n_classes=10 if dataset == "MNIST" else 2,
                        conditional_arch=conditional_arch)
    g = torch.Generator().manual_seed(seed)
    ch = 1 if dataset == "MNIST" else 3
    real = (torch.randn(B, ch, im_size, im_size, generator=g) * 0.5).clamp(-1, 1)
    fake = torch.tanh(torch.randn(B, ch, im_size, im_size, generator=g))
    labels = torch.randint(0, D.n_classes, (B,), generator=g) if conditional else None
    ptype = "WGAN-GP1" if one_sided else "WGAN-GP"
    out = {}
    for per_sample in (False, True):
        torch.manual_seed(seed + 7)
        alpha = torch.rand(B, 1)                 # what gradient_penalty.py:33 will draw next
        torch.manual_seed(seed + 7)
        pen = ref_gp.calc_penalty(D, [ptype], real, labels, fake, labels, device="cpu", per_sample=per_sample,
                                  aux_penalty=aux_penalty)
        if per_sample:
            out["penalty_per_sample"] = pen.detach().numpy()
        else:
            grads = torch.autograd.grad(pen, list(D.parameters()), allow_unused=True)
            out["penalty"] = np.float64(pen.item())
            out["grad_norms"] = np.array([0.0 if gr is None else gr.norm().item() for gr in grads])
            out["grad_heads"] = np.stack([np.zeros(8, np.float32) if gr is None else
                                          gr.reshape(-1)[:8].numpy() for gr in grads])
        out["alpha"] = alpha.reshape(-1).numpy()
    with torch.no_grad():
        d_out, d_aux = D(real, labels)
    out.update(real=real.numpy(), fake=fake.numpy(), d_out_real=d_out.numpy(),
               weight_norms=np.array([p.norm().item() for p in D.parameters()]),
               meta=np.array([B, im_size, seed, int(one_sided), int(conditional), int(aux_penalty)]))
    if labels is not None:
        out["labels"] = labels.numpy()
        if d_aux is not None:
            out["d_aux_real"] = d_aux.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "penalty", out["penalty"], "norms", np.round(out["grad_norms"], 4))


def aux_loss_cases():
    g = torch.Generator().manual_seed(5)
    out = {}
    for ncls, B in ((2, 16), (10, 32)):
        logits = torch.randn(B, ncls, generator=g)
        labels = torch.randint(0, ncls, (B,), generator=g)
        labels[:ncls] = torch.arange(ncls)        # every class present (the reference divides by class counts)
        for typ in ("wasserstein", "cross_entropy"):
            d = ref_models.Discriminator(n_classes=ncls, conditional_arch="ACGAN", aux_loss_type=typ, aux_loss_scalar=0.5)
            out[f"{typ}_{ncls}"] = np.float64(d.aux_loss(logits, labels, "cpu").item())
        out[f"logits_{ncls}"] = logits.numpy()
        out[f"labels_{ncls}"] = labels.numpy()
    np.savez_compressed(os.path.join(HERE, "aux_loss.npz"), **out)
    print("aux_loss", {k: v for k, v in out.items() if k[0] in "wc"})


def logger_case():
    path = os.path.join(HERE, "_tmp_logger.csv")
    if os.path.exists(path):
        os.remove(path)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        lg = ref_logger.Logger("A: {:4.4f} | B: {:3.1f}", ["A", "B"], 4, path)
        for i in range(8):
            lg.stats["A"] += 0.25 * i
            lg.stats["B"] += 10.0 + i
            if (i + 1) % 4 == 0:
                lg.log(i // 4, 50.0 * (i // 4))
        lg.close()
    with open(path) as f:
        csv_text = f.read()
    os.remove(path)
    with open(os.path.join(HERE, "logger_expected.txt"), "w") as f:
        f.write("#CSV\n" + csv_text + "#STDOUT\n" + buf.getvalue())
    print("logger ok")


if __name__ == "__main__":
    gp_case("gp_mnist_dcrn_b6", "MNIST", 28, 6, seed=11)
    gp_case("gp_mnist_dcrn_b6_onesided", "MNIST", 28, 6, seed=12, one_sided=True)
    gp_case("gp_celeba64_b4", "CelebA", 64, 4, seed=13)
    gp_case("gp_celeba64_cond_aux_b3", "CelebA", 64, 3, seed=14, conditional=True, aux_penalty=True)
    aux_loss_cases()
    logger_case()
