#!/usr/bin/env python3
"""Generate golden vectors by running the reference's own importable modules.

Runs ONLY in the build container (needs /root/reference); the GPU box uses the committed
``*.npz`` / ``*.txt`` outputs.  Imported from the reference, unmodified and without any
stand-in modules:  ``gradient_penalty`` , ``models`` , ``logger``  — the three files whose
imports resolve here.

The model files (DCResNet_models / MNIST_models / CelebA_models) start with ``import util``, which pulls
in torchvision and the opacus fork (absent; no stand-ins are fabricated).  None of their classes uses
``util``, so ``reference_model_classes()`` parses those files with ``ast``, keeps the ``class`` statements
only and executes them — the reference's own class bodies, read from /root/reference at run time, never
stored — in a namespace holding ``torch, nn, F`` and the ``Generator`` / ``Discriminator`` bases of the
directly importable ``models.py``.  ``model_case`` then builds G and D exactly as ``init_util.py:44-71``
does (seed ``weights_seed``, G first, then D, reseed) and records what the REFERENCE classes compute:
G(z, y), D(G(z)), ``G.loss``, D on a real batch, and the norms / leading entries of the generator
gradients of one ``train_G`` backward (train.py:502-511).  Fixtures hold seeds, small inputs and
expected outputs — weights are regenerated from the seed and pinned by per-tensor norms.

The discriminator handed to the reference's ``calc_penalty`` is the reference's own class as well.

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


REF = "/root/reference"


def reference_model_classes():
    """{name: class} for every class of the reference's model files, executed from their own source."""
    import ast
    import torch.nn.functional as F
    from torch import nn
    ns = {"torch": torch, "nn": nn, "F": F, "Generator": ref_models.Generator, "Discriminator": ref_models.Discriminator,
          "__name__": "reference_models"}
    for fname in ("DCResNet_models.py", "MNIST_models.py", "CelebA_models.py"):
        path = os.path.join(REF, fname)
        with open(path) as f:
            tree = ast.parse(f.read(), filename=path)
        tree.body = [n for n in tree.body if isinstance(n, ast.ClassDef)]
        exec(compile(tree, path, "exec"), ns)
    return {k: v for k, v in ns.items() if isinstance(v, type)}


def reference_init_models(dataset, model="DeepConvResNet", im_size=64, *, weights_seed=42, manual_seed=1, conditional=False,
                          n_classes=2, per_sample_grad=True, g_latent_dim=128, g_label_emb_mode="concat",
                          d_label_emb_mode="concat", conditional_arch="ACGAN", aux_loss_type="wasserstein", aux_loss_scalar=1,
                          init_G=True, init_D=True):
    """init_util.py:44-71 on the reference's classes (that file itself imports torchvision)."""
    cls = reference_model_classes()
    ncls = n_classes if conditional else 0
    bn = not per_sample_grad
    if dataset == "MNIST":
        GObj, DObj = (cls["MNIST_DCRN_G"], cls["MNIST_DCRN_D"]) if model == "DeepConvResNet" else (cls["MNISTVanillaG"], cls["MNISTVanillaD"])
    else:
        GObj = cls["CelebA_DCRN_G48"] if im_size == 48 else cls["CelebA_DCRN_G64"]
        DObj = cls["CelebA_DCRN_D48"] if im_size == 48 else cls["CelebA_DCRN_D64"]
    torch.manual_seed(weights_seed)
    G = GObj(z_dim=g_latent_dim, bn=bn, n_classes=ncls, emb_mode=g_label_emb_mode) if init_G else None
    D = DObj(n_classes=ncls, emb_mode=d_label_emb_mode, conditional_arch=conditional_arch, aux_loss_type=aux_loss_type,
             aux_loss_scalar=aux_loss_scalar) if init_D else None
    torch.manual_seed(manual_seed)
    return G, D


def gp_case(name, dataset, im_size, B, seed, one_sided=False, conditional=False, aux_penalty=False,
            conditional_arch="ACGAN"):
    _, D = reference_init_models(dataset, "DeepConvResNet", im_size, init_G=False, conditional=conditional,
                                 n_classes=10 if dataset == "MNIST" else 2, conditional_arch=conditional_arch)
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
            out["grad_norms"] = np.array([0.0 if gr is None else gr.double().norm().item() for gr in grads])
            out["grad_heads"] = np.stack([np.zeros(8, np.float32) if gr is None else
                                          gr.reshape(-1)[:8].numpy() for gr in grads])
        out["alpha"] = alpha.reshape(-1).numpy()
    with torch.no_grad():
        d_out, d_aux = D(real, labels)
    out.update(real=real.numpy(), fake=fake.numpy(), d_out_real=d_out.numpy(),
               weight_norms=np.array([p.detach().double().norm().item() for p in D.parameters()]),
               meta=np.array([B, im_size, seed, int(one_sided), int(conditional), int(aux_penalty)]))
    if labels is not None:
        out["labels"] = labels.numpy()
        if d_aux is not None:
            out["d_aux_real"] = d_aux.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "penalty", out["penalty"], "norms", np.round(out["grad_norms"], 4))


def upsample_conv_case():
    """The op itself (DCResNet_models.py:8-17) on a small input, with the index law the HIP path relies on."""
    cls = reference_model_classes()
    torch.manual_seed(3)
    out = {}
    for C, K, k, H in ((8, 6, 5, 3), (16, 4, 1, 2), (4, 3, 3, 4)):
        m = cls["UpsampleConv"](C, K, k, bias=(k != 5))
        x = torch.randn(2, C, H, H + 1)
        with torch.no_grad():
            y = m(x)
        tag = "c%d_k%d_f%d" % (C, K, k)
        out["x_" + tag], out["y_" + tag], out["w_" + tag] = x.numpy(), y.numpy(), m.conv.weight.detach().numpy()
        if m.conv.bias is not None:
            out["b_" + tag] = m.conv.bias.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "upsample_conv.npz"), **out)
    print("upsample_conv ok")


def _head8(gr):
    h = np.zeros(8, np.float32)
    if gr is not None:
        v = gr.reshape(-1)[:8].numpy()
        h[:v.size] = v
    return h


def model_case(name, dataset, model, im_size, B, seed, latent=128, **kw):
    """Forward / loss / train_G-gradient vectors of the reference's own G and D classes."""
    G, D = reference_init_models(dataset, model, im_size, g_latent_dim=latent, **kw)
    g = torch.Generator().manual_seed(seed)
    ch = 1 if dataset == "MNIST" else 3
    ncls = kw.get("n_classes", 2) if kw.get("conditional") else 0
    z = torch.randn(B, latent, generator=g)
    real = (torch.randn(B, ch, im_size, im_size, generator=g) * 0.5).clamp(-1, 1)
    y = torch.randint(0, ncls, (B,), generator=g) if ncls else None
    if y is not None:
        y[:min(ncls, B)] = torch.arange(min(ncls, B))      # every class present (aux_loss divides by class counts)
    G.train(); D.train()
    fake = G(z, y)                                  # training-mode forward (BatchNorm uses batch statistics)
    d_fake, d_fake_aux = D(fake, y)
    g_loss = G.loss(d_fake, "cpu")
    total = g_loss
    if d_fake_aux is not None and D.conditional_arch == "ACGAN":
        total = total + D.aux_loss(d_fake_aux, y, "cpu")        # train.py:506-509
    gparams = [(n, p) for n, p in G.named_parameters()]
    grads = torch.autograd.grad(total, [p for _, p in gparams], allow_unused=True)
    with torch.no_grad():
        d_real, d_real_aux = D(real, y)
    out = dict(z=z.numpy(), real=real.numpy(), fake=fake.detach().numpy(), d_fake=d_fake.detach().numpy(),
               d_real=d_real.numpy(), g_loss=np.float64(g_loss.item()), g_total_loss=np.float64(total.item()),
               d_real_loss=np.float64(D.real_loss(d_real, "cpu").item()), d_fake_loss=np.float64(D.fake_loss(d_fake, "cpu").item()),
               g_weight_norms=np.array([p.detach().double().norm().item() for _, p in gparams]),
               d_weight_norms=np.array([p.detach().double().norm().item() for p in D.parameters()]),
               g_param_names=np.array([n for n, _ in gparams]),
               g_grad_norms=np.array([0.0 if gr is None else gr.double().norm().item() for gr in grads]),      # float64 reductions
               g_grad_heads=np.stack([_head8(gr) for gr in grads]),
               meta=np.array([B, im_size, seed, latent, ncls, int(kw.get("per_sample_grad", True))]))
    if y is not None:
        out["labels"] = y.numpy()
    if d_fake_aux is not None:
        out["d_fake_aux"], out["d_real_aux"] = d_fake_aux.detach().numpy(), d_real_aux.numpy()
    if not kw.get("per_sample_grad", True) and model == "DeepConvResNet":
        # BatchNorm running statistics after this one training-mode forward, and the eval-mode output that uses them
        out["bn_running_mean0"] = G.blocks[0].bn1.running_mean.numpy().copy()
        out["bn_running_var0"] = G.blocks[0].bn1.running_var.numpy().copy()
        G.eval()
        with torch.no_grad():
            out["fake_eval"] = G(z, y).numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "g_loss %.6f" % out["g_loss"], "fake range", float(fake.min()), float(fake.max()))


def reference_bpc_namespace():
    """Functions and classes of the reference's backprop_clip.py, executed from its own source (its imports need torchinfo, absent)."""
    import ast
    from torch import nn
    ns = {"torch": torch, "nn": nn, "np": np, "List": list, "os": os, "sys": sys, "__name__": "reference_backprop_clip"}
    path = os.path.join(REF, "backprop_clip.py")
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    tree.body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef))]
    exec(compile(tree, path, "exec"), ns)
    return ns


def _sample(t, n=4096):
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)][:n].numpy().copy()


def bpc_case(name, model, B, seed, back=None, fwd=None, aas=0.2, awgs=1e-3, conditional=False):
    """The reference's PGCWrapper / BackpropClipper.convert on its own MNIST discriminators.  BackpropClipper.__init__ calls
    torchinfo.summary (absent) only to read each leaf layer's input / output size; forward hooks on one zero 1x1x28x28 batch
    (its probe, backprop_clip.py:123) give the same sizes, then the object is filled in as its __init__ does (backprop_clip.py:
    107-120) and its own convert() wraps the layers."""
    ns = reference_bpc_namespace()
    _, D = reference_init_models("MNIST", model=model, im_size=28, init_G=False, weights_seed=seed, conditional=conditional,
                                 n_classes=10, conditional_arch="CGAN", aux_loss_type="cross_entropy")
    hs = [m.register_forward_hook(lambda m, i, o: (setattr(m, "in_shape", list(i[0].shape[1:])), setattr(m, "out_shape", list(o.shape[1:])))[0])
          for m in D.modules() if len(list(m.children())) < 1]
    g = torch.Generator().manual_seed(seed)
    y1 = torch.zeros(1, dtype=torch.long) if conditional else None
    with torch.no_grad():
        D(torch.zeros(1, 1, 28, 28), y1)
    for h in hs:
        h.remove()
    pgc = object.__new__(ns["BackpropClipper"])
    pgc.hooks_enabled, pgc.parameter_ind, pgc.layer_ind, pgc.device = True, 0, 0, "cpu"
    pgc.back_clip_params = [] if back is None else list(back)
    pgc.input_clip_params = [] if fwd is None else list(fwd)
    pgc.auto_activation_scale, pgc.auto_weight_grad_scale, pgc.grad_l2_bounds = aas, awgs, []
    pgc.convert(D, auto_params=(back is None or fwd is None))
    # rows of very different size: some inputs / gradients are clipped, some are not
    x = torch.rand(B, 1, 28, 28, generator=g) * torch.logspace(-2, 0, B).view(B, 1, 1, 1)
    y = torch.randint(0, 10, (B,), generator=g) if conditional else None
    w = torch.logspace(-3, 1, B)                     # per-sample loss weights: output gradients of very different size
    res = dict(seed=seed, B=B, x=x.numpy(), w=w.numpy(), aas=aas, awgs=awgs,
               grad_l2_bounds=np.asarray(pgc.grad_l2_bounds, dtype=np.float64),
               back_clip_params=np.asarray(pgc.back_clip_params, dtype=np.float64),
               input_clip_params=np.asarray(pgc.input_clip_params, dtype=np.float64),
               wnorms=np.asarray([p.double().norm().item() for p in D.parameters()]))
    if y is not None:
        res["y"] = y.numpy()
    if back is not None:
        res["back_in"], res["fwd_in"] = np.asarray(back, dtype=np.float64), np.asarray(fwd, dtype=np.float64)
    for tag, on in (("on", True), ("off", False)):
        pgc.hooks_enabled = on
        for p in D.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        out, _ = D(xi, y)
        ((out.reshape(B) * w).sum() / B).backward()
        res["out"] = out.detach().numpy()
        res["gx_" + tag] = xi.grad.numpy()
        res["gnorm_" + tag] = np.asarray([p.grad.double().norm().item() for p in D.parameters()])
        for i, p in enumerate(D.parameters()):
            res["g%d_%s" % (i, tag)] = _sample(p.grad)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **res)
    print(name, "bounds", pgc.grad_l2_bounds, "gnorm on", res["gnorm_on"], "off", res["gnorm_off"])


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
    bpc_case("bpc_mnist_dcrn_auto_b6", "DeepConvResNet", 6, seed=31)
    bpc_case("bpc_mnist_vanilla_cond_auto_b8", "Vanilla", 8, seed=32, conditional=True, aas=0.05, awgs=1e-4)
    bpc_case("bpc_mnist_dcrn_explicit_b5", "DeepConvResNet", 5, seed=33, back=[0.02, 0.01, 0.5], fwd=[3.0, 20.0, 10.0])
    upsample_conv_case()
    model_case("model_celeba64_gn_b2", "CelebA", "DeepConvResNet", 64, 2, seed=21)
    model_case("model_celeba64_bn_b3", "CelebA", "DeepConvResNet", 64, 3, seed=22, per_sample_grad=False)
    model_case("model_celeba48_gn_b2", "CelebA", "DeepConvResNet", 48, 2, seed=23)
    model_case("model_celeba64_cond_acgan_b4", "CelebA", "DeepConvResNet", 64, 4, seed=24, conditional=True, n_classes=2)
    model_case("model_mnist_dcrn_gn_b4", "MNIST", "DeepConvResNet", 28, 4, seed=25, latent=16)
    model_case("model_mnist_dcrn_cond_cgan_bn_b4", "MNIST", "DeepConvResNet", 28, 4, seed=26, latent=16, conditional=True,
               n_classes=10, conditional_arch="CGAN", per_sample_grad=False)
    model_case("model_mnist_vanilla_b8", "MNIST", "Vanilla", 28, 8, seed=27, latent=100)
    model_case("model_mnist_vanilla_cond_b8", "MNIST", "Vanilla", 28, 8, seed=28, latent=100, conditional=True, n_classes=10,
               aux_loss_type="cross_entropy")
