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
            conditional_arch="ACGAN", weight_scale=1.0):
    """weight_scale k multiplies every parameter of the reference's D after construction.  At its initial weights the critic
    has ||dD/dx|| ~ 0.01-0.03 << 1, so the penalty (||g||-1)^2 barely depends on ||g|| and the one-sided clamp
    (gradient_penalty.py:54) zeroes everything; with k chosen so the per-sample input-gradient norms straddle 1 the
    fixture's penalty, its clamp and its gradients are all sensitive.  The tests apply the same k to their D."""
    _, D = reference_init_models(dataset, "DeepConvResNet", im_size, init_G=False, conditional=conditional,
                                 n_classes=10 if dataset == "MNIST" else 2, conditional_arch=conditional_arch)
    g = torch.Generator().manual_seed(seed)
    ch = 1 if dataset == "MNIST" else 3
    real = (torch.randn(B, ch, im_size, im_size, generator=g) * 0.5).clamp(-1, 1)
    fake = torch.tanh(torch.randn(B, ch, im_size, im_size, generator=g))
    labels = torch.randint(0, D.n_classes, (B,), generator=g) if conditional else None
    if weight_scale == "straddle":      # k such that the median per-sample ||dD/dx_hat|| is 1 (D is nearly homogeneous of degree n_layers in k)
        torch.manual_seed(seed + 7)
        a4 = torch.rand(B, 1).view(B, 1, 1, 1)
        w0 = [p.detach().clone() for p in D.parameters()]
        weight_scale, n_layers = 1.0, len(D.blocks) + 1
        for _ in range(6):
            with torch.no_grad():
                for p, w in zip(D.parameters(), w0):
                    p.copy_(w).mul_(weight_scale)
            xh = (a4 * real + (1 - a4) * fake).detach().requires_grad_(True)
            gx, = torch.autograd.grad(D(xh, labels)[0].sum(), xh)
            med = gx.reshape(B, -1).norm(2, dim=1).median().item()
            weight_scale = float("%.4f" % (weight_scale * med ** (-1.0 / n_layers)))
        with torch.no_grad():
            for p, w in zip(D.parameters(), w0):
                p.copy_(w)
    if weight_scale != 1.0:
        with torch.no_grad():
            for p in D.parameters():
                p.mul_(weight_scale)
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
    # ||dD(x_hat)/dx_hat|| per sample (gradient_penalty.py:48-52) on the reference's D, with the alpha recorded above
    a4 = alpha.view(B, 1, 1, 1)
    xh = (a4 * real + (1 - a4) * fake).detach().requires_grad_(True)
    o_, _ = D(xh, labels)
    gx, = torch.autograd.grad(o_, xh, torch.ones_like(o_))
    out["input_grad_norms"] = gx.reshape(B, -1).double().norm(2, dim=1).numpy()
    out["weight_scale"] = np.float64(weight_scale)
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
    print(name, "penalty", out["penalty"], "norms", np.round(out["grad_norms"], 4), "|dD/dx|", np.round(out["input_grad_norms"], 3))


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


# ---------------------------------------------------------------------------------------------------------------------
# D-step observables ("per-sample norms [layers,B], clip factors, pre-noise clipped sums") through the
# reference's OWN G / D classes and loss methods.  The per-sample-gradient engine (the Opacus fork) is absent, so the
# per-sample gradient is taken by its definition — one autograd call per sample through the reference's D — with the
# fork's documented scaling for a mean-reduced loss (grad_sample_b = B * d(batch loss)/d theta through sample b alone;
# for the separable mean losses of DCResNet_models.py:149-153 / MNIST_models.py:48-52 that IS the gradient of sample
# b's own loss, asserted below).  The clip rule min(1, C/(||g||+1e-6)), the split / accumulated pass conventions and the
# adaptive statistics of train.py:204-245 are then plain arithmetic on those reference-class gradients, written out here
# (float64 reductions) without going through oracle/.
# ---------------------------------------------------------------------------------------------------------------------
def _per_sample_grads(D, x, y, batch_loss, want_aux=True):
    """[B, *p.shape] per parameter (+ loss, out, aux): B autograd calls through the reference's D."""
    params = list(D.parameters())
    B = x.size(0)
    out, aux = D(x, y) if want_aux else D(x, y, aux=False)
    L = batch_loss(out, aux)
    leaves = [out] + ([aux] if aux is not None else [])
    cots = torch.autograd.grad(L, leaves, allow_unused=True)
    res = [torch.zeros((B,) + tuple(p.shape)) for p in params]
    for b in range(B):
        yb = None if y is None else y[b:b + 1]
        ob, ab = D(x[b:b + 1], yb) if want_aux else D(x[b:b + 1], yb, aux=False)
        s = (ob * cots[0][b:b + 1]).sum()
        if ab is not None and len(cots) > 1 and cots[1] is not None:
            s = s + (ab * cots[1][b:b + 1]).sum()
        gs = torch.autograd.grad(s * B, params, allow_unused=True)
        for r, g_ in zip(res, gs):
            if g_ is not None:
                r[b] = g_
    return res, L.detach(), out.detach(), None if aux is None else aux.detach()


def _norms(gs):
    """[L, B] float64 per-layer per-sample L2 norms."""
    return torch.stack([g_.reshape(g_.size(0), -1).double().norm(2, dim=1) for g_ in gs])


def _factors(norms, C):
    """min(1, C/(norm + 1e-6)); C scalar (norms [B]) or per layer (norms [L,B], C [L])."""
    C = torch.as_tensor(C, dtype=torch.float64)
    if C.dim() == 1:
        C = C.view(-1, 1)
    return (C / (norms + 1e-6)).clamp(max=1.0)


def _put_grads(out, key, tensors, scale=None):
    """norms, strided entry samples and the per-tensor scale entry errors are measured against: max |entry|, or for a sum over
    samples the largest entry of sum_b |f_b g_b| (a sum that cancels, e.g. an auxiliary-head bias, has rounding error
    proportional to its terms, not to its result)."""
    from dstep_inputs import sample_idx
    out[key + "_norms"] = np.array([t.double().norm().item() for t in tensors])
    out[key + "_absmax"] = np.array([t.abs().max().item() for t in tensors]) if scale is None else np.array([float(v) for v in scale])
    for i, t in enumerate(tensors):
        f = t.detach().reshape(-1)
        out["%s_s%d" % (key, i)] = f[torch.from_numpy(sample_idx(f.numel()))].float().numpy()


def dstep_case(name, dataset, model, im_size, B, seed, latent=128, penalty=True, geometry=None, adaptive_scalar=1.5,
               private_penalty=False, **kw):
    sys.path.insert(0, HERE)
    from dstep_inputs import checksums, dstep_inputs
    if geometry is None:
        G, D = reference_init_models(dataset, model, im_size, g_latent_dim=latent, **kw)
    else:       # the 128x128 extension: the reference's generic DCResNet classes with the build's size table (init_util order)
        cls = reference_model_classes()
        torch.manual_seed(42)
        G = cls["DCResNetGenerator"](z_dim=latent, channels=list(geometry["g_channels"]), first_filter_size=geometry["first"], out_ch=3,
                                     bn=False, n_classes=0, emb_mode="concat")
        D = cls["DCResNetDiscriminator"](channels=list(geometry["d_channels"]), last_filter_size=geometry["last"], n_classes=0,
                                         emb_mode="concat", conditional_arch="ACGAN", aux_loss_type="wasserstein", aux_loss_scalar=1)
        torch.manual_seed(1)
    ncls = kw.get("n_classes", 2) if kw.get("conditional") else 0
    ch = 1 if dataset == "MNIST" else 3
    inp = dstep_inputs(seed, B, ch, im_size, latent, ncls, unit_range=(model == "Vanilla"))
    use_aux = bool(ncls) and D.conditional_arch in ("ACGAN", "WCGAN")
    params = list(D.parameters())
    L = len(params)
    G.train(); D.train()
    out = dict(meta=np.array([B, im_size, seed, latent, ncls, ch]), input_checksums=checksums(inp),
               d_weight_norms=np.array([p.detach().double().norm().item() for p in params]),
               g_weight_norms=np.array([p.detach().double().norm().item() for p in G.parameters()]),
               d_param_names=np.array([n for n, _ in D.named_parameters()]))

    def real_loss(o, a, labels):
        l = D.real_loss(o, "cpu")
        return l + D.aux_loss(a, labels, "cpu", fake=False) if use_aux else l            # train.py:354-358

    def fake_loss(o, a, y):
        l = D.fake_loss(o, "cpu")
        return l + D.aux_loss(a, y, "cpu", fake=True) if use_aux else l                  # train.py:345-351 (d_fake_aux_loss=True)

    # ---- the two passes of train.py:382-387: pass 0 = generated batch, pass 1 = private batch ----
    with torch.no_grad():
        fake = G(inp["z"], inp["y"])
    g_fake, l_fake, d_fake, d_fake_aux = _per_sample_grads(D, fake, inp["y"], lambda o, a: fake_loss(o, a, inp["y"]))
    g_real, l_real, d_real, d_real_aux = _per_sample_grads(D, inp["img"], inp["labels"], lambda o, a: real_loss(o, a, inp["labels"]))
    if not use_aux:         # separable mean loss: the hook-convention gradient is the gradient of sample b's own loss
        for b in (0, B - 1):
            ob, _ = D(inp["img"][b:b + 1], None if inp["labels"] is None else inp["labels"][b:b + 1])
            gb = torch.autograd.grad(D.real_loss(ob, "cpu"), params, allow_unused=True)
            for gs, g_ in zip(g_real, gb):
                if g_ is not None:
                    assert (gs[b] - g_).abs().max().item() <= 1e-5 * (g_.abs().max().item() + 1e-12)
    n_fake, n_real = _norms(g_fake), _norms(g_real)
    out.update(fake_sample=fake.reshape(-1)[::max(1, fake.numel() // 4096)][:4096].numpy(), fake_norm=fake.double().norm().item(),
               d_real=d_real.numpy(), d_fake=d_fake.numpy(),
               d_real_loss=np.float64(D.real_loss(d_real, "cpu").item()), d_fake_loss=np.float64(D.fake_loss(d_fake, "cpu").item()),
               d_real_pass_loss=np.float64(l_real.item()), d_fake_pass_loss=np.float64(l_fake.item()),
               layer_norms=torch.stack([n_fake, n_real], dim=1).numpy(),                  # [L, pass, B]   (train.py:311-315)
               flat_norms=torch.stack([n_fake.norm(2, dim=0), n_real.norm(2, dim=0)]).numpy())   # [pass, B]
    if use_aux:
        out.update(d_real_aux=d_real_aux.numpy(), d_fake_aux=d_fake_aux.numpy())

    # ---- adaptive statistics (train.py:204-245, split mode: the real loss only, on a public / mean-sample batch) ----
    g_ad, _, _, _ = _per_sample_grads(D, inp["ms_adapt"], inp["ms_adapt_labels"], lambda o, a: real_loss(o, a, inp["ms_adapt_labels"]))
    n_ad = _norms(g_ad)
    r_mean, r_max = n_ad.mean(dim=1), n_ad.max(dim=1).values
    out.update(adaptive_norms=n_ad.numpy(), adaptive_mean=r_mean.numpy(), adaptive_max=r_max.numpy(),
               adaptive_scalar=np.float64(adaptive_scalar), c_adaptive_pl=(adaptive_scalar * r_mean).numpy(),
               c_adaptive_flat=np.float64(adaptive_scalar * r_mean.norm(2).item()))
    del g_ad

    # ---- clip factors + pre-noise clipped sums ----
    flat_real = n_real.norm(2, dim=0)
    c_flat = float("%.3g" % flat_real.median().item())          # a constant C that clips about half of the private samples
    c_pl = adaptive_scalar * r_mean
    f_flat = _factors(flat_real, c_flat)                        # [B]
    f_pl = _factors(n_real, c_pl)                               # [L, B]
    f_aflat = _factors(flat_real, out["c_adaptive_flat"])
    out.update(c_flat=np.float64(c_flat), factors_flat=f_flat.numpy(), factors_pl=f_pl.numpy(), factors_adaptive_flat=f_aflat.numpy())

    def wsum(gs, f):            # sum_b f_b g_b in float64; f [B] or None
        res = []
        for i, g_ in enumerate(gs):
            g64 = g_.double()
            if f is not None:
                fi = f[i] if f.dim() == 2 else f
                g64 = g64 * fi.view(-1, *([1] * (g64.dim() - 1)))
            res.append(g64.sum(dim=0))
        return res

    def term_scale(*gss):       # per tensor: largest entry of sum_b |g_b| over the given passes (>= any clipped sum's terms)
        return [sum(g_.double().abs().sum(dim=0) for g_ in gs).max().item() for gs in zip(*gss)]

    s_fake = wsum(g_fake, None)                                  # generated pass: never clipped in split mode (train.py:112-113)
    sum_flat = [a + b for a, b in zip(s_fake, wsum(g_real, f_flat))]
    sum_pl = [a + b for a, b in zip(s_fake, wsum(g_real, f_pl))]
    tscale = term_scale(g_fake, g_real)
    _put_grads(out, "sum_flat_split", sum_flat, tscale)
    _put_grads(out, "sum_pl_split", sum_pl, tscale)
    # accumulated passes (-gcs False): per-sample sum over passes, clipped together with one flat C
    g_acc = [a + b for a, b in zip(g_fake, g_real)]
    n_acc = _norms(g_acc).norm(2, dim=0)
    c_acc = float("%.3g" % n_acc.median().item())
    f_acc = _factors(n_acc, c_acc)
    out.update(accum_flat_norms=n_acc.numpy(), c_accum=np.float64(c_acc), factors_accum=f_acc.numpy())
    _put_grads(out, "sum_flat_accum", wsum(g_acc, f_acc), tscale)
    del g_acc, g_fake, g_real

    # ---- WGAN-GP on the public batch + parameter gradients (train.py:423-431) ----
    if penalty:
        a4 = inp["alpha"].view(B, 1, 1, 1)
        # the reference draws alpha itself (gradient_penalty.py:33): hand it OUR alpha by patching torch.rand for this one call
        real_rand = torch.rand
        torch.rand = lambda *a, **k: inp["alpha"].view(B, 1).clone()
        try:
            pen = ref_gp.calc_penalty(D, ["WGAN-GP"], inp["ms_pen"], inp["ms_pen_labels"], fake, inp["y"], device="cpu", aux_penalty=True)
        finally:
            torch.rand = real_rand
        pg = torch.autograd.grad(pen, params, allow_unused=True)
        pg = [torch.zeros_like(p) if g_ is None else g_ for g_, p in zip(pg, params)]
        xh = (a4 * inp["ms_pen"] + (1 - a4) * fake).detach().requires_grad_(True)
        o_, _ = D(xh, inp["ms_pen_labels"])
        gx, = torch.autograd.grad(o_, xh, torch.ones_like(o_))
        out.update(penalty=np.float64(pen.item()), pen_input_grad_norms=gx.reshape(B, -1).double().norm(2, dim=1).numpy())
        _put_grads(out, "pen_grad", pg)
        sg_scale = [a + B * g_.abs().max().item() for a, g_ in zip(tscale, pg)]
        _put_grads(out, "summed_grad_pl", [s_ + g_.double() * B for s_, g_ in zip(sum_pl, pg)], sg_scale)          # train.py:431
        _put_grads(out, "summed_grad_flat", [s_ + g_.double() * B for s_, g_ in zip(sum_flat, pg)], sg_scale)
        # ---- the penalty evaluated on the PRIVATE batch per sample (train.py:433-450): the reference's own loop — one
        # autograd.grad(penalties[i], D.parameters()) per sample — with penalties from its calc_penalty(per_sample=True); each
        # gradient is added to p.grad_sample[0, i] (pass 0, as written there) and the batch is clipped again.  In split mode pass 0
        # is never clipped and the private pass's norms are unchanged, so the second clip's sum is the first one + sum_i grad_i.
        if private_penalty:
            torch.rand = lambda *a, **k: inp["alpha"].view(B, 1).clone()
            try:
                pens = ref_gp.calc_penalty(D, ["WGAN-GP"], inp["img"], inp["labels"], fake, inp["y"], device="cpu", per_sample=True,
                                           aux_penalty=True)
            finally:
                torch.rand = real_rand
            tot = [torch.zeros_like(p, dtype=torch.float64) for p in params]
            tot_abs = [torch.zeros_like(p, dtype=torch.float64) for p in params]
            for i in range(B):
                gi = torch.autograd.grad(pens[i], params, retain_graph=True, allow_unused=True)
                for t_, ta_, g_ in zip(tot, tot_abs, gi):
                    if g_ is not None:
                        t_ += g_.double()
                        ta_ += g_.double().abs()
            out.update(private_penalties=pens.detach().numpy(), private_penalty_mean=np.float64(pens.mean().item()))
            pscale = [a + t_.max().item() for a, t_ in zip(tscale, tot_abs)]
            _put_grads(out, "sum_flat_split_private_pen", [s_ + t_ for s_, t_ in zip(sum_flat, tot)], pscale)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "L=%d" % L, "layer-norm means (real pass)", np.round(n_real.mean(dim=1).numpy(), 3), "c_flat", c_flat,
          "clipped frac flat %.2f pl %.2f" % ((f_flat < 0.999).double().mean().item(), (f_pl < 0.999).double().mean().item()),
          "penalty", out.get("penalty"))


def survey_probe():
    """SURVEY.md §8c quotes per-layer means of the per-sample gradient norms [0.146,0.024,0.694,0.059,0.969,0.137,1.402,0.410,
    1.623] for the imported reference D64 'at seed 42/1, B=16' without recording the input batch.  Recovered: D built ALONE under
    weights_seed 42 (no generator before it), manual_seed 1, x = randn(16,3,64,64).clamp(-1,1), the real loss — reproduces every
    entry to the three decimals quoted.  Stored with full precision for the oracle / HIP tests."""
    target = np.array([0.146, 0.024, 0.694, 0.059, 0.969, 0.137, 1.402, 0.410, 1.623])
    _, D = reference_init_models("CelebA", "DeepConvResNet", 64, init_G=False)
    torch.manual_seed(1)
    x = torch.randn(16, 3, 64, 64).clamp(-1, 1)
    gs, _, out, _ = _per_sample_grads(D, x, None, lambda o, a: D.real_loss(o, "cpu"))
    n = _norms(gs)
    m = n.mean(dim=1).numpy()
    assert np.abs(m - target).max() < 1.5e-3, (m, target)
    np.savez_compressed(os.path.join(HERE, "dstep_survey_probe.npz"), layer_norms=n.numpy(), layer_norm_means=m, survey_quote=target,
                        d_real=out.numpy(), x_checksum=np.array([x.double().sum().item(), (x.double() ** 2).sum().item()]),
                        d_weight_norms=np.array([p.detach().double().norm().item() for p in D.parameters()]))
    print("survey probe reproduced:", np.round(m, 4), "max abs diff to the quoted vector", np.abs(m - target).max())


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
    only = set(sys.argv[1:])

    def want(name):
        return not only or name in only or any(name.startswith(o.rstrip("*")) for o in only if o.endswith("*"))

    class _Run:
        def __getattr__(self, fn):
            return lambda name, *a, **k: globals()[fn](name, *a, **k) if want(name) else None
    run = _Run()
    run.gp_case("gp_mnist_dcrn_b6", "MNIST", 28, 6, seed=11)
    run.gp_case("gp_mnist_dcrn_b6_onesided", "MNIST", 28, 6, seed=12, one_sided=True)
    run.gp_case("gp_celeba64_b4", "CelebA", 64, 4, seed=13)
    run.gp_case("gp_celeba64_cond_aux_b3", "CelebA", 64, 3, seed=14, conditional=True, aux_penalty=True)
    # weights scaled so that ||dD/dx|| straddles 1: penalty, one-sided clamp and gradients all bite
    # (two-sided: norms ~1.3, away from both 0 and 1; one-sided: the median norm is 1, so the clamp splits the batch)
    run.gp_case("gp_mnist_dcrn_b6_scaled", "MNIST", 28, 6, seed=15, weight_scale=2.5)
    run.gp_case("gp_mnist_dcrn_b6_onesided_scaled", "MNIST", 28, 6, seed=16, one_sided=True, weight_scale="straddle")
    run.gp_case("gp_celeba64_b4_scaled", "CelebA", 64, 4, seed=17, weight_scale=2.5)
    run.gp_case("gp_celeba64_b4_onesided_scaled", "CelebA", 64, 4, seed=18, one_sided=True, weight_scale="straddle")
    run.gp_case("gp_celeba64_cond_aux_b3_scaled", "CelebA", 64, 3, seed=19, conditional=True, aux_penalty=True, weight_scale=2.5)
    if want("aux_loss"):
        aux_loss_cases()
    if want("logger"):
        logger_case()
    run.bpc_case("bpc_mnist_dcrn_auto_b6", "DeepConvResNet", 6, seed=31)
    run.bpc_case("bpc_mnist_vanilla_cond_auto_b8", "Vanilla", 8, seed=32, conditional=True, aas=0.05, awgs=1e-4)
    run.bpc_case("bpc_mnist_dcrn_explicit_b5", "DeepConvResNet", 5, seed=33, back=[0.02, 0.01, 0.5], fwd=[3.0, 20.0, 10.0])
    if want("upsample_conv"):
        upsample_conv_case()
    run.model_case("model_celeba64_gn_b2", "CelebA", "DeepConvResNet", 64, 2, seed=21)
    run.model_case("model_celeba64_bn_b3", "CelebA", "DeepConvResNet", 64, 3, seed=22, per_sample_grad=False)
    run.model_case("model_celeba48_gn_b2", "CelebA", "DeepConvResNet", 48, 2, seed=23)
    run.model_case("model_celeba64_cond_acgan_b4", "CelebA", "DeepConvResNet", 64, 4, seed=24, conditional=True, n_classes=2)
    run.model_case("model_mnist_dcrn_gn_b4", "MNIST", "DeepConvResNet", 28, 4, seed=25, latent=16)
    run.model_case("model_mnist_dcrn_cond_cgan_bn_b4", "MNIST", "DeepConvResNet", 28, 4, seed=26, latent=16, conditional=True,
                   n_classes=10, conditional_arch="CGAN", per_sample_grad=False)
    run.model_case("model_mnist_vanilla_b8", "MNIST", "Vanilla", 28, 8, seed=27, latent=100)
    run.model_case("model_mnist_vanilla_cond_b8", "MNIST", "Vanilla", 28, 8, seed=28, latent=100, conditional=True, n_classes=10,
                   aux_loss_type="cross_entropy")
    # D-step observables through the reference's classes
    run.dstep_case("dstep_celeba64_b8", "CelebA", "DeepConvResNet", 64, 8, seed=41, private_penalty=True)         # configs[2] geometry
    run.dstep_case("dstep_celeba64_cond_acgan_b8", "CelebA", "DeepConvResNet", 64, 8, seed=42, conditional=True, n_classes=2)
    run.dstep_case("dstep_mnist_vanilla_cond_b16", "MNIST", "Vanilla", 28, 16, seed=43, latent=100, penalty=False,     # configs[1]
                   adaptive_scalar=1.0, conditional=True, n_classes=10, aux_loss_type="cross_entropy")
    run.dstep_case("dstep_mnist_vanilla_b16", "MNIST", "Vanilla", 28, 16, seed=44, latent=100, penalty=False, adaptive_scalar=1.0)
    run.dstep_case("dstep_mnist_dcrn_b6", "MNIST", "DeepConvResNet", 28, 6, seed=45, latent=16, private_penalty=True)
    run.dstep_case("dstep_celeba128_b4", "CelebA", "DeepConvResNet", 128, 4, seed=46,                               # configs[4] geometry
                   geometry=dict(g_channels=(512, 512, 256, 128, 64, 64), first=4, d_channels=(3, 64, 128, 256, 512), last=8))
    if want("survey_probe"):
        survey_probe()
