"""Architecture facts read from the reference's SOURCE TEXT (never imported or executed) and compared with
the instantiated csl_gan_amd modules.  The model files of the reference cannot be imported here (they pull
in torchvision / the Opacus fork), so this is how their hyper-parameters are pinned without fabricating
those libraries.  Runs only where /root/reference exists (the build container)."""
import ast
import os
import re
from types import SimpleNamespace as NS

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")


def _src(name):
    with open(os.path.join(REF, name)) as f:
        return f.read()


def _class_defaults(src, cls):
    """keyword defaults of <cls>.__init__ as python literals"""
    tree = ast.parse(src)
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for fn in node.body:
                if isinstance(fn, ast.FunctionDef) and fn.name == "__init__":
                    names = [a.arg for a in fn.args.args][-len(fn.args.defaults):] if fn.args.defaults else []
                    return {n: ast.literal_eval(d) for n, d in zip(names, fn.args.defaults)}
    raise KeyError(cls)


def _opt(dataset, model, im_size=64, conditional=False):
    return NS(dataset=dataset, model=model, im_size=im_size, conditional=conditional, n_classes=10 if dataset == "MNIST" else 2,
              per_sample_grad=True, weights_seed=42, manual_seed=1, g_latent_dim=128, g_label_emb_mode="concat",
              d_label_emb_mode="concat", conditional_arch="ACGAN", aux_loss_type="wasserstein", aux_loss_scalar=1,
              g_device="cpu", d_device="cpu")


def test_celeba_and_mnist_size_tables_match_reference_text():
    from csl_gan_amd import init_util
    from csl_gan_amd.nn import HipConv2d
    cel, mn = _src("CelebA_models.py"), _src("MNIST_models.py")
    for im, gname, dname in ((64, "CelebA_DCRN_G64", "CelebA_DCRN_D64"), (48, "CelebA_DCRN_G48", "CelebA_DCRN_D48")):
        gd, dd = _class_defaults(cel, gname), _class_defaults(cel, dname)
        G, D = init_util.init_models(_opt("CelebA", "DeepConvResNet", im))
        assert G.z_dim == gd["z_dim"] and G.first_filter_size == gd["first_filter_size"]
        assert [G.linIn.out_features // gd["first_filter_size"] ** 2] + [b.conv.out_channels for b in G.blocks] == gd["channels"]
        assert [D.blocks[0].in_channels] + [b.out_channels for b in D.blocks] == dd["channels"]
        assert D.linOut.in_features == dd["channels"][-1] * dd["last_filter_size"] ** 2
    gd, dd = _class_defaults(mn, "MNIST_DCRN_G"), _class_defaults(mn, "MNIST_DCRN_D")
    G, D = init_util.init_models(_opt("MNIST", "DeepConvResNet", 28))
    assert [G.linIn.out_features // gd["first_filter_size"] ** 2] + [b.conv.out_channels for b in G.blocks] == gd["channels"]
    assert [D.blocks[0].in_channels] + [b.out_channels for b in D.blocks] == dd["channels"]
    assert D.linOut.in_features == dd["channels"][-1] * dd["last_filter_size"] ** 2
    assert all(isinstance(b, HipConv2d) for b in D.blocks)


def test_layer_hyperparameters_match_reference_text():
    from csl_gan_amd import init_util, ops
    from csl_gan_amd.nn import HipGroupNormAct
    src = _src("DCResNet_models.py")
    m = re.search(r"nn\.Conv2d\(channels\[i-1\], channels\[i\], (\d+), stride=(\d+), padding=(\d+)\)", src)
    k, s, p = (int(v) for v in m.groups())
    slope = float(re.search(r"F\.leaky_relu\(block\(o\), ([0-9.]+)\)", src).group(1))
    groups = int(re.search(r"nn\.GroupNorm\((\d+), in_ch\)", src).group(1))
    up_k = int(re.search(r"ResBlockUp\(channels\[i-1\], channels\[i\], (\d+), bn=self\.bn\)", src).group(1))
    out_k = int(re.search(r"self\.convOut = nn\.Conv2d\(channels\[-1\], self\.out_ch, (\d+), padding=\"same\"\)", src).group(1))
    assert "self.linOut = nn.Linear(size, 1, bias=False)" in src and "torch.tanh(x)" in src
    assert 'self.shortcut = UpsampleConv(in_ch, out_ch, 1)' in src and "bias=False)" in src
    G, D = init_util.init_models(_opt("CelebA", "DeepConvResNet", 64))
    for b in D.blocks:
        assert (b.kernel_size, b.stride, b.padding) == ((k, k), (s, s), (p, p)) and b.act == ops.ACT_LRELU02
    assert slope == 0.2 and D.linOut.bias is None
    for blk in G.blocks:
        assert blk.convUp.conv.kernel_size == (up_k, up_k) and blk.convUp.conv.bias is None
        assert blk.conv.kernel_size == (up_k, up_k) and blk.conv.bias is not None
        assert blk.shortcut.conv.kernel_size == (1, 1) and blk.shortcut.conv.bias is not None
        assert isinstance(blk.bn1, HipGroupNormAct) and blk.bn1.num_groups == groups and blk.bn1.relu
    assert G.convOut.kernel_size == (out_k, out_k) and G.convOut.act == ops.ACT_TANH
    # construction order inside ResBlockUp fixes the RNG stream (init parity): shortcut, bn1, convUp, bn2, conv
    body = src[src.index("class ResBlockUp"):]
    order = [body.index("self." + n + " =") for n in ("shortcut", "bn1", "convUp", "bn2", "conv")]
    assert order == sorted(order)
    assert list(dict(G.blocks[0].named_children())) == ["shortcut", "bn1", "convUp", "bn2", "conv"]


def test_option_defaults_match_reference_text():
    from csl_gan_amd import options
    src = _src("options.py")
    tree = ast.parse(src)
    tables = {}
    for node in tree.body:
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and node.targets[0].id in ("MNIST_DEFAULTS", "CELEBA_DEFAULTS"):
            tables[node.targets[0].id] = ast.literal_eval(node.value)
    assert tables["MNIST_DEFAULTS"] == options.MNIST_DEFAULTS
    assert tables["CELEBA_DEFAULTS"] == options.CELEBA_DEFAULTS
    # every flag of the reference parser exists with the same default
    ref_flags = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument":
            names = [a.value for a in node.args if isinstance(a, ast.Constant)]
            kw = {k.arg: k.value for k in node.keywords}
            default = ast.literal_eval(kw["default"]) if "default" in kw and isinstance(kw["default"], (ast.Constant, ast.List, ast.UnaryOp)) else None
            ref_flags[names[-1]] = (names, default, "default" in kw)
    parser = options.build_parser()
    mine = {a.option_strings[-1] if a.option_strings else a.dest: a for a in parser._actions}
    missing = [n for n in ref_flags if n not in mine]
    assert not missing, missing
    for n, (names, default, has_default) in ref_flags.items():
        act = mine[n]
        if act.option_strings:
            assert set(names) <= set(act.option_strings), n
        if has_default and n != "--im_size":
            assert act.default == default, (n, act.default, default)


def test_training_defaults_used_by_the_step_match_reference_text():
    """Constants the D-step hard-codes in the reference: penalty weight 10, alphas list, Adam arguments."""
    gp, tr = _src("gradient_penalty.py"), _src("train.py")
    assert "weight=10.0" in gp and "torch.rand(batch_size, 1)" in gp
    assert '[1 + x / 10.0 for x in range(1, 100)] + list(range(12, 400))' in tr
    from csl_gan_amd import accountant
    assert accountant.DEFAULT_ALPHAS == [1 + x / 10.0 for x in range(1, 100)] + list(range(12, 400))
    assert "optim.Adam(D.parameters(), lr=opt.d_lr, betas=(opt.adam_b1, opt.adam_b2), weight_decay=opt.weight_decay)" in tr
    assert "p.summed_grad += 0 if penalty_grad[j] is None else penalty_grad[j] * opt.batch_size" in tr
