"""CPU tests of the host-side logic: options surface, accountant known answers, logger format
(against the reference's own output), mean sampler, model factory / state_dict keys, CLI plumbing
for BASELINE configs[0] (MNIST vanilla GAN, cpu/cpu, no DP)."""
import io
import json
import math
import os
import contextlib

import numpy as np
import pytest
import torch

from csl_gan_amd import accountant, options


def test_options_defaults_and_quirks(tmp_path):
    o = options.parse(["CelebA", "-o", str(tmp_path), "-dpm", "gc", "-nms", "32", "-gcm", "adaptive-pl"])
    assert (o.batch_size, o.n_d_steps, o.g_latent_dim, o.sigma, o.delta) == (128, 5, 128, 0.5, 1e-6)
    assert o.penalty == ["WGAN-GP"] and o.use_dp and o.per_sample_grad and o.use_grad_clip_per_layer
    assert o.train_d_until_threshold == -1                      # forced for DCResNet + DP (options.py:240-242)
    assert o.imm_sens_per_param is True                          # False-as-unset quirk (options.py:95,77)
    assert o.clipping_param_per_layer == [1000, 200, 1000, 100, 1000, 100, 1000, 5, 2500]
    assert os.path.isdir(o.output_dir + "saves/")
    m = options.parse(["MNIST", "-o", str(tmp_path / "m"), "--conditional", "-dpm", "gc", "--sigma", "10", "-bs", "600"])
    assert (m.model, m.n_classes, m.is_acgan, m.use_aux_loss, m.sigma) == ("Vanilla", 10, True, True, 10.0)
    assert m.log_every_epochs == 1 and m.log_every == 99600      # 100000 rounded to a multiple of 600


def test_options_incompatibilities(tmp_path):
    with pytest.raises(Exception, match="mean sampling"):
        options.parse(["CelebA", "-o", str(tmp_path), "-dpm", "gc"])      # penalty on public data without mean samples
    with pytest.raises(Exception, match="select only one"):
        options.parse(["CelebA", "-o", str(tmp_path), "-pss", "10", "-nms", "4"])
    with pytest.raises(Exception, match="IS per parameter"):
        options.parse(["CelebA", "-o", str(tmp_path), "-dpm", "is", "-nms", "4", "-issm", "constant-pl"])
    with pytest.raises(SystemExit):
        options.parse(["ImageNet"])


def test_options_resume_roundtrip(tmp_path):
    o = options.parse(["MNIST", "-o", str(tmp_path), "--manual_seed", "5"])
    with open(o.output_dir + "opt.txt", "w") as f:
        json.dump(o.__dict__, f)
    r = options.parse(["MNIST", "-rp", str(tmp_path), "-re", "3", "-dd", "cuda:0"])
    assert r.manual_seed == 5 and r.resume_epochs == 3 and r.d_device == "cuda:0" and r.output_dir == o.output_dir


def test_accountant_known_answers():
    # q = 1: plain Gaussian mechanism, RDP(alpha) = alpha / (2 sigma^2) per step
    for sigma in (0.5, 1.1, 4.0):
        for a in (1.5, 2, 8, 64.5):
            assert accountant.compute_rdp(1.0, sigma, 10, a) == pytest.approx(10 * a / (2 * sigma ** 2))
    assert accountant.compute_rdp(0.0, 1.0, 5, 4.0) == 0.0
    # integer order: closed form  1/(a-1) log sum_i C(a,i) (1-q)^(a-i) q^i exp((i^2-i)/(2 s^2))
    q, s, a = 0.01, 1.1, 6
    exact = math.log(sum(math.comb(a, i) * (1 - q) ** (a - i) * q ** i * math.exp((i * i - i) / (2 * s * s)) for i in range(a + 1))) / (a - 1)
    assert accountant.compute_rdp(q, s, 1, a) == pytest.approx(exact, rel=1e-10)
    # fractional orders interpolate smoothly between the neighbouring integers and RDP grows with the order
    orders = [5.0, 5.5, 6.0]
    r = accountant.compute_rdp(q, s, 1, orders)
    assert r[0] < r[1] < r[2] and abs(r[1] - 0.5 * (r[0] + r[2])) < 0.05 * r[2]
    # composition is linear in steps; epsilon decreases with sigma and increases with steps
    alphas = accountant.DEFAULT_ALPHAS
    e = [accountant.get_privacy_spent(alphas, accountant.compute_rdp(128 / 180000, sg, 1000, alphas), 1e-6)[0] for sg in (0.5, 1.0, 2.0)]
    assert e[0] > e[1] > e[2] > 0
    e1 = accountant.get_privacy_spent(alphas, accountant.compute_rdp(0.01, 1.1, 100, alphas), 1e-5)[0]
    e2 = accountant.get_privacy_spent(alphas, accountant.compute_rdp(0.01, 1.1, 1000, alphas), 1e-5)[0]
    assert e2 > e1
    # classic conversion is never tighter than the improved one by more than the log terms allow
    rdp = accountant.compute_rdp(0.01, 1.1, 1000, alphas)
    assert accountant.get_privacy_spent(alphas, rdp, 1e-5, improved=False)[0] >= accountant.get_privacy_spent(alphas, rdp, 1e-5)[0]


def test_logger_matches_reference_output(tmp_path, golden_dir):
    from csl_gan_amd.logger import Logger
    path = str(tmp_path / "log.csv")
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        lg = Logger("A: {:4.4f} | B: {:3.1f}", ["A", "B"], 4, path)
        for i in range(8):
            lg.stats["A"] += 0.25 * i
            lg.stats["B"] += 10.0 + i
            if (i + 1) % 4 == 0:
                lg.log(i // 4, 50.0 * (i // 4))
        lg.close()
    exp = open(os.path.join(golden_dir, "logger_expected.txt")).read()
    csv_exp, out_exp = exp.split("#STDOUT\n")
    assert open(path).read().replace("\r\n", "\n") == csv_exp[len("#CSV\n"):].replace("\r\n", "\n")
    assert buf.getvalue() == out_exp


def test_mean_sampler_sample_and_cost():
    from csl_gan_amd.mean_sampler import MeanSampler
    ms = MeanSampler(noise_std=0.12, num_samples=4, mean_size=1000, dataset_size=180000, n_classes=2, smallest_class_size=70000)
    ms.mean_samples = torch.arange(2 * 4, dtype=torch.float32).view(2, 4, 1, 1, 1).expand(2, 4, 3, 8, 8).clone()
    labels = torch.tensor([0, 1, 1, 0, 1, 0])
    r, y = ms.sample(6, noise_std=0, noise_mean_std=0, requested_labels=labels)
    assert torch.equal(y, labels) and r.shape == (6, 3, 8, 8)
    base = r[:, 0, 0, 0]
    assert all(4 * int(l) <= v < 4 * int(l) + 4 for v, l in zip(base.tolist(), labels.tolist()))   # right class row
    assert sorted(base[:4].tolist() - 4 * labels[:4].float().numpy()) == [0, 1, 2, 3]                # a permutation per 4 draws
    r2, _ = ms.sample(6, requested_labels=labels)
    assert 0 < (r2 - ms.mean_samples[labels, (r2[:, 0, 0, 0] * 0).long()]).abs().mean() < 10      # jitter applied
    eps, alpha = ms.get_privacy_cost(target_delta=1e-6)
    assert eps > 0 and alpha > 1
    ms1 = MeanSampler(noise_std=0.24, num_samples=4, mean_size=1000, dataset_size=180000, n_classes=2, smallest_class_size=70000)
    assert ms1.get_privacy_cost(1e-6)[0] < eps
    # the device-side draw stream survives a checkpoint (saved inside the engine-state JSON by csl_gan_amd.train)
    ms._seed, ms._draws = 0x1234ABCD5678, 17
    ms1.load_state_dict(json.loads(json.dumps(ms.state_dict())))
    assert (ms1._seed, ms1._draws) == (0x1234ABCD5678, 17)


def test_state_dict_keys_and_shapes_match_reference_layout():
    from types import SimpleNamespace as NS
    from csl_gan_amd import init_util
    opt = NS(dataset="CelebA", model="DeepConvResNet", im_size=64, conditional=False, n_classes=2, per_sample_grad=True,
             weights_seed=42, manual_seed=1, g_latent_dim=128, g_label_emb_mode="concat", d_label_emb_mode="concat",
             conditional_arch="ACGAN", aux_loss_type="wasserstein", aux_loss_scalar=1, g_device="cpu", d_device="cpu")
    G, D = init_util.init_models(opt)
    assert list(D.state_dict().keys()) == ["blocks.%d.%s" % (i, n) for i in range(4) for n in ("weight", "bias")] + ["linOut.weight"]
    assert [tuple(p.shape) for p in D.parameters()] == [(64, 3, 5, 5), (64,), (128, 64, 5, 5), (128,), (256, 128, 5, 5), (256,),
                                                        (512, 256, 5, 5), (512,), (1, 8192)]
    assert sum(p.numel() for p in D.parameters()) == 4314752 and sum(p.numel() for p in G.parameters()) == 21057859
    gk = list(G.state_dict().keys())
    assert gk[:2] == ["linIn.weight", "linIn.bias"] and "blocks.0.shortcut.conv.weight" in gk and "blocks.3.convUp.conv.weight" in gk
    assert "blocks.0.convUp.conv.bias" not in gk and gk[-2:] == ["convOut.weight", "convOut.bias"]


def test_cli_config0_mnist_vanilla_cpu_no_dp(tmp_path):
    """BASELINE configs[0]: MNIST vanilla GAN, -gd cpu -dd cpu, bs=64, no DP — plumbing without a GPU."""
    from csl_gan_amd import train
    tr = train.main(["MNIST", "-bs", "64", "-gd", "cpu", "-dd", "cpu", "-o", str(tmp_path), "--max_iters", "12", "--synthetic",
                     "--log_every", "256", "--manual_seed", "3"])
    rows = open(str(tmp_path / "log.csv")).read().strip().splitlines()
    assert rows[0].startswith("Epoch,Batch,G Adv Loss,D Adv Loss") and len(rows) >= 3
    vals = [float(v) for v in rows[1].split(",")[2:]]
    assert all(np.isfinite(vals)) and 0 < vals[1] < 5
    assert os.path.exists(str(tmp_path / "saves" / "D-1")) and os.path.exists(str(tmp_path / "opt.txt"))
    ck = torch.load(str(tmp_path / "saves" / "D-1"), weights_only=True)       # the no-code loader reads our checkpoints
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "loss"}
    # resume through util.load_model (weights_only=True inside): weights and Adam state come back
    from csl_gan_amd import util
    from csl_gan_amd.MNIST_models import MNISTVanillaD
    from csl_gan_amd.engine import HipAdam
    D2 = MNISTVanillaD()
    opt2 = HipAdam(D2.parameters(), lr=1e-3)
    assert util.load_model(str(tmp_path / "saves" / "D-1"), D2, "cpu", opt2) == ck["epoch"]
    for a, b in zip(D2.parameters(), tr.D.parameters()):
        assert torch.equal(a, b)
    assert len(opt2.state_dict()["state"]) == len(list(D2.parameters()))


def test_cli_profile_flag_runs_torch_profiler(tmp_path, capsys):
    """-p (train.py:555-563): the loop runs under torch.profiler and the key-averages table is printed."""
    from csl_gan_amd import train
    train.main(["MNIST", "-bs", "16", "-gd", "cpu", "-dd", "cpu", "-o", str(tmp_path), "--max_iters", "9", "--synthetic",
                "--log_every", "4096", "--manual_seed", "3", "-p"])
    out = capsys.readouterr().out
    assert "Self CPU" in out and "Finished training." in out


def test_dist_sampler_partitions_the_private_set():
    """--dist: every rank draws a disjoint share of one shared permutation per epoch (csl_gan_amd.data._private_loader); the
    union of the ranks' batches of a step is world x batch_size DISTINCT private samples, the assumption behind q = R*B/N."""
    from types import SimpleNamespace as NS
    from csl_gan_amd import data
    o = NS(dataset="MNIST", train_set_size=256, public_set_size=0, im_size=28, manual_seed=5, batch_size=16, synthetic=True,
           data_path=None, synthetic_cap=256, dist_data_seed=5)
    loaders = [data.init_data(o, rank=r, world=4)[1] for r in range(4)]
    x_all = loaders[0].dataset.tensors[0]
    key = {tuple(x_all[i].flatten()[:6].tolist()): i for i in range(len(x_all))}

    def epoch_indices(dl, epoch):
        dl.sampler.set_epoch(epoch)
        return [[key[tuple(img.flatten()[:6].tolist())] for img in batch] for batch, _ in dl]
    per_rank = [epoch_indices(dl, 0) for dl in loaders]
    assert all(len(b) == 4 for b in per_rank)                                   # 256 / 4 ranks / 16 per batch
    for step in range(4):
        idx = [i for r in range(4) for i in per_rank[r][step]]
        assert len(idx) == 64 and len(set(idx)) == 64                             # one global batch: all distinct
    flat = [i for r in range(4) for b in per_rank[r] for i in b]
    assert sorted(flat) == list(range(256))                                      # an epoch covers the set exactly once
    again = [epoch_indices(dl, 1) for dl in loaders]
    assert again != per_rank                                                      # set_epoch reshuffles
    one = data.init_data(o)[1]
    assert not hasattr(one.sampler, "set_epoch") or one.sampler.__class__.__name__ == "RandomSampler"


def test_moving_avg_scaling_needs_the_beta_the_reference_never_defines(tmp_path):
    from types import SimpleNamespace as NS
    from csl_gan_amd.trainer import Trainer
    t = Trainer.__new__(Trainer)
    t.opt = NS(moving_avg_beta=None)
    t.privacy_engine, t.D = NS(scaling_vec=[1.0, 2.0]), None
    with pytest.raises(AttributeError, match="moving_avg_beta"):
        t.update_sens_moving_avg()
    p1, p2 = torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(9))
    p1.grad, p2.grad = torch.full((4,), 3.0), None
    t.D = NS(parameters=lambda: [p1, p2])
    got = {}
    t.privacy_engine = NS(scaling_vec=[1.0, 2.0], set_scaling_vec=lambda v: got.setdefault("v", v))
    t.opt = NS(moving_avg_beta=0.9)
    t.update_sens_moving_avg()
    assert got["v"] == pytest.approx([0.9 * 1.0 + 0.1 * 6.0, 0.9 * 2.0])


def test_dp_engine_refuses_cpu_modules():
    from csl_gan_amd.engine import PrivacyEngine
    from csl_gan_amd.MNIST_models import MNISTVanillaD
    with pytest.raises(RuntimeError, match="HIP device"):
        PrivacyEngine(MNISTVanillaD(), batch_size=4, sample_size=100, alphas=[2, 3], noise_multiplier=1.0, max_grad_norm=1.0)


def test_budget_analysis_and_engine_state_roundtrip(tmp_path, capsys):
    from csl_gan_amd import budget_analysis
    o = options.parse(["CelebA", "-o", str(tmp_path), "-dpm", "gc", "-nms", "4", "--sigma", "1.0"])
    with open(o.output_dir + "opt.txt", "w") as f:
        json.dump(o.__dict__, f)
    capsys.readouterr()
    budget_analysis.main([str(tmp_path), "2"])
    eps2 = float(capsys.readouterr().out.strip().strip("()").split(",")[0])
    budget_analysis.main([str(tmp_path), "4"])
    eps4 = float(capsys.readouterr().out.strip().strip("()").split(",")[0])
    assert 0 < eps2 < eps4


def test_real_data_loaders_on_generated_files(tmp_path):
    """CelebA-style JPEG folder + attribute file and MNIST idx files, generated here (no datasets in the image)."""
    import struct
    from PIL import Image
    from csl_gan_amd import datasets as ds
    root = tmp_path / "celeba"
    root.mkdir()
    rng = np.random.default_rng(0)
    for i in range(1, 7):
        arr = np.full((218, 178, 3), 20 * i, dtype=np.uint8)
        arr[:, :89, 0] = 255                      # left half red: a flip is detectable
        Image.fromarray(arr).save(str(root / ("%06d.jpg" % i)), quality=95)
    attr = tmp_path / "attr.txt"
    names = ["A", "Male", "Z"]
    with open(attr, "w") as f:
        f.write("6\n" + " ".join(names) + "\n")
        for i in range(1, 7):
            f.write("%06d.jpg  %d %d %d\n" % (i, -1, 1 if i % 2 == 0 else -1, 1))
    d = ds.CelebADataset(str(root), im_size=64, length=4, offset=2, attr_file=str(attr), attr="Male", flip=False)
    assert len(d) == 4 and d.labels.tolist() == [0, 1, 0, 1] and d.label_true_count == 2
    img, lab = d[0]                                # file 000003.jpg, attribute row 3 -> odd -> 0
    assert img.shape == (3, 64, 64) and lab == 0 and -1.0 <= img.min() and img.max() <= 1.0
    assert abs(img[1, 32, 48].item() - (60 / 255 - 0.5) / 0.5) < 0.05          # right half keeps the grey level of file 3
    assert img[0, 32, 5].item() > 0.9                                          # left half red, not flipped
    got, lab1 = d.get_item_with_label(1)
    assert lab1 == 1
    flips = ds.CelebADataset(str(root), im_size=64, length=6, flip=True, seed=1)
    assert flips.labels is None and len({float(flips[0][0][0, 32, 5] > 0.9) for _ in range(20)}) == 2

    mroot = tmp_path / "mnist"
    mroot.mkdir()
    x = rng.integers(0, 256, size=(40, 28, 28), dtype=np.uint8)
    y = (np.arange(40) % 10).astype(np.uint8)
    for stem in ("train", "t10k"):
        with open(mroot / (stem + "-images-idx3-ubyte"), "wb") as f:
            f.write(struct.pack(">IIII", 0x00000803, 40, 28, 28) + x.tobytes())
        with open(mroot / (stem + "-labels-idx1-ubyte"), "wb") as f:
            f.write(struct.pack(">II", 0x00000801, 40) + y.tobytes())
    m = ds.MNISTDataset(str(mroot), train=True, per_class=2)
    assert len(m) == 20 and sorted(m.y.tolist()) == sorted(list(range(10)) * 2)
    xi, yi = m[3]
    assert xi.shape == (1, 28, 28) and 0.0 <= xi.min() and xi.max() <= 1.0
    assert m.get_item_with_label(7)[1] == 7


def test_dense_wgrad_group_and_gram_policy():
    """Host-side launch policy (pure arithmetic, no device): slab sizes divide the batch, small launches keep one slab per
    sample, and the Gram-norm path is chosen only for few-pixel layers."""
    from csl_gan_amd import ops
    for N in (1, 6, 128, 384):
        for (K, C, R, PQ) in ((512, 256, 5, 16), (64, 3, 5, 1024), (1, 8192, 1, 1), (64, 32, 5, 4096)):
            g = ops.dense_wgrad_group(N, K, C, R, R, PQ)
            assert g >= 1 and N % g == 0
    assert ops.dense_wgrad_group(6, 128, 64, 5, 5, 256) == 1
    assert ops.dense_wgrad_group(128, 512, 256, 5, 5, 16) > 1
    # critic head conv: 8x8x256 -> 4x4x512, stride 2
    assert ops.gram_norms_preferred((128, 4, 4, 512), (128, 8, 8, 256), 2)
    assert ops.gram_norms_preferred((128, 1, 1, 1), (128, 1, 1, 8192), 1)            # linear layer
    assert ops.gram_norms_preferred((128, 8, 8, 256), (128, 16, 16, 128), 2)          # conv3: 64 output / 64 class pixels (cls64 kernel)
    assert not ops.gram_norms_preferred((128, 16, 16, 128), (128, 32, 32, 64), 2)     # 256 output pixels: product kernel
    assert not ops.gram_norms_preferred((128, 4, 4, 100), (128, 8, 8, 256), 2)        # K % 64 != 0


def test_trainer_reset_stats_drops_pending_device_sums(tmp_path):
    """train.py:566,576 reset the logger at every epoch start; the build keeps running sums on the device between log lines,
    so the reset must drop those too (they used to leak into the next epoch's first log line: 160 % accuracies)."""
    import torch
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    opt = options.parse(["MNIST", "--model", "Vanilla", "-bs", "4", "-gd", "cpu", "-dd", "cpu", "-o", str(tmp_path), "--g_latent_dim", "8"])
    G, D = init_util.init_models(opt)
    tr = Trainer(opt, G, D, log_to=str(tmp_path / "log.csv"))
    tr._acc("D Real Acc", torch.tensor(50.0))
    tr._acc("_d_adv_gate", torch.tensor(1.0))
    tr.reset_stats()
    tr.flush_stats()
    assert tr.logger.stats["D Real Acc"] == 0
    assert "_d_adv_gate" in tr.dev_stats            # the G-step gate is not a logged statistic
    tr._acc("D Real Acc", torch.tensor(50.0))
    tr.flush_stats()
    assert tr.logger.stats["D Real Acc"] == 50.0


def test_make_grid_layout_and_png(tmp_path):
    """torchvision.utils.make_grid / save_image restated (torchvision absent -> PNG bytes are parity-unpinned; the layout is the
    published one: nrow images per row, 2-pixel zero border, x255 + 0.5 rounding)."""
    from PIL import Image
    from csl_gan_amd import util
    imgs = torch.arange(5 * 1 * 4 * 3, dtype=torch.float32).reshape(5, 1, 4, 3) / 60.0
    g = util.make_grid(imgs, nrow=2)
    assert g.shape == (3, 3 * (4 + 2) + 2, 2 * (3 + 2) + 2)
    assert torch.equal(g[0, 2:6, 2:5], imgs[0, 0]) and torch.equal(g[2, 2 + 6:6 + 6, 2 + 5:5 + 5], imgs[3, 0])
    assert g[:, :2].abs().sum() == 0 and g[:, 14:18, 7:].abs().sum() == 0        # border; the empty 6th cell
    p = str(tmp_path / "g.png")
    util.save_image(imgs, p, nrow=2)
    arr = np.asarray(Image.open(p))
    assert arr.shape == (20, 12, 3) and arr.dtype == np.uint8
    assert arr[2, 3, 0] == int(imgs[0, 0, 0, 1] * 255 + 0.5)


def test_cli_writes_sample_grids(tmp_path):
    """train.py:545-546,584-585: sample() every sample_every images and at the epoch boundary; conditional -> n_classes per row."""
    import os
    from PIL import Image
    from csl_gan_amd import train as T
    out = str(tmp_path / "run")
    T.main(["MNIST", "--conditional", "-bs", "8", "-gd", "cpu", "-dd", "cpu", "-o", out, "--synthetic", "--max_iters", "4", "--manual_seed", "3",
            "--sample_every", "16", "--sample_num", "20", "--log_every", "4096"])
    files = sorted(os.listdir(os.path.join(out, "samples")))
    assert files == ["1-1.png", "1-3.png"], files
    im = Image.open(os.path.join(out, "samples", "1-1.png"))
    assert im.size == (10 * 30 + 2, 2 * 30 + 2)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` outside a torchrun environment starts N ranks itself: rank 0's single JSON
    line reports the world size the process group saw, and a failing rank makes the parent exit non-zero.  CPU rehearsal on
    gloo through the hidden --launcher-selftest mode (the real step needs a GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest", "ok"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert json.loads(lines[0]) == {"n_gpus": 2, "sum": 2.0}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launcher-selftest", "fail"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    # inside a torchrun environment a --gpus / WORLD_SIZE mismatch is an error
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], env=env1, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 2 and "process group has 1 rank" in r.stderr
