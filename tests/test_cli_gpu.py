"""The command line on one MI355X (-m gpu): the reference's smoke matrix (test_configs.sh: {MNIST, CelebA} x {gc, is} x
{unconditional, conditional}) on synthetic data with the DEFAULT launch mode (HIP-graph replay of the gc D-step, eager generator
steps in between), and checkpoint + resume of a graph-replayed run."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(tmp_path, argv):
    from csl_gan_amd import train
    return train.main(argv + ["-gd", "cuda:0", "-dd", "cuda:0", "-o", str(tmp_path), "--synthetic", "--manual_seed", "7"])


@pytest.mark.parametrize("dataset", ["MNIST", "CelebA"])
@pytest.mark.parametrize("mode", ["gc", "is"])
@pytest.mark.parametrize("cond", [False, True])
def test_cli_smoke_matrix(tmp_path, dataset, mode, cond):
    bs, iters = 32, 9
    tr = _run(tmp_path, [dataset, "-tss", "1000", "-dpm", mode, "-nms", "1", "--mean_sample_size", "10", "-bs", str(bs), "--max_iters", str(iters),
                         "--log_every", str(bs * 3)] + (["--conditional"] if cond else []))
    torch.cuda.synchronize()
    assert tr.privacy_engine.steps == iters
    assert tr.graphed is not None and tr.graphed.graph is not None, "the D-step should have been recorded and replayed"
    for p in list(tr.D.parameters()) + list(tr.G.parameters()):
        assert torch.isfinite(p).all()
    rows = open(os.path.join(str(tmp_path), "log.csv")).read().strip().splitlines()
    assert len(rows) >= 3 and "nan" not in rows[-1].lower()      # header + one line per 3 iterations


def test_cli_checkpoint_and_resume_with_graph_replay(tmp_path):
    """resuming a --hip_graph run raised at the first D-step (the device step counter came back from the checkpoint as a
    float).  Train one epoch with replayed D-steps, save, resume for another epoch: the optimizer state has torch's Adam layout, the
    engine's step count and Philox counter and the mean sampler's draw stream continue."""
    base = ["MNIST", "--model", "DeepConvResNet", "--penalty", "WGAN-GP", "-tss", "256", "-dpm", "gc", "-nms", "2", "--mean_sample_size", "10",
            "-bs", "32", "--save_every", "1", "--log_every", "64", "--g_latent_dim", "16"]
    tr = _run(tmp_path, base + ["-ne", "1"])
    assert tr.graphed is not None and tr.graphed.graph is not None
    steps1, draws1 = tr.privacy_engine.steps, tr.mean_sampler._draws
    assert steps1 == 8 and draws1 > 0
    ck = torch.load(os.path.join(str(tmp_path), "saves", "D-1"), weights_only=True)
    for st in ck["optimizer_state_dict"]["state"].values():
        assert set(st) == {"step", "exp_avg", "exp_avg_sq"}
    with open(os.path.join(str(tmp_path), "saves", "PE-1.json")) as f:
        pe_state = json.load(f)
    assert pe_state["steps"] == steps1 and pe_state["mean_sampler"]["draws"] == draws1
    from csl_gan_amd import train
    tr2 = train.main(["MNIST", "-rp", str(tmp_path), "-re", "1", "-gd", "cuda:0", "-dd", "cuda:0", "-ne", "2", "-ka", "n_epochs"])
    torch.cuda.synchronize()
    assert tr2.privacy_engine.steps == 2 * steps1 and tr2.mean_sampler._draws > draws1
    assert tr2.graphed is not None and tr2.graphed.graph is not None
    # the reference re-creates both optimizers after the warm-up loop (train.py:572), i.e. AFTER load_model restored their state
    # (train.py:79-82): Adam's moments restart on resume there and here (the optimizer round trip itself is
    # test_graph_gpu.py::test_capturable_adam_checkpoint_roundtrip)
    assert all(st["step"] == steps1 for st in tr2.d_optimizer.state.values())
    assert all(torch.isfinite(p).all() for p in tr2.D.parameters())


@pytest.mark.parametrize("extra", [["--im_size", "128"], ["--conditional"], ["-dpm", "is"]])
def test_cli_bf16_storage(tmp_path, extra):
    """BASELINE configs[4] from the command line: --compute_dtype bf16 --storage_dtype bf16 (bf16-stored activations of the critic and of
    the frozen generator, eager generator steps differentiating through them) — the 128x128 extension, the conditional (ACGAN) critic
    whose first layer sees label planes, and the immediate-sensitivity engine's double backward."""
    bs, iters = 16, 7
    argv = ["CelebA", "-tss", "1000", "-nms", "1", "--mean_sample_size", "10", "-bs", str(bs), "--max_iters", str(iters), "--log_every", str(bs * 3),
            "--compute_dtype", "bf16", "--storage_dtype", "bf16"]
    if "-dpm" not in extra:
        argv += ["-dpm", "gc"]
    tr = _run(tmp_path, argv + extra)
    torch.cuda.synchronize()
    from csl_gan_amd import ops
    assert ops.get_storage_dtype() == "bf16" and tr.privacy_engine.steps == iters
    for p in list(tr.D.parameters()) + list(tr.G.parameters()):
        assert torch.isfinite(p).all()
    rows = open(os.path.join(str(tmp_path), "log.csv")).read().strip().splitlines()
    assert len(rows) >= 3 and "nan" not in rows[-1].lower()


def test_storage_dtype_needs_bf16_compute(tmp_path):
    from csl_gan_amd import options
    with pytest.raises(Exception, match="needs --compute_dtype bf16"):
        options.parse(["CelebA", "-dpm", "gc", "-nms", "1", "-bs", "4", "-gd", "cuda:0", "-dd", "cuda:0", "-o", str(tmp_path), "--storage_dtype", "bf16"])
