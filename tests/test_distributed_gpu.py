"""The N > 1 path through the REAL engine (-m gpu): two ranks (gloo, both on cuda:0 — the box has one GPU) drive
Trainer.train_D -> PrivacyEngine._before_step -> FlatGradReducer with world_size=2 on their own shards; the reduced
gradient must equal ONE process running the concatenated batch:

    rank r:  (sum_b f_b g_b + B*pen_grad_r + (sigma*C/sqrt(R)) z_r) / (B*R)   --all-reduce(SUM)-->
    single:  (sum over all 2B samples + 2B*pen_grad + sigma*C * (z_0 + z_1)/sqrt(2)) / (2B)

with the adaptive clip statistics averaged across ranks (mean over the global batch), the accountant told the GLOBAL
sample rate, rank-keyed Philox seeds, and — immediate-sensitivity mode — the batch maximum taken over all ranks' samples
(all-reduce MAX) before the noise is scaled.  SURVEY.md §8(e); csl_gan_amd/distributed.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

BL, LATENT = 4, 16


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _inputs(world):
    g = torch.Generator().manual_seed(123)
    n = world * BL
    return dict(img=torch.rand(n, 1, 28, 28, generator=g), ms_a=torch.rand(n, 1, 28, 28, generator=g) * 0.6,
                ms_p=torch.rand(n, 1, 28, 28, generator=g) * 0.6, z=torch.randn(n, LATENT, generator=g), alpha=torch.rand(n, generator=g))


def _trainer(out, mode, B, world, rank, reducer):
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    extra = ["-gcm", "adaptive-pl", "--materialize", "all"] if mode == "gc" else ["-ispp", "True"]
    opt = options.parse(["MNIST", "--model", "DeepConvResNet", "-dpm", mode, "-nms", "4", "-bs", str(B), "-gd", "cuda:0", "-dd", "cuda:0",
                         "-o", out, "--manual_seed", "1", "--g_latent_dim", str(LATENT), "--sigma", "0.8", "--penalty", "WGAN-GP"] + extra)
    G, D = init_util.init_models(opt)
    tr = Trainer(opt, G, D, log_to=os.path.join(out, "log%d.csv" % rank), world_size=world, rank=rank, grad_reducer=reducer)
    pe = tr.setup_privacy_engine()
    return opt, tr, pe, D


def _noise(D, rank):
    return [torch.randn(p.numel(), generator=torch.Generator().manual_seed(1000 + 17 * rank + i)) for i, p in enumerate(D.parameters())]


def _step(tr, pe, D, inp, sl, noise):
    tr.explicit = dict(ms_adapt=inp["ms_a"][sl], pen_real=inp["ms_p"][sl], alpha=inp["alpha"][sl], keep=True)
    pe.host_noise = noise
    tr.train_D(inp["img"][sl].cuda(), None, inp["z"][sl].cuda(), None, use_dp=True)
    torch.cuda.synchronize()
    return torch.cat([p.grad.detach().reshape(-1).cpu() for p in D.parameters()])


def _worker(rank, world, port, out, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    from csl_gan_amd import distributed as Dist
    w, r, _ = Dist.init("gloo")
    assert (w, r) == (world, rank)
    red = Dist.FlatGradReducer()
    opt, tr, pe, D = _trainer(os.path.join(out, "r%d" % rank), mode, BL, world, rank, red)
    assert pe.world_size == world and pe.grad_reducer is red
    assert pe.sample_rate == pytest.approx(world * BL / opt.train_set_size)         # the accountant sees the GLOBAL batch
    inp = _inputs(world)
    flat = _step(tr, pe, D, inp, slice(rank * BL, (rank + 1) * BL), _noise(D, rank))
    assert red.bytes_reduced == flat.numel() * 4 and pe.steps == 1
    res = {"grad": flat.numpy(), "seed": pe.seed}
    if mode == "gc":
        res["C"] = np.asarray(pe.max_grad_norm, dtype=np.float64)
    else:
        res["sens"] = np.asarray(pe.batch_sensitivity, dtype=np.float64)
    np.savez(os.path.join(out, "rank%d.npz" % rank), **res)
    Dist.barrier()
    torch.distributed.destroy_process_group()


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_two_rank_gc_step_equals_one_process_on_the_concatenated_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "gc"), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["grad"], r1["grad"]), "ranks hold different gradients after the all-reduce"
    assert np.array_equal(r0["C"], r1["C"]), "ranks clipped with different adaptive norms"
    assert int(r0["seed"]) != int(r1["seed"]), "Philox streams must be keyed by rank"
    # one process, global batch; its unit normals are the ranks' normals summed / sqrt(R)
    opt, tr, pe, D = _trainer(str(tmp_path / "single"), "gc", world * BL, 1, 0, None)
    z = [(a + b) / world ** 0.5 for a, b in zip(_noise(D, 0), _noise(D, 1))]
    ref = _step(tr, pe, D, _inputs(world), slice(0, world * BL), z).numpy()
    assert _rel(r0["C"], np.asarray(pe.max_grad_norm)) <= 1e-4
    off = 0
    for n, p in D.named_parameters():              # per tensor, against its own scale
        seg = slice(off, off + p.numel())
        off += p.numel()
        assert _rel(r0["grad"][seg], ref[seg]) <= 1e-3, "%s: %.3e" % (n, _rel(r0["grad"][seg], ref[seg]))


def test_two_rank_immediate_sensitivity_takes_the_maximum_over_all_ranks(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), "is"), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["grad"], r1["grad"]) and np.array_equal(r0["sens"], r1["sens"])
    # each shard alone (world 1): its local sensitivities; the distributed value must be their element-wise maximum
    local = []
    inp = _inputs(world)
    for r in range(world):
        opt, tr, pe, D = _trainer(str(tmp_path / ("solo%d" % r)), "is", BL, 1, r, None)
        _step(tr, pe, D, inp, slice(r * BL, (r + 1) * BL), _noise(D, r))
        local.append(np.asarray(pe.batch_sensitivity, dtype=np.float64))
    assert _rel(r0["sens"], np.maximum(local[0], local[1])) <= 1e-3
    assert (np.abs(local[0] - local[1]) > 1e-3 * np.abs(local[0]).max()).any(), "shards should differ for the test to mean anything"


def _graph_rank(port, out, q):
    """ONE rank over RCCL (backend nccl), the reducer forced to issue its all-reduce: the D-step is recorded into a HIP graph WITH the
    collective in it (trainer.setup_privacy_engine, N > 1 path) and replayed; an eager twin runs the same inputs."""
    import torch.distributed as dist
    from csl_gan_amd.distributed import FlatGradReducer, collectives_capturable
    from csl_gan_amd.trainer import GraphedDStep
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert collectives_capturable()
        res = {}
        for use_graph in (False, True):
            torch.manual_seed(11); torch.cuda.manual_seed(11)
            red = FlatGradReducer(always=True)
            from csl_gan_amd import init_util, options
            from csl_gan_amd.mean_sampler import MeanSampler
            from csl_gan_amd.trainer import Trainer
            o = os.path.join(out, "g%d" % use_graph)
            opt = options.parse(["MNIST", "--model", "DeepConvResNet", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "4", "-bs", "8", "-gd", "cuda:0",
                                 "-dd", "cuda:0", "-o", o, "--manual_seed", "1", "--g_latent_dim", str(LATENT), "--sigma", "0.8", "--penalty", "WGAN-GP"])
            G, D = init_util.init_models(opt)
            ms = MeanSampler(num_samples=4, mean_size=10, device="cuda:0", res=28, ch=1)
            ms.mean_samples = (torch.randn((1, 4, 1, 28, 28), generator=torch.Generator().manual_seed(9)) * 0.2).cuda()
            tr = Trainer(opt, G, D, mean_sampler=ms, log_to=os.path.join(o, "log.csv"), world_size=1, rank=0, grad_reducer=red)
            pe = tr.setup_privacy_engine()
            step = GraphedDStep(tr, use_graph=use_graph, warmup=1)
            g = torch.Generator().manual_seed(5)
            for i in range(4):
                step(torch.rand(8, 1, 28, 28, generator=g).cuda(), None)
            torch.cuda.synchronize()
            assert (step.graph is not None) == use_graph, step.capture_error
            res[use_graph] = ([p.detach().cpu().clone() for p in D.parameters()], red.bytes_reduced, pe.steps)
        q.put(("ok", res))
    except Exception as e:               # noqa: BLE001 — reported to the parent
        import traceback
        q.put(("err", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_graph_replay_records_the_rccl_all_reduce(tmp_path):
    """With RCCL the multi-GPU D-step replays from a HIP graph like the single-GPU one: the all-reduce of the flat gradient bucket
    and of the adaptive statistics are stream-ordered RCCL launches recorded with the step.  One rank on the box's one GPU, the
    collective forced: four steps recorded-and-replayed equal four eager steps (same seeds; weights after Adam at lr level)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_graph_rank, args=(_free_port(), str(tmp_path), q))
    p.start()
    status, res = q.get(timeout=300)
    p.join(60)
    assert status == "ok", res
    (we, be, se), (wg, bg, sg) = res[False], res[True]
    assert se == sg == 4 and be > 0 and bg > 0          # the eager twin reduced 4 buckets; the recorded one issued the call while recording
    for a, b in zip(we, wg):
        assert (a - b).abs().max().item() <= 2.5 * 1e-4 * 4 * 0.5 + 1e-7, (a - b).abs().max().item()   # d_lr 2e-4 (MNIST): a fraction of the 4 Adam steps


def _segments_worker(rank, world, port, out, mode):
    """One of two gloo ranks on the box's GPU: the D-step as GraphedDStep runs it for N > 1 by default — recorded as graphs that END at
    each collective — and an eager twin on the same inputs."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    torch.cuda.set_device(0)
    from csl_gan_amd import distributed as Dist, init_util, options
    from csl_gan_amd.mean_sampler import MeanSampler
    from csl_gan_amd.trainer import GraphedDStep, Trainer
    Dist.init("gloo")
    assert not Dist.collectives_capturable() and Dist.segments_enabled()
    res = {}
    for use_graph in (False, True):
        torch.manual_seed(11 + rank); torch.cuda.manual_seed(11 + rank)
        red = Dist.FlatGradReducer()
        o = os.path.join(out, "r%d_g%d" % (rank, use_graph))
        extra = ["-dpm", "gc", "-gcm", "adaptive-pl"] if mode == "gc" else ["-dpm", "is", "-ispp", "True"]
        opt = options.parse(["MNIST", "--model", "DeepConvResNet", "-nms", "4", "-bs", "8", "-gd", "cuda:0",
                             "-dd", "cuda:0", "-o", o, "--manual_seed", "1", "--g_latent_dim", str(LATENT), "--sigma", "0.8", "--penalty", "WGAN-GP",
                             "--hip_graph", "True"] + extra)
        G, D = init_util.init_models(opt)
        ms = MeanSampler(num_samples=4, mean_size=10, device="cuda:0", res=28, ch=1)
        ms.mean_samples = (torch.randn((1, 4, 1, 28, 28), generator=torch.Generator().manual_seed(9)) * 0.2).cuda()
        tr = Trainer(opt, G, D, mean_sampler=ms, log_to=os.path.join(o, "log.csv"), world_size=world, rank=rank, grad_reducer=red)
        pe = tr.setup_privacy_engine()
        assert tr.graphed is not None                  # N > 1 gets the replayed step by default now
        step = GraphedDStep(tr, use_graph=use_graph, warmup=1)
        g = torch.Generator().manual_seed(5 + rank)
        for i in range(5):
            step(torch.rand(8, 1, 28, 28, generator=g).cuda(), None)
        torch.cuda.synchronize()
        if use_graph:
            assert step.graph is not None and step.segmented, step.capture_error
            # gc: adaptive statistic, gradient bucket; is: batch-maximum sensitivities, gradient bucket -> two collectives, three graphs
            assert len(step.graphs) == 3 and len(step.between) == 2, (len(step.graphs), len(step.between))
        res[use_graph] = torch.cat([p.detach().reshape(-1).cpu() for p in D.parameters()]).numpy()
        assert pe.steps == 5 and red.bytes_reduced > 0
    np.savez(os.path.join(out, "seg_rank%d.npz" % rank), eager=res[False], graph=res[True])
    Dist.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("mode", ["gc", "is"])
def test_two_rank_step_replays_as_graph_segments_around_its_collectives(tmp_path, mode):
    """N > 1 without recording a collective: the step's two collectives (adaptive statistic, flat gradient bucket) end the capture;
    replays run graph - collective - graph - collective - graph.  Two gloo ranks (gloo cannot be captured at all) on the box's GPU:
    five replayed steps leave the critic where five eager steps leave it, and both ranks hold the same weights."""
    world = 2
    mp.spawn(_segments_worker, args=(world, _free_port(), str(tmp_path), mode), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "seg_rank0.npz"), np.load(tmp_path / "seg_rank1.npz")
    assert np.array_equal(r0["graph"], r1["graph"]) and np.array_equal(r0["eager"], r1["eager"]), "ranks diverged"
    assert np.abs(r0["graph"] - r0["eager"]).max() <= 2.5 * 1e-4 * 5 * 0.5 + 1e-7, np.abs(r0["graph"] - r0["eager"]).max()


@pytest.mark.parametrize("segments", [True, False], ids=["segments", "eager"])
def test_bench_two_ranks_reports_the_eager_step_by_default(tmp_path, segments):
    """`python bench.py --gpus 2` (two gloo ranks rehearsed on the box's one GPU): the line is printed once by rank 0, says n_gpus = 2,
    aggregates both ranks' images, and is labelled `launch: "hip_graph_segments"` (the default: no collective inside a graph) or, with
    CSLGAN_GRAPH_SEGMENTS=0, `launch: "eager"` — with no graph error either way."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CSLGAN_DIST_BACKEND="gloo", CSLGAN_FORCE_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
               CSLGAN_GRAPH_SEGMENTS="1" if segments else "0")
    env.pop("CSLGAN_GRAPH_DIST", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--loop-steps", "0", "--no-variants"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["launch"] == ("hip_graph_segments" if segments else "eager") and line["graph_error"] is None, line
    assert line["config"]["global_batch"] == 256 and abs(line["value"] - 2 * line["per_gpu"]) <= 0.02
    assert "cpu_baseline" not in line                       # rank 0 at N = 1 only
