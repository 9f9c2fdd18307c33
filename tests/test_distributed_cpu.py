"""world_size-2 gloo tests (CPU) of the N>1 path: the flat-bucket all-reduce of clipped+noised
gradients, the per-rank noise scaling, and the adaptive-clip statistic averaging.
The per-rank compute is the CPU oracle (tests may use it); what is under test is
csl_gan_amd.distributed and the sharding arithmetic of SURVEY.md §8e."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from csl_gan_amd import distributed as D
    from oracle import dp_engine as E
    from oracle.nets import build_models
    w, r, _ = D.init("gloo")
    assert (w, r) == (world, rank)
    _, Dm = build_models(dataset="MNIST", model="Vanilla", init_G=False)
    params = list(Dm.parameters())
    Bl, C, sigma = 4, 0.7, 2.0
    g = torch.Generator().manual_seed(100)
    x_all = torch.rand(world * Bl, 1, 28, 28, generator=g)
    x = x_all[rank * Bl:(rank + 1) * Bl]
    gs = E.per_sample_grads_microbatch(Dm, lambda M, xb, yb: M.real_loss(M(xb)[0]), x)
    summed = E.clip_and_sum([t.unsqueeze(0) for t in gs], C, accum_passes=False, num_private_passes=None)
    # per-rank noise: unit normals z_r scaled by sigma*C/sqrt(R); pre-scale by 1/(B_local*R); one flat bucket
    zs = [torch.randn(p.shape, generator=torch.Generator().manual_seed(7 + 13 * rank + i)) for i, p in enumerate(params)]
    flat = torch.cat([((s + z * sigma * C / world ** 0.5) / (Bl * world)).reshape(-1) for s, z in zip(summed, zs)])
    red = D.FlatGradReducer()
    red(flat)
    assert red.bytes_reduced == flat.numel() * 4
    stat = torch.tensor([float(rank + 1), 10.0 * (rank + 1)])
    D.average_across_ranks(stat)
    mx = torch.tensor([float(rank)])
    D.average_across_ranks(mx, use_max=True)
    D.barrier()
    if rank == 0:
        # single-process reference on the concatenated batch with the summed noise
        gs_all = E.per_sample_grads_microbatch(Dm, lambda M, xb, yb: M.real_loss(M(xb)[0]), x_all)
        summed_all = E.clip_and_sum([t.unsqueeze(0) for t in gs_all], C, accum_passes=False, num_private_passes=None)
        z_tot = []
        for i, p in enumerate(params):
            z_tot.append(sum(torch.randn(p.shape, generator=torch.Generator().manual_seed(7 + 13 * rr + i)) for rr in range(world)) / world ** 0.5)
        ref = torch.cat([((s + z * sigma * C) / (Bl * world)).reshape(-1) for s, z in zip(summed_all, z_tot)])
        np.save(os.path.join(out_dir, "res.npy"), np.stack([flat.numpy(), ref.numpy()]))
        np.save(os.path.join(out_dir, "stat.npy"), np.concatenate([stat.numpy(), mx.numpy()]))
    dist.destroy_process_group()


def test_flat_allreduce_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got, ref = np.load(str(tmp_path / "res.npy"))
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-7)
    st = np.load(str(tmp_path / "stat.npy"))
    np.testing.assert_allclose(st, [1.5, 15.0, 1.0])


def test_single_process_helpers_are_noops():
    from csl_gan_amd import distributed as D
    assert D.env_world()[0] >= 1
    t = torch.tensor([1.0, 2.0])
    assert torch.equal(D.average_across_ranks(t.clone()), t)
    r = D.FlatGradReducer()
    assert r.world == 1 and torch.equal(r(t.clone()), t)


def _g_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from csl_gan_amd import distributed as D
    from csl_gan_amd import init_util, options
    from csl_gan_amd.trainer import Trainer
    D.init("gloo")
    opt = options.parse(["MNIST", "--model", "Vanilla", "-bs", "4", "-gd", "cpu", "-dd", "cpu", "-o", out_dir,
                         "--manual_seed", "1", "--g_latent_dim", "8", "--log_every", "100000"])
    G, Dm = init_util.init_models(opt)
    tr = Trainer(opt, G, Dm, log_to=os.path.join(out_dir, "log%d.csv" % rank), world_size=world, rank=rank)
    g = torch.Generator().manual_seed(5 + rank)                # each rank sees its own shard and its own z
    torch.manual_seed(100 + rank)
    before = [p.detach().clone() for p in G.parameters()]
    img = torch.rand(4, 1, 28, 28, generator=g) * 2 - 1
    tr.train(0, 0, img, torch.zeros(4, dtype=torch.long), use_dp=False)
    assert any(not torch.equal(a, b.detach()) for a, b in zip(before, G.parameters())), "the G step did not run"
    flat = torch.cat([p.detach().reshape(-1) for p in G.parameters()])
    grads = torch.cat([p.grad.reshape(-1) for p in G.parameters()])
    both = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(both, flat)
    gb = [torch.empty_like(grads) for _ in range(world)]
    dist.all_gather(gb, grads)
    if rank == 0:
        np.save(os.path.join(out_dir, "g.npy"), np.stack([t.numpy() for t in both] + [t.numpy() for t in gb]))
    dist.destroy_process_group()


def test_G_step_keeps_replicas_identical(tmp_path):
    """G is replicated; after a G step on different per-rank shards every rank holds the same G
    (one flat all-reduce of the G gradients, and a rank-consistent threshold gate)."""
    world = 2
    mp.spawn(_g_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a = np.load(tmp_path / "g.npy")
    assert np.array_equal(a[0], a[1]), "generator replicas diverged"
    assert np.array_equal(a[2], a[3]) and np.abs(a[2]).max() > 0, "ranks stepped with different G gradients"


def test_multi_rank_graph_capture_is_opt_in(monkeypatch):
    """Recording the step's RCCL collectives into a HIP graph is opt-in for N > 1 (CSLGAN_GRAPH_DIST=1): it has run on a one-rank
    RCCL group only, so `Trainer.setup_privacy_engine` must not pick it by default when a real multi-GPU job starts."""
    from csl_gan_amd import distributed as D
    monkeypatch.delenv("CSLGAN_GRAPH_DIST", raising=False)
    assert D.collectives_capturable()                       # one process: nothing to record but kernels
    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 8)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    assert not D.collectives_capturable()                   # RCCL, 8 ranks, not asked for -> eager step
    monkeypatch.setenv("CSLGAN_GRAPH_DIST", "1")
    assert D.collectives_capturable()
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "gloo")
    assert not D.collectives_capturable()                   # gloo stages through the host: never capturable
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 1)
    assert D.collectives_capturable()
