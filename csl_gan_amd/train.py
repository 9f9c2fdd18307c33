"""Command-line entry: `python -m csl_gan_amd.train {MNIST,CelebA} [flags]` — the reference's
train.py (README.md:18-45 invocations) with its flags, opt.txt, output layout, warm-up, epoch loop,
epsilon log / budget stop and checkpoints (train.py:34-44, 53-73, 555-603).  The step functions live
in csl_gan_amd.trainer.Trainer.  Image-grid sampling and the code snapshot (train.py:41-44, 298-308)
are I/O cosmetics and are not carried."""
import csv
import json
import os
import sys

import torch

from . import data as syn_data
from . import distributed as dist_util
from . import init_util, options, util
from .mean_sampler import MeanSampler
from .trainer import Trainer


def build_mean_sampler(opt, dataset, rank=0, world=1):
    """train.py:53-73.  --dist: the mean samples are a DP release of the private data, paid for once in the privacy cost —
    so they are built ONCE (rank 0, over the whole private set) and broadcast; every rank then perturbs the same released
    samples (post-processing)."""
    if opt.num_mean_samples <= 0:
        return None, 0.0
    if world > 1 and rank != 0:
        n_cls = opt.n_classes if opt.conditional else 1
        ms = MeanSampler(noise_std=opt.mean_sample_noise_std, num_samples=opt.num_mean_samples, mean_size=opt.mean_sample_size,
                         dataset_size=opt.train_set_size, default_batch_size=opt.batch_size, n_classes=n_cls, res=opt.im_size,
                         ch=1 if opt.dataset == "MNIST" else 3, device=opt.d_device)
        shape = dist_util.broadcast_object(None)
        ms.mean_samples = dist_util.broadcast_tensor(torch.empty(shape, device=opt.d_device))
        cost = dist_util.broadcast_object(None)
        return ms, cost
    print("Generating mean samples...")
    keep = opt.batch_size
    n_cls = opt.n_classes if opt.conditional else 1
    opt.batch_size = opt.mean_sample_size * n_cls
    opt.batch_size = min(opt.batch_size, len(dataset))
    mean_loader = syn_data.init_data(opt)[1]
    opt.batch_size = keep
    smallest = None
    if opt.conditional:
        ltc = getattr(dataset, "label_true_count", None)
        smallest = min(ltc, opt.train_set_size - ltc) if (opt.dataset == "CelebA" and ltc is not None) else opt.train_set_size / opt.n_classes
    ms = MeanSampler(dataloader=mean_loader, dataset_size=opt.train_set_size, save_path=opt.output_dir + "mean_samples/",
                     noise_std=opt.mean_sample_noise_std, num_samples=opt.num_mean_samples, mean_size=opt.mean_sample_size,
                     default_batch_size=opt.batch_size, n_classes=n_cls, smallest_class_size=smallest, res=opt.im_size,
                     ch=1 if opt.dataset == "MNIST" else 3, device=opt.d_device)
    cost, _ = ms.get_privacy_cost(target_delta=opt.delta)
    print("Privacy Cost from Mean Samples:", cost)
    if world > 1:
        ms.mean_samples = ms.mean_samples.to(opt.d_device).contiguous()
        dist_util.broadcast_object(tuple(ms.mean_samples.shape))
        dist_util.broadcast_tensor(ms.mean_samples)
        dist_util.broadcast_object(cost)
    return ms, cost


def _save_engine(tr, path):
    if tr.privacy_engine is not None:
        st = tr.privacy_engine.state_dict()
        if tr.mean_sampler is not None:
            st["mean_sampler"] = tr.mean_sampler.state_dict()
        with open(path, "w") as f:
            json.dump(st, f)


def main(argv=None):
    opt = options.parse(argv)
    world, rank, local = dist_util.init() if getattr(opt, "dist", False) else (1, 0, 0)
    if world > 1:
        opt.g_device = opt.d_device = "cuda:%d" % local
        # one seed for the shared data permutation (--manual_seed -1 draws a different seed in every process)
        opt.dist_data_seed = dist_util.broadcast_object(int(opt.manual_seed))
    if rank == 0:
        with open(opt.output_dir + "opt.txt", "w") as f:
            json.dump(opt.__dict__, f)

    G, D = init_util.init_models(opt)
    if not getattr(opt, "synthetic", False) and not os.path.isdir(opt.data_path or ""):
        print("data_path %r not found: using the synthetic dataset (--synthetic)" % opt.data_path)
    dataset, dataloader, public_dataset, public_dataloader = syn_data.init_data(opt, rank, world)
    mean_sampler, mean_cost = build_mean_sampler(opt, dataset, rank, world)
    if world > 1:
        # init_models left every rank on the same RNG state: z, penalty alpha and mean-sample jitter must differ per rank
        # (the private batches already do, through the partitioned sampler)
        torch.manual_seed(int(opt.dist_data_seed) + 7919 * rank)
        torch.cuda.manual_seed(int(opt.dist_data_seed) + 7919 * rank)
    reducer = dist_util.FlatGradReducer() if world > 1 else None
    tr = Trainer(opt, G, D, dataset=dataset, public_dataloader=public_dataloader, public_dataset=public_dataset,
                 mean_sampler=mean_sampler, world_size=world, rank=rank, grad_reducer=reducer,
                 log_to=None if rank == 0 else opt.output_dir + "log_rank%d.csv" % rank)

    start_epoch = 0
    if opt.resume_epochs > 0:
        util.load_model(opt.resume_path + "saves/G-" + str(opt.resume_epochs), G, opt.g_device, tr.g_optimizer)
        start_epoch = util.load_model(opt.resume_path + "saves/D-" + str(opt.resume_epochs), D, opt.d_device, tr.d_optimizer)

    privacy_log = privacy_writer = None
    if opt.use_dp and rank == 0:
        privacy_log = open(opt.output_dir + "privacy_log.csv", "a")
        privacy_writer = csv.writer(privacy_log)
        if opt.resume_path is None:
            privacy_writer.writerow(["Epoch", "Epsilon"])
            privacy_log.flush()

    tr.init_fixed_samples()                                # train.py:256-261
    print("\nStarting training...\n")
    tr.reset_stats()
    for it in range(opt.warmup_iter):                      # train.py:567-569: public / mean samples, no DP
        img, labels = next(iter(public_dataloader)) if opt.public_set_size > 0 else mean_sampler.sample(opt.batch_size)
        tr.train(-1, it, img, labels if labels is not None else torch.zeros(len(img), dtype=torch.long), use_dp=False)
    tr.g_optimizer, tr.d_optimizer = tr.init_optimizers()  # train.py:572
    if opt.use_dp:
        pe = tr.setup_privacy_engine()
        pe_path = (opt.resume_path or "") + "saves/PE-" + str(opt.resume_epochs) + ".json"
        if opt.resume_epochs > 0 and os.path.exists(pe_path):      # extension: the reference restarts epsilon at 0
            with open(pe_path) as f:
                st = json.load(f)
            pe.load_state_dict(st)
            if mean_sampler is not None and "mean_sampler" in st:
                mean_sampler.load_state_dict(st["mean_sampler"])

    # train.py:555-563, 538-539: -p wraps the training loop in torch.profiler (wait 1 / warm-up 1 / active 5 steps, one
    # profiler.step() per batch) and prints the key-averages table sorted by self CPU time, row_limit = n_classes
    profiler = None
    if opt.profile_training:
        from torch.profiler import ProfilerActivity, profile, schedule

        def trace_handler(p):
            print(p.key_averages().table(sort_by="self_cpu_time_total", row_limit=opt.n_classes))
        acts = [ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if torch.cuda.is_available() else [])
        profiler = profile(activities=acts, schedule=schedule(wait=1, warmup=1, active=5), on_trace_ready=trace_handler)
        profiler.__enter__()

    iters, epoch, eps = 0, start_epoch, 0.0
    for epoch in range(opt.resume_epochs, opt.n_epochs):
        tr.reset_stats()
        if hasattr(dataloader.sampler, "set_epoch"):
            dataloader.sampler.set_epoch(epoch)        # --dist: a new shared permutation, still disjoint across ranks
        batch_i = 0
        for batch_i, (img, labels) in enumerate(dataloader):
            tr.train(epoch, batch_i, img, labels, use_dp=opt.use_dp)
            if profiler is not None:
                profiler.step()
            iters += 1
            if opt.max_iters and iters >= opt.max_iters:
                break
        if opt.log_every_epochs > 0 and (epoch + 1) % opt.log_every_epochs == 0:
            tr.log(epoch, 100)
        if opt.sample_every_epochs > 0 and (epoch + 1) % opt.sample_every_epochs == 0:
            tr.sample(epoch, batch_i)
        if opt.use_dp:
            eps, _ = tr.privacy_engine.get_privacy_spent(opt.delta)
            if privacy_writer is not None:
                privacy_writer.writerow([epoch, eps + mean_cost])
                privacy_log.flush()
            if opt.epsilon_budget is not None and eps > opt.epsilon_budget:
                break
        if (epoch + 1) % opt.save_every == 0 and rank == 0:
            util.save_model(epoch, D, tr.d_optimizer, 0, opt.output_dir + "saves/D-" + str(epoch + 1))
            util.save_model(epoch, G, tr.g_optimizer, 0, opt.output_dir + "saves/G-" + str(epoch + 1))
            _save_engine(tr, opt.output_dir + "saves/PE-" + str(epoch + 1) + ".json")
        if opt.max_iters and iters >= opt.max_iters:
            break

    if profiler is not None:
        profiler.__exit__(None, None, None)
    print("Finished training.")
    if rank == 0:
        util.save_model(opt.n_epochs, D, tr.d_optimizer, 0, opt.output_dir + "saves/D-" + str(epoch + 1))
        util.save_model(opt.n_epochs, G, tr.g_optimizer, 0, opt.output_dir + "saves/G-" + str(epoch + 1))
        _save_engine(tr, opt.output_dir + "saves/PE-" + str(epoch + 1) + ".json")
    tr.flush_stats()
    tr.logger.close()
    return tr


if __name__ == "__main__":
    main(sys.argv[1:])
