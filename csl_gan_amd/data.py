"""Synthetic in-memory datasets with the reference DataLoader contract (init_util.py:13-42 returns
(dataset, dataloader, public_dataset, public_dataloader); batches are (images, labels)).

The container and the GPU box hold no MNIST / CelebA files and there is no network, so the CLI and
the benchmark train on synthetic tensors of the right shape and range:
CelebA-like: clamp(N(0, 0.5^2), -1, 1), 3 x im_size x im_size, binary label ~ Bernoulli(0.42);
MNIST-like:  U[0,1], 1 x 28 x 28, label ~ randint(10).  Real-data loaders are §8f item 4 (out of scope).
"""
import os

import torch
from torch.utils.data import DataLoader, TensorDataset
from torch.utils.data.distributed import DistributedSampler


class SyntheticImages(TensorDataset):
    def __init__(self, dataset, n, im_size, seed=1234, offset=0):
        g = torch.Generator().manual_seed(seed + offset)
        if dataset == "MNIST":
            x = torch.rand(n, 1, 28, 28, generator=g)
            y = torch.randint(0, 10, (n,), generator=g)
        else:
            x = (torch.randn(n, 3, im_size, im_size, generator=g) * 0.5).clamp(-1, 1)
            y = (torch.rand(n, generator=g) < 0.42).long()
        super().__init__(x, y)
        self.label_true_count = int(y.sum()) if dataset != "MNIST" else None

    def get_item_with_label(self, label):
        x, y = self.tensors
        idx = torch.nonzero(y == int(label)).flatten()
        i = idx[torch.randint(0, len(idx), (1,))].item()
        return x[i], int(y[i])


def _private_loader(ds, opt, rank, world, **kw):
    """The loader over the PRIVATE training set.  One process: the reference's DataLoader(shuffle=True).  --dist: every
    rank draws from its own disjoint 1/world share of a shared per-epoch permutation (DistributedSampler seeded with the
    same seed on every rank; call loader.sampler.set_epoch(epoch)), so a private sample is seen by exactly one rank per
    epoch and the global batch of a step is world x batch_size DISTINCT samples — what the accountant's sample rate
    q = world * batch_size / N (engine.PrivacyEngine.sample_rate) assumes."""
    if world <= 1:
        return DataLoader(ds, batch_size=opt.batch_size, shuffle=True, **kw)
    sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, seed=int(getattr(opt, "dist_data_seed", 0)),
                                 drop_last=True)
    kw.pop("drop_last", None)
    return DataLoader(ds, batch_size=opt.batch_size, sampler=sampler, drop_last=True, **kw)


def init_cached_data(opt, rank=0, world=1):
    """The real files through the preprocessed-tensor cache and the device prefetcher (csl_gan_amd/pipeline.py; `--data_cache PATH`):
    the cache is built once from the same dataset classes (flip off — the flip is drawn per batch on the host and applied by the
    conversion kernel), then every epoch streams uint8 NHWC rows to the device.  Same return tuple as init_real_data; the loaders
    yield DEVICE tensors (channels-last images)."""
    from . import datasets as ds, pipeline as pl
    dev = opt.d_device if torch.cuda.is_available() and str(opt.d_device).startswith("cuda") else "cpu"

    def cached(tag, make):
        path = "%s.%s" % (opt.data_cache, tag)
        if not os.path.exists(pl.cache_paths(path)[2]):
            if rank == 0:
                pl.build_cache(make(), path)
            if world > 1:
                torch.distributed.barrier()
        return pl.CachedImages(path)

    if opt.dataset == "MNIST":
        tr = ds.MNISTDataset(opt.data_path, train=True, per_class=opt.train_set_size // 10)
        data = pl.CachedImages.from_arrays((tr.x.squeeze(1).numpy() * 255.0).round().astype("uint8")[..., None], tr.y.numpy(), signed=False)
        pub = None
        if opt.public_set_size > 0:
            te = ds.MNISTDataset(opt.data_path, train=False)
            pub = pl.CachedImages.from_arrays((te.x.squeeze(1).numpy() * 255.0).round().astype("uint8")[..., None], te.y.numpy(), signed=False)
        flip = False
    else:
        data = cached("train%d_%d" % (opt.train_set_size, opt.im_size), lambda: ds.CelebADataset(
            opt.data_path, im_size=opt.im_size, length=opt.train_set_size, attr_file=opt.label_path, attr=opt.label_attr, flip=False))
        pub = cached("public%d_%d" % (opt.public_set_size, opt.im_size), lambda: ds.CelebADataset(
            opt.data_path, im_size=opt.im_size, length=opt.public_set_size, offset=opt.train_set_size, attr_file=opt.label_path,
            attr=opt.label_attr, flip=False)) if opt.public_set_size > 0 else None
        flip = True
    seed = int(getattr(opt, "dist_data_seed", 0)) if world > 1 else int(opt.manual_seed)
    dl = pl.DevicePrefetcher(data, pl.EpochSampler(len(data), opt.batch_size, rank, world, seed=seed), device=dev, flip=flip, seed=int(opt.manual_seed))
    pdl = pl.DevicePrefetcher(pub, pl.EpochSampler(len(pub), opt.batch_size, seed=seed + 1), device=dev, flip=flip, seed=int(opt.manual_seed) + 1) \
        if pub is not None else None
    return data, dl, pub, pdl


def init_real_data(opt, rank=0, world=1):
    """init_util.init_data (init_util.py:13-42) on the real files: returns (dataset, loader, public set, loader)."""
    from . import datasets as ds
    if getattr(opt, "data_cache", None):
        return init_cached_data(opt, rank, world)
    pub = None
    if opt.dataset == "MNIST":
        data = ds.MNISTDataset(opt.data_path, train=True, per_class=opt.train_set_size // 10)
        if opt.public_set_size > 0:
            pub = ds.MNISTDataset(opt.data_path, train=False)
    else:
        data = ds.CelebADataset(opt.data_path, im_size=opt.im_size, length=opt.train_set_size, attr_file=opt.label_path, attr=opt.label_attr)
        if opt.public_set_size > 0:
            pub = ds.CelebADataset(opt.data_path, im_size=opt.im_size, length=opt.public_set_size, offset=opt.train_set_size,
                                   attr_file=opt.label_path, attr=opt.label_attr)
    dl = _private_loader(data, opt, rank, world, num_workers=opt.num_workers, pin_memory=torch.cuda.is_available())
    pdl = DataLoader(pub, batch_size=opt.batch_size, num_workers=opt.num_workers, shuffle=True) if pub is not None else None
    return data, dl, pub, pdl


def init_data(opt, rank=0, world=1):
    """Real files when --data_path exists and --synthetic is not given; otherwise the synthetic counterpart of
    init_util.init_data: same return tuple, shuffle=True loaders (rank-partitioned under --dist, see _private_loader)."""
    if not getattr(opt, "synthetic", False) and opt.data_path and os.path.isdir(opt.data_path):
        return init_real_data(opt, rank, world)
    n = min(opt.train_set_size, getattr(opt, "synthetic_cap", 4096))
    ds = SyntheticImages(opt.dataset, n, opt.im_size, seed=opt.manual_seed)
    pub = SyntheticImages(opt.dataset, min(opt.public_set_size, 2048), opt.im_size, seed=opt.manual_seed, offset=1) \
        if opt.public_set_size > 0 else None
    dl = _private_loader(ds, opt, rank, world, drop_last=True, num_workers=0)
    pdl = DataLoader(pub, batch_size=opt.batch_size, shuffle=True, num_workers=0) if pub is not None else None
    return ds, dl, pub, pdl
