"""Input pipeline (csl_gan_amd/pipeline.py; SURVEY.md §8f item 4, reference datasets.py:20-63): the preprocessed-tensor cache holds exactly
what CelebADataset computes, the prefetcher walks every sample once per epoch in the rank's share, and — on the device — the one
conversion kernel equals ToTensor + RandomHorizontalFlip + Normalize."""
import numpy as np
import pytest
import torch


def _make_jpegs(root, n, seed=0):
    from PIL import Image
    rng = np.random.default_rng(seed)
    root.mkdir(parents=True, exist_ok=True)
    for i in range(n):
        w, h = (89, 109) if i % 2 == 0 else (120, 96)          # portrait like CelebA (178x218 halved) and a landscape one
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(root / ("%06d.jpg" % (i + 1)), quality=92)
    with open(root.parent / "attr.txt", "w") as f:
        f.write("%d\nSmiling Male\n" % n)
        for i in range(n):
            f.write("%06d.jpg %d %d\n" % (i + 1, 1 if i % 3 == 0 else -1, 1 if i % 2 else -1))


def test_cache_holds_the_dataset_transform_and_the_prefetcher_covers_an_epoch(tmp_path):
    from csl_gan_amd import datasets as ds, pipeline as pl
    n, B = 37, 8
    _make_jpegs(tmp_path / "img", n)
    data = ds.CelebADataset(str(tmp_path / "img"), im_size=32, length=n, attr_file=str(tmp_path / "attr.txt"), attr="Smiling", flip=False)
    hdr = pl.build_cache(data, str(tmp_path / "cache" / "celeba"))
    assert (hdr["n"], hdr["H"], hdr["W"], hdr["C"], hdr["signed"]) == (n, 32, 32, 3, True)
    cache = pl.CachedImages(str(tmp_path / "cache" / "celeba"))
    assert cache.label_true_count == data.label_true_count == 13
    for i in (0, 1, 17, 36):
        x, y = data[i]
        got = cache.to_float(cache.x[i:i + 1])[0]
        assert int(cache.labels[i]) == y
        assert (got - x).abs().max().item() <= 1e-6                 # the dataset quantises to uint8 before dividing by 255 as well
    # two ranks: disjoint halves of one shared permutation per epoch, ragged tail dropped, a new permutation every epoch
    seen = []
    for rank in (0, 1):
        samp = pl.EpochSampler(n, B, rank=rank, world=2, seed=5)
        batches = list(samp)
        assert len(batches) == len(samp) == (n // 2) // B
        seen.append(np.concatenate(batches))
        samp.set_epoch(1)
        assert not np.array_equal(np.concatenate(list(samp)), seen[-1])
    assert len(np.intersect1d(seen[0], seen[1])) == 0
    # the host form of the prefetcher (no GPU): every image of the rank's share exactly once, unflipped rows equal the cache
    pf = pl.DevicePrefetcher(cache, pl.EpochSampler(n, B, seed=3), device="cpu", flip=False)
    idx = np.concatenate(list(pl.EpochSampler(n, B, seed=3)))
    k = 0
    for img, lab in pf:
        assert img.shape == (B, 3, 32, 32) and lab.dtype == torch.int64
        ref = cache.to_float(cache.x[np.sort(idx[k:k + B])])
        assert torch.equal(img, cache.to_float(cache.x[idx[k:k + B]])) and ref.shape == img.shape
        assert torch.equal(lab, torch.from_numpy(cache.labels[idx[k:k + B]]))
        k += B
    assert k == (n // B) * B
    # with flips: every image is its source or the mirror of it, and about half are mirrored
    pf = pl.DevicePrefetcher(cache, pl.EpochSampler(n, B, seed=3), device="cpu", flip=True, seed=1)
    flips, k = 0, 0
    for img, _ in pf:
        src = cache.to_float(cache.x[idx[k:k + B]])
        for a, b in zip(img, src):
            same, mirrored = torch.equal(a, b), torch.equal(a, b.flip(2))
            assert same or mirrored
            flips += int(mirrored and not same)
        k += B
    assert 4 <= flips <= 28


@pytest.mark.gpu
def test_device_prefetcher_and_conversion_kernel(tmp_path):
    from csl_gan_amd import _lib, ops, pipeline as pl
    rng = np.random.default_rng(1)
    n, B, H, W, C = 70, 16, 64, 64, 3
    cache = pl.CachedImages.from_arrays(rng.integers(0, 256, (n, H, W, C), dtype=np.uint8), rng.integers(0, 2, n), signed=True)
    # the kernel against the host reference, explicit flip flags
    u8 = torch.from_numpy(cache.x[:B].copy())
    flip = torch.from_numpy((rng.random(B) < 0.5).astype(np.uint8))
    out = torch.empty((B, H, W, C), device="cuda")
    d_u8, d_flip = u8.cuda(), flip.cuda()
    ops.check(_lib.lib().cslgan_u8_to_f32_nhwc(ops._p(d_u8), ops._p(d_flip), B, H, W, C, cache.scale, cache.bias, ops._p(out),
                                               torch.cuda.current_stream().cuda_stream), "u8_to_f32_nhwc")
    ref = cache.to_float(u8.numpy(), flip.numpy())
    assert (out.permute(0, 3, 1, 2).cpu() - ref).abs().max().item() <= 2e-7        # (the kernel contracts x * scale + bias into one fma)
    # the prefetcher: three epochs' worth of batches arrive on the device in sampler order, channels-last, while slots are recycled
    samp = pl.EpochSampler(n, B, seed=9)
    pf = pl.DevicePrefetcher(cache, samp, device="cuda:0", flip=False, depth=2)
    for epoch in range(3):
        samp.set_epoch(epoch)
        idx = np.concatenate(list(samp))
        k = 0
        for img, lab in pf:
            assert img.is_cuda and img.shape == (B, C, H, W) and img.permute(0, 2, 3, 1).is_contiguous()
            got = img.cpu()                                   # (synchronises: the slot may be recycled after this)
            assert (got - cache.to_float(cache.x[idx[k:k + B]])).abs().max().item() <= 2e-7, (epoch, k)
            assert torch.equal(lab.cpu(), torch.from_numpy(cache.labels[idx[k:k + B]]))
            k += B
        assert k == (n // B) * B
