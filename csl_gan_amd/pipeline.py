"""Input pipeline that can feed the D-step (reference datasets.py:20-63, init_util.py:13-42).

The reference decodes, resizes, crops, flips and normalises every image with PIL / torchvision inside DataLoader workers on every
epoch and copies fp32 batches to the device synchronously (train.py:566-575).  At the step rate of this build (~18 k images/s per
GPU) that path is input-bound by two orders of magnitude, so the work is split by how often it has to happen:

  once per dataset   `build_cache`: decode + resize + centre-crop (the deterministic part of datasets.py:41-45) into a uint8
                     memory-mapped array [N, H, W, C] — ALREADY in the device's NHWC layout — plus labels and a JSON header;
  once per batch     `DevicePrefetcher`: a background thread gathers the batch's rows from the memmap into one of two pinned
                     staging buffers (12 KB per 64x64 RGB image: a quarter of the fp32 bytes) and draws the per-image flip flags;
                     the H2D copy is enqueued `non_blocking` on a side stream; ONE kernel (cslgan_u8_to_f32_nhwc) turns the bytes
                     into the normalised fp32 channels-last batch the critic's first conv reads in place (ToTensor, flip, Normalize);
                     the consumer's stream waits on an event, never on the host.

The random horizontal flip stays random per epoch (host RNG, one byte per image); everything else is bit-identical to what
`CelebADataset` computes (its transform quantises to uint8 before the division by 255 too).  MNIST needs no cache file: the idx
arrays are the cache.
"""
from __future__ import annotations

import json
import os
import threading
import queue

import numpy as np
import torch

CACHE_VERSION = 1


def cache_paths(path):
    return path + ".u8", path + ".labels.npy", path + ".json"


def build_cache(dataset, path, progress=None):
    """One pass over an image dataset whose __getitem__ returns ((C,H,W) float tensor in [-1,1] or [0,1], label) with the flip
    switched OFF (CelebADataset(flip=False)): writes the uint8 NHWC memmap + labels + header and returns the header.
    The value range is recorded so the device kernel restores exactly the dataset's own floats."""
    n = len(dataset)
    x0, _ = dataset[0]
    C, H, W = x0.shape
    lo = float(x0.min()) < -1e-6                     # Normalize(0.5, 0.5) data lives in [-1, 1], ToTensor data in [0, 1]
    u8p, labp, hdrp = cache_paths(path)
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    mm = np.lib.format.open_memmap(u8p, mode="w+", dtype=np.uint8, shape=(n, H, W, C))
    labels = np.zeros(n, dtype=np.int64)
    for i in range(n):
        x, y = dataset[i]
        a = x.numpy()
        a = (a * 0.5 + 0.5) if lo else a
        mm[i] = np.clip(np.rint(a * 255.0), 0, 255).astype(np.uint8).transpose(1, 2, 0)
        labels[i] = int(y)
        if progress is not None and (i + 1) % 1000 == 0:
            progress(i + 1, n)
    mm.flush()
    del mm
    np.save(labp, labels)
    hdr = {"version": CACHE_VERSION, "n": n, "H": H, "W": W, "C": C, "signed": bool(lo), "dtype": "uint8", "layout": "NHWC"}
    with open(hdrp, "w") as f:
        json.dump(hdr, f)
    return hdr


class CachedImages:
    """The memory-mapped cache as a dataset: len(), labels, `gather(indices, out)` into a (pinned) uint8 host tensor, and the
    reference's `get_item_with_label` / `label_true_count` for the conditional paths."""

    def __init__(self, path):
        u8p, labp, hdrp = cache_paths(path)
        with open(hdrp) as f:
            self.hdr = json.load(f)
        if self.hdr.get("version") != CACHE_VERSION or self.hdr.get("layout") != "NHWC":
            raise RuntimeError("unsupported image cache %s" % path)
        self.x = np.load(u8p, mmap_mode="r")
        self.labels = np.load(labp)
        self.n, self.H, self.W, self.C = (self.hdr[k] for k in ("n", "H", "W", "C"))
        self.signed = bool(self.hdr["signed"])
        self.label_true_count = int((self.labels == 1).sum())
        # uint8 -> float: x/255 for [0,1] data, x/127.5 - 1 for Normalize(0.5, 0.5)
        self.scale, self.bias = (1.0 / 127.5, -1.0) if self.signed else (1.0 / 255.0, 0.0)

    @classmethod
    def from_arrays(cls, x_u8_nhwc, labels, signed):
        """An in-memory cache (MNIST: the idx file IS the uint8 array)."""
        self = cls.__new__(cls)
        self.x, self.labels = np.ascontiguousarray(x_u8_nhwc), np.asarray(labels, dtype=np.int64)
        self.n, self.H, self.W, self.C = self.x.shape
        self.signed = bool(signed)
        self.hdr = {"version": CACHE_VERSION, "n": self.n, "H": self.H, "W": self.W, "C": self.C, "signed": self.signed}
        self.label_true_count = int((self.labels == 1).sum())
        self.scale, self.bias = (1.0 / 127.5, -1.0) if self.signed else (1.0 / 255.0, 0.0)
        return self

    def __len__(self):
        return self.n

    def gather(self, idx, out):
        """rows `idx` (sorted internally for memmap locality, returned in the caller's order) into out[len(idx), H, W, C] (uint8)."""
        idx = np.asarray(idx)
        order = np.argsort(idx, kind="stable")
        dst = out.numpy() if torch.is_tensor(out) else out
        dst[order] = self.x[idx[order]]
        return out

    def to_float(self, u8_rows, flip=None):
        """Host reference of the device kernel: normalised fp32 NCHW batch from uint8 NHWC rows (tests, CPU runs)."""
        a = torch.from_numpy(np.array(u8_rows, dtype=np.uint8)).float() * self.scale + self.bias
        if flip is not None:
            f = torch.as_tensor(np.asarray(flip)).bool()
            a = torch.where(f.view(-1, 1, 1, 1), a.flip(2), a)
        return a.permute(0, 3, 1, 2).contiguous()

    def get_item_with_label(self, label, number=None):
        idx = np.nonzero(self.labels == int(label))[0]
        i = int(idx[np.random.randint(0, len(idx))]) if number is None else int(number)
        return self.to_float(self.x[i:i + 1])[0], int(self.labels[i])


class EpochSampler:
    """Index stream of one rank: a fresh shared permutation per epoch (seed + epoch), this rank's contiguous 1/world share, batches of
    `batch_size`, the ragged tail dropped — the contract of data._private_loader (DistributedSampler, drop_last)."""

    def __init__(self, n, batch_size, rank=0, world=1, seed=0, shuffle=True):
        self.n, self.bs, self.rank, self.world, self.seed, self.shuffle = n, batch_size, rank, world, seed, shuffle
        self.epoch = 0
        self.per_rank = n // world

    def set_epoch(self, e):
        self.epoch = int(e)

    def __len__(self):
        return self.per_rank // self.bs

    def __iter__(self):
        perm = np.random.default_rng(self.seed + self.epoch).permutation(self.n) if self.shuffle else np.arange(self.n)
        mine = perm[self.rank * self.per_rank:(self.rank + 1) * self.per_rank]
        for b in range(len(self)):
            yield mine[b * self.bs:(b + 1) * self.bs]


class DevicePrefetcher:
    """Iterates (images, labels) batches ALREADY ON THE DEVICE: channels-last fp32 images (logical NCHW), int64 labels.

    depth staging slots (pinned uint8 + flags + labels on the host, uint8 on the device): a worker thread fills slot k+1 from the memmap
    while the device consumes slot k; uploads and the conversion kernel run on a side stream; the consumer's stream only waits on the
    slot's event.  On a CPU-only host (tests, configs[0]) the same iterator yields host tensors through `CachedImages.to_float`."""

    def __init__(self, cache: CachedImages, sampler: EpochSampler, device="cuda:0", flip=True, depth=3, seed=0):
        self.cache, self.sampler, self.flip, self.depth = cache, sampler, bool(flip), max(2, int(depth))
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        self.rng = np.random.default_rng(seed + 7919 * sampler.rank)
        B, H, W, C = sampler.bs, cache.H, cache.W, cache.C
        pin = self.on_gpu
        self.slots = []
        for _ in range(self.depth):
            s = {"u8": torch.empty((B, H, W, C), dtype=torch.uint8, pin_memory=pin), "flip": torch.zeros(B, dtype=torch.uint8, pin_memory=pin),
                 "lab": torch.empty(B, dtype=torch.int64, pin_memory=pin)}
            if self.on_gpu:
                s.update(d_u8=torch.empty((B, H, W, C), dtype=torch.uint8, device=self.device), d_flip=torch.empty(B, dtype=torch.uint8, device=self.device),
                         d_lab=torch.empty(B, dtype=torch.int64, device=self.device),
                         out=torch.empty((B, H, W, C), dtype=torch.float32, device=self.device), ready=torch.cuda.Event(), free=torch.cuda.Event())
            self.slots.append(s)
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.sampler_iter = None

    dataset = property(lambda self: self.cache)

    def __len__(self):
        return len(self.sampler)

    def _fill_host(self, slot, idx):
        self.cache.gather(idx, slot["u8"])
        slot["lab"].copy_(torch.from_numpy(self.cache.labels[idx]))
        if self.flip:
            slot["flip"].copy_(torch.from_numpy((self.rng.random(len(idx)) < 0.5).astype(np.uint8)))

    def _upload(self, slot):
        from . import _lib, ops
        B, H, W, C = slot["u8"].shape
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(slot["free"])            # the consumer has finished with this slot's previous batch
            slot["d_u8"].copy_(slot["u8"], non_blocking=True)
            slot["d_flip"].copy_(slot["flip"], non_blocking=True)
            slot["d_lab"].copy_(slot["lab"], non_blocking=True)
            ops.check(_lib.lib().cslgan_u8_to_f32_nhwc(ops._p(slot["d_u8"]), ops._p(slot["d_flip"]) if self.flip else None, B, H, W, C,
                                                       float(self.cache.scale), float(self.cache.bias), ops._p(slot["out"]),
                                                       torch.cuda.current_stream().cuda_stream), "u8_to_f32_nhwc")
            slot["ready"].record(self.stream)

    def __iter__(self):
        batches = iter(self.sampler)
        q = queue.Queue(maxsize=self.depth - 1)

        def worker():
            try:
                for k, idx in enumerate(batches):
                    slot = self.slots[k % self.depth]
                    if self.on_gpu and k >= self.depth:
                        slot["host_free"].wait()            # the upload of batch k - depth has left the pinned buffers ...
                        slot["host_free"].clear()           # ... and the next "set" can only come from THIS batch's upload
                    self._fill_host(slot, idx)
                    if not self.on_gpu:             # host form: hand over finished tensors (the slot is refilled right away)
                        q.put((k, (self.cache.to_float(slot["u8"].numpy(), slot["flip"].numpy() if self.flip else None), slot["lab"].clone())))
                    else:
                        q.put((k, slot))
                q.put(None)
            except BaseException as e:                      # surfaced in the consumer
                q.put(e)

        for s in self.slots:
            s["host_free"] = threading.Event()
        th = threading.Thread(target=worker, daemon=True)
        th.start()
        while True:
            item = q.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            k, slot = item
            if not self.on_gpu:
                yield slot
                continue
            self._upload(slot)
            # the pinned buffers may be refilled once the copies have been issued AND executed: the side stream's event tells
            up = torch.cuda.Event()
            up.record(self.stream)
            threading.Thread(target=lambda e=up, ev=slot["host_free"]: (e.synchronize(), ev.set()), daemon=True).start()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(slot["ready"])
            yield slot["out"].permute(0, 3, 1, 2), slot["d_lab"]
            slot["free"].record(cur)                        # everything the consumer enqueued on the batch precedes this point
        th.join()
