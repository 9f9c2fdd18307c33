"""Multi-GPU data parallelism for the D-step: one process per GPU, torch.distributed over RCCL/xGMI.

The path shards by sample (per-sample gradients and clip factors are independent per sample,
SURVEY.md §8e): every rank runs the whole D-step on its own B_local images; the only data-path
exchange is ONE all-reduce(SUM) of the already-clipped-and-noised gradients — 17.3 MB fp32 for D64,
sent as a single flat bucket (xGMI is point-to-point, 7 links x ~153 GB/s: at this size the
collective is latency-bound, so fewer/larger messages, never per-tensor calls).  Each rank adds
Gaussian noise of variance (sigma*C)^2 / R so the reduced sum carries exactly (sigma*C)^2, and
pre-scales by 1/(B_local*R), so nothing follows the collective.  Adaptive clip norms are averaged
across ranks (9 floats) so every rank clips and noises with the same C.

The reference has no distributed code at all; this module is the build's extension
for BASELINE.json configs 4-5.  `backend="nccl"` is RCCL on ROCm; CPU tests use gloo.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the default process group from the torchrun environment (no-op for WORLD_SIZE=1)."""
    world, rank, local = env_world()
    if world == 1:
        return world, rank, local
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("CSLGAN_FORCE_DEVICE", local)))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


# Segmented graph replay (round 4): while GraphedDStep records or replays a multi-rank step whose collectives must NOT go into a HIP
# graph, it installs a callable here.  Every collective of the step is handed to it as a closure: at recording time it ends the graph
# being captured, issues the collective eagerly (every rank does, so the ranks stay in step) and begins the next graph; a replay runs
# graph, collective, graph, ...  None: the collective is simply issued.
_boundary = None


def _collective(fn):
    if _boundary is not None:
        return _boundary(fn)
    return fn()


def segments_enabled():
    """Multi-rank steps whose collectives cannot be recorded replay as SEGMENTS around eagerly issued collectives (CSLGAN_GRAPH_SEGMENTS=0:
    such steps are launched eagerly, the round-3 behaviour)."""
    return os.environ.get("CSLGAN_GRAPH_SEGMENTS", "1") == "1"


class FlatGradReducer:
    """all-reduce(SUM) of one flat gradient bucket.  The engine hands over the flat fp32 buffer its
    per-parameter .grad tensors alias, already scaled by 1/(B_local*R)."""

    def __init__(self, group=None, always=False):
        """always: issue the collective on a one-rank group too (tests: the RCCL call as it is recorded into a HIP graph)."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.always = bool(always) and dist.is_initialized()
        self.bytes_reduced = 0

    def __call__(self, flat: torch.Tensor):
        if self.world > 1 or self.always:
            _collective(lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group))
            self.bytes_reduced += flat.numel() * flat.element_size()      # (host counter: counts recorded calls, not replays)
        return flat


def average_across_ranks(t: torch.Tensor, use_max=False, group=None):
    """In-place mean (or max) of a small tensor over ranks: adaptive clipping statistics."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        _collective(lambda: dist.all_reduce(t, op=dist.ReduceOp.MAX if use_max else dist.ReduceOp.SUM, group=group))
        if not use_max:
            t /= dist.get_world_size(group)
    return t


def collectives_capturable():
    """True when the step's collectives may be recorded into a HIP graph: always in a single process; with several ranks only
    over RCCL (backend "nccl" — its all-reduce is a stream-ordered kernel launch that torch registers with the capture;
    scripts/rccl_capture_probe.py; gloo stages through the host and cannot be captured) AND only when asked for with
    CSLGAN_GRAPH_DIST=1.  Multi-rank capture is OPT-IN: it has been verified on a one-rank RCCL group only (the pool this was
    built on has one GPU per box), so until a multi-GPU run has shown it working the default N > 1 step is launched eagerly."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return True
    return dist.get_backend() == "nccl" and os.environ.get("CSLGAN_GRAPH_DIST", "0") == "1"


def broadcast_object(obj, src=0):
    """A small picklable host object from rank `src` to every rank (seeds, the mean-sample tensor's shape)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        box = [obj]
        dist.broadcast_object_list(box, src=src)
        return box[0]
    return obj


def broadcast_tensor(t, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
    return t


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":      # name the device: RCCL otherwise guesses it from the rank
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()
