"""DP engine with the surface the reference's train.py uses from the twosixlabs Opacus fork.

The fork is absent and unpinned (requirements.txt:9); what is implemented is the specification
SURVEY.md §8 (a7)-(a10),(b) derives from the reference's call sites (cited per method).  All
arithmetic on device tensors runs on the HIP kernels of libcslgan_hip.so.

Data layout in HBM (288 GB per GPU — per-sample gradients are materialised, not recomputed):
  grad_sample buffer of a parameter : [n_passes, B, numel(p)] fp32 in the parameter's own memory
      order (conv filters KRSC), exposed as ``p.grad_sample`` with logical shape [n_passes, B, *p.shape];
  squared norms                     : [n_layers, n_passes * B] — accumulated by the wgrad kernel's
      epilogue while it writes grad_sample (no separate norm pass over HBM);
  p.summed_grad                     : same shape/strides as p.
"""
from __future__ import annotations

import os
import types
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import accountant, ops
from .nn import HipConv2d, HipLinear, PerSampleSink

CLIP_EPS = 1e-6


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (train.py:76) on the cslgan_adam_multi_f32 kernel for device parameters — every tensor of a
    parameter group in ONE launch; CPU parameters (configs[0] plumbing) use the same update written with torch ops.
    optimizer.state keeps torch's layout (step, exp_avg, exp_avg_sq) so checkpoints interoperate."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        # capturable: the step count the bias corrections need lives in HBM (one int32 per lock-step set of parameters, advanced
        # by a device op) instead of in the launch arguments, so a step recorded in a HIP graph stays correct on replay.  The
        # counters are private to the object — NOT optimizer.state: load_state_dict casts state tensors to the parameter's dtype
        # and a checkpoint must hold nothing but torch.optim.Adam's keys — and are rebuilt from state["step"] when missing.
        self.capturable = False
        self._step_dev = {}

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._step_dev = {}
        for st in self.state.values():          # a checkpoint written by an older build may carry the device counter
            st.pop("step_dev", None)
            if isinstance(st.get("step"), torch.Tensor):
                st["step"] = int(st["step"].item())

    def prepare_capture(self):
        """Create the device step counters NOW (eagerly), so that a capture records only their increment."""
        self.capturable = True
        for grp in self.param_groups:
            ps = [p for p in grp["params"] if p.is_cuda]
            by_step = {}
            for p in ps:
                by_step.setdefault(self.state[p].get("step", 0) if p in self.state else 0, []).append(p)
            for step, plist in by_step.items():
                self._counter(plist, step)

    def _counter(self, plist, step_before):
        key = tuple(id(p) for p in plist)
        c = self._step_dev.get(key)
        if c is None:
            c = self._step_dev[key] = torch.full((1,), int(step_before), device=plist[0].device, dtype=torch.int32)
        return c

    @torch.no_grad()
    def step(self, closure=None):
        for grp in self.param_groups:
            b1, b2 = grp["betas"]
            dev = {}                                         # step value -> device parameters taking that step together
            for p in grp["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad
                if p.is_cuda:
                    if g.stride() != p.stride() or not _dense(p):
                        g = _like_layout(g, p)
                    dev.setdefault(st["step"], []).append((p, g, st))
                else:
                    if grp["weight_decay"]:
                        g = g.add(p, alpha=grp["weight_decay"])
                    m, v, t = st["exp_avg"], st["exp_avg_sq"], st["step"]
                    m.mul_(b1).add_(g, alpha=1 - b1)
                    v.mul_(b2).addcmul_(g, g, value=1 - b2)
                    denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(grp["eps"])
                    p.addcdiv_(m, denom, value=-grp["lr"] / (1 - b1 ** t))
            for step, items in dev.items():
                s = step
                if self.capturable:
                    s = self._counter([p for p, _, _ in items], step - 1)
                    s.add_(1)
                ops.adam_multi([_flat(p) for p, _, _ in items], [_flat(g) for _, g, _ in items],
                               [_flat(st["exp_avg"]) for _, _, st in items], [_flat(st["exp_avg_sq"]) for _, _, st in items],
                               grp["lr"], b1, b2, grp["eps"], grp["weight_decay"], s)
                for p, _, _ in items:
                    # the kernel wrote p through its raw pointer: tell autograd (and ops.repack_cache) it changed
                    torch.autograd.graph.increment_version(p)

    def bump_versions(self):
        """After a HIP-graph replay of a recorded step (the replay runs no Python): mark every parameter as changed, so the
        repack caches of ops.py and autograd's saved-tensor checks see the new weights."""
        for grp in self.param_groups:
            for p in grp["params"]:
                if p.is_cuda:
                    torch.autograd.graph.increment_version(p)


def _dense(t):
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


def _flat(t: torch.Tensor) -> torch.Tensor:
    """1-D alias of a dense tensor's memory (contiguous or channels-last), no copy."""
    if t.is_contiguous():
        return t.view(-1)
    if t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last):
        return t.permute(0, 2, 3, 1).reshape(-1)
    raise RuntimeError("tensor is not dense in memory")


def _like_layout(g, p):
    out = torch.empty_like(p, memory_format=torch.preserve_format)
    out.copy_(g)
    return out


def calc_sample_norms(named_params, flat=True):
    """opacus.utils.tensor_utils.calc_sample_norms (train.py:311-314): list of [n_passes, B] norms,
    one per parameter or a single all-parameter norm when flat."""
    gs = [g for _, g in named_params]
    if not gs:
        return []
    if not gs[0].is_cuda:
        per = [g.reshape(g.size(0), g.size(1), -1).norm(2, dim=2) for g in gs]
    else:
        shp = gs[0].shape[:2]
        sq = ops.sample_sqnorm([_rows(g) for g in gs])
        per = [s.reshape(shp) for s in sq.sqrt()]
    if flat:
        return [torch.stack(per, 0).norm(2, dim=0)]
    return per


def _rows(g: torch.Tensor) -> torch.Tensor:
    """[n_passes, B, ...] grad_sample (possibly a permuted view of a dense buffer) -> dense [n_passes*B, numel]."""
    base = getattr(g, "_cslgan_rows", None)
    if base is not None:
        return base
    n = g.size(0) * g.size(1)
    if g.dim() == 6 and not g.is_contiguous():
        g = g.permute(0, 1, 2, 4, 5, 3)     # logical [P,B,K,C,R,S] view of KRSC memory -> memory order
    return g.contiguous().reshape(n, -1)


class _NormClipper:
    def __init__(self, engine):
        self._e = engine

    @property
    def is_per_layer(self):
        return self._e._per_layer

    def calc_clipping_factors(self, norms):
        """train.py:324-328: one [n_passes, B] factor tensor per entry of `norms`."""
        C = self._e.max_grad_norm
        if self.is_per_layer:
            return [(float(c) / (n + CLIP_EPS)).clamp(max=1.0) for n, c in zip(norms, C)]
        return [(float(C) / (n + CLIP_EPS)).clamp(max=1.0) for n in norms]


class _Clipper:
    def __init__(self, engine):
        self._e = engine
        self.norm_clipper = _NormClipper(engine)

    def _named_grad_samples(self):
        return [(n, p.grad_sample) for n, p in self._e.module.named_parameters() if hasattr(p, "grad_sample")]


class _LayerCollector:
    """Receives (gz, x) from a layer's backward and launches the per-sample wgrad kernels."""

    def __init__(self, engine, layer):
        self.e, self.layer = engine, layer

    def side_stream(self):
        return self.e._side_stream()

    def collect(self, pass_idx, gz, x, R, S, stride, pad, has_bias):
        e, layer = self.e, self.layer
        B = x.shape[0]
        n_pass = e._fwd_count[layer]
        scale = float(B) if e.loss_reduction == "mean" else 1.0
        w = layer.weight
        K, Cc = gz.shape[-1], x.shape[-1]
        if e.row_roles is not None:
            return self._collect_roles(gz, x, R, S, stride, pad, has_bias)
        if e.norms_only:
            # adaptive-clipping pass (train.py:204-245): only the per-sample norms are consumed
            _, sq = e._buffers(w, n_pass, B, 0)
            _weight_sqnorms(gz, x, R, S, stride, pad, scale, sq[pass_idx])
            if has_bias:
                _, bsq = e._buffers(layer.bias, n_pass, B, 0)
                ops.bias_grad_grouped(gz, group=1, alpha=scale, want_gb=False, sq=bsq[pass_idx])
            return
        n_private = e._n_private(n_pass)
        if e.lean and pass_idx < n_pass - n_private:
            # a pass that is never clipped (generated data in split mode): only its SUM is needed
            e._add_dense(w, _dense_wgrad(gz, x, R, S, stride, pad, scale))
            if has_bias:
                e._add_dense_rows(layer.bias, _dense_bgrad(gz, scale))
            return
        if e.lean:
            pass_idx, n_pass = pass_idx - (n_pass - n_private), n_private
        if e._ghost_layer(gz, x, stride):
            self._ghost_rows(pass_idx, n_pass, gz, x, R, S, stride, pad, scale, has_bias)
            return
        buf, sq = e._buffers(w, n_pass, B, K * R * S * Cc, e._gs_dtype)
        ops.conv2d_wgrad_grouped(gz, x, R, S, stride=stride, pad=pad, group=1, alpha=scale,
                                 out=buf[pass_idx].view(B, K, R, S, Cc), sq=sq[pass_idx])
        if isinstance(layer, nn.Conv2d):
            view = buf.view(n_pass, B, K, R, S, Cc).permute(0, 1, 2, 5, 3, 4)
        else:
            view = buf.view(n_pass, B, K, Cc)
        view._cslgan_rows = buf.view(n_pass * B, -1)
        w.grad_sample = view
        if has_bias:
            b = layer.bias
            bbuf, bsq = e._buffers(b, n_pass, B, K)
            ops.bias_grad_grouped(gz, group=1, alpha=scale, out=bbuf[pass_idx], sq=bsq[pass_idx])
            bview = bbuf.view(n_pass, B, K)
            bview._cslgan_rows = bbuf.view(n_pass * B, K)
            b.grad_sample = bview


def _weight_sqnorms(gz, x, R, S, stride, pad, scale, sq_row):
    """sq_row[n] += ||scale * per-sample weight gradient||^2 without storing the gradient: from the two pixel-Gram
    matrices where that is the cheaper form (few output pixels), else from the product kernel's epilogue."""
    if ops.gram_norms_preferred(gz.shape, x.shape, stride):
        ops.conv2d_wgrad_sqnorm_gram(gz, x, R, S, stride=stride, pad=pad, alpha=scale, sq=sq_row)
    else:
        ops.conv2d_wgrad_grouped(gz, x, R, S, stride=stride, pad=pad, group=1, alpha=scale, want_gw=False, sq=sq_row)


def _ghost_rows(self, pass_idx, n_pass, gz, x, R, S, stride, pad, scale, has_bias, joint=None):
    """materialize="ghost": a clipped pass of a layer whose norms come from the Gram kernel.  Only the norms are
    computed now; (gz, x) are kept until clip() knows the factors and forms sum_b f_b g_b with ONE weighted dense
    wgrad — the per-sample gradient tensor (1.7 GB for the critic's last conv at B=128) is never written or re-read."""
    e, layer = self.e, self.layer
    w = layer.weight
    n = x.shape[0]
    _, sq = e._buffers(w, n_pass, n, 0)
    ops.conv2d_wgrad_sqnorm_gram(gz, x, R, S, stride=stride, pad=pad, alpha=scale, sq=sq[pass_idx])
    # joint = (gz, x, n_dense, scale_dense): the never-clipped rows that sit right before these rows in a fused batch;
    # their dense sum rides in the same clip-weighted launch (weight scale_dense instead of f_b * scale)
    e._ghost.setdefault(id(w), {})[pass_idx] = (gz, x, R, S, stride, pad, scale, joint)
    if has_bias:
        b = layer.bias
        K = gz.shape[-1]
        bbuf, bsq = e._buffers(b, n_pass, n, K)
        ops.bias_grad_grouped(gz, group=1, alpha=scale, out=bbuf[pass_idx], sq=bsq[pass_idx])
        bview = bbuf.view(n_pass, n, K)
        bview._cslgan_rows = bbuf.view(n_pass * n, K)
        b.grad_sample = bview


_LayerCollector._ghost_rows = _ghost_rows


def _dense_wgrad(gz, x, R, S, stride, pad, scale, row_scale=None, out=None, rows=False):
    """rows: the un-summed slabs [n, numel] (for PrivacyEngine._add_dense_rows) where the layer's dense gradient is made of slabs."""
    r = ops.conv2d_wgrad_dense(gz, x, R, S, stride=stride, pad=pad, alpha=scale, row_scale=row_scale, out=out, want_rows=rows)
    return r if (rows and r.dim() == 2 and r.shape[0] > 1) else r.reshape(-1)


def _dense_bgrad(gz, scale):
    """Per-sample bias gradients [N, K] of a never-clipped row block: their column sum is taken by the launch that folds the dense
    sums into summed_grad (PrivacyEngine._add_dense_rows), not by a launch of its own."""
    return ops.bias_grad_grouped(gz, group=1, alpha=scale)


def _collect_roles(self, gz, x, R, S, stride, pad, has_bias):
    """One fused forward carried several logical passes as consecutive row blocks (Trainer.train_D_fused):
    ("norms", n) rows feed only per-sample norms, ("dense", n) rows only their sum, ("private", n) rows are
    materialised per sample.  Each block is its own mean-reduced loss, hence its own x n scaling."""
    e, layer = self.e, self.layer
    w = layer.weight
    K, Cc = gz.shape[-1], x.shape[-1]
    r0 = 0
    ghost = e._ghost_layer(gz, x, stride)
    held = None                       # ghost layer: ("dense" rows start, count, scale) waiting for the private rows after them
    # equal row blocks of a layer the LDS-resident kernel takes: ONE launch for all blocks (three launches of 640 workgroups fill
    # the chip's 512 slots 62 %, one of 1920 fills them 94 %); the blocks' outputs are collected here and launched after the loop
    blocks = None
    if (not ghost and len(e.row_roles) > 1 and e._gs_dtype == torch.float32 and isinstance(layer, nn.Conv2d)
            and len({n for _, n in e.row_roles}) == 1 and len(e.row_roles) <= 4 and ops.wgrad_blocks_eligible(gz.shape, x.shape, R, S, stride)):
        blocks = []
    for role, n in e.row_roles:
        g_, x_ = gz[r0:r0 + n], x[r0:r0 + n]
        row0 = r0
        r0 += n
        scale = float(n) if e.loss_reduction == "mean" else 1.0
        if role == "norms":
            _, sq = e._buffers(("norms", id(w)), 1, n, 0)
            if blocks is not None:
                blocks.append((n, None, sq[0], None))
            else:
                _weight_sqnorms(g_, x_, R, S, stride, pad, scale, sq[0])
            if has_bias:
                _, bsq = e._buffers(("norms", id(layer.bias)), 1, n, 0)
                ops.bias_grad_grouped(g_, group=1, alpha=scale, want_gb=False, sq=bsq[0])
        elif role == "dense":
            if held is not None:
                e._add_dense(w, _dense_wgrad(gz[held[0]:held[0] + held[1]], x[held[0]:held[0] + held[1]], R, S, stride, pad, held[2]))
                held = None
            if ghost:
                held = (row0, n, scale)
            elif blocks is not None:
                blocks.append((n, torch.empty((n, K * R * S * Cc), device=gz.device, dtype=torch.float32), None, "dense"))
            else:
                r = _dense_wgrad(g_, x_, R, S, stride, pad, scale, rows=True)
                (e._add_dense_rows if r.dim() == 2 else e._add_dense)(w, r)
            if has_bias:
                e._add_dense_rows(layer.bias, _dense_bgrad(g_, scale))
        elif ghost:
            joint = None
            if held is not None and held[0] + held[1] == row0:
                joint = (gz[held[0]:r0], x[held[0]:r0], held[1], held[2])
                held = None
            self._ghost_rows(0, 1, g_, x_, R, S, stride, pad, scale, has_bias, joint=joint)
        else:
            buf, sq = e._buffers(w, 1, n, K * R * S * Cc, e._gs_dtype)
            if blocks is not None:
                blocks.append((n, buf[0], sq[0], None))
            else:
                ops.conv2d_wgrad_grouped(g_, x_, R, S, stride=stride, pad=pad, group=1, alpha=scale,
                                         out=buf[0].view(n, K, R, S, Cc), sq=sq[0])
            view = buf.view(1, n, K, R, S, Cc).permute(0, 1, 2, 5, 3, 4) if isinstance(layer, nn.Conv2d) else buf.view(1, n, K, Cc)
            view._cslgan_rows = buf.view(n, -1)
            w.grad_sample = view
            if has_bias:
                b = layer.bias
                bbuf, bsq = e._buffers(b, 1, n, K)
                ops.bias_grad_grouped(g_, group=1, alpha=scale, out=bbuf[0], sq=bsq[0])
                bview = bbuf.view(1, n, K)
                bview._cslgan_rows = bbuf.view(n, K)
                b.grad_sample = bview

    if held is not None:
        e._add_dense(w, _dense_wgrad(gz[held[0]:held[0] + held[1]], x[held[0]:held[0] + held[1]], R, S, stride, pad, held[2]))
    if blocks:
        n = blocks[0][0]
        ops.conv2d_wgrad_blocks(gz, x, R, S, stride, pad, float(n) if e.loss_reduction == "mean" else 1.0, [(bl[0], bl[1], bl[2]) for bl in blocks])
        for _, slabs, _, kind in blocks:
            if kind == "dense":              # the block's sum: its per-sample slabs are column-summed by the launch that folds the dense sums
                e._add_dense_rows(w, slabs)


_LayerCollector._collect_roles = _collect_roles


class PrivacyEngine(PerSampleSink):
    """Gradient-clipping DP engine (train.py:110-116 constructor call).

    materialize="all"     every pass's per-sample gradients are written to p.grad_sample[pass] (fork layout);
    materialize="private" only the passes that are clipped are materialised; never-clipped passes (generated
                          data under grad_clip_split) contribute a dense sum computed by the same MFMA kernel
                          with coarse groups — 2.2 GB less written and re-read per pass for D64 at B=128.
    materialize="ghost"   as "private", and layers with at most 64 output pixels per sample (the critic's last two convs and
                          its linear head, 95 % of its parameters) are never materialised at all: their norms come from the pixel-Gram kernels and their
                          clipped sum from one clip-weighted dense wgrad inside clip() ("ghost clipping").  Those
                          parameters have no p.grad_sample.  Needs split clipping (accum_passes=False).
    `norms_only` (set around the adaptive-clipping pass) computes per-sample norms without storing gradients.
    """

    def __init__(self, module, batch_size, sample_size, alphas, noise_multiplier, max_grad_norm,
                 accum_passes=True, num_private_passes=None, auto_clip_and_accum_on_step=True,
                 loss_reduction="mean", world_size=1, materialize="all", grad_sample_dtype="fp32", **_unused):
        if materialize not in ("all", "private", "ghost"):
            raise ValueError("materialize must be 'all', 'private' or 'ghost'")
        if grad_sample_dtype not in ("fp32", "bf16"):
            raise ValueError("grad_sample_dtype must be 'fp32' or 'bf16'")
        # bf16: weight-tensor per-sample gradients are STORED as bfloat16 (fp32 MFMA accumulate, round-to-nearest
        # on store; norms / clip read the rounded values) — half the HBM bytes of the clip passes.  Bias gradients
        # (a few KB per sample) stay fp32.
        self._gs_dtype = torch.bfloat16 if grad_sample_dtype == "bf16" else torch.float32
        self.materialize, self.norms_only = materialize, False
        self.row_roles = None                   # set by Trainer.train_D_fused for one fused forward/backward
        self._sq_arena, self._sq_off = None, 0
        self._dense = {}
        self._ghost = {}                        # id(weight) -> {pass: (gz, x, R, S, stride, pad, scale)} awaiting clip()
        self.use_side_stream = os.environ.get("CSLGAN_SIDE_STREAM", "0") == "1"
        self._clip_side = None
        self._side, self._side_dirty = None, False
        self.module = module
        self.batch_size, self.sample_size = batch_size, sample_size
        self.alphas = list(alphas)
        self.noise_multiplier = noise_multiplier
        self.accum_passes, self.num_private_passes = accum_passes, num_private_passes
        self.auto_clip_and_accum_on_step = auto_clip_and_accum_on_step
        self.loss_reduction = loss_reduction
        self.world_size = world_size            # ranks sharing the step
        self.sample_rate = batch_size * world_size / sample_size
        self.steps = 0
        self.enabled = True                     # PerSampleSink.enabled
        self.params = list(module.parameters())
        if any(not p.is_cuda for p in self.params):
            raise RuntimeError("PrivacyEngine needs the discriminator on a HIP device: per-sample gradients, clip and "
                               "noise run on libcslgan_hip.so and there is no CPU path")
        self.layers = [m for m in module.modules() if isinstance(m, (HipConv2d, HipLinear))]
        covered = {id(p) for l in self.layers for p in l.parameters(recurse=False)}
        missing = [n for n, p in module.named_parameters() if id(p) not in covered]
        if missing:
            raise RuntimeError("PrivacyEngine: parameters outside HipConv2d/HipLinear layers: %s" % missing)
        for l in self.layers:
            l._sink = self
        self._collectors = {l: _LayerCollector(self, l) for l in self.layers}
        self._fwd_count = {l: 0 for l in self.layers}
        self._bufs = {}
        self.clipper = _Clipper(self)
        self.set_max_grad_norm(max_grad_norm)
        self.seed, self._noise_calls = 0, 0
        self._noise_ctr = None                  # device mirror of _noise_calls: the Philox offset is read from HBM (graph replays)
        self._idx_cache = {}
        self.host_noise_generator: Optional[torch.Generator] = None   # parity mode: noise drawn on the host
        self.host_noise = None                  # parity tests: explicit unit normals, one flat tensor per parameter
        self.optimizer = None
        self._accumulated = False
        self.grad_reducer = None                # set by csl_gan_amd.distributed for N>1

    # -- PerSampleSink ----------------------------------------------------------------------
    def next_pass(self, layer):
        i = self._fwd_count[layer]
        self._fwd_count[layer] = i + 1
        return i

    def collector(self, layer):
        return self._collectors[layer]

    @property
    def lean(self):
        return self.materialize in ("private", "ghost")

    def _side_stream(self):
        """Stream for the weight-gradient side work of the backward hooks (None = same stream)."""
        if not self.use_side_stream:
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.params[0].device)
        self._side_dirty = True
        return self._side

    def join_side_stream(self):
        """Make the current stream wait for the side work: call before anything reads norms / per-sample buffers."""
        if self._side is not None and self._side_dirty:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_dirty = False

    def _ghost_layer(self, gz, x, stride):
        return self.materialize == "ghost" and not self.accum_passes and ops.gram_norms_preferred(gz.shape, x.shape, stride)

    def _n_private(self, n_pass):
        return n_pass if (self.accum_passes or self.num_private_passes is None) else min(self.num_private_passes, n_pass)

    def _add_dense(self, p, flat):
        cur = self._dense.get(id(p))
        if cur is not None and cur.dim() == 2:          # un-reduced rows queued earlier: reduce them now
            tot = torch.empty(cur.shape[1], device=cur.device, dtype=torch.float32)
            ops.clip_accum_noise([cur], [tot])
            cur = tot
        self._dense[id(p)] = flat if cur is None else cur.add_(flat)

    def _add_dense_rows(self, p, rows):
        """A never-clipped contribution handed over as UN-REDUCED rows [n, numel(p)] (per-sample slabs, per-sample bias gradients):
        clip() column-sums them in the same launch that adds the dense sums into summed_grad (segments of different heights), so the
        five to six per-layer column sums of a step are not launches of their own."""
        if id(p) in self._dense:
            tot = torch.empty(rows.shape[1], device=rows.device, dtype=torch.float32)
            ops.clip_accum_noise([rows], [tot])
            return self._add_dense(p, tot)
        self._dense[id(p)] = rows

    def adaptive_clip_fused(self, stat, scalar, per_layer):
        """update_adaptive_clipping_params + calc_clipping_factors of a fused pass as ONE launch (ops.adaptive_clip): the adaptive
        statistic r of every layer from the "norms" rows, the clip norm(s) r * scalar, the clip factors of the clipped rows, the
        factor rows of the materialised layers and the row weights of the joint clip-weighted launches — what clip() would otherwise
        assemble from a dozen small launches.  Returns r, or None when this step's state does not fit (then the caller runs the
        separate ops).  clip() picks the results up (self._pre) if nothing invalidated them in between."""
        ps = self.params
        if not ps or not ps[0].is_cuda or any(("norms", id(p)) not in self._bufs or id(p) not in self._bufs for p in ps):
            return None
        n_pass, B = self._bufs[id(ps[0])][1].shape
        if (self.accum_passes and n_pass > 1) or len(ps) > 32:
            return None
        self.join_side_stream()
        adapt = [self._bufs[("norms", id(p))][1].reshape(-1) for p in ps]
        rows = [self._bufs[id(p)][1].reshape(-1) for p in ps]
        mat_idx = [i for i, p in enumerate(ps) if id(p) not in self._ghost]
        jobs, done = [], set()
        for i, p in enumerate(ps):
            stash = self._ghost.get(id(p))
            for k, (gz, x, R, S, stride, pad, sc, joint) in sorted((stash or {}).items()):
                if joint is not None:
                    n_d, scale_d = joint[2], joint[3]
                    key = ("rs", id(p), n_d, float(scale_d), B, str(gz.device))
                    rs = self._idx_cache.get(key)
                    if rs is None:
                        rs = self._idx_cache[key] = torch.full((n_d + B,), float(scale_d), device=gz.device, dtype=torch.float32)
                    jobs.append((rs[n_d:], i, k * B, float(sc)))
                    done.add((id(p), k))
        if len(jobs) > 16:
            return None
        n_private = self._n_private(n_pass)
        r, c, sq, f, f_mat = ops.adaptive_clip(adapt, rows, stat == "max", scalar, per_layer, CLIP_EPS, (n_pass - n_private) * B,
                                               mat_layers=mat_idx if (per_layer and len(mat_idx) < len(ps)) else (), jobs=jobs)
        self.set_max_grad_norm_device(c)
        self._per_layer = bool(per_layer)
        self._pre = dict(sq=sq, f=f, f_mat=f_mat, mat_idx=mat_idx, jobs=done, per_layer=bool(per_layer))
        return r

    def norms_rows_sqnorms(self) -> torch.Tensor:
        """[n_params, n] squared norms of the "norms" row block of a fused pass."""
        self.join_side_stream()
        return torch.stack([self._bufs[("norms", id(p))][1].reshape(-1) for p in self.params])

    def _buffers(self, p, n_pass, B, numel, dtype=torch.float32):
        key = p if isinstance(p, tuple) else id(p)
        cur = self._bufs.get(key)
        if cur is None or cur[0].shape != (n_pass, B, numel):
            dev = self.params[0].device
            cur = (torch.empty((n_pass, B, numel), device=dev, dtype=dtype), self._sq_alloc(n_pass * B, dev).view(n_pass, B))
            self._bufs[key] = cur
        return cur

    def _sq_alloc(self, n, dev):
        """Zeroed [n] floats for a squared-norm accumulator.  A step asks for about twenty of these (one per parameter tensor and
        row block); they are slices of ONE arena zeroed once per step (_reset_samples) instead of twenty 4-us fill launches.  The
        first step — and any step that needs more than the arena holds — falls back to torch.zeros and sizes the next arena."""
        n4 = (n + 3) & ~3
        a, off = self._sq_arena, self._sq_off
        self._sq_off = off + n4
        if a is None or off + n4 > a.numel() or a.device != dev:
            return torch.zeros(n, device=dev, dtype=torch.float32)
        return a[off:off + n]

    # -- train.py:117,373,389 -----------------------------------------------------------------
    def enable_hooks(self):
        self.enabled = True

    def disable_hooks(self):
        self.enabled = False

    def zero_grad(self):
        """Drop per-sample state (the fork patches optimizer.zero_grad to do this; train.py:245)."""
        self._reset_samples()
        for p in self.params:
            if hasattr(p, "summed_grad"):
                del p.summed_grad
            p.grad = None

    # -- train.py:241-243, 321 -------------------------------------------------------------------
    def set_max_grad_norm(self, v):
        """float -> one flat clip norm; list / 1-D tensor -> one per parameter tensor."""
        if isinstance(v, torch.Tensor):
            v = v.detach().reshape(-1).tolist() if v.numel() > 1 else float(v)
        if isinstance(v, (list, tuple, np.ndarray)):
            v = [float(x) for x in v]
            if len(v) != len(self.params):
                raise ValueError("per-layer max_grad_norm needs %d entries, got %d" % (len(self.params), len(v)))
            self._per_layer = True
        else:
            v = float(v)
            self._per_layer = False
        self._C_host, self._C_dev = v, None

    def set_max_grad_norm_device(self, t: torch.Tensor):
        """Same, from a device tensor ([1] flat or [n_params] per layer) without a host sync: adaptive
        clipping (train.py:233-243) stays on the GPU; the floats are fetched only if someone reads
        ``max_grad_norm``."""
        t = t.detach().reshape(-1).to(torch.float32).contiguous()
        if t.numel() not in (1, len(self.params)):
            raise ValueError("max_grad_norm tensor must have 1 or %d entries" % len(self.params))
        self._per_layer = t.numel() > 1 or (len(self.params) == 1 and self._per_layer)
        self._C_dev, self._C_host = t, None

    @property
    def max_grad_norm(self):
        if self._C_host is None:
            vals = self._C_dev.cpu().tolist()
            self._C_host = vals if self._per_layer else vals[0]
        return self._C_host

    def max_grad_norm_device(self) -> torch.Tensor:
        return self._C_device(self.params[0].device)

    def _C_device(self, device):
        if self._C_dev is None or self._C_dev.device != device:
            c = self._C_host if isinstance(self._C_host, list) else [self._C_host]
            self._C_dev = torch.tensor(c, dtype=torch.float32, device=device)
        return self._C_dev

    # -- norms ------------------------------------------------------------------------------------
    def sample_sqnorms(self, recompute=False) -> torch.Tensor:
        """[n_params, n_passes*B] squared per-sample norms.  Default: the values the wgrad epilogue
        accumulated; recompute=True re-reads the materialised grad_sample (cslgan_sample_sqnorm_f32),
        which is what must be used after a caller edited p.grad_sample in place (train.py:447)."""
        self.join_side_stream()
        stored = [self._bufs[id(p)][1].reshape(-1) for p in self.params]
        if recompute:
            mat = [i for i, p in enumerate(self.params) if id(p) not in self._ghost]
            fresh = ops.sample_sqnorm([_rows(self.params[i].grad_sample) for i in mat])
            for j, i in enumerate(mat):
                stored[i] = fresh[j]
        return torch.stack(stored)

    # -- train.py:399-402, 417 -----------------------------------------------------------------
    def clip(self, recompute_norms=False):
        """Per-sample clip factors + clipped sum into p.summed_grad (a SUM over samples)."""
        self.join_side_stream()
        ps = self.params
        mat_idx = [i for i, p in enumerate(ps) if id(p) not in self._ghost]
        mats = [_rows(ps[i].grad_sample) for i in mat_idx]
        n_pass, B = self._bufs[id(ps[0])][1].shape
        if self.accum_passes and n_pass > 1:
            # passes are added per sample before clipping: column-sum over the pass axis
            summed_ps = [torch.empty((B, m.shape[1]), device=m.device, dtype=torch.float32) for m in mats]
            ops.clip_accum_noise([m.view(n_pass, -1) for m in mats], [s.view(-1) for s in summed_ps])
            mats, n_pass = summed_ps, 1
            sq = ops.sample_sqnorm(mats)
            self._pre = None
        else:
            sq = None
        n_private = self._n_private(n_pass)
        per_layer = self._per_layer
        pre, self._pre = getattr(self, "_pre", None), None
        if pre is not None and (recompute_norms or pre["mat_idx"] != mat_idx or pre["per_layer"] != per_layer or pre["sq"].shape[1] != n_pass * B):
            pre = None
        if pre is not None:        # adaptive_clip_fused computed them in the launch that made the clip norms
            sq, f = pre["sq"], pre["f"]
        else:
            if sq is None:
                sq = self.sample_sqnorms(recompute=recompute_norms)
            f = ops.clip_factors(sq, self._C_device(sq.device), flat=not per_layer, eps=CLIP_EPS,
                                 first_private_row=(n_pass - n_private) * B)
        self.last_factors, self.last_sq = f, sq
        outs = []
        for p in ps:
            p.summed_grad = torch.empty_like(p, memory_format=torch.preserve_format)
            outs.append(_flat(p.summed_grad))
        if len(mat_idx) == len(ps):
            ops.clip_accum_noise(mats, outs, factors=f)
        else:
            if pre is not None and per_layer and pre["f_mat"] is not None:
                f_mat = pre["f_mat"]
            else:
                f_mat = f[self._index_tensor(mat_idx, f.device)].contiguous() if per_layer else f
            self._jobs_done = pre["jobs"] if pre is not None else ()
            ops.clip_accum_noise(mats, [outs[i] for i in mat_idx], factors=f_mat)
            # ghost layers: sum_b f_b g_b as one clip-weighted dense wgrad per pass.  The layers' launches are independent and each
            # under-fills the chip (128 - 640 workgroups): every second one goes to a second stream (CSLGAN_CLIP_STREAM=0: off)
            ghosts = [i for i, p in enumerate(ps) if self._ghost.get(id(p)) is not None]
            two = os.environ.get("CSLGAN_CLIP_STREAM", "1") == "1" and len(ghosts) > 1
            cur = torch.cuda.current_stream()
            if two:
                if self._clip_side is None:
                    self._clip_side = torch.cuda.Stream(device=ps[0].device)
                self._clip_side.wait_stream(cur)
            for n_g, i in enumerate(ghosts):
                p = ps[i]
                stash = self._ghost.get(id(p))
                with torch.cuda.stream(self._clip_side if (two and n_g % 2 == 1) else cur):
                    self._clip_ghost_layer(i, p, stash, f, per_layer, n_pass, B, outs)
            if two:
                cur.wait_stream(self._clip_side)
        if self._dense:       # sums of the never-clipped passes (lean modes)
            idx = [i for i, p in enumerate(ps) if id(p) in self._dense]
            segs = [self._dense[id(ps[i])] for i in idx]
            ops.clip_accum_noise([t if t.dim() == 2 else t.view(1, -1) for t in segs], [outs[i] for i in idx], beta=1.0, ragged=True)
        self._accumulated = False

    def _clip_ghost_layer(self, i, p, stash, f, per_layer, n_pass, B, outs):
        """One ghost layer of clip(): sum_b f_b g_b as clip-weighted dense weight gradient(s) into outs[i]."""
        fi = (f[i] if per_layer else f).reshape(n_pass, B)
        total = None
        single = len(stash) == 1            # one clipped pass: the weighted sum is written straight into summed_grad
        for k, (gz, x, R, S, stride, pad, scale, joint) in sorted(stash.items()):
            dst = outs[i] if single else None
            if joint is None:
                part = _dense_wgrad(gz, x, R, S, stride, pad, scale, row_scale=fi[k].contiguous(), out=dst)
            else:
                gzj, xj, n_d, scale_d = joint
                # row weights of the joint launch: [scale_d] * n_d (never-clipped rows; a constant prefix kept across
                # steps) followed by f_b * scale — one launch per step instead of fill + mul + cat
                key = ("rs", id(p), n_d, float(scale_d), B, str(f.device))
                rs = self._idx_cache.get(key)
                if rs is None:
                    rs = self._idx_cache[key] = torch.full((n_d + B,), float(scale_d), device=f.device, dtype=torch.float32)
                if (id(p), k) not in getattr(self, "_jobs_done", ()):      # (adaptive_clip_fused wrote the suffix already)
                    torch.mul(fi[k], float(scale), out=rs[n_d:])
                part = _dense_wgrad(gzj, xj, R, S, stride, pad, 1.0, row_scale=rs, out=dst)
            total = part if total is None else total.add_(part)
        if not single:
            outs[i].copy_(total)

    def _index_tensor(self, idx, device):
        """Device copy of a small index list, uploaded once (a host->device copy per step is also not graph-capturable)."""
        key = (tuple(idx), str(device))
        t = self._idx_cache.get(key)
        if t is None:
            t = self._idx_cache[key] = torch.tensor(list(idx), device=device)
        return t

    def add_to_grad_sample(self, param, rows, pass_idx=0):
        """p.grad_sample[pass_idx, i] += rows[i] (train.py:447) on the dense [passes*B, numel] buffer behind the view; rows are
        [B, numel] in the parameter's memory order.  Needs the parameter materialised (--materialize all)."""
        gs = getattr(param, "grad_sample", None)
        if gs is None:
            raise RuntimeError("add_to_grad_sample: the parameter has no p.grad_sample (use --materialize all)")
        dense = _rows(gs)
        B = gs.shape[1]
        if rows.shape != (B, dense.shape[1]):
            raise RuntimeError("add_to_grad_sample: rows %s, expected %s" % (tuple(rows.shape), (B, dense.shape[1])))
        dense[pass_idx * B:(pass_idx + 1) * B].add_(rows.to(dense.dtype))

    def accum_grads_across_passes(self):
        """The cross-pass sum (train.py:402) already happened inside clip(): rows = passes x samples."""
        return None

    def accumulate_batch(self):
        """train.py:417: the clipped batch becomes the pending update."""
        self._accumulated = True

    # -- train.py:135-136, 484 ------------------------------------------------------------------
    def _set_seed(self, seed):
        self.seed = int(seed)
        self._set_noise_calls(0)

    def _set_noise_calls(self, n):
        self._noise_calls = int(n)
        if self._noise_ctr is not None:
            self._noise_ctr.fill_(int(n))

    def ensure_noise_counter(self):
        """Device mirror of _noise_calls (the Philox call index is read from HBM so a graph replay advances it)."""
        if self._noise_ctr is None:
            self._noise_ctr = torch.full((1,), self._noise_calls, device=self.params[0].device, dtype=torch.int64)
        return self._noise_ctr

    def attach(self, optimizer):
        self.optimizer = optimizer
        engine = self
        orig_step, orig_zero = optimizer.step, optimizer.zero_grad

        def dp_step(self_opt, closure=None):
            engine._before_step()
            return orig_step(closure) if closure is not None else orig_step()

        def dp_zero_grad(self_opt, *a, **k):
            engine.zero_grad()
            return orig_zero(*a, **k)

        optimizer.privacy_engine = self
        optimizer._orig_step, optimizer._orig_zero_grad = orig_step, orig_zero
        optimizer.step = types.MethodType(dp_step, optimizer)
        optimizer.zero_grad = types.MethodType(dp_zero_grad, optimizer)

    def detach(self):
        if self.optimizer is not None:
            self.optimizer.step, self.optimizer.zero_grad = self.optimizer._orig_step, self.optimizer._orig_zero_grad
            self.optimizer = None
        for l in self.layers:
            l._sink = None

    def noise_stds(self) -> List[float]:
        s = self.noise_multiplier
        if isinstance(self.max_grad_norm, list):
            return [s * c for c in self.max_grad_norm]
        return [s * self.max_grad_norm] * len(self.params)

    def _before_step(self):
        """grad = (summed_grad + N(0,(sigma*C)^2)) / B  for every parameter, one launch."""
        ps = self.params
        if not all(hasattr(p, "summed_grad") for p in ps):
            if self.auto_clip_and_accum_on_step and all(hasattr(p, "grad_sample") for p in ps):
                self.clip()
            else:
                return          # non-DP step (warm-up iterations call the wrapped optimizer too)
        R = self.world_size
        denom = float(self.batch_size * R)
        # R ranks each add noise of variance (sigma*C)^2 / R, so the all-reduced sum has (sigma*C)^2
        dev = ps[0].device
        ins = [_flat(p.summed_grad).view(1, -1) for p in ps]
        # one flat fp32 bucket; every p.grad aliases its slice (same strides as p), so the all-reduce
        # needs no gather/scatter copies
        flat = torch.empty(sum(p.numel() for p in ps), device=dev, dtype=torch.float32)
        grads, off = [], 0
        for p in ps:
            p.grad = torch.as_strided(flat, p.size(), p.stride(), storage_offset=off)
            grads.append(flat[off:off + p.numel()])
            off += p.numel()
        noises = None
        if self.host_noise is not None:
            noises = [z.to(dev) for z in self.host_noise]
        elif self.host_noise_generator is not None:
            noises = [torch.randn(p.numel(), generator=self.host_noise_generator).to(dev) for p in ps]
        std_dev = None
        if self.noise_multiplier > 0:
            std_dev = (self._C_device(dev) * (self.noise_multiplier / (R ** 0.5))).expand(len(ps)).contiguous()
        self.ensure_noise_counter()
        # Philox stream (seed, call, tensor, column): the call index is read from HBM, so a graph replay advances it too
        ops.clip_accum_noise(ins, grads, noise_std=std_dev, noises=noises, seed=self.seed, offset=0, call_counter=self._noise_ctr,
                             scale=1.0 / denom)
        self._noise_ctr.add_(1)
        if self.grad_reducer is not None:
            self.grad_reducer(flat)
        self._noise_calls += 1
        self.steps += 1
        for p in ps:
            del p.summed_grad
        self._reset_samples()

    def _reset_samples(self):
        """Per-sample state lives for one step: drop it once the noised gradient exists."""
        for l in self.layers:
            self._fwd_count[l] = 0
        self._bufs.clear()
        # squared-norm arena: sized by the largest step seen, zeroed here once for the next step
        if self._sq_off > 0:
            if self._sq_arena is None or self._sq_off > self._sq_arena.numel():
                self._sq_arena = torch.zeros(self._sq_off, device=self.params[0].device, dtype=torch.float32)
            else:
                self._sq_arena.zero_()
        self._sq_off = 0
        self._dense.clear()
        self._ghost.clear()
        self._pre, self._jobs_done = None, ()
        self.row_roles = None
        for p in self.params:
            if hasattr(p, "grad_sample"):
                del p.grad_sample
        self._accumulated = False

    # -- train.py:294-295, 588 ------------------------------------------------------------------
    def get_privacy_spent(self, target_delta=None):
        delta = 1e-6 if target_delta is None else target_delta
        rdp = accountant.compute_rdp(self.sample_rate, self.noise_multiplier, self.steps, self.alphas)
        return accountant.get_privacy_spent(self.alphas, rdp, delta)

    # -- checkpoint: the reference drops the engine state on resume (epsilon restarts from 0);
    #    steps / seed / Philox call counter / clip norms are what must survive to keep the accounting honest
    def state_dict(self):
        return {"steps": self.steps, "seed": self.seed, "noise_calls": self._noise_calls, "max_grad_norm": self.max_grad_norm,
                "noise_multiplier": self.noise_multiplier, "sample_rate": self.sample_rate}

    def load_state_dict(self, st):
        self.steps, self.seed = st["steps"], int(st["seed"])
        self._set_noise_calls(st["noise_calls"])
        self.set_max_grad_norm(st["max_grad_norm"])
