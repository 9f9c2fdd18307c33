"""Tensor-level wrappers over the C-ABI (include/cslgan.h).  torch is plumbing here: it owns the
device memory and the stream; every kernel launched is hand-written HIP from csl_gan_amd/csrc.

Layout convention: activations are NHWC-contiguous fp32 tensors of shape [N,H,W,C]; conv filters
are [K,R,S,C].  Device tensors only — there is no CPU path in this module.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import ConvT, SegsT, check

ACT_NONE, ACT_LRELU02, ACT_RELU, ACT_TANH = 0, 1, 2, 3
COMPUTE_F32, COMPUTE_BF16, COMPUTE_BF16X3 = 0, 1, 2          # cslgan_conv_t.compute (include/cslgan.h)

_compute = COMPUTE_F32
_auto = False


def set_compute_dtype(name):
    """Arithmetic of every MFMA conv / linear / weight-gradient launch of this process (tensors are fp32 in HBM either way):
      "fp32"    exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the default;
      "bf16"    operands rounded to bfloat16 in the kernels, fp32 accumulate (BASELINE configs[4]);
      "bf16x3"  fp32 emulated from three bfloat16 pieces per operand on the bf16 matrix cores (fp32-accurate, see
                csrc/igemm_bf16.hip) for every launch;
      "fp32_auto"  fp32 accuracy on whichever of the two fp32-accurate paths is faster for the launch: bf16x3 for forward /
                data-gradient launches that fill the chip (the generator's 5x5 convs, the critic's fused 384-row passes),
                exact fp32 MFMA for weight gradients and small launches.
    One process drives one GPU and one training run, so this is process-wide state set once from --compute_dtype."""
    global _compute, _auto
    if name not in ("fp32", "bf16", "bf16x3", "fp32_auto"):
        raise ValueError("compute dtype must be 'fp32', 'bf16', 'bf16x3' or 'fp32_auto', got %r" % (name,))
    _auto = name == "fp32_auto"
    _compute = {"fp32": COMPUTE_F32, "bf16": COMPUTE_BF16, "bf16x3": COMPUTE_BF16X3, "fp32_auto": COMPUTE_F32}[name]


def get_compute_dtype():
    return "fp32_auto" if _auto else {COMPUTE_F32: "fp32", COMPUTE_BF16: "bf16", COMPUTE_BF16X3: "bf16x3"}[_compute]


def _kc_compute(rows, n_out, kdim):
    """compute field for a forward / data-gradient launch of `rows` output rows x n_out output channels with reduction
    length kdim.  fp32_auto: measured same-box (scripts/compute_modes.py) the three-piece path wins once the launch has
    >= 128 tiles of 128x128 and a long reduction (150-185 vs 110-135 TF on the generator's convs, 100-155 vs 81-112 TF on
    the critic's 384-row passes) and loses on the 128-row launches of the small layers."""
    if _auto and n_out >= 64 and kdim >= _AUTO_MIN_K and ((rows + 127) // 128) * ((n_out + 127) // 128) >= _AUTO_MIN_TILES:
        return COMPUTE_BF16X3
    return _compute


_F32_HALO = os.environ.get("CSLGAN_F32_HALO", "1") == "1"        # A/B: exact-fp32 launches the round-4 LDS-halo kernel takes run on it (igemm_x3h<., 0, .>)


# Largest channel split of a low-fill halo launch (<= 1: never split).  OFF by default: measured alone the critic's last convs gain
# (conv4 forward, 384 rows: 167 -> 200 TF; 128 rows: 86 -> 140 TF), but inside the two-stream recorded step the other branch
# already fills the idle CUs and the step does not move (6.797 ms split, 6.731 ms unsplit, 6.723 ms split + 128-wide tiles: noise)
# while every split launch adds a reduce launch.  CSLGAN_X3_SPLIT=8 switches it on (single-stream / eager runs).
_X3_SPLIT = int(os.environ.get("CSLGAN_X3_SPLIT", "0"))


def _split_scratch(d, rows, cols, red_channels, out):
    """Scratch for the channel split of the LDS-halo kernel (cslgan_conv_t.split_ws): a launch of fewer than 512 128x128 tiles may
    divide its reduction channels over up to _X3_SPLIT workgroups per tile, at least two 16-channel chunks each.  The tensor comes
    from the caching allocator like the output (stream-ordered, graph-pool safe); the C side decides the actual split."""
    s = min(_X3_SPLIT, (red_channels // 16) // 2)
    if s < 2 or -(-rows // 128) * -(-cols // 128) >= 512:
        return None
    ws = torch.empty(s * out.numel(), device=out.device, dtype=torch.float32)
    d.split_ws, d.split_ws_floats = ws.data_ptr(), ws.numel()
    return ws


_GN_FUSE = os.environ.get("CSLGAN_GN_FUSE", "1") == "1"            # A/B: GroupNorm statistics from the producing conv's epilogue


class gn_partials:
    """Ask the NEXT conv2d_fwd on this thread to leave GroupNorm(groups) partial statistics of its output behind
    (cslgan_conv_t.gn_part): `with ops.gn_partials(32) as cell: y = conv(...)`, then `groupnorm_act(y, ..., part=cell.part)`.
    cell.part stays None when the conv's shape or route cannot produce them — the normalisation then runs its own statistics pass."""
    _req = None

    def __init__(self, groups):
        self.groups, self.part = int(groups), None

    def __enter__(self):
        self._prev, gn_partials._req = gn_partials._req, (self if _GN_FUSE else None)
        return self

    def __exit__(self, *a):
        gn_partials._req = self._prev


_ones_cache = {}


def ones_like_const(t):
    """A persistent tensor of ones with t's shape / dtype / device: the constant cotangent of a backward() or autograd.grad call
    (torch.ones_like there is one fill launch per call, re-run by every replay of a recorded step).  Never written to."""
    key = (tuple(t.shape), t.dtype, str(t.device))
    c = _ones_cache.get(key)
    if c is None:
        c = _ones_cache[key] = torch.ones(t.shape, dtype=t.dtype, device=t.device)
    return c


def set_f32_halo(on):
    """Switch the exact-fp32 form of the round-4 halo kernel on or off at run time (returns the previous setting): the same products
    summed in a different order, so the golden tests use it to prove which activation units sit within rounding of zero."""
    global _F32_HALO
    prev, _F32_HALO = _F32_HALO, bool(on)
    return prev


_X3_WGRAD = os.environ.get("CSLGAN_X3_WGRAD", "1") == "1"      # A/B: fp32_auto sends eligible weight gradients to the three-piece kernel
_AUTO_WG_MIN_FLOP = float(os.environ.get("CSLGAN_AUTO_WG_MIN_GFLOP", "0.5")) * 1e9
_X3_S2 = os.environ.get("CSLGAN_X3_S2", "1") == "1"            # A/B: the LDS-halo form of the bf16 paths for stride-2 forward convs
_X3_DGRAD = os.environ.get("CSLGAN_X3_DGRAD", "1") == "1"      # ... and for data gradients (0: the gather kernels, as in round 3)
# fp32_auto thresholds (A/B switches; scripts/compute_modes.py): 128x128 tiles of the launch, reduction length
_AUTO_MIN_TILES = int(os.environ.get("CSLGAN_AUTO_MIN_TILES", "32"))
_AUTO_MIN_K = int(os.environ.get("CSLGAN_AUTO_MIN_K", "512"))


class compute_dtype:
    """Context manager form of set_compute_dtype (tests)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = get_compute_dtype()
        set_compute_dtype(self.name)

    def __exit__(self, *a):
        set_compute_dtype(self.prev)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_storage_bf16 = False


def set_storage_dtype(name):
    """Element type of the ACTIVATIONS and ACTIVATION GRADIENTS between the critic's conv layers in HBM (BASELINE configs[4],
    csrc/igemm_bf16s.hip):
      "fp32"  the default: every tensor is fp32;
      "bf16"  the critic's conv layers write bfloat16 outputs; every conv / data-gradient / weight-gradient call whose activation
              operands are bfloat16 runs on the bf16-stored kernels (bf16 filter copies cached per parameter version, fp32
              accumulation, fp32 weight gradients).  Parameters, weight gradients, per-sample gradients, norms, clip, noise and
              Adam stay fp32.  The dtype then FOLLOWS the tensors: an op handed bf16 activations answers in kind.
    Process-wide like set_compute_dtype, set once from --storage_dtype."""
    global _storage_bf16
    if name not in ("fp32", "bf16"):
        raise ValueError("storage dtype must be 'fp32' or 'bf16', got %r" % (name,))
    _storage_bf16 = name == "bf16"


def get_storage_dtype():
    return "bf16" if _storage_bf16 else "fp32"


def storage_bf16():
    return _storage_bf16


class storage_dtype:
    """Context manager form of set_storage_dtype (tests)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.prev = get_storage_dtype()
        set_storage_dtype(self.name)

    def __exit__(self, *a):
        set_storage_dtype(self.prev)


def cast_bf16(t):
    """fp32 -> bfloat16 (round to nearest even), same shape and strides; one pass of cslgan_cast_f32_bf16."""
    if t.dtype == torch.bfloat16:
        return t
    _chk_dense(t, "t")
    out = torch.empty_like(t, dtype=torch.bfloat16)
    check(_lib.lib().cslgan_cast_f32_bf16(_p(t), _p(out), t.numel(), _stream()), "cast_f32_bf16")
    return out


def cast_f32(t):
    """bfloat16 -> fp32 (exact), same shape and strides."""
    if t is None or t.dtype == torch.float32:
        return t
    _chk_dense(t, "t", allow_bf16=True)
    out = torch.empty_like(t, dtype=torch.float32)
    check(_lib.lib().cslgan_cast_bf16_f32(_p(t), _p(out), t.numel(), _stream()), "cast_bf16_f32")
    return out


def _chk_dense(t, name, allow_bf16=False):
    """A device tensor whose memory is one dense block in some dimension order (casts are layout-agnostic)."""
    if not t.is_cuda:
        raise RuntimeError("%s must be a device tensor (csl_gan_amd.ops has no CPU path)" % name)
    if t.dtype != torch.float32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise RuntimeError("%s must be float32%s, got %s" % (name, " or bfloat16" if allow_bf16 else "", t.dtype))
    if not (t.is_contiguous() or _dense_block(t)):
        raise RuntimeError("%s must be dense" % name)
    return t


def _dense_block(t):
    """True when t's elements fill one block of memory exactly once (any dimension order: e.g. channels-last views)."""
    dims = sorted(((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1))
    run = 1
    for st, sz in dims:
        if st != run:
            return False
        run *= sz
    return True


def _is_bf16(t):
    return t is not None and t.dtype == torch.bfloat16


class _RepackCache:
    """Repacked filter matrices (data-gradient classes, stride-2 parity classes, the channel-folded filters of the
    generator's UpsampleConv layers) keyed by the filter tensor's storage and autograd version counter: any in-place
    torch op bumps the counter, and HipAdam — which writes the weights through raw pointers — bumps it explicitly
    (torch.autograd.graph.increment_version).  A D-step reuses each critic layer's repack four times and the
    generator's folded filters until the next generator step.

    Pinned entries (round 4): a recorded HIP graph bakes the ADDRESS of every workspace it reads.  The generator is frozen during
    D-steps, so its folded / pre-split filters need not be re-made by every replay: GraphedDStep pins the generator's entries (made
    by the eager warm-up steps), records the step with cache hits on them, and calls refresh_pinned() before each replay — an entry
    whose source parameter has a new version (a train_G step ran) is rebuilt IN PLACE by the closure it was made with.  A pinned
    entry is never evicted or replaced; a version mismatch seen by get() reuses its buffer (repack = 1)."""

    def __init__(self, max_entries=256):
        self.d, self.max = {}, max_entries      # callers pass wkey (a never-reused per-module token, csl_gan_amd.nn) only for module-owned filters
        self.pinned = {}                         # key -> number of recorded graphs that read the entry's buffer

    def get(self, kind, w, numel, wkey=None, version=None):
        """version: the autograd version to key on when `w` is itself derived from a parameter (its own counter is always 0): an int,
        or an object with a `.version` property (csl_gan_amd.nn.VersionRef) that refresh_pinned() can read again later.
        A hit from another stream than the one that packed the buffer waits for that pack (the D-step runs its gradient-penalty
        branch on a second stream, and both branches read the critic's re-packed filters)."""
        if wkey is None:          # not known to be a live parameter (a temporary may reuse an address): never cache
            return torch.empty(numel, device=w.device, dtype=torch.float32), 1
        key = (kind, wkey, w.data_ptr(), tuple(w.shape))
        src = version if hasattr(version, "version") else (w if version is None else None)
        ver = w._version if version is None else (version.version if hasattr(version, "version") else version)
        hit = self.d.get(key)
        cur = torch.cuda.current_stream()
        if hit is not None and hit[1].numel() == numel and hit[1].device == w.device:
            if hit[0] == ver:
                if hit[2] != cur.cuda_stream and hit[3] is not None:
                    cur.wait_event(hit[3])
                self._fresh = None
                return hit[1], 0
            if key in self.pinned:        # a recorded graph reads this buffer: rebuild it where it is
                hit[0], hit[2], hit[3] = ver, cur.cuda_stream, None
                self._fresh = key
                return hit[1], 1
        if len(self.d) >= self.max:
            self.clear()
        ws = torch.empty(numel, device=w.device, dtype=torch.float32)
        self.d[key] = [ver, ws, cur.cuda_stream, None, None, src]
        self._fresh = key
        return ws, 1

    def set_rebuild(self, fn):
        """Attach to the entry get() has just made (a miss) the closure that re-makes its contents in place: fn() launches the
        packing kernels on the current stream.  Only entries with a rebuild closure and a version source can be pinned."""
        key = getattr(self, "_fresh", None)
        if key is not None and key in self.d:
            self.d[key][4] = fn

    def packed(self):
        """Called right after the launch that filled the most recent fresh buffer: marks the point other streams must wait for."""
        key = getattr(self, "_fresh", None)
        if key is not None and key in self.d and self.multi_stream:
            ev = torch.cuda.Event()
            ev.record()
            self.d[key][3] = ev
        self._fresh = None

    multi_stream = False        # set while a second stream is in use (Trainer): events are only recorded then

    def pin(self, tokens):
        """Pin every entry that belongs to a module token in `tokens` (key[1] is the token or a tuple starting with it) and can be
        rebuilt in place.  Returns the pinned keys (hand them back to unpin())."""
        keys = []
        for key, e in self.d.items():
            tok = key[1][0] if isinstance(key[1], tuple) else key[1]
            if tok in tokens and e[4] is not None and e[5] is not None:
                self.pinned[key] = self.pinned.get(key, 0) + 1
                e[3] = None           # (the caller has synchronised the device: no cross-stream wait is owed any more)
                keys.append(key)
        return keys

    def unpin(self, keys):
        for key in keys:
            n = self.pinned.get(key, 0) - 1
            if n > 0:
                self.pinned[key] = n
            else:
                self.pinned.pop(key, None)

    def refresh_pinned(self):
        """Rebuild (in place, in insertion order: a folded filter before its pieces) every pinned entry whose source has a new
        version.  Returns how many were rebuilt."""
        n = 0
        for key in self.d:
            if key not in self.pinned:
                continue
            e = self.d[key]
            ver = e[5].version if hasattr(e[5], "version") else e[5]._version
            if ver != e[0]:
                e[4]()
                e[0], e[2], e[3] = ver, torch.cuda.current_stream().cuda_stream, None
                n += 1
        return n

    def clear(self):
        """Drop every entry that is not pinned."""
        self.d = {k: v for k, v in self.d.items() if k in self.pinned}


repack_cache = _RepackCache()


class LaunchTimer:
    """HIP-event timing of individual C-ABI launches on the stream they are enqueued on (torch's
    current stream), for bench.py's live roofline figures.  Events are resolved once, after the
    caller has synchronised.  Each record carries the name of the device kernel the entry dispatched to
    (cslgan_last_kernel), so figures can be grouped per KERNEL as rocprofv3 lists them."""

    def __init__(self, only=None):
        """only: optional set of (entry name, shape tag) pairs — launches outside it are not instrumented (two event records
        per launch cost the host ~5 us; bench.py watches just the dominant kernel's launches inside its timed region)."""
        self.records = []
        self.only = only

    def begin(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, name, flop, nbytes, start, exec_flop=None, tag=None, kernel=None):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((name, flop, nbytes, start, e, flop if exec_flop is None else exec_flop, tag, kernel or name))

    def keys_of_kernel(self, kernel_name):
        """(entry name, shape tag) pairs whose launches dispatched to `kernel_name`."""
        return {(name, tag) for name, _, _, _, _, _, tag, kernel in self.records if kernel == kernel_name}

    def summary(self, by_shape=False, by_kernel=False):
        out = {}
        for name, flop, nbytes, s, e, xf, tag, kernel in self.records:
            if by_kernel:
                name = kernel
            if by_shape and tag:
                name = "%s %s" % (name, tag)
            d = out.setdefault(name, {"name": name, "ms": 0.0, "flop": 0.0, "exec_flop": 0.0, "bytes": 0.0, "n": 0})
            d["ms"] += s.elapsed_time(e)
            d["flop"] += flop
            d["exec_flop"] += xf
            d["bytes"] += nbytes
            d["n"] += 1
        return out


_timer = None


def set_launch_timer(t):
    global _timer
    _timer = t


def _timed(name, flop, nbytes, fn, exec_flop=None, tag=None):
    """flop = algorithmic FLOP of the op as the reference executes it; exec_flop = FLOP the kernel really issues
    (differs for the UpsampleConv layers, which run on C/4 folded channels, and for the zero-padded RGB input)."""
    if _timer is None:
        return fn()
    tg = tag() if callable(tag) else tag
    if _timer.only is not None and (name, tg) not in _timer.only:
        return fn()
    s = _timer.begin()
    r = fn()
    _timer.end(name, flop, nbytes, s, exec_flop, tg, _lib.lib().cslgan_last_kernel().decode(errors="replace"))
    return r


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk(t: torch.Tensor, name: str, allow_bf16=False):
    if not t.is_cuda:
        raise RuntimeError("%s must be a device tensor (csl_gan_amd.ops has no CPU path)" % name)
    if t.dtype != torch.float32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise RuntimeError("%s must be float32%s, got %s" % (name, " or bfloat16" if allow_bf16 else "", t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def conv_out_size(H, R, stride, pad):
    return (H + 2 * pad - R) // stride + 1


def _conv_desc(N, H, W, Cc, K, R, S, stride, pad, kind="wgrad", group=None):
    """kind: "fwd" / "dgrad" may take the bf16x3 path under fp32_auto; weight gradients never do."""
    P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
    comp = _compute
    if kind == "fwd":
        comp = _kc_compute(N * P * Q, K, R * S * Cc)
    elif kind == "dgrad":
        comp = _kc_compute(N * H * W, Cc, (R * S * K) // (stride * stride))
    elif (kind == "wgrad" and _auto and _X3_WGRAD and S == 5 and stride in (1, 2) and K % 64 == 0 and Cc % 64 == 0 and 2.0 * N * P * Q * K * R * S * Cc >= _AUTO_WG_MIN_FLOP
          and ((P % 8 == 0 and Q % 8 == 0) or
               ((P, Q, H, W, stride, pad) == (4, 4, 8, 8, 2, 2) and group is not None and group % 2 == 0))):
        comp = COMPUTE_BF16X3          # weight gradient on the LDS-resident three-piece kernel (csrc/igemm_wgh.hip: igemm_x3w_kernel)
    if comp == COMPUTE_BF16 and H == 1 and W == 1 and R == 1 and S == 1 and 2.0 * N * K * Cc < 1e9:
        # small linear layers (the critic's head, the generator's first layer): a few hundred MFLOP on skinny GEMMs where the bf16
        # gather kernels ran at < 1 TF (0.35 ms for 268 MFLOP); the fp32 kernels take 15-30 us and are exact
        comp = COMPUTE_F32
    return ConvT(N, H, W, Cc, K, R, S, stride, pad, comp, P, Q), P, Q


def _conv2d_fwd_stored(x, w, bias, stride, pad, residual, act, out, wkey, alg_scale, wversion, out_dtype):
    """conv2d_fwd with bf16-stored activations (x and / or y bfloat16): the bf16-stored kernel when x is bf16 with C % 8 == 0,
    otherwise the fp32 kernels between casts."""
    _chk(x, "x", allow_bf16=True); _chk(w, "w")
    N, H, W, Cc = x.shape
    K, R, S, C2 = w.shape
    if C2 != Cc:
        raise RuntimeError("conv2d_fwd: channel mismatch x C=%d, w C=%d" % (Cc, C2))
    P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
    if out_dtype is None:           # follow the input: conv layers answer bf16 activations in kind; heads and 1..4-channel images stay fp32
        out_dtype = torch.bfloat16 if (x.dtype == torch.bfloat16 and P * Q > 1 and K > 4) else torch.float32
    y_bf16 = out_dtype == torch.bfloat16
    if x.dtype == torch.bfloat16 and Cc % 8 == 0 and (K > 4 or P * Q == 1):      # 1..4 output channels of an image: the fp32 vector-ALU kernel below
        d = ConvT(N, H, W, Cc, K, R, S, stride, pad, COMPUTE_BF16, P, Q)
        y = out if out is not None else torch.empty((N, P, Q, K), device=x.device, dtype=out_dtype)
        if y.dtype != out_dtype:
            raise RuntimeError("conv2d_fwd: out has dtype %s, expected %s" % (y.dtype, out_dtype))
        if bias is not None:
            _chk(bias, "bias")
        if residual is not None:
            _chk(residual, "residual", allow_bf16=True)
            if tuple(residual.shape) != (N, P, Q, K):
                raise RuntimeError("conv2d_fwd: residual shape %s, expected %s" % (tuple(residual.shape), (N, P, Q, K)))
        ws, repack = repack_cache.get("bf16s_fwd", w, (w.numel() + 1) // 2, wkey, version=wversion)
        flop = 2.0 * N * P * Q * K * R * S * Cc * alg_scale
        nbytes = 2.0 * (N * H * W * Cc + K * R * S * Cc) + y.element_size() * float(N * P * Q * K)
        _timed("conv2d_fwd", flop, nbytes, lambda: check(
            _lib.lib().cslgan_conv2d_fwd_bf16s(C.byref(d), _p(x), _p(w), _p(ws), repack, _p(bias), _p(residual), 1 if _is_bf16(residual) else 0,
                                               act, _p(y), 1 if y_bf16 else 0, _stream()), "conv2d_fwd_bf16s"),
            exec_flop=2.0 * N * P * Q * K * R * S * Cc, tag=lambda: "N%d %dx%d C%d K%d R%d s%d bf16s" % (N, H, W, Cc, K, R, stride))
        repack_cache.packed()
        return y
    if (x.dtype == torch.bfloat16 and not y_bf16 and K <= 4 and Cc == 64 and stride == 1 and R * S <= 9 and P % 8 == 0 and Q % 8 == 0
            and residual is None and out is None):
        # the generator's output conv (64 -> 3 channels): the vector-ALU kernel reads the bf16-stored input as it is
        d = ConvT(N, H, W, Cc, K, R, S, stride, pad, COMPUTE_F32, P, Q)
        y = torch.empty((N, P, Q, K), device=x.device, dtype=torch.float32)
        if bias is not None:
            _chk(bias, "bias")
        _timed("conv2d_fwd", 2.0 * N * P * Q * K * R * S * Cc, 2.0 * N * H * W * Cc + 4.0 * N * P * Q * K, lambda: check(
            _lib.lib().cslgan_conv2d_fwd_skinny_bf16in(C.byref(d), _p(x), _p(w), _p(bias), act, _p(y), _stream()), "conv2d_fwd_skinny_bf16in"),
            tag=lambda: "N%d %dx%d C%d K%d R%d s%d bf16in" % (N, H, W, Cc, K, R, stride))
        return y
    if (x.dtype == torch.float32 and y_bf16 and residual is None and out is None
            and Cc == 3 and _c3_layer(H, W, K, R, S, stride, pad, H % 16 == 0 and W % 32 == 0)):
        # the RGB first layer: fp32 image in, bf16-stored activations out of the same kernel (no cast pass)
        d = ConvT(N, H, W, Cc, K, R, S, stride, pad, COMPUTE_F32, P, Q)
        y = torch.empty((N, P, Q, K), device=x.device, dtype=torch.bfloat16)
        if bias is not None:
            _chk(bias, "bias")
        _timed("conv2d_fwd", 2.0 * N * P * Q * K * R * S * Cc, 4.0 * N * H * W * Cc + 2.0 * N * P * Q * K, lambda: check(
            _lib.lib().cslgan_conv2d_c3_fwd_bf16out(C.byref(d), _p(x), _p(w), _p(bias), act, _p(y), _stream()), "conv2d_c3_fwd_bf16out"),
            tag=lambda: "N%d %dx%d C%d K%d R%d s%d bf16out" % (N, H, W, Cc, K, R, stride))
        return y
    yf = conv2d_fwd(cast_f32(x), w, bias, stride=stride, pad=pad, residual=cast_f32(residual), act=act, wkey=wkey, alg_scale=alg_scale,
                    wversion=wversion)
    y = cast_bf16(yf) if y_bf16 else yf
    if out is not None:
        out.copy_(y)
        return out
    return y


def in_affine_ok(x, w, stride, pad):
    """Can conv2d_fwd(x, w, in_affine=...) fold a per-(image, channel) affine map into its staging?  The stride-1 launches of the
    LDS-halo kernel (fp32 / three-piece arithmetic) and the 1..4-output-channel kernel (64 input channels)."""
    if not (_GN_FUSE and x.is_cuda and x.dtype == torch.float32 and stride == 1 and x.dim() == 4):
        return False
    N, H, W, Cc = x.shape
    K, R, S, _ = w.shape
    P, Q = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
    if R * S == 1 or P % 8 or Q % 8:
        return False
    if K <= 4:
        return Cc == 64 and R * S <= 9
    comp = _kc_compute(N * P * Q, K, R * S * Cc)
    return (comp == COMPUTE_BF16X3 or (comp == COMPUTE_F32 and _F32_HALO)) and Cc % 16 == 0 and K >= 64 and K % 4 == 0


def groupnorm_affine(part, gamma, beta, groups, eps, N, HW, Cc):
    """(scale, shift)[N, C] of GroupNorm(groups) from a conv epilogue's partial statistics (gn_partials): the normalisation as the
    affine map conv2d_fwd(in_affine=...) applies while it stages its input."""
    pt, n_part = part
    sc = torch.empty((N, Cc), device=pt.device, dtype=torch.float32)
    sh = torch.empty((N, Cc), device=pt.device, dtype=torch.float32)
    check(_lib.lib().cslgan_groupnorm_affine_parts_f32(_p(pt), n_part, _p(gamma), _p(beta), N, HW, Cc, groups, float(eps), _p(sc), _p(sh),
                                                       _stream()), "groupnorm_affine_parts")
    return sc, sh


def conv2d_fwd(x, w, bias=None, stride=1, pad=0, residual=None, act=ACT_NONE, out=None, wkey=None, alg_scale=1.0, wversion=None,
               out_dtype=None, in_affine=None):
    """y[N,P,Q,K] = act(conv(x[N,H,W,C], w[K,R,S,C]) + bias [+ residual[N,P,Q,K]]).

    in_affine = (scale[N,C], shift[N,C], relu): x is replaced by max(scale * x + shift, 0 if relu else -inf) while it is staged
    (cslgan_conv_t.in_scale; only where in_affine_ok says so — an error otherwise).

    alg_scale: FLOP the reference spends on this layer / FLOP of this call (4 for an UpsampleConv's conv, which the
    reference runs over four identical channel groups) — bench accounting only.
    out_dtype: torch.bfloat16 stores the output as bfloat16 (set_storage_dtype); None follows the input's element type."""
    if x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16 or _is_bf16(residual):
        return _conv2d_fwd_stored(x, w, bias, stride, pad, residual, act, out, wkey, alg_scale, wversion, out_dtype)
    _chk(x, "x"); _chk(w, "w")
    N, H, W, Cc = x.shape
    K, R, S, C2 = w.shape
    if C2 != Cc:
        raise RuntimeError("conv2d_fwd: channel mismatch x C=%d, w C=%d" % (Cc, C2))
    if Cc == 3 and R * S > 1 and not _c3_layer(H, W, K, R, S, stride, pad, residual is None and H % 16 == 0 and W % 32 == 0):
        # RGB input on a shape the first-layer kernel (csrc/conv_c3.hip) does not take: a zero 4th channel makes every tap one
        # aligned 16-byte load (scalar gathers ran at 22-32 TF); the zero channel adds nothing to the sums
        x, w, Cc, wkey = _pad_c4(x), _pad_c4(w), 4, None
    c_alg = C2                    # channels the reference convolves (FLOP accounting)
    d, P, Q = _conv_desc(N, H, W, Cc, K, R, S, stride, pad, kind="fwd")
    y = out if out is not None else torch.empty((N, P, Q, K), device=x.device, dtype=torch.float32)
    if bias is not None:
        _chk(bias, "bias")
    if residual is not None:
        _chk(residual, "residual")
        if tuple(residual.shape) != (N, P, Q, K):
            raise RuntimeError("conv2d_fwd: residual shape %s, expected %s" % (tuple(residual.shape), (N, P, Q, K)))
    if in_affine is not None:
        a_sc, a_sh, a_relu = in_affine
        _chk(a_sc, "in_affine scale"); _chk(a_sh, "in_affine shift")
        if tuple(a_sc.shape) != (N, Cc) or tuple(a_sh.shape) != (N, Cc):
            raise RuntimeError("conv2d_fwd: in_affine tables must be [N, C] = [%d, %d]" % (N, Cc))
        d.in_scale, d.in_shift, d.in_relu = a_sc.data_ptr(), a_sh.data_ptr(), 1 if a_relu else 0
    flop = 2.0 * N * P * Q * K * R * S * c_alg * alg_scale    # the dense conv the reference executes
    nbytes = 4.0 * (N * H * W * c_alg + K * R * S * c_alg + N * P * Q * K)
    xflop = 2.0 * N * P * Q * K * R * S * Cc
    halo_arith = d.compute in (COMPUTE_BF16X3, COMPUTE_BF16) or (d.compute == COMPUTE_F32 and _F32_HALO)
    kind_sfx = {COMPUTE_BF16X3: "x3", COMPUTE_BF16: "b16", COMPUTE_F32: "f32h"}[d.compute]
    if (halo_arith and stride == 2 and R == S and R % 2 == 1 and R > 1 and Cc % 16 == 0 and K >= 64
            and residual is None and w.numel() % 8 == 0 and _X3_S2):
        # parity sub-images through the LDS-halo kernel of the bf16 matrix cores (csrc/igemm_x3.hip): one workspace holds the fp32
        # class matrices and, behind them, their bfloat16 pieces in step-major order
        nw = w.numel()
        ws, repack = repack_cache.get("s2_fwd_" + kind_sfx, w, nw + (3 * nw + 1) // 2, wkey)
        part = _split_scratch(d, N * P * Q, K, Cc, y)
        _timed("conv2d_fwd", flop, nbytes, lambda: check(
            _lib.lib().cslgan_conv2d_s2_fwd_x3_f32(C.byref(d), _p(x), _p(w), _p(ws), C.c_void_p(ws.data_ptr() + 4 * nw), repack, _p(bias), act,
                                                   _p(y), _stream()),
            "conv2d_s2_fwd_x3"), exec_flop=xflop, tag=lambda: "N%d %dx%d C%d K%d R%d s2 %s" % (N, H, W, Cc, K, R, kind_sfx))
        del part
        repack_cache.packed()
        return y
    if stride == 2 and R == S and R % 2 == 1 and R > 1 and Cc % 32 == 0 and K >= 64 and residual is None:
        # parity sub-images through the LDS-halo kernel (the C entry falls back to the generic kernel for other grids)
        ws, repack = repack_cache.get("s2_fwd", w, w.numel(), wkey)
        _timed("conv2d_fwd", flop, nbytes, lambda: check(
            _lib.lib().cslgan_conv2d_s2_fwd_f32(C.byref(d), _p(x), _p(w), _p(ws), repack, _p(bias), act, _p(y), _stream()),
            "conv2d_s2_fwd"), exec_flop=xflop, tag=lambda: "N%d %dx%d C%d K%d R%d s2 halo" % (N, H, W, Cc, K, R))
        repack_cache.packed()
        return y
    if halo_arith and stride == 1 and R * S > 1 and Cc % 16 == 0 and K >= 64 and P % 8 == 0 and Q % 8 == 0 and K % 4 == 0:
        # the round-4 LDS-halo kernel reads the filter in step-major order: pre-split into bfloat16 pieces / pre-rounded / as an fp32 copy
        # (cached per parameter version)
        ws, repack = repack_cache.get({COMPUTE_BF16X3: "x3w", COMPUTE_BF16: "bf16w", COMPUTE_F32: "f32w"}[d.compute], w, (3 * w.numel() + 1) // 2, wkey, version=wversion)
        req, gpart = gn_partials._req, None
        if req is not None:
            gn_partials._req = None         # one conv per request
            cpg = K // req.groups if K % req.groups == 0 else 0
            if act == ACT_NONE and cpg in (1, 2, 4, 8, 16, 32) and P * Q // 64 <= 64 and K % 64 == 0:
                gpart = torch.empty(N * (P * Q // 64) * req.groups * 2, device=x.device, dtype=torch.float32)
                d.gn_part, d.gn_groups = gpart.data_ptr(), req.groups
                req.part = (gpart, P * Q // 64)
        _timed("conv2d_fwd", flop, nbytes, lambda: check(
            _lib.lib().cslgan_conv2d_fwd_x3_f32(C.byref(d), _p(x), _p(w), _p(ws), repack, _p(bias), _p(residual), act, _p(y), _stream()),
            "conv2d_fwd_x3"), exec_flop=xflop, tag=lambda: "N%d %dx%d C%d K%d R%d s%d" % (N, H, W, Cc, K, R, stride))
        if repack:
            pieces = {COMPUTE_BF16X3: 3, COMPUTE_BF16: 1, COMPUTE_F32: 0}[d.compute]
            repack_cache.set_rebuild(lambda: check(_lib.lib().cslgan_split_filter_x3_f32(_p(w), K, R * S, Cc, _p(ws), pieces, _stream()), "split_filter_x3"))
        repack_cache.packed()
        return y
    _timed("conv2d_fwd", flop, nbytes, lambda: check(
        _lib.lib().cslgan_conv2d_fwd_f32(C.byref(d), _p(x), _p(w), _p(bias), _p(residual), act, _p(y), _stream()),
        "conv2d_fwd"), exec_flop=xflop, tag=lambda: "N%d %dx%d C%d K%d R%d s%d" % (N, H, W, Cc, K, R, stride))
    return y


def depth_to_space(x, inverse=False):
    """UpsampleConv's data movement (DCResNet_models.py:13-15) on NHWC data: [N,H,W,C] -> [N,2H,2W,C/4] with
    out[n,2h+i,2w+j,c'] = x[n,h,w,4c'+2i+j]; inverse=True maps [N,2H,2W,C/4] back to [N,H,W,C] (its gradient)."""
    if x.dtype == torch.bfloat16:          # parity-test route only (the product path shuffles inside the normalisation kernel)
        return cast_bf16(depth_to_space(cast_f32(x), inverse=inverse))
    _chk(x, "x")
    if inverse:
        N, H2, W2, Cq = x.shape
        if H2 % 2 or W2 % 2:
            raise RuntimeError("depth_to_space(inverse): spatial dims must be even, got %dx%d" % (H2, W2))
        H, W, Cc = H2 // 2, W2 // 2, Cq * 4
        out = torch.empty((N, H, W, Cc), device=x.device, dtype=torch.float32)
    else:
        N, H, W, Cc = x.shape
        if Cc % 4:
            raise RuntimeError("depth_to_space: C=%d is not a multiple of 4 (the HIP UpsampleConv path needs C %% 4 == 0)" % Cc)
        out = torch.empty((N, 2 * H, 2 * W, Cc // 4), device=x.device, dtype=torch.float32)
    check(_lib.lib().cslgan_depth_to_space_f32(_p(x), N, H, W, Cc, 1 if inverse else 0, _p(out), _stream()), "depth_to_space")
    return out


def fold_channels4(w, wkey=None):
    """[K,R,S,C] -> [K,R,S,C/4]: wf[...,c'] = sum_q w[...,c'+q*C/4] — the filter UpsampleConv's conv applies to the
    depth-to-space tensor (whose four channel groups are identical).  Cached per parameter version when wkey is given."""
    _chk(w, "w")
    K, R, S, Cc = w.shape
    if Cc % 4:
        raise RuntimeError("fold_channels4: C=%d is not a multiple of 4" % Cc)
    wf, fresh = repack_cache.get("fold4", w, w.numel() // 4, wkey)
    if fresh:
        fold = lambda: check(_lib.lib().cslgan_fold_channels4_f32(_p(w), K * R * S, Cc, 0, _p(wf), _stream()), "fold_channels4")
        fold()
        repack_cache.set_rebuild(fold)
        repack_cache.packed()
    return wf.view(K, R, S, Cc // 4)


def unfold_channels4(gwf):
    """Gradient of fold_channels4: [K,R,S,C/4] -> [K,R,S,C] with gw[...,c'+q*C/4] = gwf[...,c']."""
    _chk(gwf, "gwf")
    K, R, S, Cq = gwf.shape
    gw = torch.empty((K, R, S, 4 * Cq), device=gwf.device, dtype=torch.float32)
    check(_lib.lib().cslgan_fold_channels4_f32(_p(gwf), K * R * S, 4 * Cq, 1, _p(gw), _stream()), "unfold_channels4")
    return gw


def _c3_layer(H, W, K, R, S, stride, pad, grid_ok):
    """The critic's first conv as csrc/conv_c3.hip takes it (c3_shape there): 3 -> 64 channels, 5x5, stride 2, pad 2, even image."""
    return K == 64 and R == 5 and S == 5 and stride == 2 and pad == 2 and H % 2 == 0 and W % 2 == 0 and grid_ok


def _pad_c4(t):
    """[..., 3] -> [..., 4] with a zero last channel (one small elementwise pass)."""
    out = torch.zeros(t.shape[:-1] + (4,), device=t.device, dtype=t.dtype)
    out[..., :3] = t
    return out


def _conv2d_dgrad_stored(gy, w, in_hw, stride, pad, mask, wkey, out_dtype):
    """conv2d_dgrad with bf16-stored activation gradients: the bf16-stored kernel when gy is bf16 with K % 8 == 0, otherwise the
    fp32 kernels between casts."""
    _chk(gy, "gy", allow_bf16=True); _chk(w, "w")
    N, P, Q, K = gy.shape
    K2, R, S, Cc = w.shape
    H, W = in_hw
    if out_dtype is None:
        out_dtype = torch.bfloat16 if (gy.dtype == torch.bfloat16 and Cc > 4) else torch.float32
    if mask is not None and mask.dtype != out_dtype:
        mask = cast_bf16(mask) if out_dtype == torch.bfloat16 else cast_f32(mask)
    if (gy.dtype == torch.float32 and out_dtype == torch.bfloat16 and K == 1 and (P, Q, R, S, H, W) == (1, 1, 1, 1, 1, 1) and Cc % 8 == 0
            and N <= 65535 and stride == 1 and pad == 0):
        # the critic's head: fp32 loss cotangent, bf16 features
        gx = torch.empty((N, 1, 1, Cc), device=gy.device, dtype=torch.bfloat16)
        if mask is not None:
            _chk(mask, "mask", allow_bf16=True)
        _timed("conv2d_dgrad", 2.0 * N * Cc, 2.0 * N * Cc * (2 if mask is not None else 1) + 4.0 * Cc, lambda: check(
            _lib.lib().cslgan_linear_k1_dgrad_bf16s(_p(gy), _p(w), _p(mask), N, Cc, _p(gx), _stream()), "linear_k1_dgrad_bf16s"),
            tag=lambda: "N%d 1x1 C%d K1 R1 s1 bf16s" % (N, Cc))
        return gx
    if gy.dtype == torch.bfloat16 and K % 8 == 0 and stride in (1, 2) and Cc > 4:
        P2, Q2 = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
        if K2 != K or (P2, Q2) != (P, Q):
            raise RuntimeError("conv2d_dgrad: gy shape %s inconsistent with input %dx%d" % (tuple(gy.shape), H, W))
        d = ConvT(N, H, W, Cc, K, R, S, stride, pad, COMPUTE_BF16, P, Q)
        gx = torch.empty((N, H, W, Cc), device=gy.device, dtype=out_dtype)
        if mask is not None:
            _chk(mask, "mask", allow_bf16=True)
            if tuple(mask.shape) != tuple(gx.shape):
                raise RuntimeError("conv2d_dgrad: mask shape mismatch")
        ws, repack = repack_cache.get("bf16s_dgrad%d" % stride, w, (w.numel() + 1) // 2, wkey)
        flop = 2.0 * N * P * Q * K * R * S * Cc
        nbytes = 2.0 * (K * R * S * Cc + N * P * Q * K) + gx.element_size() * float(N * H * W * Cc)
        _timed("conv2d_dgrad", flop, nbytes, lambda: check(
            _lib.lib().cslgan_conv2d_dgrad_bf16s(C.byref(d), _p(gy), _p(w), _p(ws), repack, _p(mask), _p(gx),
                                                 1 if out_dtype == torch.bfloat16 else 0, _stream()), "conv2d_dgrad_bf16s"),
            tag=lambda: "N%d %dx%d C%d K%d R%d s%d bf16s" % (N, H, W, Cc, K, R, stride))
        repack_cache.packed()
        return gx
    if (gy.dtype == torch.bfloat16 and out_dtype == torch.float32 and mask is None and Cc <= 4 and K == 64 and stride in (1, 2) and R * S <= 25
            and H % stride == 0 and W % stride == 0 and (H // stride) % 8 == 0 and (W // stride) % 8 == 0 and (R <= 3 or stride == 2)):
        # the critic's first layer: the image gradient from a bf16-stored output gradient on the vector-ALU kernel
        d = ConvT(N, H, W, Cc, K, R, S, stride, pad, COMPUTE_F32, P, Q)
        gx = torch.empty((N, H, W, Cc), device=gy.device, dtype=torch.float32)
        ws, repack = repack_cache.get("dgrad%d" % stride, w, w.numel(), wkey)
        _timed("conv2d_dgrad", 2.0 * N * P * Q * K * R * S * Cc, 2.0 * N * P * Q * K + 4.0 * N * H * W * Cc, lambda: check(
            _lib.lib().cslgan_conv2d_dgrad_skinny_bf16in(C.byref(d), _p(gy), _p(w), _p(ws), repack, _p(gx), _stream()), "conv2d_dgrad_skinny_bf16in"),
            tag=lambda: "N%d %dx%d C%d K%d R%d s%d bf16in" % (N, H, W, Cc, K, R, stride))
        repack_cache.packed()
        return gx
    gx = conv2d_dgrad(cast_f32(gy), w, in_hw, stride=stride, pad=pad, mask=cast_f32(mask), wkey=wkey)
    return cast_bf16(gx) if out_dtype == torch.bfloat16 else gx


def conv2d_dgrad(gy, w, in_hw, stride=1, pad=0, mask=None, wkey=None, out_dtype=None):
    """gx[N,H,W,C] = conv_transpose(gy[N,P,Q,K], w[K,R,S,C]) (* lrelu'(mask)).  out_dtype as for conv2d_fwd."""
    if gy.dtype == torch.bfloat16 or out_dtype == torch.bfloat16 or _is_bf16(mask):
        return _conv2d_dgrad_stored(gy, w, in_hw, stride, pad, mask, wkey, out_dtype)
    _chk(gy, "gy"); _chk(w, "w")
    N, P, Q, K = gy.shape
    K2, R, S, Cc = w.shape
    H, W = in_hw
    d, P2, Q2 = _conv_desc(N, H, W, Cc, K, R, S, stride, pad, kind="dgrad")
    if K2 != K or (P2, Q2) != (P, Q):
        raise RuntimeError("conv2d_dgrad: gy shape %s inconsistent with input %dx%d" % (tuple(gy.shape), H, W))
    gx = torch.empty((N, H, W, Cc), device=gy.device, dtype=torch.float32)
    if mask is not None:
        _chk(mask, "mask")
        if tuple(mask.shape) != tuple(gx.shape):
            raise RuntimeError("conv2d_dgrad: mask shape mismatch")
    flop = 2.0 * N * P * Q * K * R * S * Cc
    nbytes = 4.0 * (N * H * W * Cc + K * R * S * Cc + N * P * Q * K)
    if ((d.compute in (COMPUTE_BF16X3, COMPUTE_BF16) or (d.compute == COMPUTE_F32 and _F32_HALO)) and K % 16 == 0 and Cc >= 64 and Cc % 4 == 0
            and R * S > 1 and w.numel() % 8 == 0
            and (H // stride) % 4 == 0 and (W // stride) % 4 == 0 and H % stride == 0 and W % stride == 0 and _X3_DGRAD):
        # LDS-halo kernel of the bf16 matrix cores (csrc/igemm_x3.hip): the repacked class matrices and, behind them, their bfloat16
        # pieces in step-major order share one cached workspace
        nw = w.numel()
        sfx = {COMPUTE_BF16X3: "x3", COMPUTE_BF16: "b16", COMPUTE_F32: "f32h"}[d.compute]
        ws, repack = repack_cache.get("dgrad%d_%s" % (stride, sfx), w, nw + (3 * nw + 1) // 2, wkey)
        part = _split_scratch(d, N * H * W, Cc, K, gx)
        _timed("conv2d_dgrad", flop, nbytes, lambda: check(
            _lib.lib().cslgan_conv2d_dgrad_x3_f32(C.byref(d), _p(gy), _p(w), _p(ws), C.c_void_p(ws.data_ptr() + 4 * nw), repack, _p(mask), _p(gx),
                                                  _stream()), "conv2d_dgrad_x3"),
            tag=lambda: "N%d %dx%d C%d K%d R%d s%d %s" % (N, H, W, Cc, K, R, stride, sfx))
        del part
        repack_cache.packed()
        return gx
    ws, repack = repack_cache.get("dgrad%d" % stride, w, w.numel(), wkey)
    _timed("conv2d_dgrad", flop, nbytes, lambda: check(
        _lib.lib().cslgan_conv2d_dgrad_f32(C.byref(d), _p(gy), _p(w), _p(ws), repack, _p(mask), _p(gx), _stream()), "conv2d_dgrad"),
        tag=lambda: "N%d %dx%d C%d K%d R%d s%d" % (N, H, W, Cc, K, R, stride))
    repack_cache.packed()
    return gx


def _wgh_arith(S):
    """The LDS-resident weight-gradient kernels exist in exact fp32 (2..5 filter columns) and in the three-piece form (5 columns)."""
    return _compute == COMPUTE_F32 or (_compute == COMPUTE_BF16X3 and S == 5)


def dense_wgrad_group(N, K, Cc, R, S, PQ, stride=1, out_hw=None):
    """Samples per slab for a dense (summed) weight gradient.  The slab count sets the workgroup count, and the launch
    time follows how well that count fills 256 CUs x 3 resident workgroups (800 workgroups take two rounds, 3200 take
    4.2: measured 1.45 vs 1.16 ms on the same 55 GFLOP); more slabs cost their write + re-read by the column sum.
    Model: t(g) = FLOP / (100 TF x fill(g)) + 2 x slab bytes / 4 TB/s, minimised over g | N."""
    if (_wgh_arith(S) and out_hw is not None and stride in (1, 2) and 2 <= S <= 5 and K % 64 == 0 and Cc % 64 == 0
            and out_hw[0] % 8 == 0 and out_hw[1] % 8 == 0):
        # igemm_wgh (LDS-resident operands): (K/128)(C/64)R tiles per slab.  Big launches (the generator's convs, >= 40 GFLOP):
        # ONE slab — the kernel splits the patch loop over workgroups itself (atomic adds), no slab traffic.  Small ones:
        # the largest group that keeps >= 1024 workgroups (scripts/wgrad_group_sweep.py).
        if 2.0 * N * PQ * K * R * S * Cc >= 40e9:
            return N
        tiles = (K // 128 if K % 128 == 0 else K // 64) * (Cc // 64) * R
        best = 1
        g = 1
        while g <= N:
            if N % g == 0 and (N // g) * tiles >= 1024:
                best = g
            g *= 2
        return best
    ndim = R * S * Cc
    bn = 256 if (32 < K <= 64 and ndim >= 1024) else 128
    tiles = ((K + 127) // 128 if K > 64 else 1) * ((ndim + bn - 1) // bn)
    flop = 2.0 * N * PQ * K * ndim
    out_bytes = 4.0 * K * ndim
    best, best_t = 1, None
    g = 1
    while g <= N:
        if N % g == 0:
            blocks = (N // g) * tiles
            if blocks >= 1024 or g == 1:
                rounds = -(-blocks // 768)
                t = flop / (100e12 * blocks / (rounds * 768.0)) + (2.0 * (N // g) * out_bytes / 4e12 if N // g > 1 else 0.0)
                if best_t is None or t < best_t:
                    best, best_t = g, t
        g *= 2
    return best


def gram_norms_eligible(gy_shape, x_shape):
    """Shapes cslgan_conv2d_wgrad_sqnorm_gram_f32 accepts (and where the Gram form is cheaper than the product)."""
    _, P, Q, K = gy_shape
    return P * Q <= 64 and K % 32 == 0 and x_shape[-1] % 32 == 0


def gram_norms_preferred(gy_shape, x_shape, stride):
    """Shapes where the Gram form runs on the pixel-pair kernels (output pixels and input pixels per stride-parity class both
    <= 16: gram_sqnorm_small_kernel; both <= 64: gram_sqnorm_cls64_kernel): there it is far cheaper than the product (the critic's
    conv3: 6.3 vs 105 MFLOP per sample, and no 419 MB of per-sample gradients), so the engine uses it for norms and ghost clipping."""
    _, P, Q, K = gy_shape
    _, H, W, Cc = x_shape
    if P * Q == 1 and H * W == 1:
        return True              # a linear layer: ||gy_b x_b^T||^2 = ||gy_b||^2 ||x_b||^2
    if stride not in (1, 2) or P * Q > _GRAM_MAX_PIX or K % 64 or Cc % 32:
        return False
    return ((H + stride - 1) // stride) * ((W + stride - 1) // stride) <= _GRAM_MAX_PIX


_GRAM_MAX_PIX = int(os.environ.get("CSLGAN_GHOST_MAX_PIX", "64"))     # 16 restores the round-1 rule (ghost clipping for conv4 + linear only)


def conv2d_wgrad_sqnorm_gram(gy, x, R, S, stride=1, pad=0, alpha=1.0, sq=None):
    """sq[N] += ||alpha * per-sample weight gradient||^2 from the two PQ x PQ Gram matrices (no gradient formed)."""
    gy, x = cast_f32(gy), cast_f32(x)       # fp32 kernels only (ghost clipping is not combined with bf16 storage)
    _chk(gy, "gy"); _chk(x, "x")
    N, H, W, Cc = x.shape
    N2, P, Q, K = gy.shape
    d, P2, Q2 = _conv_desc(N, H, W, Cc, K, R, S, stride, pad, kind="gram")
    if N2 != N or (P2, Q2) != (P, Q):
        raise RuntimeError("conv2d_wgrad_sqnorm_gram: gy %s inconsistent with x %s" % (tuple(gy.shape), tuple(x.shape)))
    if sq is None:
        sq = torch.zeros(N, device=x.device, dtype=torch.float32)
    _chk(sq, "sq")
    if sq.numel() != N:
        raise RuntimeError("conv2d_wgrad_sqnorm_gram: sq needs %d entries" % N)
    if P * Q == 1 and H * W == 1 and R == 1 and S == 1:
        # linear layer: the 1x1 Gram matrices are the two row norms (one pass of the contract norm kernel over gy and x)
        both = sample_sqnorm([gy.reshape(N, K), x.reshape(N, Cc)])
        sq.view(-1).addcmul_(both[0], both[1], value=float(alpha) ** 2)
        return sq
    flop = 2.0 * N * (P * Q) ** 2 * (K + R * S * Cc)          # the tap-by-tap Gram form
    cls_pix = ((H + stride - 1) // stride) * ((W + stride - 1) // stride)
    if stride in (1, 2) and P * Q <= 16 and cls_pix <= 16:   # pixel-pair kernels: s^2 class Gram matrices (+ GY GY^T once / per class)
        xflop = 2.0 * N * ((P * Q) ** 2 * K + stride * stride * cls_pix ** 2 * Cc)
    elif stride in (1, 2) and P * Q <= 64 and cls_pix <= 64:
        xflop = 2.0 * N * stride * stride * 64 * 64 * (K + Cc)
    else:
        xflop = flop
    _timed("conv2d_wgrad_gram_norms", flop, 4.0 * (gy.numel() + x.numel()), lambda: check(
        _lib.lib().cslgan_conv2d_wgrad_sqnorm_gram_f32(C.byref(d), _p(gy), _p(x), float(alpha), _p(sq), _stream()),
        "conv2d_wgrad_sqnorm_gram"), exec_flop=xflop, tag=lambda: "N%d %dx%d C%d K%d R%d s%d" % (N, H, W, Cc, K, R, stride))
    return sq


def conv2d_wgrad_grouped(gy, x, R, S, stride=1, pad=0, group=1, alpha=1.0, want_gw=True, sq=None, out=None, row_scale=None):
    """gw[N/group,K,R,S,C] (per-group weight gradients) and/or sq[N/group] += ||alpha*gw_g||^2.
    row_scale [N]: gy of sample n is weighted by row_scale[n] (clip-weighted sums; fp32 output, no sq)."""
    if gy.dtype == torch.bfloat16 or x.dtype == torch.bfloat16:
        if (gy.dtype == x.dtype and gy.shape[-1] % 8 == 0 and x.shape[-1] % 8 == 0 and (want_gw or sq is not None)
                and (row_scale is None or _scaled_stored_ok(gy, sq, want_gw, out))):
            return _conv2d_wgrad_grouped_stored(gy, x, R, S, stride, pad, group, alpha, want_gw, sq, out, row_scale)
        if (gy.dtype == torch.float32 and x.dtype == torch.bfloat16 and gy.shape[-1] == 1 and R == 1 and S == 1
                and tuple(x.shape[1:3]) == (1, 1) and x.shape[-1] % 8 == 0 and (out is None or out.dtype == torch.float32)
                and x.shape[0] % group == 0 and x.shape[0] // group <= 65535 and (row_scale is None or (sq is None and want_gw))):
            # the critic's head: fp32 loss cotangent times bf16 feature rows
            N, Cc = x.shape[0], x.shape[-1]
            _chk(gy, "gy"); _chk(x, "x", allow_bf16=True)
            gw = None
            if want_gw:
                gw = out if out is not None else torch.empty((N // group, 1, 1, 1, Cc), device=x.device, dtype=torch.float32)
                _chk(gw, "gw")
            if sq is not None:
                _chk(sq, "sq")
            if row_scale is not None:
                _chk(row_scale, "row_scale")
                if row_scale.numel() != N:
                    raise RuntimeError("conv2d_wgrad: row_scale needs [N] factors")
            _timed("conv2d_wgrad_grouped" + ("" if want_gw else "_normonly"), 2.0 * N * Cc, 2.0 * N * Cc + (4.0 * (N // group) * Cc if want_gw else 0.0),
                   lambda: check(_lib.lib().cslgan_linear_k1_wgrad_bf16s(_p(gy), _p(x), _p(row_scale), N, Cc, group, float(alpha), _p(gw), _p(sq), _stream()),
                                 "linear_k1_wgrad_bf16s"), tag=lambda: "N%d 1x1 C%d K1 R1 s1 g%d bf16s%s" % (N, Cc, group, " scaled" if row_scale is not None else ""))
            return gw
        if (gy.dtype == torch.bfloat16 and x.dtype == torch.float32 and x.shape[-1] == 3 and group == 1 and row_scale is None
                and (out is None or out.dtype == torch.float32) and (want_gw or sq is not None)
                and _c3_layer(x.shape[1], x.shape[2], gy.shape[-1], R, S, stride, pad, gy.shape[2] in (16, 32, 64) and gy.shape[1] % (128 // gy.shape[2]) == 0)):
            # the RGB first layer: bf16-stored output gradient, fp32 image, the first-layer kernel reads gy as stored
            N, H, W, Cc = x.shape
            _, P, Q, K = gy.shape
            _chk(gy, "gy", allow_bf16=True); _chk(x, "x")
            d = ConvT(N, H, W, Cc, K, R, S, stride, pad, COMPUTE_F32, P, Q)
            gw = None
            if want_gw:
                gw = out if out is not None else torch.empty((N, K, R, S, Cc), device=x.device, dtype=torch.float32)
                _chk(gw, "gw")
            if sq is not None:
                _chk(sq, "sq")
            _timed("conv2d_wgrad_grouped" + ("" if want_gw else "_normonly"), 2.0 * N * P * Q * K * R * S * Cc, 4.0 * N * H * W * Cc + 2.0 * N * P * Q * K,
                   lambda: check(_lib.lib().cslgan_conv2d_c3_wgrad_bf16gy(C.byref(d), _p(gy), _p(x), float(alpha), _p(gw), _p(sq), _stream()),
                                 "conv2d_c3_wgrad_bf16gy"), tag=lambda: "N%d %dx%d C%d K%d R%d s%d g1 bf16gy" % (N, H, W, Cc, K, R, stride))
            return gw
        gy, x = cast_f32(gy), cast_f32(x)       # mixed element types / shapes the bf16-stored kernel does not take
    _chk(gy, "gy"); _chk(x, "x")
    N, H, W, Cc = x.shape
    N2, P, Q, K = gy.shape
    c3 = (Cc == 3 and group == 1 and row_scale is None and (out is None or out.dtype == torch.float32)
          and _c3_layer(H, W, K, R, S, stride, pad, Q in (16, 32, 64) and P % (128 // Q) == 0))
    if Cc == 3 and R * S > 1 and row_scale is None and (out is None or out.dtype == torch.float32) and not c3:
        # RGB input on a shape the first-layer kernel does not take: run on a zero-padded 4th channel (aligned 16-byte gathers),
        # then drop that channel's (zero) gradients
        g4 = conv2d_wgrad_grouped(gy, _pad_c4(x), R, S, stride=stride, pad=pad, group=group, alpha=alpha, want_gw=want_gw, sq=sq)
        if g4 is None:
            return None
        if out is not None:
            out.view(g4.shape[:-1] + (3,)).copy_(g4[..., :3])
            return out
        return g4[..., :3].contiguous()
    d, P2, Q2 = _conv_desc(N, H, W, Cc, K, R, S, stride, pad, group=group)
    if N2 != N or (P2, Q2) != (P, Q):
        raise RuntimeError("conv2d_wgrad: gy %s inconsistent with x %s" % (tuple(gy.shape), tuple(x.shape)))
    if N % group:
        raise RuntimeError("conv2d_wgrad: N=%d not divisible by group=%d" % (N, group))
    G = N // group
    if row_scale is not None:
        _chk(row_scale, "row_scale")
        if row_scale.numel() != N or sq is not None or not want_gw:
            raise RuntimeError("conv2d_wgrad: row_scale needs [N] factors, a gradient output and no sq")
        gw = out if out is not None else torch.empty((G, K, R, S, Cc), device=x.device, dtype=torch.float32)
        _chk(gw, "gw")
        flop = 2.0 * N * P * Q * K * R * S * Cc
        _timed("conv2d_wgrad_grouped", flop, 4.0 * (N * H * W * Cc + N * P * Q * K + G * K * R * S * Cc), lambda: check(
            _lib.lib().cslgan_conv2d_wgrad_scaled_f32(C.byref(d), _p(gy), _p(x), _p(row_scale), group, float(alpha), _p(gw), _stream()),
            "conv2d_wgrad_scaled"), tag=lambda: "N%d %dx%d C%d K%d R%d s%d g%d scaled" % (N, H, W, Cc, K, R, stride, group))
        return gw
    gw = None
    scratch = False
    if not want_gw and sq is not None:
        # norms only, but a layer with a handful of tiles and a long pixel loop (the 3-channel first conv) runs several
        # times faster when its pixels are split over workgroups, which needs a (small) output to accumulate into
        tiles = ((K + 63) // 64) * ((R * S * Cc + 127) // 128) * G
        if not c3 and tiles < 192 and group * P * Q >= 512 and G * K * R * S * Cc <= (1 << 22):
            want_gw, scratch = True, True
    if want_gw:
        gw = out if out is not None else torch.empty((G, K, R, S, Cc), device=x.device, dtype=torch.float32)
        _chk(gw, "gw", allow_bf16=True)
    if sq is not None:
        _chk(sq, "sq")
    flop = 2.0 * N * P * Q * K * R * S * Cc
    esz = gw.element_size() if want_gw else 0
    nbytes = 4.0 * (N * H * W * Cc + N * P * Q * K) + float(esz) * (G * K * R * S * Cc)
    L = _lib.lib()
    fn = L.cslgan_conv2d_wgrad_grouped_bf16out_f32 if (want_gw and gw.dtype == torch.bfloat16) else L.cslgan_conv2d_wgrad_grouped_f32
    _timed("conv2d_wgrad_grouped" + ("" if (want_gw and not scratch) else "_normonly"), flop, nbytes, lambda: check(
        fn(C.byref(d), _p(gy), _p(x), group, float(alpha), _p(gw), _p(sq), _stream()), "conv2d_wgrad_grouped"),
        tag=lambda: "N%d %dx%d C%d K%d R%d s%d g%d" % (N, H, W, Cc, K, R, stride, group))
    return None if scratch else gw


def _scaled_stored_ok(gy, sq, want_gw, out):
    """Shapes cslgan_conv2d_wgrad_scaled_bf16s takes: a K tile (64 pixels) must lie inside one sample, fp32 gradient out, no norms."""
    return (gy.shape[1] * gy.shape[2]) % 64 == 0 and sq is None and want_gw and (out is None or out.dtype == torch.float32)


def _conv2d_wgrad_grouped_stored(gy, x, R, S, stride, pad, group, alpha, want_gw, sq, out, row_scale=None):
    """conv2d_wgrad_grouped on bf16 gy and bf16 x (cslgan_conv2d_wgrad_grouped_bf16s): fp32 (or bf16) gw and / or sq.
    row_scale [N] (clip-weighted sums): cslgan_conv2d_wgrad_scaled_bf16s — the fp32 weight meets each sample's accumulated product."""
    _chk(gy, "gy", allow_bf16=True); _chk(x, "x", allow_bf16=True)
    N, H, W, Cc = x.shape
    N2, P, Q, K = gy.shape
    P2, Q2 = conv_out_size(H, R, stride, pad), conv_out_size(W, S, stride, pad)
    if N2 != N or (P2, Q2) != (P, Q):
        raise RuntimeError("conv2d_wgrad: gy %s inconsistent with x %s" % (tuple(gy.shape), tuple(x.shape)))
    if N % group:
        raise RuntimeError("conv2d_wgrad: N=%d not divisible by group=%d" % (N, group))
    G = N // group
    d = ConvT(N, H, W, Cc, K, R, S, stride, pad, COMPUTE_BF16, P, Q)
    gw = None
    if want_gw:
        gw = out if out is not None else torch.empty((G, K, R, S, Cc), device=x.device, dtype=torch.float32)
        _chk(gw, "gw", allow_bf16=True)
    if sq is not None:
        _chk(sq, "sq")
    flop = 2.0 * N * P * Q * K * R * S * Cc
    nbytes = 2.0 * (N * H * W * Cc + N * P * Q * K) + (float(gw.element_size()) * (G * K * R * S * Cc) if want_gw else 0.0)
    if row_scale is not None:
        _chk(row_scale, "row_scale")
        if row_scale.numel() != N:
            raise RuntimeError("conv2d_wgrad: row_scale needs [N] factors")
        _timed("conv2d_wgrad_grouped", flop, nbytes, lambda: check(
            _lib.lib().cslgan_conv2d_wgrad_scaled_bf16s(C.byref(d), _p(gy), _p(x), _p(row_scale), group, float(alpha), _p(gw), _stream()),
            "conv2d_wgrad_scaled_bf16s"), tag=lambda: "N%d %dx%d C%d K%d R%d s%d g%d bf16s scaled" % (N, H, W, Cc, K, R, stride, group))
        return gw
    _timed("conv2d_wgrad_grouped" + ("" if want_gw else "_normonly"), flop, nbytes, lambda: check(
        _lib.lib().cslgan_conv2d_wgrad_grouped_bf16s(C.byref(d), _p(gy), _p(x), group, float(alpha), _p(gw),
                                                     1 if _is_bf16(gw) else 0, _p(sq), _stream()), "conv2d_wgrad_grouped_bf16s"),
        tag=lambda: "N%d %dx%d C%d K%d R%d s%d g%d bf16s" % (N, H, W, Cc, K, R, stride, group))
    return gw


def wgrad_blocks_eligible(gy_shape, x_shape, R, S, stride):
    """Shapes cslgan_conv2d_wgrad_blocks_f32 takes (the LDS-resident fp32 kernel): mirrors wgh_eligible in csrc/igemm_wgh.hip."""
    _, P, Q, K = gy_shape
    Cc = x_shape[-1]
    return (_wgh_arith(S) and stride in (1, 2) and 2 <= S <= 5 and K % 64 == 0 and Cc % 64 == 0 and P % 8 == 0 and Q % 8 == 0)


def conv2d_wgrad_blocks(gy, x, R, S, stride, pad, alpha, blocks):
    """Per-sample weight gradients of consecutive row blocks in ONE launch.  blocks: [(n_rows, gw_out or None, sq or None)] —
    gw_out [n_rows, K*R*S*C] fp32 (None: nothing stored), sq [n_rows] accumulated (None: no norms)."""
    _chk(gy, "gy"); _chk(x, "x")
    N, H, W, Cc = x.shape
    N2, P, Q, K = gy.shape
    d, P2, Q2 = _conv_desc(N, H, W, Cc, K, R, S, stride, pad)
    if N2 != N or (P2, Q2) != (P, Q) or sum(b[0] for b in blocks) != N:
        raise RuntimeError("conv2d_wgrad_blocks: blocks / gy %s inconsistent with x %s" % (tuple(gy.shape), tuple(x.shape)))
    nb = len(blocks)
    first = (C.c_int32 * nb)()
    gws, sqs = (C.c_void_p * nb)(), (C.c_void_p * nb)()
    r0 = 0
    for i, (n, gw, sq) in enumerate(blocks):
        first[i] = r0
        r0 += n
        if gw is not None:
            _chk(gw, "gw")
            if gw.numel() != n * K * R * S * Cc:
                raise RuntimeError("conv2d_wgrad_blocks: gw of block %d has %d elements, expected %d" % (i, gw.numel(), n * K * R * S * Cc))
        if sq is not None:
            _chk(sq, "sq")
            if sq.numel() != n:
                raise RuntimeError("conv2d_wgrad_blocks: sq of block %d needs %d entries" % (i, n))
        gws[i] = None if gw is None else gw.data_ptr()
        sqs[i] = None if sq is None else sq.data_ptr()
    flop = 2.0 * N * P * Q * K * R * S * Cc
    nbytes = 4.0 * (x.numel() + gy.numel() + sum(b[1].numel() for b in blocks if b[1] is not None))
    _timed("conv2d_wgrad_grouped", flop, nbytes, lambda: check(
        _lib.lib().cslgan_conv2d_wgrad_blocks_f32(C.byref(d), _p(gy), _p(x), float(alpha), nb, first, gws, sqs, _stream()),
        "conv2d_wgrad_blocks"), tag=lambda: "N%d %dx%d C%d K%d R%d s%d g1 blocks%d" % (N, H, W, Cc, K, R, stride, nb))


def conv2d_wgrad_dense(gy, x, R, S, stride=1, pad=0, alpha=1.0, row_scale=None, out=None, want_rows=False):
    """The summed weight gradient [K,R,S,C] of a batch: slabs of the grouped MFMA kernel + a column sum, or the
    vector-ALU kernel for 1..4 output channels.  out (optional, flat fp32 [K*R*S*C]): destination of the sum.
    want_rows: return the UN-SUMMED slabs [n_slabs, K*R*S*C] instead (the caller column-sums them together with other
    contributions in one launch: PrivacyEngine._add_dense_rows)."""
    if (gy.dtype == torch.float32 and x.dtype == torch.bfloat16 and gy.shape[-1] == 1 and R == 1 and S == 1
            and tuple(x.shape[1:3]) == (1, 1) and x.shape[-1] % 8 == 0):
        # the critic's head on bf16 features: weighted sums of feature rows in slabs of 8 samples + a column sum
        N = x.shape[0]
        slabs = conv2d_wgrad_grouped(gy, x, 1, 1, group=8 if N % 8 == 0 else 1, alpha=alpha, row_scale=row_scale)
        res = torch.empty(x.shape[-1], device=x.device, dtype=torch.float32) if out is None else out
        sum_rows(slabs.reshape(slabs.shape[0], -1), res.view(-1))
        return res.view(1, 1, 1, -1)
    c3_mixed = (gy.dtype == torch.bfloat16 and x.dtype == torch.float32 and x.shape[-1] == 3 and row_scale is None
                and _c3_layer(x.shape[1], x.shape[2], gy.shape[-1], R, S, stride, pad, gy.shape[2] in (16, 32, 64) and gy.shape[1] % (128 // gy.shape[2]) == 0))
    if (gy.dtype == torch.bfloat16 or x.dtype == torch.bfloat16) and not c3_mixed and not (
            gy.dtype == x.dtype and gy.shape[-1] % 8 == 0 and x.shape[-1] % 8 == 0
            and (row_scale is None or (gy.shape[1] * gy.shape[2]) % 64 == 0)):
        gy, x = cast_f32(gy), cast_f32(x)
    N, H, W, Cc = x.shape
    _, P, Q, K = gy.shape
    if (K <= 4 and Cc == 64 and stride == 1 and R * S <= 9 and P % 8 == 0 and Q % 8 == 0 and row_scale is None):
        _chk(gy, "gy"); _chk(x, "x")
        d, P2, Q2 = _conv_desc(N, H, W, Cc, K, R, S, stride, pad)
        if (P2, Q2) != (P, Q):
            raise RuntimeError("conv2d_wgrad_dense: gy %s inconsistent with x %s" % (tuple(gy.shape), tuple(x.shape)))
        nb = min(512, N * (P // 8) * (Q // 8))
        partial = torch.empty((nb, K * R * S * Cc), device=x.device, dtype=torch.float32)
        flop = 2.0 * N * P * Q * K * R * S * Cc
        _timed("conv2d_wgrad_grouped", flop, 4.0 * (x.numel() + gy.numel()), lambda: check(
            _lib.lib().cslgan_conv2d_wgrad_skinny_f32(C.byref(d), _p(gy), _p(x), float(alpha), _p(partial), nb, _stream()),
            "conv2d_wgrad_skinny"), tag=lambda: "N%d %dx%d C%d K%d R%d skinny" % (N, H, W, Cc, K, R))
        out = torch.empty(K * R * S * Cc, device=x.device, dtype=torch.float32) if out is None else out
        sum_rows(partial, out)
        return out.view(K, R, S, Cc)
    if Cc == 3 and row_scale is None and _c3_layer(H, W, K, R, S, stride, pad, Q in (16, 32, 64) and P % (128 // Q) == 0):
        group = 1            # first-layer kernel: per-image gradients (19 KB each), summed below
    else:
        group = dense_wgrad_group(N, K, Cc, R, S, P * Q, stride=stride, out_hw=(P, Q))
    slabs = conv2d_wgrad_grouped(gy, x, R, S, stride=stride, pad=pad, group=group, alpha=alpha, row_scale=row_scale)
    if want_rows and out is None:
        return slabs.reshape(slabs.shape[0], -1)
    if slabs.shape[0] == 1:
        if out is not None:
            out.copy_(slabs[0].reshape(-1))
            return out.view(slabs.shape[1:])
        return slabs[0]
    out = torch.empty(slabs[0].numel(), device=x.device, dtype=torch.float32) if out is None else out
    sum_rows(slabs.reshape(slabs.shape[0], -1), out.view(-1))
    return out.view(slabs.shape[1:])


def norm_act_bwd(x, dy, y, gamma, stats, rows_per_stat, groups, eps, relu):
    """Backward of groupnorm_act / batchnorm_act: returns (dx, dgamma, dbeta)."""
    for t, n in ((x, "x"), (dy, "dy"), (gamma, "gamma"), (stats, "stats")):
        _chk(t, n)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    L = _lib.lib()
    ws = torch.empty(L.cslgan_norm_bwd_ws_floats(rows, rows_per_stat, Cc, groups), device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    dgamma = torch.empty(Cc, device=x.device, dtype=torch.float32)
    dbeta = torch.empty(Cc, device=x.device, dtype=torch.float32)
    check(L.cslgan_norm_act_bwd_f32(_p(x), _p(dy), _p(y), _p(gamma), _p(stats), rows, rows_per_stat, Cc, groups, float(eps),
                                    1 if relu else 0, _p(ws), _p(dx), _p(dgamma), _p(dbeta), _stream()), "norm_act_bwd")
    return dx, dgamma, dbeta


def bias_grad_grouped(gy, group=1, alpha=1.0, want_gb=True, sq=None, out=None):
    N, K = gy.shape[0], gy.shape[-1]
    PQ = gy.numel() // (N * K)
    gb = None
    if want_gb:
        gb = out if out is not None else torch.empty((N // group, K), device=gy.device, dtype=torch.float32)
    if gy.dtype == torch.bfloat16:
        if K % 8 == 0 and K <= 2048 and 256 % (K // 8) == 0:
            _chk(gy, "gy", allow_bf16=True)
            check(_lib.lib().cslgan_bias_grad_grouped_bf16(_p(gy), N, PQ, K, group, float(alpha), _p(gb), _p(sq), _stream()), "bias_grad_bf16")
            return gb
        gy = cast_f32(gy)
    _chk(gy, "gy")
    check(_lib.lib().cslgan_bias_grad_grouped_f32(_p(gy), N, PQ, K, group, float(alpha), _p(gb), _p(sq), _stream()), "bias_grad")
    return gb


def _segs(ins: Sequence[torch.Tensor], outs=None, noises=None, ragged=False):
    """ragged: the segments may have different row counts (column sums only: cslgan_segs_t.rows)."""
    s = SegsT()
    if len(ins) > _lib.MAX_SEGS:
        raise RuntimeError("at most %d segments per launch" % _lib.MAX_SEGS)
    s.n_seg = len(ins)
    n_rows = ins[0].shape[0] if len(ins) else 0
    if ragged:
        n_rows = max(t.shape[0] for t in ins)
    for i, t in enumerate(ins):
        _chk(t, "segment %d" % i, allow_bf16=True)
        if t.dtype != ins[0].dtype:
            raise RuntimeError("segments of one launch must share a dtype")
        if t.dim() != 2 or (t.shape[0] != n_rows and not ragged):
            raise RuntimeError("segments must be 2-D [n_rows, len] with equal n_rows")
        if ragged:
            s.rows[i] = t.shape[0]
        s.inp[i] = t.data_ptr()
        s.len[i] = t.shape[1]
        s.row_stride[i] = t.shape[1]
        if outs is not None:
            _chk(outs[i], "out %d" % i)
            if outs[i].numel() != t.shape[1]:
                raise RuntimeError("out %d has %d elements, expected %d" % (i, outs[i].numel(), t.shape[1]))
            s.out[i] = outs[i].data_ptr()
        if noises is not None and noises[i] is not None:
            _chk(noises[i], "noise %d" % i)
            if noises[i].numel() != t.shape[1]:
                raise RuntimeError("noise %d size mismatch" % i)
            s.noise[i] = noises[i].data_ptr()
    return s, n_rows


def _take_rows(t, idx):
    """t[idx] for a list of row indices WITHOUT building an index tensor on the host (that is a host-to-device copy, which a HIP-graph
    capture refuses): a slice when the indices are consecutive, a stack of row views otherwise."""
    if all(b - a == 1 for a, b in zip(idx, idx[1:])):
        return t[idx[0]:idx[-1] + 1].contiguous()
    return torch.stack([t[j] for j in idx])


def _by_dtype(mats):
    """Index lists of the fp32 and the bf16 segments (a launch handles one element type)."""
    parts = {}
    for i, m in enumerate(mats):
        parts.setdefault(m.dtype, []).append(i)
    return parts


def sample_sqnorm(mats: Sequence[torch.Tensor]) -> torch.Tensor:
    """mats: list of [n_rows, len_i] (fp32 or bf16) -> [n_seg, n_rows] squared row norms."""
    n_rows = mats[0].shape[0]
    out = torch.empty((len(mats), n_rows), device=mats[0].device, dtype=torch.float32)
    L = _lib.lib()
    for dt, idx in _by_dtype(mats).items():
        fn = L.cslgan_sample_sqnorm_bf16 if dt == torch.bfloat16 else L.cslgan_sample_sqnorm_f32
        for i in range(0, len(idx), _lib.MAX_SEGS):
            sub = idx[i:i + _lib.MAX_SEGS]
            chunk = [mats[j] for j in sub]
            s, _ = _segs(chunk)
            o = torch.empty((len(chunk), n_rows), device=chunk[0].device, dtype=torch.float32)
            nbytes = float(sum(n_rows * m.shape[1] * m.element_size() for m in chunk))
            _timed("sample_sqnorm", 0.0, nbytes, lambda: check(fn(C.byref(s), n_rows, _p(o), _stream()), "sample_sqnorm"))
            if len(sub) == len(mats):
                return o
            for k, j in enumerate(sub):
                out[j].copy_(o[k])
    return out


def clip_factors(sq, max_norm, flat, eps=1e-6, first_private_row=0, want_norms=False):
    """sq [n_seg,n_rows], max_norm device tensor [1] (flat) or [n_seg] -> factors ([n_rows] | [n_seg,n_rows])."""
    _chk(sq, "sq"); _chk(max_norm, "max_norm")
    n_seg, n_rows = sq.shape
    if max_norm.numel() != (1 if flat else n_seg):
        raise RuntimeError("clip_factors: max_norm has %d entries" % max_norm.numel())
    shape = (n_rows,) if flat else (n_seg, n_rows)
    f = torch.empty(shape, device=sq.device, dtype=torch.float32)
    nrm = torch.empty(shape, device=sq.device, dtype=torch.float32) if want_norms else None
    check(_lib.lib().cslgan_clip_factors_f32(_p(sq), n_seg, n_rows, _p(max_norm), 1 if flat else 0, float(eps),
                                             int(first_private_row), _p(f), _p(nrm), _stream()), "clip_factors")
    return (f, nrm) if want_norms else f


def adaptive_clip(sq_adapt, sq_rows, stat_max, scalar, per_layer, eps, first_private_row, mat_layers=(), jobs=()):
    """cslgan_adaptive_clip_f32: sq_adapt / sq_rows are lists (one entry per layer) of 1-D fp32 device tensors of equal length each;
    jobs = [(dst tensor [count], layer, first row, scale)].  Returns (r [L], C [L] | [1], sq [L, n_rows], f, f_mat | None)."""
    L = len(sq_rows)
    if L > _lib.MAX_CLIP_LAYERS or len(jobs) > _lib.MAX_CLIP_JOBS or len(mat_layers) > _lib.MAX_CLIP_LAYERS:
        raise RuntimeError("adaptive_clip: too many layers / jobs for one launch")
    n_adapt, n_rows = sq_adapt[0].numel(), sq_rows[0].numel()
    a = _lib.AdaptiveClipT()
    a.n_layers, a.n_mat, a.n_jobs = L, len(mat_layers), len(jobs)
    for l in range(L):
        _chk(sq_adapt[l], "sq_adapt"); _chk(sq_rows[l], "sq_rows")
        if sq_adapt[l].numel() != n_adapt or sq_rows[l].numel() != n_rows:
            raise RuntimeError("adaptive_clip: ragged norm vectors")
        a.sq_adapt[l], a.sq_rows[l] = sq_adapt[l].data_ptr(), sq_rows[l].data_ptr()
    for m, l in enumerate(mat_layers):
        a.mat_layer[m] = int(l)
    for j, (dst, layer, first, scale) in enumerate(jobs):
        _chk(dst, "job dst")
        a.job_dst[j], a.job_layer[j], a.job_first[j], a.job_count[j], a.job_scale[j] = dst.data_ptr(), int(layer), int(first), dst.numel(), float(scale)
    dev = sq_rows[0].device
    r = torch.empty(L, device=dev, dtype=torch.float32)
    c = torch.empty(L if per_layer else 1, device=dev, dtype=torch.float32)
    sq = torch.empty((L, n_rows), device=dev, dtype=torch.float32)
    f = torch.empty((L, n_rows) if per_layer else (n_rows,), device=dev, dtype=torch.float32)
    f_mat = torch.empty((len(mat_layers), n_rows), device=dev, dtype=torch.float32) if (mat_layers and per_layer) else None
    check(_lib.lib().cslgan_adaptive_clip_f32(C.byref(a), n_adapt, n_rows, 1 if stat_max else 0, float(scalar), 1 if per_layer else 0, float(eps),
                                              int(first_private_row), _p(r), _p(c), _p(sq), _p(f), _p(f_mat), _stream()), "adaptive_clip")
    return r, c, sq, f, f_mat


class deferred_sums:
    """Inside this context the column sums requested through sum_rows() are only queued; they run as ONE multi-segment launch
    (segments of different heights: cslgan_segs_t.rows) at exit.  For callers that take several dense weight gradients and read
    none of them before the block ends — the parameter gradients of the gradient penalty (train.py:427), which were nine
    latency-sized launches per step.  A queued destination holds garbage until the flush: never use this around code that consumes
    a sum inside the block (autograd accumulating two contributions into one parameter, a double backward)."""

    def __enter__(self):
        global _pending_sums
        self._prev, _pending_sums = _pending_sums, []
        return self

    def __exit__(self, *exc):
        global _pending_sums
        pend, _pending_sums = _pending_sums, self._prev
        if exc[0] is None:
            flush_sums(pend)


_pending_sums = None


def flush_sums(pend):
    for i in range(0, len(pend), _lib.MAX_SEGS):
        chunk = pend[i:i + _lib.MAX_SEGS]
        clip_accum_noise([m for m, _ in chunk], [o for _, o in chunk], ragged=True)


def sum_rows(slabs, out):
    """out[len] = sum over the rows of slabs [n, len] (fp32) — queued when a deferred_sums block is open."""
    if _pending_sums is not None and slabs.dtype == torch.float32:
        _pending_sums.append((slabs, out))
        return out
    clip_accum_noise([slabs], [out])
    return out


def clip_accum_noise(mats, outs, factors=None, noise_std=None, noises=None, seed=0, offset=0, scale=1.0, beta=0.0, call_counter=None,
                     ragged=False):
    """outs[i] = beta*outs[i] + scale*(sum_r f_r * mats[i][r] + noise_std[i]*z_i).  mats may mix fp32 and bf16
    segments (one launch per element type, fp32 accumulation either way).  call_counter: device int64 [1] whose value is
    added to `offset` inside the kernel (graph-captured steps: the counter lives in HBM, not in the launch arguments).
    ragged: the segments may have different row counts (plain column sums: no factors)."""
    if ragged and factors is not None:
        raise RuntimeError("clip_accum_noise: ragged segments take no clip factors")
    L = _lib.lib()
    if factors is not None:
        _chk(factors, "factors")
    if noise_std is not None:
        _chk(noise_std, "noise_std")
    for dt, idx in _by_dtype(mats).items():
        fn = L.cslgan_clip_accum_noise_bf16 if dt == torch.bfloat16 else L.cslgan_clip_accum_noise_f32
        for i in range(0, len(idx), _lib.MAX_SEGS):
            sub = idx[i:i + _lib.MAX_SEGS]
            whole = len(sub) == len(mats)
            s, n_rows = _segs([mats[j] for j in sub], [outs[j] for j in sub], None if noises is None else [noises[j] for j in sub], ragged=ragged)
            if call_counter is not None:
                if call_counter.dtype != torch.int64 or not call_counter.is_cuda:
                    raise RuntimeError("call_counter must be a device int64 tensor")
                s.call_counter = call_counter.data_ptr()
            per_seg, f = 0, None
            if factors is not None:
                if factors.dim() == 2:
                    per_seg = 1
                    f = factors if whole else _take_rows(factors, sub)
                else:
                    f = factors
            ns = None
            if noise_std is not None:
                ns = noise_std if whole else _take_rows(noise_std, sub)
            nbytes = float(sum(mats[j].shape[0] * mats[j].shape[1] * mats[j].element_size() + 4 * mats[j].shape[1] for j in sub))
            # the Philox stream is keyed by (offset, segment index within the launch): make it unique per launch
            _timed("clip_accum_noise", 0.0, nbytes, lambda: check(
                fn(C.byref(s), n_rows, _p(f), per_seg, _p(ns), int(seed), int(offset) * 64 + sub[0], float(scale), float(beta),
                   _stream()), "clip_accum_noise"))


def l2_clip_rows(t, Cval):
    _chk(t, "t")
    n = t.shape[0]
    flat = t.reshape(n, -1)
    out = torch.empty_like(flat)
    ws = torch.empty(n, device=t.device, dtype=torch.float32)
    check(_lib.lib().cslgan_l2_clip_rows_f32(_p(flat), _p(out), n, flat.shape[1], float(Cval), _p(ws), _stream()), "l2_clip_rows")
    return out.reshape(t.shape)


def mean_sample(mean_samples, labels, perms, noise_mean_std, noise_std, seed, offset, n=None, want_labels=False, out=None):
    """MeanSampler.sample's gather + per-image jitter + per-pixel noise (mean_sampler.py:75-84) in one pass.
    mean_samples [n_classes, num_samples, ...] fp32, labels / perms [n] int64 or None: missing permutations (and, with several
    classes, missing labels) are drawn inside the kernel; n is then required.  want_labels: also return the labels used."""
    _chk(mean_samples, "mean_samples")
    n_cls, num = mean_samples.shape[0], mean_samples.shape[1]
    if perms is not None:
        n = perms.numel()
    elif n is None:
        raise RuntimeError("mean_sample: n is required when the kernel draws the permutations")
    elif num > 1024:
        raise RuntimeError("mean_sample: in-kernel permutations need num_samples <= 1024")
    ln = mean_samples[0, 0].numel()
    for t, nm in ((labels, "labels"), (perms, "perms")):
        if t is not None and (t.dtype != torch.int64 or not t.is_cuda or t.numel() != n):
            raise RuntimeError("mean_sample: %s must be an int64 device tensor of %d entries" % (nm, n))
    if out is None:
        out = torch.empty((n,) + tuple(mean_samples.shape[2:]), device=mean_samples.device, dtype=torch.float32)
    else:           # caller-owned destination (a slice of the trainer's fused critic batch): n dense rows of ln floats
        _chk(out, "out")
        if out.numel() != n * ln:
            raise RuntimeError("mean_sample: out has %d elements, expected %d" % (out.numel(), n * ln))
    lab_out = torch.empty(n, device=mean_samples.device, dtype=torch.int64) if (want_labels and labels is None and n_cls > 1) else None
    check(_lib.lib().cslgan_mean_sample_f32(_p(mean_samples), n_cls, num, ln, None if labels is None else _p(labels.contiguous()),
                                            None if perms is None else _p(perms.contiguous()), n,
                                            float(noise_mean_std or 0.0), float(noise_std or 0.0), int(seed) & (2 ** 64 - 1),
                                            int(offset) & (2 ** 64 - 1), _p(out), _p(lab_out), _stream()), "mean_sample")
    if want_labels:
        return out, (labels if labels is not None else lab_out)
    return out


def row_l2norm(t2d):
    _chk(t2d, "t")
    n, L = t2d.shape
    out = torch.empty(n, device=t2d.device, dtype=torch.float32)
    check(_lib.lib().cslgan_row_l2norm_f32(_p(t2d), n, L, _p(out), _stream()), "row_l2norm")
    return out


def row_l2norm_bwd(t2d, norm, gnorm):
    _chk(t2d, "t"); _chk(norm, "norm"); _chk(gnorm, "gnorm")
    n, L = t2d.shape
    out = torch.empty_like(t2d)
    check(_lib.lib().cslgan_row_l2norm_bwd_f32(_p(t2d), _p(norm), _p(gnorm), n, L, _p(out), _stream()), "row_l2norm_bwd")
    return out


def act_bwd(g, y, slope):
    if g.dtype == torch.bfloat16 or y.dtype == torch.bfloat16:
        if g.dtype == y.dtype and g.numel() % 8 == 0:
            _chk(g, "g", allow_bf16=True); _chk(y, "y", allow_bf16=True)
            out = torch.empty_like(g)
            check(_lib.lib().cslgan_act_bwd_bf16(_p(g), _p(y), g.numel(), float(slope), _p(out), _stream()), "act_bwd_bf16")
            return out
        out = act_bwd(cast_f32(g), cast_f32(y), slope)
        return cast_bf16(out) if g.dtype == torch.bfloat16 else out
    _chk(g, "g"); _chk(y, "y")
    out = torch.empty_like(g)
    check(_lib.lib().cslgan_act_bwd_f32(_p(g), _p(y), g.numel(), float(slope), _p(out), _stream()), "act_bwd")
    return out


def _d2s_out(x, d2s, want_raw, dtype=torch.float32):
    N, H, W, Cc = x.shape
    if not d2s:
        if want_raw:
            raise RuntimeError("want_raw needs d2s=True")
        return torch.empty_like(x, dtype=dtype), None, 0
    if Cc % 4:
        raise RuntimeError("depth-to-space output needs C %% 4 == 0 (C=%d)" % Cc)
    shp = (N, 2 * H, 2 * W, Cc // 4)
    y = torch.empty(shp, device=x.device, dtype=dtype)
    return y, (torch.empty(shp, device=x.device, dtype=dtype) if want_raw else None), W


_norm_scratch = {}


_NORM_PARTIALS = os.environ.get("CSLGAN_NORM_PARTIALS", "1") == "1"
NORM_PARTIAL_BLOCKS = 64            # include/cslgan.h CSLGAN_NORM_PARTIAL_BLOCKS


def _scratch(dev, n_stats_floats):
    """Workspace for the per-workgroup partial statistics of the two-launch normalisation (cslgan_groupnorm_act_f32's `scratch`):
    n_stats_floats (= 2 * statistics) x NORM_PARTIAL_BLOCKS floats, persistent per (device, stream) — launches on one stream are
    ordered, and nothing in it outlives the apply launch that follows.  None with CSLGAN_NORM_PARTIALS=0 (A/B: the four-launch form)."""
    if not _NORM_PARTIALS:
        return None
    need = int(n_stats_floats) * NORM_PARTIAL_BLOCKS
    key = (dev, torch.cuda.current_stream().cuda_stream)
    t = _norm_scratch.get(key)
    if t is None or t.numel() < need:
        t = _norm_scratch[key] = torch.empty(max(need, 1 << 16), device=dev, dtype=torch.float32)
    return t


def groupnorm_act(x, gamma, beta, groups, eps=1e-5, relu=True, return_stats=False, d2s=False, want_raw=False, out_dtype=None, part=None):
    """GroupNorm(+ReLU).  d2s=True: the output is written depth-to-space shuffled ([N,2H,2W,C/4], see depth_to_space);
    want_raw=True additionally returns the raw x in that layout (one read of x feeds ResBlockUp's convUp and shortcut).
    out_dtype torch.bfloat16 (or a bfloat16 x): bf16-stored outputs (set_storage_dtype), fp32 statistics and arithmetic."""
    if x.dtype == torch.bfloat16 or out_dtype == torch.bfloat16:
        _chk(x, "x", allow_bf16=True); _chk(gamma, "gamma"); _chk(beta, "beta")
        N, H, W, Cc = x.shape
        y, xs, dW = _d2s_out(x, d2s, want_raw, torch.bfloat16)
        ws = torch.empty(2 * N * groups, device=x.device, dtype=torch.float32)
        check(_lib.lib().cslgan_groupnorm_act_bf16s(_p(x), 1 if x.dtype == torch.bfloat16 else 0, _p(gamma), _p(beta), N, H * W, Cc, groups,
                                                    float(eps), 1 if relu else 0, _p(ws), _p(y), dW, _p(xs), _stream()), "groupnorm_act_bf16s")
        out = (y, xs) if want_raw else y
        return (out, ws) if return_stats else out
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta")
    N, H, W, Cc = x.shape
    y, xs, dW = _d2s_out(x, d2s, want_raw)
    ws = torch.empty(2 * N * groups, device=x.device, dtype=torch.float32)
    if part is not None:        # statistics left by the producing conv's epilogue (gn_partials): the apply launch alone
        pt, n_part = part
        check(_lib.lib().cslgan_groupnorm_apply_parts_f32(_p(x), _p(gamma), _p(beta), N, H * W, Cc, groups, float(eps), 1 if relu else 0,
                                                          _p(pt), n_part, _p(ws), _p(y), dW, _p(xs), _stream()), "groupnorm_apply_parts")
        out = (y, xs) if want_raw else y
        return (out, ws) if return_stats else out
    sc = _scratch(x.device, 2 * N * groups)
    check(_lib.lib().cslgan_groupnorm_act_f32(_p(x), _p(gamma), _p(beta), N, H * W, Cc, groups, float(eps), 1 if relu else 0,
                                              _p(ws), _p(y), dW, _p(xs), _p(sc), _stream()), "groupnorm_act")
    out = (y, xs) if want_raw else y
    return (out, ws) if return_stats else out


def batchnorm_act(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, relu=True, return_stats=False,
                  d2s=False, want_raw=False):
    """Training-mode BatchNorm (+ReLU) over all leading dims of NHWC x; updates the running stats in place.
    d2s / want_raw as for groupnorm_act (x must then be 4-D)."""
    _chk(x, "x"); _chk(gamma, "gamma"); _chk(beta, "beta")
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if d2s:
        y, xs, dW = _d2s_out(x, d2s, want_raw)
        rpi = x.shape[1] * x.shape[2]
    else:
        y, xs, dW, rpi = torch.empty_like(x), None, 0, 0
    ws = torch.empty(2 * Cc, device=x.device, dtype=torch.float32)
    check(_lib.lib().cslgan_batchnorm_act_f32(_p(x), _p(gamma), _p(beta), rows, Cc, float(eps), 1 if relu else 0, float(momentum),
                                              _p(running_mean), _p(running_var), _p(ws), _p(y), rpi, dW, _p(xs),
                                              _p(_scratch(x.device, 2 * Cc)), _stream()), "batchnorm_act")
    out = (y, xs) if want_raw else y
    return (out, ws) if return_stats else out


def batchnorm_eval_act(x, gamma, beta, running_mean, running_var, eps=1e-5, relu=True, d2s=False, want_raw=False):
    """Eval-mode BatchNorm (+ReLU) from the running statistics (the generator in eval(), train.py:298-308)."""
    for t, n in ((x, "x"), (gamma, "gamma"), (beta, "beta"), (running_mean, "running_mean"), (running_var, "running_var")):
        _chk(t, n)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if d2s:
        y, xs, dW = _d2s_out(x, d2s, want_raw)
        rpi = x.shape[1] * x.shape[2]
    else:
        y, xs, dW, rpi = torch.empty_like(x), None, 0, 0
    ws = torch.empty(2 * Cc, device=x.device, dtype=torch.float32)
    check(_lib.lib().cslgan_batchnorm_eval_act_f32(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), rows, Cc, float(eps),
                                                   1 if relu else 0, _p(ws), _p(y), rpi, dW, _p(xs), _stream()), "batchnorm_eval_act")
    return (y, xs) if want_raw else y


def adam_step(p, g, m, v, lr, b1, b2, eps, weight_decay, step):
    """step: int (1-based, passed by value) or a device int32 tensor [1] read by the kernel (graph-captured steps)."""
    for t, n in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk(t, n)
    if isinstance(step, torch.Tensor):
        if step.dtype != torch.int32 or not step.is_cuda:
            raise RuntimeError("adam_step: a tensor step must be a device int32 tensor")
        check(_lib.lib().cslgan_adam_step_dev_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(b1), float(b2), float(eps),
                                                  float(weight_decay), _p(step), _stream()), "adam_step_dev")
        return
    check(_lib.lib().cslgan_adam_step_f32(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(b1), float(b2), float(eps),
                                          float(weight_decay), int(step), _stream()), "adam_step")


def adam_multi(ps, gs, ms, vs, lr, b1, b2, eps, weight_decay, step):
    """Adam over lists of flat fp32 tensors in ONE launch per <= 16 tensors.  step: int (1-based) or a device int32 [1] tensor."""
    dev_step = isinstance(step, torch.Tensor)
    if dev_step and (step.dtype != torch.int32 or not step.is_cuda):
        raise RuntimeError("adam_multi: a tensor step must be a device int32 tensor")
    for lst, nm in ((ps, "p"), (gs, "g"), (ms, "m"), (vs, "v")):
        for t in lst:
            _chk(t, nm)
    L = _lib.lib()
    VP = C.c_void_p
    for i in range(0, len(ps), _lib.MAX_SEGS):
        sub = range(i, min(i + _lib.MAX_SEGS, len(ps)))
        n = len(sub)
        arr = lambda lst: (VP * n)(*[lst[j].data_ptr() for j in sub])
        check(L.cslgan_adam_multi_f32(n, arr(ps), arr(gs), arr(ms), arr(vs), (C.c_int64 * n)(*[ps[j].numel() for j in sub]), float(lr),
                                      float(b1), float(b2), float(eps), float(weight_decay), 0 if dev_step else int(step),
                                      _p(step) if dev_step else None, _stream()), "adam_multi")


def _roles(sizes, scale):
    n = len(sizes)
    if not 1 <= n <= 8 or len(scale) != n:
        raise RuntimeError("segment means: 1..8 row blocks with one scale each, got %d / %d" % (n, len(scale)))
    return n, (C.c_int32 * n)(*[int(s) for s in sizes]), (C.c_float * n)(*[float(s) for s in scale])


def segment_means(x, sizes, scale):
    """vec[s] = scale[s] * sum(x[block s]), total = sum(vec): the critic's +-mean losses over the row blocks of one pass."""
    _chk(x, "x")
    if x.numel() != sum(sizes):
        raise RuntimeError("segment_means: %d values for blocks %s" % (x.numel(), list(sizes)))
    n, cs, sc = _roles(sizes, scale)
    out = torch.empty(n + 1, device=x.device, dtype=torch.float32)
    check(_lib.lib().cslgan_segment_means_f32(_p(x), n, cs, sc, _p(out), C.c_void_p(out.data_ptr() + 4 * n), _stream()), "segment_means")
    return out[:n], out[n]


def segment_means_bwd(g_total, g_vec, sizes, scale, shape, device):
    n, cs, sc = _roles(sizes, scale)
    gx = torch.empty(shape, device=device, dtype=torch.float32)
    check(_lib.lib().cslgan_segment_means_bwd_f32(_p(g_total), _p(g_vec), n, cs, sc, _p(gx), _stream()), "segment_means_bwd")
    return gx


def dstep_stats(d_real, d_fake, real_loss, fake_loss, penalty, acc7):
    """train.py:488-496 in one launch; acc7 (persistent, device) += the seven statistics (see include/cslgan.h)."""
    for t, nm in ((d_real, "d_real"), (d_fake, "d_fake"), (real_loss, "real_loss"), (fake_loss, "fake_loss"), (acc7, "acc7")):
        _chk(t, nm)
    if acc7.numel() != 7:
        raise RuntimeError("dstep_stats: acc7 must hold 7 floats")
    if penalty is not None:
        _chk(penalty, "penalty")
    check(_lib.lib().cslgan_dstep_stats_f32(_p(d_real), d_real.numel(), _p(d_fake), d_fake.numel(), _p(real_loss), _p(fake_loss),
                                            _p(penalty), _p(acc7), _stream()), "dstep_stats")


def grad_log_stats(sq, col0, B, max_norm, per_layer, eps, acc):
    """update_grad_logging (train.py:310-329) from the clip's squared norms sq [L, ld]; acc: [5, L or 1] persistent sums
    (means, stds, maxes, clip norms, clipped fraction)."""
    _chk(sq, "sq"); _chk(max_norm, "max_norm"); _chk(acc, "acc")
    Lr, ld = sq.shape
    rows = Lr if per_layer else 1
    if tuple(acc.shape) != (5, rows) or max_norm.numel() < (rows if per_layer else 1):
        raise RuntimeError("grad_log_stats: acc must be [5, %d] and max_norm hold %d value(s)" % (rows, rows))
    a = [C.c_void_p(acc.data_ptr() + 4 * rows * i) for i in range(5)]
    check(_lib.lib().cslgan_grad_log_stats_f32(_p(sq), Lr, ld, int(col0), int(B), _p(max_norm), 1 if per_layer else 0, float(eps),
                                               a[0], a[1], a[2], a[3], a[4], _stream()), "grad_log_stats")


def _mem_order(t):
    return tuple(st for sz, st in zip(t.shape, t.stride()) if sz > 1)


def lerp_rows(real, fake, alpha):
    """alpha[b] * real[b] + (1 - alpha[b]) * fake[b] (gradient_penalty.py:36).  real / fake: same shape, same dense memory order
    (contiguous or channels-last), batch outermost; the result has real's strides."""
    _chk(alpha, "alpha")
    for t, nm in ((real, "real"), (fake, "fake")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError("lerp_rows: %s must be a float32 device tensor" % nm)
        if not (t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))):
            raise RuntimeError("lerp_rows: %s is not dense" % nm)
    if real.shape != fake.shape or _mem_order(real) != _mem_order(fake):
        raise RuntimeError("lerp_rows: real and fake must share shape and memory order")
    B = real.shape[0]
    if alpha.numel() != B or (B > 1 and real.stride(0) * B != real.numel()):
        raise RuntimeError("lerp_rows: alpha must hold one weight per row of a batch-outermost tensor")
    out = torch.empty_strided(real.shape, real.stride(), device=real.device, dtype=torch.float32)
    check(_lib.lib().cslgan_lerp_rows_f32(_p(real), _p(fake), _p(alpha), B, real.numel() // B, _p(out), _stream()), "lerp_rows")
    return out


_tickets = {}


def _ticket(dev):
    t = _tickets.get(dev)
    if t is None:
        t = _tickets[dev] = torch.zeros(1, device=dev, dtype=torch.int32)
    return t


def lipschitz_term(t2d, one_sided, coef):
    """(norm [B], per [B], total []) with per = coef * (||t_b|| - 1)^2 (clamped at 0 from below when one_sided)."""
    _chk(t2d, "t")
    n, Ln = t2d.shape
    buf = torch.empty(2 * n + 1, device=t2d.device, dtype=torch.float32)
    check(_lib.lib().cslgan_lipschitz_term_f32(_p(t2d), n, Ln, 1 if one_sided else 0, float(coef), _p(buf),
                                               C.c_void_p(buf.data_ptr() + 4 * n), C.c_void_p(buf.data_ptr() + 8 * n),
                                               _p(_ticket(t2d.device)), _stream()), "lipschitz_term")
    return buf[:n], buf[n:2 * n], buf[2 * n]


def lipschitz_term_bwd(t2d, norm, g_total, g_per, one_sided, coef):
    _chk(t2d, "t"); _chk(norm, "norm")
    n, Ln = t2d.shape
    out = torch.empty_like(t2d)
    check(_lib.lib().cslgan_lipschitz_term_bwd_f32(_p(t2d), _p(norm), _p(g_total), _p(g_per), n, Ln, 1 if one_sided else 0, float(coef),
                                                   _p(out), _stream()), "lipschitz_term_bwd")
    return out
