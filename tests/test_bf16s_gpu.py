"""bf16 STORAGE path (-m gpu; csrc/igemm_bf16s.hip, `--storage_dtype bf16`, BASELINE.json configs[4]).

Kernel level: every bf16-stored entry against plain PyTorch fp32 math on the SAME bf16-rounded operands (the kernels multiply
bf16 values exactly and accumulate in fp32, so only the summation order differs: 1e-4 of scale for fp32 outputs; a bf16 output
may in addition sit half a bfloat16 spacing — up to 2^-8 relative — from the reference).
Step level: Trainer.train_D with bf16-stored critic activations against the FP32 oracle, at tolerances derived from an error
model stated in the test."""
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF_ULP = 2.0 ** -7          # largest spacing of bfloat16 (8 significant bits) relative to the value: a correctly rounded result is within half of it


def _ops():
    from csl_gan_amd import ops
    return ops


def _seed(case):
    return zlib.crc32(repr(case).encode()) % 100000


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _nhwc16(t):     # NCHW cpu fp32 -> NHWC device bf16
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def _krsc(w):
    return w.permute(0, 2, 3, 1).contiguous().cuda()


def _close(got, exp, rtol=1e-4, what="", ulps=0.0):
    """|got - exp| <= rtol * scale (+ ulps bf16 rounding steps of the entry itself)."""
    got = got.detach().float().cpu().double()
    exp = exp.detach().float().cpu().double()
    assert got.shape == exp.shape, (what, got.shape, exp.shape)
    scale = exp.abs().max().item() + 1e-12
    excess = ((got - exp).abs() - ulps * BF_ULP * exp.abs()).max().item()
    assert excess <= rtol * scale, "%s: error beyond the rounding allowance %.3e vs scale %.3e (rel %.3e)" % (what, excess, scale, excess / scale)


def test_casts_round_trip():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    for n in (8, 1000, 4099, 1 << 20):
        x = (torch.randn(n, generator=g) * 3).cuda()
        h = ops.cast_bf16(x)
        assert h.dtype == torch.bfloat16 and torch.equal(h, x.to(torch.bfloat16)), "round-to-nearest-even cast, n=%d" % n
        assert torch.equal(ops.cast_f32(h), h.float())
    x4 = torch.randn(3, 5, 6, 8, generator=g).cuda().permute(0, 3, 1, 2)        # channels-last strides are kept
    h4 = ops.cast_bf16(x4)
    assert h4.stride() == x4.stride() and torch.equal(h4, x4.to(torch.bfloat16))


FWD_CASES = [
    # N, H, W, C, K, R, stride, pad, bias, act, residual (None / "f32" / "bf16"), bf16 output
    (2, 8, 8, 8, 16, 5, 2, 2, True, 1, None, True),
    (4, 32, 32, 64, 128, 5, 2, 2, True, 1, None, True),           # the critic's conv2 (64x64 images: 32x32 here)
    (3, 16, 16, 128, 256, 5, 2, 2, True, 1, None, True),
    (16, 8, 8, 256, 512, 5, 2, 2, True, 1, None, True),           # conv4: 64-row... 4x4 output grids (not patchable)
    (2, 16, 16, 64, 64, 5, 1, 2, True, 2, "bf16", True),          # stride 1, residual, ReLU
    (2, 16, 16, 32, 96, 3, 1, 1, False, 0, "f32", False),         # ragged N tile, fp32 residual and output
    (2, 7, 9, 16, 24, 3, 1, 1, True, 0, None, True),              # ragged grid
    (130, 1, 1, 8192, 1, 1, 1, 0, False, 0, None, False),         # the critic's head on bf16 features
    (6, 1, 1, 512, 10, 1, 1, 0, True, 0, None, False),            # auxiliary head
    (128, 8, 8, 512, 512, 5, 1, 2, False, 0, None, True),         # > 256 tiles of 128x128
    (4, 32, 32, 64, 3, 3, 1, 1, True, 3, None, False),            # the generator's output conv: bf16 in, tanh, fp32 image out
    (2, 64, 64, 32, 64, 5, 1, 2, False, 0, None, True),           # LDS-halo form, 64 filters
    (2, 16, 16, 128, 256, 5, 1, 2, True, 2, "bf16", True),        # LDS-halo form, 128-wide filter tiles, bf16 residual
    (2, 32, 32, 16, 64, 5, 1, 2, False, 0, "f32", False),         # LDS-halo form on one 16-channel chunk, fp32 residual / output
    (128, 32, 32, 64, 128, 1, 1, 0, True, 0, None, True),         # 1x1 stream kernel (the generator's shortcut convs): 64 -> 128
    (64, 64, 64, 16, 64, 1, 1, 0, False, 0, None, True),          # ... 16 -> 64: the whole reduction is one MFMA k-step
    (70, 31, 31, 32, 96, 1, 1, 0, True, 2, None, False),          # ... ragged pixel count, three filter tiles, ReLU, fp32 output
    (2, 8, 8, 16, 6, 3, 1, 1, True, 1, "bf16", True),             # 6 filters: the epilogue's element-wise path (channel count not a multiple of 4)
    (2, 8, 8, 16, 20, 3, 1, 1, True, 2, "f32", True),             # 20 filters: 4-channel vector groups, the last 32-wide tile partly filled
    (3, 8, 8, 16, 36, 3, 1, 1, True, 1, "bf16", False),           # ... fp32 output with a bf16 residual
]


@pytest.mark.parametrize("case", FWD_CASES)
def test_conv2d_fwd_bf16_stored(case):
    ops = _ops()
    N, H, W, C, K, R, s, p, has_b, act, res, out16 = case
    g = torch.Generator().manual_seed(_seed(case))
    x = _bf(torch.randn(N, C, H, W, generator=g))
    w = torch.randn(K, C, R, R, generator=g) / (C * R * R) ** 0.5
    b = torch.randn(K, generator=g) if has_b else None
    y0 = F.conv2d(x, w if (K <= 4 and H * W > 1) else _bf(w), b, stride=s, padding=p)      # 1..4 output channels of an image: fp32 filter, vector ALU
    rs = None
    if res is not None:
        rs = torch.randn(y0.shape, generator=g)
        rs = _bf(rs) if res == "bf16" else rs
        y0 = y0 + rs
    ref = F.leaky_relu(y0, 0.2) if act == 1 else (F.relu(y0) if act == 2 else (torch.tanh(y0) if act == 3 else y0))
    rdev = None if rs is None else (_nhwc16(rs) if res == "bf16" else _nhwc(rs))
    y = ops.conv2d_fwd(_nhwc16(x), _krsc(w), None if b is None else b.cuda(), stride=s, pad=p, residual=rdev, act=act,
                       out_dtype=torch.bfloat16 if out16 else torch.float32)
    assert y.dtype == (torch.bfloat16 if out16 else torch.float32)
    _close(y.permute(0, 3, 1, 2), ref, rtol=1e-4, ulps=0.51 if out16 else 0.0, what="bf16-stored fwd %s" % (case,))


DGRAD_CASES = [
    # N, H, W, C, K, R, stride, pad, mask, bf16 output
    (2, 8, 8, 8, 16, 5, 2, 2, False, True),
    (4, 32, 32, 64, 128, 5, 2, 2, True, True),
    (3, 16, 16, 128, 256, 5, 2, 2, True, True),
    (16, 8, 8, 256, 512, 5, 2, 2, True, True),
    (3, 32, 32, 3, 64, 5, 2, 2, False, False),             # the first layer's data gradient: 3 output channels, fp32 image gradient
    (2, 16, 16, 64, 64, 5, 1, 2, False, True),             # stride 1 (the generator's convs under train_G)
    (2, 9, 7, 16, 24, 3, 1, 1, True, True),
    (5, 15, 15, 8, 16, 5, 2, 2, True, True),               # odd image: ragged parity classes
    (2, 64, 64, 128, 256, 5, 2, 2, True, True),            # parity classes on 32x32 grids, 128 output channels: LDS-halo form, wide tiles
    (3, 32, 32, 64, 64, 3, 2, 1, False, False),            # 3x3 stride 2: a class with a single tap keeps the launch on the gather form
    (2, 12, 12, 20, 16, 3, 1, 1, True, False),             # 20 input channels, fp32 mask and fp32 gradient: vector groups, partly filled tile
    (2, 10, 10, 6, 16, 5, 2, 2, True, False),              # 6 input channels: element-wise epilogue with an fp32 mask
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv2d_dgrad_bf16_stored(case):
    ops = _ops()
    N, H, W, C, K, R, s, p, use_mask, out16 = case
    g = torch.Generator().manual_seed(_seed(case))
    w = torch.randn(K, C, R, R, generator=g) / (K * R * R) ** 0.5
    P, Q = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    gy = _bf(torch.randn(N, K, P, Q, generator=g))
    wr = w if C <= 4 else _bf(w)        # 1..4 image channels: the vector-ALU kernel keeps the fp32 filter and fp32 arithmetic
    ref = F.conv_transpose2d(gy, wr, None, stride=s, padding=p, output_padding=(H + 2 * p - R - (P - 1) * s, W + 2 * p - R - (Q - 1) * s))
    mask = None
    if use_mask:
        mask = torch.randn(N, C, H, W, generator=g)
        ref = ref * torch.where(mask > 0, 1.0, 0.2)
    mdev = None if mask is None else (_nhwc16(mask) if out16 else _nhwc(mask))
    gx = ops.conv2d_dgrad(_nhwc16(gy), _krsc(w), (H, W), stride=s, pad=p, mask=mdev, out_dtype=torch.bfloat16 if out16 else torch.float32)
    assert gx.dtype == (torch.bfloat16 if out16 else torch.float32)
    _close(gx.permute(0, 3, 1, 2), ref, rtol=1e-4, ulps=0.51 if out16 else 0.0, what="bf16-stored dgrad %s" % (case,))


WGRAD_CASES = [(2, 8, 8, 8, 16, 5, 2, 2), (4, 32, 32, 64, 128, 5, 2, 2), (8, 8, 8, 256, 512, 5, 2, 2), (2, 16, 16, 128, 256, 5, 2, 2),
               (2, 7, 9, 16, 24, 3, 1, 1), (4, 64, 64, 8, 64, 5, 2, 2), (6, 1, 1, 800, 128, 1, 1, 0), (2, 16, 16, 64, 64, 5, 1, 2)]


@pytest.mark.parametrize("case", WGRAD_CASES)
@pytest.mark.parametrize("group", [1, 2, 0])
def test_conv2d_wgrad_grouped_bf16_stored(case, group):
    """Grouped weight gradient (group=1: per-sample gradients) from bf16 gy and bf16 x: gradients, the fused per-group squared
    norms, bf16 gradient storage, the split-reduction form of few-tile launches."""
    ops = _ops()
    N, H, W, C, K, R, s, p = case
    grp = N if group == 0 else group
    if N % grp:
        pytest.skip("N not divisible by group")
    g = torch.Generator().manual_seed(_seed(case) + grp)
    x = _bf(torch.randn(N, C, H, W, generator=g))
    P, Q = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    gy = _bf(torch.randn(N, K, P, Q, generator=g))
    alpha = 1.75
    wz = torch.zeros(K, C, R, R, requires_grad=True)
    refs = []
    for b0 in range(0, N, grp):
        y = F.conv2d(x[b0:b0 + grp], wz, None, stride=s, padding=p)
        refs.append(torch.autograd.grad(y, wz, gy[b0:b0 + grp])[0] * alpha)
    ref = torch.stack(refs)
    sq = torch.zeros(N // grp, device="cuda")
    gw = ops.conv2d_wgrad_grouped(_nhwc16(gy), _nhwc16(x), R, R, stride=s, pad=p, group=grp, alpha=alpha, sq=sq)
    assert gw.dtype == torch.float32
    sq2 = torch.zeros(N // grp, device="cuda")
    none = ops.conv2d_wgrad_grouped(_nhwc16(gy), _nhwc16(x), R, R, stride=s, pad=p, group=grp, alpha=alpha, want_gw=False, sq=sq2)
    assert none is None
    _close(gw.permute(0, 1, 4, 2, 3), ref, rtol=1e-4, what="bf16-stored wgrad %s g%d" % (case, grp))
    exp_sq = ref.reshape(N // grp, -1).double().pow(2).sum(1).float()
    _close(sq, exp_sq, rtol=2e-4, what="bf16-stored wgrad sq")
    _close(sq2, exp_sq, rtol=2e-4, what="bf16-stored wgrad sq (norms only)")
    out16 = torch.empty((N // grp, K, R, R, C), device="cuda", dtype=torch.bfloat16)
    sq3 = torch.zeros(N // grp, device="cuda")
    ops.conv2d_wgrad_grouped(_nhwc16(gy), _nhwc16(x), R, R, stride=s, pad=p, group=grp, alpha=alpha, sq=sq3, out=out16)
    _close(out16.permute(0, 1, 4, 2, 3), ref, rtol=1e-4, ulps=0.51, what="bf16-stored wgrad, bf16 gradient storage")
    _close(sq3, out16.float().reshape(N // grp, -1).double().pow(2).sum(1).float(), rtol=2e-4, what="sq of the ROUNDED gradients")
    dense = ops.conv2d_wgrad_dense(_nhwc16(gy), _nhwc16(x), R, R, stride=s, pad=p, alpha=alpha)
    _close(dense.permute(0, 3, 1, 2), ref.sum(0), rtol=2e-4, what="bf16-stored dense wgrad %s" % (case,))


@pytest.mark.parametrize("case", [(8, 16, 16, 256, 512, 5, 2, 2, 4), (6, 32, 32, 64, 128, 5, 2, 2, 2), (4, 8, 8, 64, 64, 3, 1, 1, 4), (6, 16, 16, 128, 256, 5, 2, 2, 6)])
def test_conv2d_wgrad_clip_weighted_bf16_stored(case):
    """Clip-weighted sums of ghost-clipped layers from bf16 gy / x: sum_b f_b g_b with the fp32 weight applied to each sample's
    ACCUMULATED product — equal to weighting the materialised per-sample gradients (fp32 math on the stored values) at 1e-4, i.e. no
    bfloat16 rounding of f_b * gy anywhere."""
    ops = _ops()
    N, H, W, C, K, R, s, p, grp = case
    g = torch.Generator().manual_seed(_seed(case))
    x = _bf(torch.randn(N, C, H, W, generator=g))
    P, Q = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    assert (P * Q) % 64 == 0
    gy = _bf(torch.randn(N, K, P, Q, generator=g))
    f = torch.rand(N, generator=g) * 0.9 + 0.1
    wz = torch.zeros(K, C, R, R, requires_grad=True)
    per = torch.stack([torch.autograd.grad(F.conv2d(x[i:i + 1], wz, None, stride=s, padding=p), wz, gy[i:i + 1])[0] for i in range(N)])
    ref = (per * f.view(N, 1, 1, 1, 1)).reshape(N // grp, grp, K, C, R, R).sum(1) * 1.5
    got = ops.conv2d_wgrad_grouped(_nhwc16(gy), _nhwc16(x), R, R, stride=s, pad=p, group=grp, alpha=1.5, row_scale=f.cuda())
    assert got.dtype == torch.float32
    _close(got.permute(0, 1, 4, 2, 3), ref, rtol=1e-4, what="clip-weighted bf16-stored wgrad %s" % (case,))
    dense = ops.conv2d_wgrad_dense(_nhwc16(gy), _nhwc16(x), R, R, stride=s, pad=p, alpha=1.5, row_scale=f.cuda())
    _close(dense.permute(0, 3, 1, 2), ref.sum(0), rtol=2e-4, what="clip-weighted dense sum %s" % (case,))
    # the weight really is exact: weighting gy in bfloat16 first would differ at the 2^-9 level
    rounded = torch.stack([torch.autograd.grad(F.conv2d(x[i:i + 1], wz, None, stride=s, padding=p), wz, _bf(gy[i:i + 1] * f[i]))[0] for i in range(N)])
    assert ((rounded.sum(0) * 1.5 - ref.sum(0)).abs().max() > 3e-4 * ref.sum(0).abs().max()).item()


def test_mixed_element_types_fall_back_to_the_fp32_kernels():
    """fp32 gy with bf16 x (the critic's head: fp32 loss cotangent, bf16 features) and the RGB first layer (fp32 image, bf16
    output gradient) run on the fp32 kernels between casts — same numbers as fp32 math on the values as stored."""
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(6, 512, 1, 1, generator=g))
    gy = torch.randn(6, 1, 1, 1, generator=g)
    gw = ops.conv2d_wgrad_grouped(_nhwc(gy), _nhwc16(x), 1, 1, group=1, alpha=2.0)
    _close(gw.reshape(6, 512), 2.0 * gy.reshape(6, 1) * x.reshape(6, 512), what="linear head per-sample gradient")
    x8 = _bf(torch.randn(16, 4096, 1, 1, generator=g))
    gy8 = torch.randn(16, 1, 1, 1, generator=g)
    sq = torch.zeros(8, device="cuda")
    gw2 = ops.conv2d_wgrad_grouped(_nhwc(gy8), _nhwc16(x8), 1, 1, group=2, alpha=0.5, sq=sq)
    ref2 = 0.5 * (gy8.reshape(16, 1) * x8.reshape(16, 4096)).reshape(8, 2, 4096).sum(1)
    _close(gw2.reshape(8, 4096), ref2, what="linear head grouped gradient")
    _close(sq, ref2.pow(2).sum(1), rtol=2e-4, what="linear head grouped sq")
    dense = ops.conv2d_wgrad_dense(_nhwc(gy8), _nhwc16(x8), 1, 1, alpha=0.5)
    _close(dense.reshape(-1), ref2.sum(0), rtol=2e-4, what="linear head dense gradient")
    f8 = torch.rand(16, generator=g) + 0.1
    densef = ops.conv2d_wgrad_dense(_nhwc(gy8), _nhwc16(x8), 1, 1, alpha=0.5, row_scale=f8.cuda())
    _close(densef.reshape(-1), 0.5 * ((gy8.reshape(16, 1) * f8.reshape(16, 1)) * x8.reshape(16, 4096)).sum(0), rtol=2e-4, what="linear head clip-weighted sum")
    w8 = torch.randn(1, 4096, 1, 1, generator=g)
    m8 = torch.randn(16, 4096, 1, 1, generator=g)
    gx8 = ops.conv2d_dgrad(_nhwc(gy8), _krsc(w8), (1, 1), mask=_nhwc16(m8), out_dtype=torch.bfloat16)
    assert gx8.dtype == torch.bfloat16
    _close(gx8.reshape(16, 4096), gy8.reshape(16, 1) * _bf(w8).reshape(1, 4096) * torch.where(m8.reshape(16, 4096) > 0, 1.0, 0.2), ulps=0.51,
           what="linear head data gradient to bf16 features")
    img = torch.randn(3, 3, 32, 32, generator=g)
    gy1 = _bf(torch.randn(3, 64, 16, 16, generator=g))
    wz = torch.zeros(64, 3, 5, 5, requires_grad=True)
    ref = torch.stack([torch.autograd.grad(F.conv2d(img[i:i + 1], wz, None, stride=2, padding=2), wz, gy1[i:i + 1])[0] for i in range(3)])
    gw1 = ops.conv2d_wgrad_grouped(_nhwc16(gy1), _nhwc(img), 5, 5, stride=2, pad=2, group=1)
    _close(gw1.permute(0, 1, 4, 2, 3), ref, what="first-layer per-sample gradient from a bf16 output gradient")
    dense1 = ops.conv2d_wgrad_dense(_nhwc16(gy1), _nhwc(img), 5, 5, stride=2, pad=2)
    _close(dense1.permute(0, 3, 1, 2), ref.sum(0), rtol=2e-4, what="first-layer dense gradient from a bf16 output gradient")
    sq1 = torch.zeros(3, device="cuda")
    assert ops.conv2d_wgrad_grouped(_nhwc16(gy1), _nhwc(img), 5, 5, stride=2, pad=2, group=1, want_gw=False, sq=sq1) is None
    _close(sq1, ref.reshape(3, -1).pow(2).sum(1), rtol=2e-4, what="first-layer per-sample norms from a bf16 output gradient")
    w = torch.randn(64, 3, 5, 5, generator=g) * 0.1
    y = ops.conv2d_fwd(_nhwc(img), _krsc(w), None, stride=2, pad=2, act=1, out_dtype=torch.bfloat16)
    assert y.dtype == torch.bfloat16
    _close(y.permute(0, 3, 1, 2), F.leaky_relu(F.conv2d(img, w, None, stride=2, padding=2), 0.2), ulps=0.51, what="first layer, bf16 output")


def test_act_bwd_and_bias_grad_bf16():
    ops = _ops()
    g = torch.Generator().manual_seed(12)
    gy = _bf(torch.randn(6, 16, 16, 128, generator=g))
    y = _bf(torch.randn(6, 16, 16, 128, generator=g))
    out = ops.act_bwd(gy.to(torch.bfloat16).cuda(), y.to(torch.bfloat16).cuda(), 0.2)
    assert out.dtype == torch.bfloat16
    _close(out, gy * torch.where(y > 0, 1.0, 0.2), ulps=0.51, what="act_bwd bf16")
    for grp in (1, 3, 6):
        sq = torch.zeros(6 // grp, device="cuda")
        gb = ops.bias_grad_grouped(gy.to(torch.bfloat16).cuda(), group=grp, alpha=1.5, sq=sq)
        ref = 1.5 * gy.reshape(6 // grp, -1, 128).sum(1)
        _close(gb, ref, what="bias grad bf16 g%d" % grp)
        _close(sq, ref.pow(2).sum(1), rtol=2e-4, what="bias grad sq")


@pytest.mark.parametrize("case", [(4, 16, 16, 64, True, True), (3, 8, 8, 128, False, False), (2, 32, 32, 32, True, False), (2, 4, 4, 512, False, True)])
def test_groupnorm_act_bf16_stored(case):
    """GroupNorm(32) + ReLU writing bf16-stored activations from an fp32 or a bf16 input, plain and depth-to-space shuffled
    (with the raw input beside it), against torch in fp32 on the input as stored."""
    ops = _ops()
    N, H, W, C, in16, d2s = case
    g = torch.Generator().manual_seed(_seed(case))
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    if in16:
        x = _bf(x)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    ref = F.relu(F.group_norm(x, 32, gamma, beta, eps=1e-5))
    xd = _nhwc16(x) if in16 else _nhwc(x)
    if d2s:
        y, xs = ops.groupnorm_act(xd, gamma.cuda(), beta.cuda(), 32, d2s=True, want_raw=True, out_dtype=torch.bfloat16)
        assert y.dtype == torch.bfloat16 and xs.dtype == torch.bfloat16
        shuf = lambda t: F.pixel_shuffle(t, 2)          # [N, 4c'+2i+j, h, w] -> [N, c', 2h+i, 2w+j]: the plain depth-to-space map
        _close(y.permute(0, 3, 1, 2), shuf(ref), ulps=0.51, what="groupnorm bf16 d2s %s" % (case,))
        _close(xs.permute(0, 3, 1, 2), shuf(x), ulps=0.51, rtol=1e-6, what="raw input, shuffled %s" % (case,))
    else:
        y = ops.groupnorm_act(xd, gamma.cuda(), beta.cuda(), 32, out_dtype=torch.bfloat16)
        assert y.dtype == torch.bfloat16
        _close(y.permute(0, 3, 1, 2), ref, ulps=0.51, what="groupnorm bf16 %s" % (case,))


def test_generator_forward_bf16_stored_matches_fp32():
    """The frozen generator of a D-step with --storage_dtype bf16 (GroupNorm writes bf16, the convs answer in kind, the output conv
    returns the fp32 image) against the same generator on the fp32 kernels: 13 convs deep, 4e-2 of scale per pixel."""
    ops = _ops()
    from csl_gan_amd import init_util, options
    import tempfile
    opt = options.parse(["CelebA", "-dpm", "gc", "-nms", "4", "-bs", "4", "-gd", "cuda:0", "-dd", "cuda:0", "-o", tempfile.mkdtemp(), "--manual_seed", "3"])
    G, D = init_util.init_models(opt)
    z = torch.randn(4, 128, generator=torch.Generator().manual_seed(4)).cuda()
    with torch.no_grad():
        ref = G(z)
        with ops.compute_dtype("bf16"), ops.storage_dtype("bf16"):
            got = G(z)
    assert got.dtype == torch.float32 and got.shape == ref.shape
    _close(got, ref, rtol=4e-2, what="generator forward, bf16 storage vs fp32")


def test_conv_autograd_closed_under_bf16_storage():
    """Conv / Dgrad / Wgrad with bf16-stored activations: first- and second-order gradients (the WGAN-GP double backward) against
    torch autograd in fp32 on a two-layer LeakyReLU-free critic, at a bf16 tolerance (three bf16-stored tensors deep)."""
    ops = _ops()
    from csl_gan_amd import functional as HF
    g = torch.Generator().manual_seed(21)
    x = torch.randn(4, 3, 32, 32, generator=g)
    w1 = (torch.randn(64, 3, 5, 5, generator=g) * 0.1)
    w2 = (torch.randn(128, 64, 5, 5, generator=g) * 0.03)

    def run_ref():
        xr, a, b = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
        y = F.conv2d(F.conv2d(xr, a, None, stride=2, padding=2), b, None, stride=2, padding=2)
        (gx,) = torch.autograd.grad(y.sum(), xr, create_graph=True)
        pen = (gx.reshape(4, -1).norm(2, dim=1) - 1).pow(2).mean()
        ga, gb = torch.autograd.grad(pen, (a, b))
        return y, gx, ga, gb

    def run_hip():
        xd = _nhwc(x).requires_grad_(True)
        a, b = _krsc(w1).requires_grad_(True), _krsc(w2).requires_grad_(True)
        h = HF.Conv.apply(xd, a, None, 2, 2, ops.ACT_NONE, None, None, 1.0, None, None, False, False, torch.bfloat16)
        y = HF.Conv.apply(h, b, None, 2, 2, ops.ACT_NONE, None, None, 1.0, None, None, False, False, torch.bfloat16)
        assert h.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
        (gx,) = torch.autograd.grad(y.float().sum(), xd, create_graph=True)
        assert gx.dtype == torch.float32
        pen = (gx.reshape(4, -1).norm(2, dim=1) - 1).pow(2).mean()
        ga, gb = torch.autograd.grad(pen, (a, b))
        return y, gx, ga, gb

    yr, gxr, gar, gbr = run_ref()
    yh, gxh, gah, gbh = run_hip()
    _close(yh.permute(0, 3, 1, 2), yr, rtol=2e-2, what="two bf16-stored convs")
    _close(gxh.permute(0, 3, 1, 2), gxr, rtol=2e-2, what="input gradient through bf16-stored gradients")
    _close(gah.permute(0, 3, 1, 2), gar, rtol=3e-2, what="second-order gradient of w1")
    _close(gbh.permute(0, 3, 1, 2), gbr, rtol=3e-2, what="second-order gradient of w2")


STEP_CASES = [
    ("CelebA", ["--im_size", "128", "-gcm", "adaptive-pl"], 4, 128),          # BASELINE configs[4] geometry (extension): 128x128
    # ghost clipping in the bf16 storage mode (the benchmarked configuration): Gram norms of the stored values, clip weights applied in
    # fp32 to each sample's accumulated product (conv4 + head at 128x128; conv3 on the scaled kernel, conv4 on the fp32 fallback at 64x64)
    ("CelebA", ["--im_size", "128", "-gcm", "adaptive-pl", "--materialize", "ghost"], 4, 128),
    ("CelebA", ["-gcm", "adaptive-pl", "--materialize", "ghost"], 8, 128),
    ("CelebA", ["-c", "2.0", "--materialize", "ghost"], 8, 128),              # flat clipping
    ("CelebA", ["-gcm", "adaptive-pl"], 8, 128),
    ("CelebA", ["-gcm", "adaptive-pl", "--grad_sample_dtype", "bf16", "--materialize", "private"], 8, 128),
    # every pass written out per sample (the reference route of tests/test_fullsize_gpu.py's route comparison)
    ("CelebA", ["--im_size", "128", "-gcm", "adaptive-pl", "--materialize", "all", "--fuse_passes", "False"], 4, 128),
]


def _rel_l2(a, b):
    a = torch.as_tensor(a).detach().float().cpu().double().reshape(-1)
    b = torch.as_tensor(b).detach().float().cpu().double().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _run_step(tmp_path, dataset, extra, B, latent, seed=78):
    from test_dstep_gpu import _masks, _setup
    from csl_gan_amd import ops
    try:
        opt, tr, pe, oracle, Do = _setup(tmp_path, dataset, extra, B, latent)
        g = torch.Generator().manual_seed(seed)
        ch, im = (1, 28) if dataset == "MNIST" else (3, opt.im_size)
        img = (torch.randn(B, ch, im, im, generator=g) * 0.5).clamp(-1, 1)
        ms_a = (torch.randn(B, ch, im, im, generator=g) * 0.3).clamp(-1, 1)
        ms_p = (torch.randn(B, ch, im, im, generator=g) * 0.3).clamp(-1, 1)
        z, alpha = torch.randn(B, latent, generator=g), torch.rand(B, generator=g)
        tr.explicit = dict(ms_adapt=ms_a, pen_real=ms_p, alpha=alpha, z_adapt=z.cuda(), keep=True)
        pe.noise_multiplier = 0.0
        with _masks(G=tr.G, D=tr.D) as rec:       # the oracle replays the device's ReLU / LeakyReLU decisions (see the test below)
            tr.train_D(img.cuda(), None, z.cuda(), None, use_dp=True)
        torch.cuda.synchronize()
        modes = (ops.get_compute_dtype(), ops.get_storage_dtype())
    finally:
        ops.set_compute_dtype("fp32")
        ops.set_storage_dtype("fp32")
    return opt, tr, oracle, Do, rec, (img, z, ms_a, ms_p, alpha), modes


@pytest.mark.parametrize("dataset,extra,B,latent", STEP_CASES)
def test_train_D_bf16_storage_matches_fp32_oracle(tmp_path, dataset, extra, B, latent):
    """--compute_dtype bf16 --storage_dtype bf16 against the FP32 oracle, activation masks shared.

    Error model.  A bfloat16 rounding is a relative perturbation uniform in +-2^-9 (rms 2^-9/sqrt(3) = 1.1e-3) of one value, and
    a sum of many independently perturbed terms keeps that relative rms (times its cancellation).  A quantity that crossed D
    rounding stages collects sqrt(D) of them.  On the path of a critic weight gradient: the forward chain to the loss (per layer
    its stored input and the filter copy) and the backward chain from it (per layer the stored output gradient and the filter
    copy) — 4 + 4 layers x 2 = 16 stages, the SAME count as bf16 compute on fp32 tensors (the operand is rounded either way;
    storage moves the rounding from the consumer's load to the producer's store), plus the generator's 13-16 convs in front of
    the generated rows.  sqrt(16..32) x 1.1e-3 = 4.5e-3..6e-3 expected relative L2 error per gradient tensor; measured
    2.1e-3..7.8e-3 (B = 4, 128x128) — held to 2e-2 (3 sigma of the model), the whole gradient to 1.5e-2.  Observables that are
    maxima over entries (generated image, critic outputs, norms, clip norms) are held to 4e-2 of scale per entry (measured
    2.2e-2 on the 128x128 image: the tail of 2 x 10^5 entries at 5e-3 rms).

    This only holds on the SAME piecewise-linear function: under bf16 rounding about 0.3 % of the LeakyReLU / ReLU units sit
    closer to zero than their own rounding error and take the other slope, each such unit is off by 0.8 of its gradient, and
    sqrt(0.003) x 0.8 = 4e-2 — exactly the free-running error this step measures against the oracle (whole gradient 4.4e-2,
    tensors 3.5e-2..8e-2).  That figure is a property of LeakyReLU under ANY bf16 arithmetic, not of these kernels, so the oracle
    replays the device's masks (csl_gan_amd.nn.ActivationMaskRecorder -> oracle.nets.MaskPlayer) as the fp32 parity tests do."""
    opt, tr, oracle, Do, rec, (img, z, ms_a, ms_p, alpha), modes = _run_step(
        tmp_path, dataset, extra + ["--compute_dtype", "bf16", "--storage_dtype", "bf16"], B, latent)
    assert modes == ("bf16", "bf16") and opt.materialize == (extra[extra.index("--materialize") + 1] if "--materialize" in extra else "all")
    last = tr.last
    oracle.cfg.sigma = 0.0
    from test_dstep_gpu import _close as close, _masked_oracle
    with _masked_oracle(rec, G=oracle.G, D=Do) as player:
        obs = oracle.step(img, None, z, None, ms_adapt=ms_a, z_adapt=z, pen_real=ms_p, alpha=alpha, apply_update=False)
        assert player.exhausted()
    T = 4e-2
    close(last["fake_img"], obs["fake_img"], "fake_img (bf16 generator)", rtol=T)
    dscale = max(abs(obs["d_real_loss"]), abs(obs["d_fake_loss"]), obs["d_real"].abs().max().item())
    assert abs(float(last["d_real_loss"]) - obs["d_real_loss"]) <= T * dscale
    assert abs(float(last["d_fake_loss"]) - obs["d_fake_loss"]) <= T * dscale
    close(last["penalty"], obs["penalty"], "penalty", rtol=T)
    Cfin = oracle.max_grad_norm
    close(last["clip_params"], torch.tensor(Cfin if isinstance(Cfin, list) else [Cfin]), "clip params", rtol=T)
    n_o = obs["norms"]
    n_h = last["norms"].reshape(n_o.shape[0], -1)
    close(n_h[:, -B:], n_o[:, 1], "per-sample norms of the clipped pass", rtol=T)
    report = []
    names = [n for n, _ in tr.D.named_parameters()]
    for i, (a, b) in enumerate(zip(last["summed_grad"], obs["summed_grad"])):
        report.append((names[i], _rel_l2(a, b), _rel_l2(last["summed_clipped"][i], obs["summed_clipped"][i])))
    whole = _rel_l2(torch.cat([a.reshape(-1).float().cpu() for a in last["summed_grad"]]), torch.cat([b.reshape(-1) for b in obs["summed_grad"]]))
    print("\nbf16 storage %s %s: whole-gradient rel L2 %.3e; norms max rel %.3e; fake_img max rel %.3e" % (
        dataset, extra, whole, ((n_h[:, -B:].cpu() - n_o[:, 1]).abs().max() / n_o[:, 1].abs().max()).item(),
        ((last["fake_img"].float().cpu() - obs["fake_img"]).abs().max() / obs["fake_img"].abs().max()).item()))
    for n, e1, e2 in report:
        print("   %-22s summed_grad %.3e   summed_clipped %.3e" % (n, e1, e2))
    assert whole <= 1.5e-2, "whole summed gradient: relative L2 error %.3e" % whole
    for n, e1, e2 in report:
        assert e1 <= 2e-2, "summed_grad %s: relative L2 error %.3e" % (n, e1)
        assert e2 <= 2e-2, "summed_clipped %s: relative L2 error %.3e" % (n, e2)
