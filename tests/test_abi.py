"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol that
include/cslgan.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from csl_gan_amd import build
    return build.build()


def test_header_symbols_are_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "cslgan.h")).read()
    declared = set(re.findall(r"\b(cslgan_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 16
    L = ctypes.CDLL(built_lib)
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    from csl_gan_amd import _lib
    assert set(_lib.EXPORTS) == declared


def test_library_loads_and_reports_version(built_lib):
    from csl_gan_amd import _lib
    L = _lib.lib()
    assert L.cslgan_version() == _lib.ABI_VERSION
    assert L.cslgan_device_count() >= 0
    assert isinstance(L.cslgan_last_error(), bytes)


def test_invalid_arguments_fail_loudly_without_gpu(built_lib):
    """Argument validation happens on the host before any launch: safe to call without a device."""
    from csl_gan_amd import _lib
    L = _lib.lib()
    rc = L.cslgan_sample_sqnorm_f32(None, 4, None, None)
    assert rc == -1 and b"null" in L.cslgan_last_error()
    d = _lib.ConvT(1, 8, 8, 4, 4, 5, 5, 2, 2, 0, 99, 99)
    rc = L.cslgan_conv2d_fwd_f32(ctypes.byref(d), 1, 1, None, None, 0, 1, None)
    assert rc == -1 and b"does not match" in L.cslgan_last_error()
    # round-2 entries
    rc = L.cslgan_mean_sample_f32(None, 1, 4, 16, None, None, 2, 0.0, 0.0, 1, 1, None, None, None)
    assert rc == -1 and b"null" in L.cslgan_last_error()
    rc = L.cslgan_mean_sample_f32(16, 2, 4, 16, None, 16, 2, 0.0, 0.0, 1, 1, 16, None, None)
    assert rc == -1 and b"give labels" in L.cslgan_last_error()
    rc = L.cslgan_mean_sample_f32(16, 1, 4096, 16, None, None, 2, 0.0, 0.0, 1, 1, 16, None, None)
    assert rc == -1 and b"in-kernel permutations" in L.cslgan_last_error()
    # round-3 entries (csrc/step_kernels.hip)
    rc = L.cslgan_segment_means_f32(16, 9, (ctypes.c_int32 * 9)(*[1] * 9), (ctypes.c_float * 9)(*[1.0] * 9), 16, None, None)
    assert rc == -1 and b"n_seg" in L.cslgan_last_error()
    rc = L.cslgan_lipschitz_term_f32(16, 4, 8, 0, 1.0, 16, 16, None, None, None)
    assert rc == -1 and b"null" in L.cslgan_last_error()
    rc = L.cslgan_grad_log_stats_f32(16, 9, 128, 64, 128, 16, 1, 1e-6, 16, 16, 16, 16, 16, None)
    assert rc == -1 and b"outside a row" in L.cslgan_last_error()
    d2 = _lib.ConvT(6, 32, 32, 64, 128, 5, 5, 2, 2, 0, 16, 16)
    first = (ctypes.c_int32 * 2)(0, 0)
    ptrs = (ctypes.c_void_p * 2)(None, None)
    rc = L.cslgan_conv2d_wgrad_blocks_f32(ctypes.byref(d2), 16, 16, 1.0, 2, first, ptrs, ptrs, None)
    assert rc == -1 and b"must increase" in L.cslgan_last_error()
    d3 = _lib.ConvT(6, 32, 32, 3, 128, 5, 5, 2, 2, 0, 16, 16)
    first = (ctypes.c_int32 * 1)(0)
    rc = L.cslgan_conv2d_wgrad_blocks_f32(ctypes.byref(d3), 16, 16, 1.0, 1, first, ptrs, ptrs, None)
    assert rc == -1 and b"not taken by the LDS-resident kernel" in L.cslgan_last_error()


def test_bf16_storage_entries_validate_their_arguments(built_lib):
    """csrc/igemm_bf16s.hip and the bf16 forms of the first-layer / vector-ALU / normalisation kernels (round 3): shapes they do not
    take are refused on the host, before any launch."""
    from csl_gan_amd import _lib
    L = _lib.lib()
    err = lambda: L.cslgan_last_error()
    d = _lib.ConvT(2, 16, 16, 12, 64, 5, 5, 2, 2, 1, 8, 8)               # 12 channels: not a multiple of 8
    assert L.cslgan_conv2d_fwd_bf16s(ctypes.byref(d), 16, 16, 16, 0, None, None, 0, 0, 16, 1, None) == -1 and b"multiple of 8" in err()
    assert L.cslgan_conv2d_fwd_bf16s(ctypes.byref(d), None, 16, 16, 0, None, None, 0, 0, 16, 1, None) == -1 and b"null" in err()
    d = _lib.ConvT(2, 16, 16, 64, 12, 5, 5, 2, 2, 1, 8, 8)               # 12 output channels: the data gradient's reduction
    assert L.cslgan_conv2d_dgrad_bf16s(ctypes.byref(d), 16, 16, 16, 0, None, 16, 1, None) == -1 and b"multiple of 8" in err()
    d = _lib.ConvT(2, 16, 16, 64, 128, 5, 5, 3, 2, 1, 6, 6)
    assert L.cslgan_conv2d_dgrad_bf16s(ctypes.byref(d), 16, 16, 16, 0, None, 16, 1, None) == -1 and b"stride" in err()
    d = _lib.ConvT(6, 16, 16, 64, 128, 5, 5, 2, 2, 1, 8, 8)
    assert L.cslgan_conv2d_wgrad_grouped_bf16s(ctypes.byref(d), 16, 16, 4, 1.0, 16, 0, None, None) == -1 and b"not divisible" in err()
    assert L.cslgan_conv2d_wgrad_grouped_bf16s(ctypes.byref(d), 16, 16, 1, 1.0, None, 0, None, None) == -1 and b"neither" in err()
    assert L.cslgan_cast_f32_bf16(16, 20, 64, None) == -1 and b"misaligned" in err()
    assert L.cslgan_act_bwd_bf16(16, 16, 12, 0.2, 16, None) == -1 and b"multiple of 8" in err()
    assert L.cslgan_bias_grad_grouped_bf16(16, 4, 16, 24, 1, 1.0, 16, None, None) == -1 and b"multiple of 8" in err()
    assert L.cslgan_linear_k1_dgrad_bf16s(16, 16, None, 4, 100, 16, None) == -1 and b"bad argument" in err()
    assert L.cslgan_linear_k1_wgrad_bf16s(16, 16, None, 6, 64, 4, 1.0, 16, None, None) == -1 and b"bad argument" in err()
    d = _lib.ConvT(4, 8, 8, 64, 128, 5, 5, 2, 2, 1, 4, 4)               # 16 output pixels per sample: a 64-pixel K tile would span samples
    assert L.cslgan_conv2d_wgrad_scaled_bf16s(ctypes.byref(d), 16, 16, 16, 2, 1.0, 16, None) == -1 and b"one sample" in err()
    d = _lib.ConvT(2, 30, 30, 3, 64, 5, 5, 2, 2, 0, 15, 15)              # not a 16x32-tileable image
    assert L.cslgan_conv2d_c3_fwd_bf16out(ctypes.byref(d), 16, 16, None, 1, 16, None) == -1 and b"first layer" in err()
    assert L.cslgan_conv2d_c3_wgrad_bf16gy(ctypes.byref(d), 16, 16, 1.0, 16, None, None) == -1 and b"first-layer kernel" in err()
    d = _lib.ConvT(2, 16, 16, 32, 3, 3, 3, 1, 1, 0, 16, 16)              # 32 input channels: the vector-ALU kernel takes 64
    assert L.cslgan_conv2d_fwd_skinny_bf16in(ctypes.byref(d), 16, 16, None, 3, 16, None) == -1 and b"64 input channels" in err()
    d = _lib.ConvT(2, 16, 16, 8, 64, 5, 5, 2, 2, 0, 8, 8)
    assert L.cslgan_conv2d_dgrad_skinny_bf16in(ctypes.byref(d), 16, 16, 16, 0, 16, None) == -1 and b"1..4 input channels" in err()
    assert L.cslgan_groupnorm_act_bf16s(16, 1, 16, 16, 2, 64, 48, 32, 1e-5, 1, 16, 16, 0, None, None) == -1 and b"not divisible" in err()
    segs = _lib.SegsT()                                                  # per-segment row counts are for plain column sums
    segs.n_seg = 1; segs.inp[0] = 16; segs.out[0] = 16; segs.len[0] = 8; segs.row_stride[0] = 8; segs.rows[0] = 4
    assert L.cslgan_clip_accum_noise_f32(ctypes.byref(segs), 4, 16, 0, None, 0, 0, 1.0, 0.0, None) == -1 and b"factors == NULL" in err()


def test_ops_refuse_cpu_tensors(built_lib):
    import torch
    from csl_gan_amd import ops
    with pytest.raises(RuntimeError, match="device tensor"):
        ops.conv2d_fwd(torch.zeros(1, 4, 4, 4), torch.zeros(4, 3, 3, 4))


def test_missing_library_raises(monkeypatch, tmp_path):
    from csl_gan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipLibraryMissing):
        _lib.lib()


def test_library_sources_do_not_use_hip_memset():
    """A hipMemsetAsync captured into a HIP graph replayed wrong fills on ROCm 7.2 (tests/test_kernels_gpu.py::
    test_zeroed_accumulators_replay_correctly_in_a_captured_graph); accumulators are zeroed by the library's own kernel."""
    import glob, os, re
    src = os.path.join(os.path.dirname(__file__), "..", "csl_gan_amd", "csrc")
    files = glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))
    assert files
    for f in files:
        code = re.sub(r"//[^\n]*", "", open(f).read())
        assert not re.search(r"\bhipMemset\w*\s*\(", code), "%s calls hipMemset*: use cslgan::zero_floats" % os.path.basename(f)
