"""Command-line surface of the reference (options.py:113-287): every flag, the per-dataset default
tables (options.py:11-91), the derived flags (options.py:230-235), the incompatibility checks
(options.py:247-256), output-dir creation, seeding and the resume merge.

Additions of this build (all optional, none changes a reference default):
  --im_size 128        the 128x128 extension of BASELINE.json config 5
  --synthetic          use the synthetic in-memory dataset (the container has no MNIST/CelebA files)
  --data_cache PATH    real data through the preprocessed-tensor cache and the device prefetcher (csl_gan_amd/pipeline.py): the files
                       under --data_path are decoded / resized / cropped ONCE into PATH.*.u8 (uint8 NHWC memmap) and every batch is
                       uploaded as bytes on a side stream and normalised + flipped by one kernel
  --max_iters N        stop after N training iterations (smoke runs)
  --dist               one process per GPU under torch.distributed (RCCL); see csl_gan_amd/distributed.py
  --fuse_passes B      run the adaptive / generated / real discriminator passes as one forward+backward over the
                       concatenated batch (same numbers, fewer and fuller launches); needs --materialize private|ghost
  --grad_sample_dtype  storage type of the materialised per-sample weight gradients (fp32 | bf16; fp32 accumulate)
  --compute_dtype T    arithmetic of the conv / linear / weight-gradient kernels: fp32 (exact fp32 MFMA, default), bf16x3 (fp32
                       emulated from three bfloat16 pieces on the bf16 matrix cores: fp32-accurate, faster on large launches),
                       fp32_auto (per launch the faster of those two) or bf16 (operands rounded to bfloat16 inside the
                       kernels, fp32 accumulate; BASELINE.json configs[4]).  With bf16
                       the per-sample norms must be norms of the gradients that are actually summed, so ghost clipping (whose
                       Gram norms are computed in fp32) is replaced by --materialize private — unless the activations are also STORED as
                       bfloat16 (--storage_dtype bf16), where the Gram norms are the norms of the stored values' gradients
  --storage_dtype T    element type of the critic's activations and activation gradients in HBM: fp32 (default) or bf16 (needs
                       --compute_dtype bf16: bf16-stored kernels of csrc/igemm_bf16s.hip, bf16 filter copies, fp32 accumulation,
                       fp32 weight gradients / norms / clip / noise / Adam; BASELINE.json configs[4])
  --hip_graph B        (default True) record the DP D-step (dp_mode=gc, full batches, one process) once in a HIP graph and replay it
                       (trainer.GraphedDStep): no per-launch host work, and the step's second stream (gradient-penalty branch)
                       overlaps by dependency instead of by host timing; partial batches and every other mode run eagerly
  --moving_avg_beta B  the smoothing factor train.py:249 reads as opt.moving_avg_beta but options.py never defines
                       (imm_sens_scaling_mode=moving-avg-pl raises AttributeError in the reference without it)
  --materialize M      per-sample gradients kept in HBM: "all" passes (the fork's p.grad_sample layout) or only
                       the "private" (clipped) passes; "ghost" additionally never materialises layers with few
                       output pixels (Gram norms + clip-weighted dense wgrad) — see csl_gan_amd.engine.PrivacyEngine
Quirks kept on purpose: fill_defaults treats False like "unset" (options.py:95), so e.g. `-ispp False`
on CelebA still becomes True; --mean_sample_noise_std is parsed as int (options.py:166).
"""
import argparse
import json
import os
import random
from argparse import Namespace
from datetime import datetime

import torch

from . import util

_COMMON = dict(g_label_emb_mode="concat", d_label_emb_mode="concat", iter_on_mean_samples=0, grad_clip_mode="standard",
               imm_sens_scaling_mode="standard", tm_m=10)
MNIST_DEFAULTS = dict(
    _COMMON, data_path="/persist/datasets/mnist/", model="Vanilla", im_size=28, n_epochs=10000, g_lr=0.0002, d_lr=0.0002,
    batch_size=600, batch_split_size=60, train_set_size=60000, g_latent_dim=100, n_d_steps=1,
    aux_loss_type="cross_entropy", adam_b1=0.9, adam_b2=0.999, penalty=[], mean_sample_size=5000,
    mean_sample_noise_std=0.22, delta=1e-5, sigma=5.0, clipping_param=4.0, tm_max_val=-1, tm_min_val=1,
    save_every=50, log_every=100000, sample_every=600000, sample_num=100, n_classes=10, weights_seed=42)
CELEBA_DEFAULTS = dict(
    _COMMON, data_path="/persist/datasets/celeba/img_align_celeba/all/",
    label_path="/persist/datasets/celeba/Anno/list_attr_celeba.txt", label_attr="Male", model="DeepConvResNet",
    im_size=64, n_epochs=1000, g_lr=0.0001, d_lr=0.0001, batch_size=128, batch_split_size=32, train_set_size=180000,
    public_set_size=0, g_latent_dim=128, n_d_steps=5, aux_loss_type="wasserstein", adam_b1=0.0, adam_b2=0.9,
    penalty=["WGAN-GP"], mean_sample_size=1000, mean_sample_noise_std=0.12, delta=1e-6, sigma=0.5,
    imm_sens_scaling_vec=[20, 2, 15, 1.5, 10, 1.5, 10, 1, 30], imm_sens_per_param=True, clipping_param=200,
    clipping_param_per_layer=[1000, 200, 1000, 100, 1000, 100, 1000, 5, 2500], tm_min_val=-1, tm_max_val=1,
    save_every=10, log_every=20000, sample_every=60000, sample_num=25, n_classes=2, gp_lambda=10)


def fill_defaults(opt, default_dict):
    d = opt.__dict__
    for key, val in default_dict.items():
        if d.get(key) is None or d.get(key) is False:
            d[key] = val


def none_or_str(value):
    return None if value == "None" else value


def str2bool(v):
    if isinstance(v, bool):
        return v
    s = v.lower()
    if s in ("yes", "true", "t", "y", "1"):
        return True
    if s in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


_PENALTIES = [None, "WGAN-GP", "WGAN-GP1", "DRAGAN", "DRAGAN1"]
# (flags, kwargs) in the reference's order
_ARGS = [
    (("--weights_seed",), dict(type=int, default=42)),
    (("--manual_seed",), dict(type=int, default=-1)),
    (("dataset",), dict(type=str, choices=["MNIST", "CelebA"])),
    (("-d", "--data_path"), dict(type=str, default=None)),
    (("-lp", "--label_path"), dict(type=str, default=None)),
    (("-la", "--label_attr"), dict(type=str, default=None)),
    (("--model",), dict(type=str, choices=["Vanilla", "DeepConvResNet"], default=None)),
    (("--im_size",), dict(type=int, default=None, choices=[64, 48, 128])),
    (("--download_mnist",), dict(default=False, action="store_true")),
    (("-o", "--output_dir"), dict(type=str, default=None)),
    (("-rp", "--resume_path"), dict(type=str, default=None)),
    (("-re", "--resume_epochs"), dict(type=int, default=0)),
    (("-ka", "--keep_args"), dict(type=str, nargs="*", default=[])),
    (("-ne", "--n_epochs"), dict(type=int, default=None)),
    (("--d_lr",), dict(type=float, default=None)),
    (("--g_lr",), dict(type=float, default=None)),
    (("-wd", "--weight_decay"), dict(type=float, default=0)),
    (("-bs", "--batch_size"), dict(type=int, default=None)),
    (("-bss", "--batch_split_size"), dict(type=int, default=None)),
    (("-tss", "--train_set_size"), dict(type=int, default=None)),
    (("-gd", "--g_device"), dict(type=str, default="cpu")),
    (("-dd", "--d_device"), dict(type=str, default="cpu")),
    (("-nw", "--num_workers"), dict(type=int, default=8)),
    (("--g_latent_dim",), dict(type=int, default=None)),
    (("--n_d_steps",), dict(type=int, default=None)),
    (("--train_d_until_threshold",), dict(type=float, default=1e10)),
    (("-cond", "--conditional"), dict(action="store_true", default=False)),
    (("--g_label_emb_mode",), dict(type=str, choices=["embed", "concat"], default=None)),
    (("--d_label_emb_mode",), dict(type=str, choices=["embed", "concat"], default=None)),
    (("--conditional_arch",), dict(type=str, choices=["CGAN", "ACGAN", "WCGAN"], default="ACGAN")),
    (("--aux_loss_type",), dict(type=str, choices=["wasserstein", "cross_entropy"], default=None)),
    (("--aux_loss_scalar",), dict(type=float, default=1)),
    (("--aux_penalty",), dict(type=str2bool, default=True)),
    (("--d_fake_aux_loss",), dict(type=str2bool, default=True)),
    (("--adam_b1",), dict(type=float, default=None)),
    (("--adam_b2",), dict(type=float, default=None)),
    (("--penalty",), dict(type=str, nargs="*", choices=_PENALTIES, default=None)),
    (("-pss", "--public_set_size"), dict(type=int, default=0)),
    (("-nms", "--num_mean_samples"), dict(type=int, default=0)),
    (("-pupd", "--penalty_use_public_data"), dict(type=str2bool, default=True)),
    (("-wi", "--warmup_iter"), dict(type=int, default=0)),
    (("--mean_sample_size",), dict(type=int, default=None)),
    (("--mean_sample_noise_std",), dict(type=int, default=None)),
    (("--delta",), dict(type=float, default=None)),
    (("--sigma",), dict(type=float, default=None)),
    (("-eb", "--epsilon_budget"), dict(type=float, default=None)),
    (("-dpm", "--dp_mode"), dict(type=str, choices=["gc", "is", "tm", "sv"], default=None)),
    (("-ispp", "--imm_sens_per_param"), dict(type=str2bool, default=False)),
    (("-issv", "--imm_sens_scaling_vec"), dict(type=float, nargs="*", default=None)),
    (("-issm", "--imm_sens_scaling_mode"), dict(type=str, choices=["standard", "constant-pl", "moving-avg-pl"], default=None)),
    (("-gcs", "--grad_clip_split"), dict(type=str2bool, default=True)),
    (("-gcm", "--grad_clip_mode"), dict(type=str, choices=["standard", "adaptive", "constant-pl", "adaptive-pl"], default=None)),
    (("-c", "--clipping_param"), dict(type=float, default=None)),
    (("-cpl", "--clipping_param_per_layer"), dict(type=float, nargs="*", default=None)),
    (("-as", "--adaptive_scalar"), dict(type=float, default=1.5)),
    (("--adaptive_stat",), dict(choices=["mean", "max"], default="mean")),
    (("--smooth_sens_t",), dict(type=float, default=0.01)),
    (("--tm_m",), dict(type=int, default=None)),
    (("--tm_max_val",), dict(type=float, default=None)),
    (("--tm_min_val",), dict(type=float, default=None)),
    (("--tm_rho_per_epoch",), dict(type=float, default=10)),
    (("--tm_sens_compute_bs",), dict(type=float, default=None)),
    (("-bpc", "--backprop_clip"), dict(type=str2bool, default=False)),
    (("--bpc_back_clip_param",), dict(type=float, default=0.01)),
    (("--bpc_back_clip_param_pl",), dict(type=float, nargs="*", default=None)),
    (("--bpc_forward_clip_param",), dict(type=float, default=20)),
    (("--bpc_forward_clip_param_pl",), dict(type=float, nargs="*", default=None)),
    (("-bpcaas", "--bpc_auto_activation_scale"), dict(type=float, default=0.2)),
    (("-bpcawgs", "--bpc_auto_weight_grad_scale"), dict(type=float, default=1e-3)),
    (("--bpc_during_g_train",), dict(type=str2bool, default=True)),
    (("--save_every",), dict(type=int, default=None)),
    (("--log_every",), dict(type=int, default=None)),
    (("--sample_every",), dict(type=int, default=None)),
    (("--sample_num",), dict(type=int, default=None)),
    (("-p", "--profile_training"), dict(default=False, action="store_true")),
    # ---- additions of this build ----
    (("--synthetic",), dict(default=False, action="store_true")),
    (("--data_cache",), dict(type=str, default=None)),
    (("--max_iters",), dict(type=int, default=0)),
    (("--dist",), dict(default=False, action="store_true")),
    (("--materialize",), dict(type=str, choices=["all", "private", "ghost"], default="ghost")),
    (("--fuse_passes",), dict(type=str2bool, default=True)),
    (("--grad_sample_dtype",), dict(type=str, choices=["fp32", "bf16"], default="fp32")),
    (("--moving_avg_beta",), dict(type=float, default=None)),
    (("--hip_graph",), dict(type=str2bool, default=True)),
    (("--compute_dtype",), dict(type=str, choices=["fp32", "bf16", "bf16x3", "fp32_auto"], default="fp32")),
    (("--storage_dtype",), dict(type=str, choices=["fp32", "bf16"], default="fp32")),
]
ALWAYS_KEEP = ["g_device", "d_device", "num_workers", "resume_path", "resume_epochs"]


def build_parser():
    parser = argparse.ArgumentParser(description="DP-GAN training (csl-gan command line) on MI355X")
    for flags, kw in _ARGS:
        parser.add_argument(*flags, **kw)
    return parser


def parse(argv=None, make_dirs=True):
    opt = build_parser().parse_args(argv)
    opt.keep_args = opt.keep_args + ALWAYS_KEEP
    for k in ("data_path", "resume_path", "output_dir"):
        setattr(opt, k, util.add_slash(getattr(opt, k)))
    if opt.resume_path is not None:
        loaded = load_opt(opt.resume_path + "opt.txt")
        for arg in opt.keep_args:
            setattr(loaded, arg, getattr(opt, arg))
        loaded.output_dir = opt.resume_path
        return loaded
    return finalize(opt, make_dirs=make_dirs)


def finalize(opt, make_dirs=True):
    """Defaults merge, derived flags, checks, output dir, seeds (options.py:216-270)."""
    fill_defaults(opt, MNIST_DEFAULTS if opt.dataset == "MNIST" else CELEBA_DEFAULTS)
    opt.log_every_epochs = -1 if opt.log_every < opt.train_set_size else opt.log_every // opt.train_set_size
    opt.sample_every_epochs = -1 if opt.sample_every < opt.train_set_size else opt.sample_every // opt.train_set_size
    opt.log_every = max((opt.log_every // opt.batch_size) * opt.batch_size, 1)
    opt.sample_every = max((opt.sample_every // opt.batch_size) * opt.batch_size, 1)

    opt.use_dp = opt.dp_mode is not None
    opt.use_grad_clip_per_layer = opt.grad_clip_mode not in ("standard", "adaptive")
    opt.per_sample_grad = opt.dp_mode in ("gc", "tm", "sv")
    opt.is_acgan = opt.conditional and opt.conditional_arch == "ACGAN"
    opt.use_aux_loss = opt.conditional and opt.conditional_arch in ("ACGAN", "WCGAN")

    if opt.conditional_arch == "WCGAN" and opt.aux_penalty:
        print("Setting aux_penalty to false due to using WCGAN.")
        opt.aux_penalty = False
    if opt.model == "DeepConvResNet" and opt.use_dp:
        print("Setting train_d_until_threshold to -1, which is generally recommended for WGAN using DP")
        opt.train_d_until_threshold = -1
    if opt.backprop_clip:
        print("Backpropogation clipping implementation is experimental and not finished.")

    pen_dp = len(opt.penalty) > 0 and opt.use_dp
    no_public = opt.public_set_size < 1 and opt.num_mean_samples < 1
    if opt.imm_sens_per_param and opt.imm_sens_scaling_mode not in (None, "standard"):
        raise Exception("Calculating IS per parameter does not require per parameter scaling. Scaling estimates per-parameter calculation.")
    if opt.public_set_size > 0 and opt.num_mean_samples > 0:
        raise Exception("Both public data partition and mean samples were configured, please select only one.")
    if pen_dp and opt.penalty_use_public_data and no_public:
        raise Exception("In order to enable gradient penalty using public data, please enable mean sampling by setting num_mean_samples or public data by setting public_set_size.")
    if pen_dp and no_public:
        print("Currently configured to calculate penalty per-sample. It is strongly recommended that you use public data or mean samples for gradient penalties when using grad clipping.")
    if opt.model == "Vanilla" and (opt.g_label_emb_mode, opt.d_label_emb_mode) != ("concat", "concat"):
        raise Exception("Vanilla model with embedded labels not implemented")

    if pen_dp and not opt.penalty_use_public_data and opt.dp_mode == "gc" and getattr(opt, "materialize", "all") != "all":
        print("penalty_use_public_data=False: using --materialize all (the per-sample penalty gradients are added to p.grad_sample of "
              "every parameter, train.py:447)")
        opt.materialize = "all"
    if getattr(opt, "compute_dtype", "fp32") == "bf16" and opt.materialize == "ghost" and getattr(opt, "storage_dtype", "fp32") != "bf16":
        # with bf16 STORAGE the operands are already the rounded values: the Gram norms (fp32 sums of exact products of the stored
        # values) ARE the norms of the gradients the bf16 matrix core sums, and the clip weights are applied in fp32 to each sample's
        # accumulated product (cslgan_conv2d_wgrad_scaled_bf16s) — ghost clipping stays on in that mode
        print("compute_dtype=bf16: using --materialize private (ghost clipping's Gram norms are fp32 norms of unrounded products)")
        opt.materialize = "private"
    if getattr(opt, "storage_dtype", "fp32") == "bf16" and getattr(opt, "compute_dtype", "fp32") != "bf16":
        raise Exception("--storage_dtype bf16 needs --compute_dtype bf16 (bf16-stored activations are multiplied on the bf16 matrix cores)")
    if getattr(opt, "storage_dtype", "fp32") == "bf16" and getattr(opt, "backprop_clip", False):
        raise Exception("--storage_dtype bf16 is not combined with --backprop_clip")

    if not opt.output_dir:
        stamp = datetime.now().strftime("output/%m-%d-%H:%M-")
        opt.output_dir = "%s%s-g%s-d%s/" % (stamp, opt.dataset, str(opt.g_device)[-1], str(opt.d_device)[-1])
    if make_dirs:
        for sub in ("", "samples/", "saves/", "code/"):
            os.makedirs(opt.output_dir + sub, exist_ok=True)

    if opt.manual_seed < 0:
        opt.manual_seed = random.randint(1, 1000000)
    random.seed(opt.manual_seed)
    torch.manual_seed(opt.manual_seed)
    return opt


def load_opt(path):
    opt = Namespace()
    with open(path, "r") as f:
        opt.__dict__ = json.load(f)
    return opt
