#!/usr/bin/env python3
"""Benchmark of the DP discriminator step (BASELINE.json metric) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms measure N ranks: started WITHOUT a torchrun environment and with --gpus N > 1, this process (which has not touched
the GPU) launches the second form itself as a child, relays rank 0's JSON line and exits with the child's status; inside a
torchrun environment --gpus must equal WORLD_SIZE (a mismatch is an error, not a warning) and the line's ``n_gpus`` is the
world size the process group reports.

One "step" = one `train_D` (train.py:360-500) of the headline configuration
    CelebA DCResNet, dp_mode=gc, -gcm adaptive-pl, -nms 32, WGAN-GP on mean samples, bs=128 per GPU
on synthetic 3x64x64 data already resident in HBM: adaptive-clipping pass on mean samples, generator
forward for the fakes, fake + real discriminator passes, per-sample gradients, norms, clip,
accumulate, WGAN-GP double backward, Gaussian noise, Adam.  Weak scaling: every rank processes its
own 128 images; the clipped+noised gradients are all-reduced over RCCL.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the roofline accounting).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic work per image as the REFERENCE executes it (SURVEY.md §8d / BASELINE.md §6), FLOP: the generator's UpsampleConv
# layers are charged at the reference's C input channels although only C/4 are distinct
FLOP_PER_IMG_STEP = 14.3e9
PEAK_FP32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense bf16 MFMA (never the 2:1-sparsity headline)
PEAK_HBM_GBPS = 8000.0               # MI355X_MICROARCH.md: HBM3E
B_PER_GPU = 128


def build_trainer(rank, world, local, batch=B_PER_GPU, outdir=None, extra=()):
    from csl_gan_amd import distributed as D, init_util, options
    from csl_gan_amd.mean_sampler import MeanSampler
    from csl_gan_amd.trainer import Trainer
    dev = "cuda:%d" % local
    outdir = outdir or tempfile.mkdtemp(prefix="cslgan_bench_")
    opt = options.parse(["CelebA", "-dpm", "gc", "-gcm", "adaptive-pl", "-nms", "32", "-bs", str(batch), "-gd", dev, "-dd", dev,
                         "-o", outdir, "--manual_seed", "1234", "--synthetic"] + list(extra))
    G, Dm = init_util.init_models(opt)
    g = torch.Generator().manual_seed(1234)
    # 32 mean samples: mean of 1000 synthetic images + N(0, 0.12^2)  (options.py:71-72)
    S = opt.im_size
    acc = torch.zeros(32, 3, S, S)
    for i in range(32):
        acc[i] = (torch.randn(250, 3, S, S, generator=g) * 0.5).clamp(-1, 1).mean(0) if S > 64 else \
            (torch.randn(1000, 3, S, S, generator=g) * 0.5).clamp(-1, 1).mean(0)
    ms = MeanSampler(noise_std=0.12, num_samples=32, mean_size=1000, dataset_size=opt.train_set_size, device=dev, res=S)
    ms.mean_samples = (acc + torch.randn(acc.shape, generator=g) * 0.12).unsqueeze(0).to(dev)
    reducer = D.FlatGradReducer() if world > 1 else None
    tr = Trainer(opt, G, Dm, mean_sampler=ms, log_to=os.path.join(outdir, "log_rank%d.csv" % rank), world_size=world,
                 rank=rank, grad_reducer=reducer)
    tr.setup_privacy_engine()
    gr = torch.Generator().manual_seed(1234 + rank)
    img = (torch.randn(batch, 3, S, S, generator=gr) * 0.5).clamp(-1, 1).to(dev)
    return opt, tr, img


def cpu_baseline():
    """The oracle's D-step (same step definition, hook-based unfold+einsum per-sample gradients —
    the algorithm family the reference's Opacus dependency uses) on the host cores."""
    from oracle import dp_engine as E
    from oracle.dstep import OracleDStep, StepConfig
    from oracle.nets import build_models
    threads = min(16, os.cpu_count() or 1)      # the GPU box gives one GPU a 16-core CPU share
    torch.set_num_threads(threads)
    # ONE fixed algorithm from round 4 on: the hook-based fp32 unfold+einsum step with fp32 norm reductions (what the reference's
    # dependency executes).  The checker's float64 norm reductions (a 4.4 GB copy per pass at bs=128) are switched off for the timed
    # steps — they are test hygiene and made this baseline drift 24.6 -> 22.1 -> 14.9 images/s over rounds 1-3.
    E.set_norm_dtype(torch.float32)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    G, Dm = build_models(weights_seed=42, manual_seed=1234)
    st = OracleDStep(G, Dm, StepConfig(grad_clip_mode="adaptive-pl", clipping_param_per_layer=[1.0] * 9, sigma=0.5))
    g = torch.Generator().manual_seed(1234)

    def one(B):
        img = (torch.randn(B, 3, 64, 64, generator=g) * 0.5).clamp(-1, 1)
        ms = torch.randn(B, 3, 64, 64, generator=g) * 0.2
        t0 = time.perf_counter()
        st.step(img, None, torch.randn(B, 128, generator=g), None, ms_adapt=ms, pen_real=ms, alpha=torch.rand(B, generator=g), noise_gen=g)
        return time.perf_counter() - t0
    try:
        one(8)                       # warm-up (allocator, thread pool)
        n_steps, dt = 0, 0.0
        while dt < 15.0 and n_steps < 5:          # about 15-25 s of host work
            dt += one(B_PER_GPU)
            n_steps += 1
    finally:
        E.set_norm_dtype(torch.float64)
    return {"value": round(n_steps * B_PER_GPU / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "cpu_model": cpu_model,
            "host_cpus": os.cpu_count(), "kind": "port", "algorithm": "hook-based fp32 unfold+einsum per-sample gradients, fp32 norm reductions",
            "sample": "%d oracle D-step(s) (config 3) at bs=128 after a bs=8 warm-up, %.1f s" % (n_steps, dt)}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """Start `n` ranks of this script under torch.distributed.run (fresh child processes: this parent never initialises the
    GPU, so nothing is exec'ed from a GPU process), relay the ranks' stdout (rank 0 prints the ONE JSON line) and stderr, and
    return the launcher's exit status — non-zero if any rank failed."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    return subprocess.run(cmd, env=env).returncode


def _launcher_selftest(mode):
    """CPU rehearsal of the launcher contract (tests/test_host_logic.py): every rank joins a gloo group, all-reduces a one, rank 0
    prints the JSON line with the world size the GROUP reports; mode "fail" makes the last rank exit non-zero first."""
    import torch.distributed as dist
    from csl_gan_amd import distributed as D
    world, rank, _ = D.init("gloo")
    if mode == "fail" and rank == world - 1:
        sys.exit(3)
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
        D.barrier()
    if rank == 0:
        print(json.dumps({"n_gpus": dist.get_world_size() if dist.is_initialized() else 1, "sum": float(t)}))
    if dist.is_initialized():
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--loop-steps", type=int, default=10, help="iterations of the full train() loop timed for the secondary metric (0 = skip)")
    ap.add_argument("--no-variants", action="store_true", help="skip the hip_graph / fp32_auto variant passes (profiling runs)")
    ap.add_argument("--dump-shapes", type=str, default="", help="write the per-kernel, per-shape launch table (HIP-event times) to this file")
    ap.add_argument("--opt", type=str, default="", help="extra train.py flags for experiments, e.g. '--grad_sample_dtype bf16' "
                    "(the headline line is the run WITHOUT this)")
    ap.add_argument("--compute", type=str, default="fp32_auto", choices=["fp32_auto", "fp32"],
                    help="arithmetic of the headline step: fp32_auto (default; fp32 results from three bfloat16 pieces per operand on the bf16 "
                         "matrix cores wherever a launch is large enough, the exact fp32 MFMA kernels elsewhere) or fp32 (exact fp32 MFMA "
                         "everywhere: the round 1-3 headline, reported as variants.fp32_exact otherwise)")
    ap.add_argument("--launcher-selftest", type=str, default="", choices=["", "ok", "fail"], help=argparse.SUPPRESS)
    a = ap.parse_args()

    in_torchrun = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if a.gpus > 1 and not in_torchrun:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    if a.launcher_selftest:
        return _launcher_selftest(a.launcher_selftest)

    from csl_gan_amd import distributed as D, ops
    # CSLGAN_DIST_BACKEND=gloo + CSLGAN_FORCE_DEVICE=0 rehearse the N>1 code path on a one-GPU box
    world, rank, local = D.init(os.environ.get("CSLGAN_DIST_BACKEND", "nccl"))
    if "CSLGAN_FORCE_DEVICE" in os.environ:
        local = int(os.environ["CSLGAN_FORCE_DEVICE"])
    if torch.distributed.is_initialized():
        world = torch.distributed.get_world_size()        # what the process group (RCCL) reports, not what the environment claims
    if world != a.gpus:
        print("error: --gpus %d but the process group has %d rank(s)" % (a.gpus, world), file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # option parsing prints notices; stdout carries ONE JSON line
        user = a.opt.split()
        opt, tr, img = build_trainer(rank, world, local, extra=(user if "--compute_dtype" in user else ["--compute_dtype", a.compute] + user))
    B = img.shape[0]

    def step():
        tr.train_D(img, None, tr.gen_z(B), None, use_dp=True)
        pass  # statistics keep accumulating in place between log lines, as in training

    def time_region(fn, n):
        """n calls of fn bracketed by barrier + synchronize on both sides; MAX over ranks of the wall time."""
        torch.cuda.synchronize()
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        D.barrier()
        dt_ = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt_], device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt_ = float(t)
        return dt_

    # Warm-up.  Its last steps are instrumented launch by launch (HIP events on the launch stream): they name the dominant
    # device kernel and give the per-kernel tables.  Inside the eager TIMED region only that kernel's launches carry events — two
    # event records per launch cost the host ~5 us, i.e. ~1 ms per step if every one of the ~170 launches were watched.
    n_probe = min(3, a.warmup)
    for _ in range(a.warmup - n_probe):
        step()
    torch.cuda.synchronize()
    probe = ops.LaunchTimer()
    ops.set_launch_timer(probe)
    # the per-kernel tables are ISOLATED measurements: the step's second stream (gradient-penalty branch beside the critic pass) is
    # and the clip phase's second stream are switched off for these steps only — with them, concurrent kernels share the CUs and each one's event time is stretched
    saved_env = {k: os.environ.get(k) for k in ("CSLGAN_GP_STREAM", "CSLGAN_CLIP_STREAM")}
    os.environ.update({k: "0" for k in saved_env})
    for _ in range(n_probe):
        # A few milliseconds of head start for the host: an event pair brackets the launch CALL, so on an idle GPU the pair also times the
        # host's way from the first record to the launch (the generator's first small kernels read 0.3-0.6 ms for 10-30 us of work).
        # With the queue ahead of the device every pair times its kernel.
        torch.cuda._sleep(int(6e-3 * 2.1e9))
        step()
    torch.cuda.synchronize()
    for k, v in saved_env.items():
        if v is None:
            del os.environ[k]
        else:
            os.environ[k] = v
    ops.set_launch_timer(None)
    pk = {k: v for k, v in probe.summary(by_kernel=True).items() if v["exec_flop"] > 0}
    dom_name = max(pk.values(), key=lambda k: k["ms"])["name"] if pk else None
    # Region A — the step launched eagerly, the dominant kernel's launches bracketed by HIP events (the roofline figure)
    timer = ops.LaunchTimer(only=probe.keys_of_kernel(dom_name) if dom_name else set())
    ops.set_launch_timer(timer)
    dt_eager = time_region(step, a.steps)
    ops.set_launch_timer(None)
    def finish(dt, launch_mode, graph_err, variant, loop, with_cpu=True):
        """Assemble and print the ONE JSON line (rank 0) from the instrumented / eager regions plus whatever came after them."""
        if rank != 0:
            return
        ips = world * B * a.steps / dt
        n_pr = max(n_probe, 1)
        entries = probe.summary()                                   # per-kernel tables: the instrumented warm-up steps
        kernels = probe.summary(by_kernel=True)
        shapes = probe.summary(by_kernel=True, by_shape=True)
        timed = timer.summary(by_kernel=True)                       # the dominant kernel, inside the timed region
        timed_shapes = timer.summary(by_kernel=True, by_shape=True)
        # roofline: the dominant device KERNEL by summed HIP-event time (names as rocprofv3 lists them)
        dom = timed.get(dom_name) if dom_name else None
        roof = None
        if dom:
            traffic, traffic_src = None, None
            try:        # fabric bytes per launch of that kernel from the newest committed rocprofv3 PMC passes (scripts/collect_profiles.sh)
                import glob
                src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
                with open(src) as f:
                    pk = {k.replace(" ", ""): v for k, v in json.load(f)["per_kernel"].items()}
                v = pk.get(dom["name"].replace(" ", ""))
                traffic = None if v is None else round((v["fetch_MB_per_launch"] + v["write_MB_per_launch"]) * 1e6)
                traffic_src = "profiles/" + os.path.basename(src) + " (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, not measured in this run)"
            except Exception:
                traffic = None
            # a bf16x3 kernel issues SIX bf16 MFMAs per logical fp32 multiply-add step: its executed matrix FLOP are 6x the logical
            x3_kernel = "bf16x3" in dom["name"] or ("igemm_x3" in dom["name"] and ",3," in dom["name"])
            mfma_mult = 6.0 if x3_kernel else 1.0
            ach = mfma_mult * dom["exec_flop"] / (dom["ms"] * 1e-3) / 1e12
            # bf16 matrix-core kernels: the fp32-tensor family of csrc/igemm_bf16.hip and the bf16-stored family of csrc/igemm_bf16s.hip
            on_bf16 = "bf16" in dom["name"] or any(t in dom["name"] for t in ("igemm_kcs_kernel", "igemm_mcs_kernel", "igemm_mcs_tr_kernel", "igemm_halos_kernel",
                                                                             "igemm_x3h_kernel", "igemm_x3w_kernel"))
            peak = PEAK_BF16_MFMA_TFLOPS if on_bf16 else PEAK_FP32_MFMA_TFLOPS
            worst = sorted((v for k, v in timed_shapes.items() if k.startswith(dom["name"])), key=lambda v: -v["ms"])
            roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": traffic, "traffic_from": traffic_src,
                    "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["n"]),
                    "flop_per_launch_executed": round(mfma_mult * dom["exec_flop"] / dom["n"]),
                    "logical_fp32_tflops": round(dom["exec_flop"] / (dom["ms"] * 1e-3) / 1e12, 2),
                    "reference_algorithmic_tflops": round(dom["flop"] / (dom["ms"] * 1e-3) / 1e12, 2),
                    "note": "achieved = FLOP the kernel executes / its summed HIP-event time" + (
                                " — a three-piece (x3) kernel issues SIX bf16 MFMAs per logical fp32 multiply-add step, so its executed matrix FLOP are "
                                "6x logical_fp32_tflops and the peak is the dense bf16 MFMA rate" if x3_kernel else "") +
                            "; reference_algorithmic_tflops charges the UpsampleConv layers at the reference's 4x redundant channel count and "
                            "is NOT a roofline fraction",
                    "launches_per_step": dom["n"] / a.steps, "avg_launch_ms": round(dom["ms"] / dom["n"], 4),
                    "share_of_step": round(dom["ms"] / (dt_eager * 1e3), 3),
                    "measured_in": "HIP events around this kernel's launches over the %d eagerly launched timed steps (%.3f ms/step)%s" % (
                        a.steps, dt_eager / a.steps * 1e3,
                        "; the headline region replays the same launches from a HIP graph, where events cannot be placed" if launch_mode.startswith("hip_graph") else ""),
                    "launch_shapes": {v["name"][len(dom["name"]) + 1:]: {"n_per_step": v["n"] / a.steps, "avg_ms": round(v["ms"] / v["n"], 4),
                                                                         "tflops": round(v["exec_flop"] / (v["ms"] * 1e-3) / 1e12, 1)}
                                      for v in worst[:8]}}
        # the HBM group: clip_accum_noise over the materialised per-sample gradients / slabs, from the instrumented steps
        roof_hbm = None
        ck = kernels.get("clip_accum_noise_kernel<float>")
        if ck and ck["ms"] > 0:
            gbs = ck["bytes"] / (ck["ms"] * 1e-3) / 1e9
            roof_hbm = {"bound": "hbm", "kernel": "clip_accum_noise_kernel<float>", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBPS, 4), "launches_per_step": ck["n"] / n_pr, "MB_per_step": round(ck["bytes"] / n_pr / 1e6, 1),
                        "avg_launch_us": round(ck["ms"] / ck["n"] * 1e3, 2),
                        "note": "algorithmic bytes of each launch (rows x row length read once + one output row) / HIP-event time, instrumented "
                                "warm-up steps; SURVEY §8d's materialised-path figure is 34.5 MB/img/clipped pass — ghost clipping materialises only "
                                "conv1 + conv2 (5 %% of the parameters), so the kernel moves %.1f MB per step instead of 4.4 GB and is launch-latency "
                                "sized" % (ck["bytes"] / n_pr / 1e6)}
        exec_flop_step = sum(v["exec_flop"] for v in kernels.values()) / n_pr
        if a.dump_shapes:
            with open(a.dump_shapes, "w") as f:
                f.write("# HIP-event times of the launch-by-launch instrumented steps (eager, one stream, %d step(s)).  An event pair brackets the launch\n"
                        "# CALL: for a few short launches at the head of the generator (its first linear layer, the first 1x1 / 5x5 convs after a\n"
                        "# normalisation) the pair reads 0.2-0.6 ms where rocprofv3 --kernel-trace of the same command shows 10-250 us\n"
                        "# (profiles/*kernel_stats*.csv, authoritative for kernel durations; the headline region replays a HIP graph).\n" % n_pr)
                for k, v in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"]):
                    f.write("%-78s n/step %5.1f  avg_ms %8.4f  ms/step %7.3f  %s\n" % (
                        k, v["n"] / n_pr, v["ms"] / v["n"], v["ms"] / n_pr,
                        ("%6.1f TF" % (v["exec_flop"] / (v["ms"] * 1e-3) / 1e12)) if v["exec_flop"] else ("%7.1f GB/s" % (v["bytes"] / (v["ms"] * 1e-3) / 1e9))))
        bf16 = getattr(opt, "compute_dtype", "fp32") == "bf16"
        step_peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_FP32_MFMA_TFLOPS
        # mixed arithmetic (fp32_auto / bf16x3): a launch on a three-piece kernel executes 6 bf16 MFMA FLOP per logical fp32 FLOP and is
        # priced against the dense bf16 peak, every other launch against the fp32 MFMA peak; the sum is the step's matrix-pipe floor
        is_x3 = lambda k: "bf16x3" in k or ("igemm_x3" in k and ",3," in k)
        x3_flop = sum(v["exec_flop"] for k, v in kernels.items() if is_x3(k)) / n_pr
        mfma_floor_s = 6.0 * x3_flop / (PEAK_BF16_MFMA_TFLOPS * 1e12) + (exec_flop_step - x3_flop) / (step_peak * 1e12)
        mode = ("dp_mode=gc -gcm %s" % opt.grad_clip_mode) if opt.dp_mode == "gc" else ("dp_mode=is -ispp %s" % bool(opt.imm_sens_per_param))
        line = {
            # the BASELINE.json metric for the default command; with --opt the line names what was actually run
            "metric": "images/sec/GPU CelebA DCResNet dp_mode=%s bs=%d at 1/2/4/8 MI355X" % (opt.dp_mode, B),
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": "CelebA DCResNet D-step: %s -nms %d, WGAN-GP on mean samples, 3x%dx%d" % (mode, opt.num_mean_samples, opt.im_size, opt.im_size),
                       "compute_dtype": getattr(opt, "compute_dtype", "fp32"), "storage_dtype": getattr(opt, "storage_dtype", "fp32"),
                       "launch": launch_mode, "batch_per_gpu": B, "global_batch": B * world, "parallelism": "dp%d" % world,
                       "materialize": getattr(opt, "materialize", "all"), "grad_sample_dtype": getattr(opt, "grad_sample_dtype", "fp32"),
                       "fuse_passes": bool(getattr(opt, "fuse_passes", False)),
                       "step": ("train_D (adaptive pass + G fwd + 2 D passes + per-sample grads + clip + GP + noise + Adam)" if opt.dp_mode == "gc" else
                                "train_D (G fwd + D passes + create_graph gradients + one double-backward sweep per sensitivity + GP + noise + Adam)")},
            "per_gpu": round(ips / world, 2),
            "step_gflop_executed_per_image": round(exec_flop_step / B / 1e9, 3),
            "step_tflops_executed": round(exec_flop_step / (dt / a.steps) / 1e12, 2),
            ("step_frac_of_bf16_mfma_peak" if bf16 else "step_frac_of_fp32_mfma_peak"): round(exec_flop_step / (dt / a.steps) / 1e12 / step_peak, 4),
            "step_share_of_flop_on_x3_kernels": round(x3_flop / max(exec_flop_step, 1.0), 4),
            "step_mfma_floor_ms": round(mfma_floor_s * 1e3, 3),
            "step_frac_of_mfma_roofline": round(mfma_floor_s / (dt / a.steps), 4),
            "step_tflops_reference_algorithmic": None if a.opt else round(FLOP_PER_IMG_STEP * ips / world / 1e12, 2),
            "roofline": roof,
            "roofline_hbm": roof_hbm,
            "secondary": loop,
            "variants": variant,
            "graph_error": graph_err,
            "tables_from": "%d launch-by-launch instrumented warm-up step(s) run on ONE stream (isolated kernel times); roofline from the "
                           "eagerly launched timed region; both timed regions run the step's two streams" % n_pr,
            "entries_ms_per_step": {k: round(v["ms"] / n_pr, 3) for k, v in sorted(entries.items(), key=lambda kv: -kv[1]["ms"])},
            "kernels_ms_per_step": {k: {"ms": round(v["ms"] / n_pr, 3), "n": v["n"] / n_pr,
                                        "tflops": round(v["exec_flop"] / (v["ms"] * 1e-3) / 1e12, 1) if v["exec_flop"] else None,
                                        "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if not v["exec_flop"] else None}
                                    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])},
        }
        if world == 1 and not a.no_cpu_baseline and with_cpu:
            line["cpu_baseline"] = cpu_baseline()
            line["gpu_over_cpu"] = round(ips / line["cpu_baseline"]["value"], 1)
        print(json.dumps(line))

    # Region B — the same step as the framework runs it by default on one GPU (--hip_graph True): recorded once in a HIP graph
    # (both streams of the step) and replayed.  This is the headline when it exists; events cannot be placed inside a replay, so the
    # roofline keeps region A's event times (same kernels, same shapes; rocprofv3 of this command covers both regions).
    dt, launch_mode, graph_err = dt_eager, "eager", None
    gs = getattr(tr, "graphed", None)
    bail = None
    if gs is not None and world > 1:
        # Default (round 4): segmented replay, no collective inside a graph; CSLGAN_GRAPH_DIST=1 records the RCCL collectives with the step
        # instead (opt-in: csl_gan_amd.distributed.collectives_capturable); CSLGAN_GRAPH_SEGMENTS=0 keeps N > 1 eager.  If recording or
        # replaying does not come back, rank 0 still prints the eager measurement, labelled — and the
        # process ends with a NON-ZERO status (3): a hang is a failure of the run, not a result.
        import threading
        limit = float(os.environ.get("CSLGAN_GRAPH_REGION_LIMIT_S", "240"))

        def _bail():
            try:
                finish(dt_eager, "eager", "the HIP-graph region did not finish within %.0f s on %d ranks; the eager region is reported" % (limit, world),
                       None, None, with_cpu=False)
                sys.stdout.flush()
            finally:
                os._exit(3)
        bail = threading.Timer(limit, _bail)
        bail.daemon = True
        bail.start()
    if gs is not None:              # N > 1: every rank records the same step, RCCL all-reduce included (trainer.setup_privacy_engine)
        try:
            for _ in range(gs.warmup + 2):
                gs(img, None)
            if gs.graph is not None:
                # N > 1 by default: "hip_graph_segments" — the step replays as graphs that end at each collective, the collectives
                # themselves issued eagerly between them (GraphedDStep._capture_segments); nothing of RCCL is inside a graph
                dt, launch_mode = time_region(lambda: gs(img, None), a.steps), ("hip_graph_segments" if getattr(gs, "segmented", False) else "hip_graph")
            else:
                graph_err = gs.capture_error or "not captured"
        except Exception as e:      # a failed capture must not cost the line: the eager region stands
            graph_err = repr(e)[:200]
    if bail is not None:
        bail.cancel()
    variant = None
    if not a.opt and world == 1 and not a.no_variants:
        variant = {}
        if launch_mode == "hip_graph":
            variant["eager"] = {"value": round(world * B * a.steps / dt_eager, 2), "unit": "images/sec", "ms_per_step": round(dt_eager / a.steps * 1e3, 3),
                                "what": "the identical step launched kernel by kernel from Python (--hip_graph False); the roofline events were taken here"}
        elif graph_err:
            variant["hip_graph"] = {"error": graph_err}
        # the same step on the OTHER fp32-accurate arithmetic — the exact fp32 MFMA kernels everywhere when the headline is fp32_auto (that
        # was the headline of rounds 1-3), fp32_auto when the headline was asked to be exact — reported beside the headline, never as it
        cur = getattr(opt, "compute_dtype", "fp32")
        other = "fp32" if cur == "fp32_auto" else "fp32_auto"
        saved_explicit, tr.explicit = tr.explicit, {}            # the eager step draws its own mean-sample batches
        ops.set_compute_dtype(other)
        ops.repack_cache.clear()
        for _ in range(2):
            step()
        dv = time_region(step, a.steps)
        dvg = None
        if launch_mode == "hip_graph":
            try:
                from csl_gan_amd.trainer import GraphedDStep
                gs2 = GraphedDStep(tr, warmup=1)
                for _ in range(3):
                    gs2(img, None)
                dvg = time_region(lambda: gs2(img, None), a.steps)
            except Exception:
                dvg = None
            finally:
                try:
                    gs2.release()
                except NameError:
                    pass
        ops.set_compute_dtype(cur)
        tr.explicit = saved_explicit
        ops.repack_cache.clear()
        best = dv if dvg is None else min(dv, dvg)
        variant["fp32_exact" if other == "fp32" else "fp32_auto"] = {
            "value": round(world * B * a.steps / best, 2), "unit": "images/sec", "ms_per_step": round(best / a.steps * 1e3, 3),
            "ms_per_step_eager": round(dv / a.steps * 1e3, 3), "ms_per_step_hip_graph": None if dvg is None else round(dvg / a.steps * 1e3, 3),
            "what": ("--compute_dtype fp32: every conv / linear / weight-gradient launch on the exact fp32 MFMA kernels "
                     "(v_mfma_f32_32x32x2_f32, 157.3 TF peak) — the headline arithmetic of rounds 1-3" if other == "fp32" else
                     "--compute_dtype fp32_auto: large forward / data-gradient launches run fp32 emulated from three bfloat16 pieces per "
                     "operand (six bf16 MFMAs per product step; error vs fp64 <= the exact-fp32 kernels', "
                     "tests/test_kernels_gpu.py::test_bf16x3_*), everything else the exact fp32 MFMA kernels")}
        # BASELINE.json configs[4] beside the headline (its own trainer: 128x128 extension, bf16 matrix cores, bf16-STORED activations —
        # csrc/igemm_bf16s.hip), so that the driver's run carries a measurement of it too; never the headline value
        try:
            with contextlib.redirect_stdout(sys.stderr):
                opt16, tr16, img16 = build_trainer(rank, world, local, extra=["--compute_dtype", "bf16", "--storage_dtype", "bf16", "--im_size", "128"])
            for _ in range(2):
                tr16.train_D(img16, None, tr16.gen_z(B), None, use_dp=True)
            d16 = time_region(lambda: tr16.train_D(img16, None, tr16.gen_z(B), None, use_dp=True), max(a.steps // 2, 5))
            n16, mode16 = max(a.steps // 2, 5), "eager"
            g16 = getattr(tr16, "graphed", None)
            if g16 is not None:
                for _ in range(g16.warmup + 2):
                    g16(img16, None)
                if g16.graph is not None:
                    d16, mode16 = time_region(lambda: g16(img16, None), n16), "hip_graph"
            variant["bf16_storage_128x128"] = {
                "value": round(world * B * n16 / d16, 2), "unit": "images/sec", "ms_per_step": round(d16 / n16 * 1e3, 3), "dtype": "bf16", "launch": mode16,
                "what": "BASELINE configs[4]: the same D-step at 3x128x128, bs=%d per GPU, --compute_dtype bf16 --storage_dtype bf16 (bfloat16 "
                        "activations / activation gradients / filter copies in HBM, fp32 accumulate, fp32 weight gradients, norms, clip, noise, Adam); "
                        "per-kernel figures: python bench.py --opt \"--compute_dtype bf16 --storage_dtype bf16 --im_size 128\"" % B}
            del tr16, img16, g16
        except Exception as e:      # the variant must not cost the headline line
            variant["bf16_storage_128x128"] = {"error": repr(e)[:200]}
        finally:
            ops.set_compute_dtype(getattr(opt, "compute_dtype", "fp32"))
            ops.set_storage_dtype(getattr(opt, "storage_dtype", "fp32"))
            ops.repack_cache.clear()
            torch.cuda.empty_cache()
    # secondary metric: the full train() loop, a G step forced on every n_d_steps-th iteration
    loop = None
    if a.loop_steps > 0:
        opt.train_d_until_threshold = float("inf")
        lbl = torch.zeros(B, dtype=torch.long)
        for i in range(opt.n_d_steps):
            tr.train(0, i, img, lbl, use_dp=True)
        dl = time_region(lambda it=iter(range(10 ** 9)): tr.train(0, next(it), img, lbl, use_dp=True), a.loop_steps)
        loop = {"metric": "full train() loop, G step every %d iterations" % opt.n_d_steps, "value": round(world * B * a.loop_steps / dl, 2),
                "unit": "images/sec", "iterations": a.loop_steps, "ms_per_iteration": round(dl / a.loop_steps * 1e3, 3)}

    finish(dt, launch_mode, graph_err, variant, loop)


if __name__ == "__main__":
    main()
