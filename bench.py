#!/usr/bin/env python3
"""bench.py -- train-step throughput of the hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = nnUNetTrainer.train_step (forward, deep-supervised DC+CE loss, backward, gradient all-reduce, global-norm
clip, SGD-Nesterov) on the reference's own synthetic benchmark batch
(nnUNetTrainerBenchmark_5epochs_noDataLoading.py:16-22).  Workload = BASELINE.json configs[1]: PlainConvUNet
3d_fullres, 6 stages, 31.2 M parameters, 4 modalities, 128^3 patch, fp32, batch 2 per GPU (weak scaling: the global
batch is 2*N, split by the reference's rule nnUNetTrainer.py:304-349).  Inputs are resident in HBM before the timed
region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def metric_name(config, precision, patch):
    """BASELINE.json's metric, qualified by what was actually run (dtype and configuration follow the flags)."""
    what = {"cfg2": "PlainConvUNet 3d_fullres", "cfg3": "PlainConvUNet 3d_fullres, mutual-distillation dual branch",
            "cfg4": "PlainConvUNet 3d_fullres, mutual-distillation dual branch + soft-clDice topology term",
            "cfg5": "PlainConvUNet 3d_fullres"}[config]
    size = "128^3" if tuple(patch) == PATCH else "x".join(map(str, patch))
    return f"train-step samples/sec on 4-modality {size} patches ({what}, {'fp32' if precision == 'fp32' else 'bf16 mixed precision'})"


PATCH = (128, 128, 128)
STRIDES = [[1, 1, 1]] + [[2, 2, 2]] * 5
IN_CH, NUM_CLASSES, PER_GPU_BATCH = 4, 5, 2
FP32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 = fp32 vector rate
BF16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PFLOP/s dense bf16
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def dataset_json():
    return {"channel_names": {str(i): f"mod{i}" for i in range(IN_CH)},
            "labels": {"background": 0, **{f"c{i}": i for i in range(1, NUM_CLASSES)}}}


def time_kernel(fn, iters, torch):
    """Average duration (ms) of `fn` over `iters` launches, HIP events on the stream the kernels run on."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def in_step_roofline(roof, timer):
    """`achieved` / `frac` from the kernel's average launch duration INSIDE the timed train steps (HIP events around its
    launches, ops.LaunchTimer); the duration of the same launch repeated back to back after the steps stays beside it as
    `ms_per_launch_isolated` (it reads 10-15 % longer for the fp32 kernel -- the chip holds a lower clock under an
    unbroken fp32-MFMA load than under the step's mix -- and ~8 % longer for the bf16 kernel, whose input the step's
    previous kernel has just written)."""
    ms = timer.mean_ms()
    if ms is None:
        return roof
    iso = roof["ms_per_launch"]
    k = iso / ms
    roof["ms_per_launch_isolated"] = iso
    roof["frac_isolated"] = roof["frac"]
    roof["ms_per_launch"] = round(ms, 4)
    roof["launches_timed_in_step"] = len(timer.pairs)
    roof["achieved"] = round(roof["achieved"] * k, 2)
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    if "effective_tflops" in roof:  # fp32 view
        roof["effective_tflops"] = round(roof["effective_tflops"] * k, 2)
        roof["effective_over_peak"] = round(roof["effective_tflops"] / roof["peak"], 4)
        roof["hbm_view"]["achieved_GBps"] = round(roof["hbm_view"]["achieved_GBps"] * k, 1)
        roof["hbm_view"]["frac_of_hbm_peak"] = round(roof["hbm_view"]["achieved_GBps"] / HBM_PEAK_GBS, 4)
    if "mfma_view" in roof:  # bf16 view
        roof["mfma_view"]["tflops"] = round(roof["mfma_view"]["tflops"] * k, 1)
        roof["mfma_view"]["frac"] = round(roof["mfma_view"]["tflops"] / roof["mfma_view"]["peak"], 4)
    return roof


def measured_traffic(key="traffic_bytes_per_launch"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in their own runs, gfx950 half-count correction applied): the newest profiles/rNN_pmc_traffic.json
    that holds `key`."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            v = json.load(open(path)).get(key)
        except (OSError, ValueError):
            continue
        if v is not None:
            return v
    return None


def bf16_block_roofline(torch, dev, patch):
    """bf16 mixed precision (BASELINE configs[3], [4]): the fused conv block of the north_star's HBM target, conv3d
    32 -> 32 at the full patch, batch 2, bf16 activations, through the entry point ops.Conv3dFn calls.
    ALGORITHMIC bytes = (C_in*N_in + C_out*N_out) * 2 B: read the producer's output once, write the raw
    conv output once.  `bound` = hbm (block arithmetic intensity 432 flop/B vs a ridge of ~312); the MFMA view is given
    beside it."""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call, i3, query
    import ctypes
    N, C, K = PER_GPU_BATCH, 32, 32
    D, H, W = patch
    x = ops.empty_cl3d((N, C, D, H, W), dev, torch.bfloat16).normal_()
    w = torch.randn(K, C, 3, 3, 3, device=dev) * 0.05
    bias = torch.zeros(K, device=dev)
    wf, _ = ops.pack_weight_bf16(w, False)
    y = ops.empty_cl3d((N, K, D, H, W), dev, torch.bfloat16)
    ws = torch.empty(max(1, query("mvd_conv_fwd_workspace_bytes", N, D * H * W, K)), dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def conv():
        call("mvd_conv3d_fwd_bf16", P(x), C, None, 0, P(wf), P(bias), P(y), N, D, H, W, K, i3((3, 3, 3)), i3((1, 1, 1)),
             P(ws), ws.numel(), s)
    ms = time_kernel(conv, 10, torch)
    V = float(N * D * H * W)
    alg_bytes = (C + K) * V * 2.0
    flops = 2.0 * 27 * C * K * V
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    roof = {"kernel": f"conv3d_fwd 32->32 @{'x'.join(map(str, patch))} bf16 (fwd-type implicit GEMM on "
                      "v_mfma_f32_32x32x16_bf16, z-marching 8x32 columns, weights resident in accumulator registers: k_fwd16z)",
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_GB_per_launch": round(alg_bytes / 1e9, 4),
            "traffic": measured_traffic("traffic_bytes_per_launch_bf16"), "ms_per_launch": round(ms, 4),
            "mfma_view": {"tflops": round(flops / (ms * 1e-3) / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                          "frac": round(flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4)}}
    g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)

    def norm():
        ops.InstanceNormLeakyReLUFn.apply(y, g, b, 1e-5, 0.01)
    nms = time_kernel(norm, 10, torch)
    nbytes = 3.0 * K * V * 2
    ngb = nbytes / (nms * 1e-3) / 1e9
    roof["instnorm_lrelu_fwd"] = {"bound": "hbm", "achieved": round(ngb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ngb / HBM_PEAK_GBS, 4), "ms_per_launch": round(nms, 4),
                                  "algorithmic_bytes": "3*C*N*2 (bf16 in, bf16 out)"}
    return roof


def dominant_kernel_roofline(torch, dev):
    """Times the dominant kernel of the step in isolation, live, with HIP events: the 3x3x3 conv forward-type
    implicit GEMM on the most expensive layer (decoder stage 5 conv 0: 64 -> 32 channels at 128^3, two input
    pointers = the eliminated torch.cat), batch 2, through the same entry point ops.Conv3dFn calls.
    ALGORITHMIC flops = 2*27*C_in*C_out*N_out: the direct-convolution count.  The kernel is the Winograd
    F(2x2,3x3) engine, which EXECUTES 4/9 of them on the MFMA pipe (12 instead of 27 multiply-adds per output and
    channel pair), so `achieved` (algorithmic flops / time) can exceed the pipe's peak; `executed_tflops` /
    `mfma_frac_executed` state what the pipe actually did."""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call, i3
    import ctypes
    N, C1, C2, K = PER_GPU_BATCH, 32, 32, 32
    D, H, W = PATCH
    x1 = ops.empty_cl3d((N, C1, D, H, W), dev).normal_()
    x2 = ops.empty_cl3d((N, C2, D, H, W), dev).normal_()
    w = torch.randn(K, C1 + C2, 3, 3, 3, device=dev) * 0.03
    bias = torch.zeros(K, device=dev)
    wf, _ = ops.pack_weight(w, False)
    y = ops.empty_cl3d((N, K, D, H, W), dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    from multimodal_mvd_seg_amd._lib import query
    wino = query("mvd_conv_wino_applicable", N, D, H, W, C1, C2, K, i3((3, 3, 3)), i3((1, 1, 1))) & 1
    uf = None
    if wino:
        uf = torch.empty(query("mvd_wino_weight_elems", C1 + C2, K), device=dev)
        call("mvd_pack_weight_wino", P(w), P(uf), None, K, C1 + C2, s)

    def conv():
        call("mvd_conv3d_fwd_wino", P(x1), C1, P(x2), C2, P(wf), P(uf) if wino else None, P(bias), P(y), N, D, H, W, K,
             i3((3, 3, 3)), i3((1, 1, 1)), None, 0, s)
    ms = time_kernel(conv, 3, torch)
    flops = 2.0 * 27 * (C1 + C2) * K * N * D * H * W
    conv_tf = flops / (ms * 1e-3) / 1e12
    mode = query("mvd_wino_mode") if wino else 0
    exec_ratio = {0: 1.0, 1: 2.0 / 3.0, 2: 4.0 / 9.0}[mode]
    exec_tf = conv_tf * exec_ratio
    alg_bytes = ((C1 + C2) + K) * N * D * H * W * 4.0
    # `achieved` / `frac` price what the MFMA pipe EXECUTED (a fraction of a roofline cannot exceed 1); the direct-conv
    # (algorithmic) rate the Winograd kernel delivers is reported beside it as `effective_tflops`
    roof = {"kernel": "conv3d_fwd 64->32 @128^3 (fwd-type implicit GEMM, " +
                      {0: "direct: k_fwd32)", 1: "Winograd F(2,3) along W: k_fwd_wino)",
                       2: "Winograd F(2x2,3x3) over H,W: k_fwd_wino2)"}[mode],
            "bound": "mfma", "achieved": round(exec_tf, 2),
            "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(exec_tf / FP32_MFMA_PEAK_TFLOPS, 4),
            "executed_flop_ratio": round(exec_ratio, 4), "effective_tflops": round(conv_tf, 2),
            "effective_over_peak": round(conv_tf / FP32_MFMA_PEAK_TFLOPS, 4),
            "algorithmic_gflop_per_launch": round(flops / 1e9, 1),
            "traffic": measured_traffic(), "ms_per_launch": round(ms, 3),
            "hbm_view": {"algorithmic_GB": round(alg_bytes / 1e9, 3),
                         "achieved_GBps": round(alg_bytes / (ms * 1e-3) / 1e9, 1),
                         "frac_of_hbm_peak": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
    # the HBM-bound kernel of the fused block: InstanceNorm+LeakyReLU forward at [2,32,128^3]
    # algorithmic bytes = 3*C*N*4 (read x for the statistics, read x, write y)
    g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)

    def norm():
        ops.InstanceNormLeakyReLUFn.apply(y, g, b, 1e-5, 0.01)
    nms = time_kernel(norm, 5, torch)
    nbytes = 3.0 * K * N * D * H * W * 4
    ngb = nbytes / (nms * 1e-3) / 1e9
    roof["instnorm_lrelu_fwd"] = {"bound": "hbm", "achieved": round(ngb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ngb / HBM_PEAK_GBS, 4), "ms_per_launch": round(nms, 3)}
    return roof


def usable_cores():
    """Host cores this process may actually use: CPU affinity capped by the cgroup quota (the GPU box exposes all
    host CPUs to os.cpu_count() but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def cpu_baseline(torch):
    """The oracle's torch-CPU train step (same ops, same order as the reference's `-device cpu` path,
    run_training.py:391-395: all usable host threads, no autocast) on a BOUNDED sample of the workload (target: 10-30 s
    of CPU work).  A batch-1 step on a 64^3 sub-patch (1/8 of the voxels; every layer is convolutional, so cost scales
    with voxels) is timed first; if it predicts <= 40 s for the real 4x128^3 patch, ONE batch-1 step at the full patch is
    timed and reported, otherwise the sub-patch figure scaled by 1/8."""
    from oracle import loss_oracle as LO, step_oracle as SO, unet_oracle as UO
    cores = usable_cores()
    torch.set_num_threads(cores)

    def one_step(sub):
        net = UO.build_plainconv_unet(IN_CH, NUM_CLASSES, 6, STRIDES, seed=0)
        batch = SO.synthetic_batch(1, IN_CH, sub, STRIDES, num_classes=NUM_CLASSES, seed=1234)
        loss_fn = LO.build_loss(len(batch["target"]))
        opt = SO.make_optimizer(net.parameters())
        t0 = time.perf_counter()
        SO.train_step(net, loss_fn, opt, batch)
        return time.perf_counter() - t0

    sub = (64, 64, 64)
    dt_sub = one_step(sub)
    frac = (sub[0] * sub[1] * sub[2]) / float(PATCH[0] * PATCH[1] * PATCH[2])
    if dt_sub / frac <= 40.0:
        dt = one_step(PATCH)
        return {"value": round(1.0 / dt, 5), "unit": "samples/s", "cores": cores, "kind": "port",
                "sample": f"1 train step (fwd+loss+bwd+clip+SGD), batch 1, the full 4x128^3 patch, torch "
                          f"{torch.__version__} CPU fp32, {cores} threads: {dt:.1f} s wall (after a {dt_sub:.1f} s batch-1 "
                          f"step on a 64^3 sub-patch that also warmed oneDNN up)"}
    return {"value": round(frac / dt_sub, 5), "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": f"1 train step (fwd+loss+bwd+clip+SGD), batch 1, 4x64^3 sub-patch of the 4x128^3 workload, torch "
                      f"{torch.__version__} CPU fp32, {cores} threads: {dt_sub:.1f} s wall (first step, includes oneDNN "
                      f"primitive creation); value = (64^3/128^3) / wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--roofline-only", action="store_true",
                    help="profiling aid: only the isolated launches of the roofline kernels (rocprofv3 --kernel-trace then "
                         "shows the timed layer alone)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--patch", type=int, nargs=3, default=list(PATCH), help="debug only; the metric is quoted at 128^3")
    ap.add_argument("--backend", default="nccl", help="debug only: 'gloo' lets several ranks share one GPU")
    ap.add_argument("--config", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="BASELINE.json configs[1..4]: cfg2 = the metric's workload (default); cfg3 = + mutual-distillation "
                         "dual branch (KL on the vessel logits + feature KL); cfg4 = cfg3 + soft-clDice topology term, "
                         "bf16; cfg5 = single branch on the 160x160x128 patch, bf16")
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "bf16"],
                    help="auto = what BASELINE.json names for the config (fp32 for cfg2/cfg3, bf16 for cfg4/cfg5); bf16 = "
                         "bf16 activations on the bf16 MFMA engine, fp32 master weights/statistics/losses/optimizer")
    args = ap.parse_args()
    if args.precision == "auto":
        args.precision = "bf16" if args.config in ("cfg4", "cfg5") else "fp32"
    if args.config == "cfg5" and tuple(args.patch) == PATCH:
        args.patch = [160, 160, 128]

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)  # nccl == RCCL over xGMI on ROCm

    from multimodal_mvd_seg_amd import trainer
    patch = tuple(args.patch)

    def roofline():
        return bf16_block_roofline(torch, dev, patch) if args.precision == "bf16" else dominant_kernel_roofline(torch, dev)

    if args.roofline_only:
        for _ in range(3):
            r = roofline()
        print(json.dumps({"roofline": r}), flush=True)
        return
    plans = trainer.make_plans(patch, STRIDES, batch_size=PER_GPU_BATCH * world)
    if args.config in ("cfg2", "cfg5"):
        tr = trainer.nnUNetTrainerMI355Benchmark_noDataLoading(plans, "3d_fullres", 0, dataset_json(), device=dev)
    else:
        tr = trainer.ContrastiveTrainerMI355(plans, "3d_fullres", 0, dataset_json(), device=dev)
        tr.use_topo = args.config == "cfg4"
    torch.manual_seed(0)
    tr.precision = args.precision
    tr.initialize()
    if args.config in ("cfg3", "cfg4"):
        tr.dummy_batch = tr.make_dummy_batch()
    assert tr.batch_size == PER_GPU_BATCH
    tr.on_train_epoch_start()
    batch = tr.dummy_batch  # resident in HBM

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the roofline kernel's launches INSIDE the timed steps, bracketed by HIP events on the stream they run on (two event
    # records per launch): fp32 = decoder stage 5 conv 0 forward (64 -> 32 at the patch), bf16 = the 32 -> 32 convs at the
    # patch (encoder stage 0 conv 1, decoder stage 5 conv 1)
    from multimodal_mvd_seg_amd import ops as _ops
    bf = args.precision == "bf16"
    timer = _ops.LaunchTimer((bf, PER_GPU_BATCH, 32, 0 if bf else 32, 32, *patch, (3, 3, 3), (1, 1, 1)))
    _ops.LAUNCH_TIMER = timer
    for _ in range(args.warmup):
        tr.train_step(batch)
    sync()
    timer.on = not args.no_roofline
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = tr.train_step(batch)
    sync()
    dt = time.perf_counter() - t0
    timer.on = False
    _ops.LAUNCH_TIMER = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        samples = PER_GPU_BATCH * world * args.steps
        out = {"metric": metric_name(args.config, args.precision, patch), "value": round(samples / dt, 4), "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16",
               "data": "synthetic",
               "config": {"workload": {"cfg2": "BASELINE configs[1]", "cfg3": "BASELINE configs[2] (dual-branch MVD step)",
                                       "cfg4": "BASELINE configs[3] (dual-branch MVD step + soft-clDice)",
                                       "cfg5": "BASELINE configs[4]"}[args.config] +
                                      ": PlainConvUNet 3d_fullres 6 stages 31.2M params, "
                                      f"{IN_CH}x{'x'.join(map(str, patch))} patch, {NUM_CLASSES} classes, "
                                      f"{'fp32' if args.precision == 'fp32' else 'bf16 mixed precision'}, "
                                      "DC+CE deep supervision, SGD-Nesterov+clip",
                          "per_gpu_batch": PER_GPU_BATCH, "global_batch": PER_GPU_BATCH * world,
                          "parallelism": f"dp{world}"},
               "final_loss": float(last["loss"])}
        if not args.no_roofline:
            out["roofline"] = in_step_roofline(roofline(), timer)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(torch)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
