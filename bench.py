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


def time_graphed(fn, iters, torch):
    """Average duration (ms) of a multi-launch sequence `fn`, replayed as a hipGraph (how the train step runs it: no host
    launch gaps between its kernels), HIP events around `iters` replays."""
    fn()
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def in_step_roofline(roof, timer):
    """Adds the kernel's average launch duration INSIDE train steps (HIP events around its launches on the launch stream,
    ops.LaunchTimer, taken over a few EAGER steps run right after the timed region -- the timed region itself is replayed
    as one hipGraph launch, inside which no event can be read) as `ms_per_launch_in_step` / `frac_in_step`.
    `ms_per_launch` / `achieved` / `frac` stay on the isolated, back-to-back launches (round 3, ADVICE r2: like-for-like
    with round 1 and with the rocprofv3 `--roofline-only` statistics under profiles/)."""
    ms = timer.mean_ms()
    if ms is None:
        return roof
    k = roof["ms_per_launch"] / ms
    roof["ms_per_launch_in_step"] = round(ms, 4)
    roof["launches_timed_in_step"] = len(timer.pairs)
    roof["achieved_in_step"] = round(roof["achieved"] * k, 2)
    roof["frac_in_step"] = round(roof["achieved_in_step"] / roof["peak"], 4)
    return roof


def measured_traffic(key="traffic_bytes_per_launch"):
    """(HBM bytes per launch of the dominant kernel, file it came from): the committed PMC passes (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in their own runs, gfx950 half-count correction applied) -- the newest
    profiles/rNN_pmc_traffic.json that holds `key`.  NOT measured in this run (PMC needs the profiler)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            v = json.load(open(path)).get(key)
        except (OSError, ValueError):
            continue
        if v is not None:
            return v, os.path.relpath(path, ROOT)
    return None, None


def bf16_block_roofline(torch, dev, patch):
    """bf16 mixed precision (BASELINE configs[3], [4]): the fused conv block of the north_star's HBM target -- Conv3d
    32 -> 32 (3x3x3) + InstanceNorm3d + LeakyReLU at the full patch, batch 2, bf16 activations -- timed as the network
    runs it: network.ConvDropoutNormReLU.forward on a resident bf16 NDHWC activation (the conv launch plus EVERY
    normalisation launch the block needs: statistics / finalize / apply, whatever the current build fuses).
    ALGORITHMIC bytes = (C_in*N_in + C_out*N_out) * 2 B: read the producer's output once, write the raw conv
    output once (that byte model ASSUMES the normalisation is fused away; un-fused passes show up as a lower fraction).
    `achieved` / `frac` = those bytes / the BLOCK's time (round 3).  The conv launch alone (what round
    2 printed as `frac`) is under `conv_only`; the MFMA view is given beside it."""
    from multimodal_mvd_seg_amd import network, ops
    from multimodal_mvd_seg_amd._lib import call, i3, query
    from torch import nn
    import ctypes
    N, C, K = PER_GPU_BATCH, 32, 32
    D, H, W = patch
    x = ops.empty_cl3d((N, C, D, H, W), dev, torch.bfloat16).normal_()
    blk = network.ConvDropoutNormReLU(nn.Conv3d, C, K, 3, 1, True, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, None,
                                      None, nn.LeakyReLU, {'inplace': True}).to(dev)
    blk.precision = "bf16"
    with torch.no_grad():
        blk.conv.weight.normal_(0, 0.05)
        blk.norm.weight.uniform_(0.5, 1.5)
        blk.norm.bias.normal_(0, 0.1)
    wf, _ = ops.pack_weight_bf16(blk.conv.weight, False)
    y = ops.empty_cl3d((N, K, D, H, W), dev, torch.bfloat16)
    ws = torch.empty(max(1, query("mvd_conv_fwd_workspace_bytes", N, D * H * W, K)), dtype=torch.uint8, device=dev)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def conv():
        call("mvd_conv3d_fwd_bf16", P(x), C, None, 0, P(wf), P(blk.conv.bias), P(y), N, D, H, W, K, i3((3, 3, 3)),
             i3((1, 1, 1)), P(ws), ws.numel(), s)

    blk2 = network.ConvDropoutNormReLU(nn.Conv3d, K, K, 3, 1, True, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, None,
                                       None, nn.LeakyReLU, {'inplace': True}).to(dev)
    blk2.precision = "bf16"
    with torch.no_grad():
        blk2.conv.weight.normal_(0, 0.05)

    def block():      # the un-fused form: conv (+ statistics epilogue) -> finalize -> apply pass
        with torch.no_grad():
            blk.norm_act(blk.conv_only(x))

    with torch.no_grad():
        y_raw = blk.conv_only(x)
    fused_ok = ops.fused_norm_conv_ok(y_raw, blk2.conv.weight, blk2.stride)

    def block_fused():  # the fused form (inference; training where k_wgrad16z serves the shape): finalize + this conv with the
        with torch.no_grad():  # InstanceNorm-apply + LeakyReLU in its loader and the statistics in its epilogue
            blk2.forward_from_raw(blk, y_raw)
    ms = time_kernel(conv, 10, torch)
    ums = time_graphed(block, 10, torch)
    fms = time_graphed(block_fused, 10, torch) if fused_ok else None
    # which form the TRAIN step runs for this block (a 32 -> 32 conv fed by a 32-channel conv: stage 0 / the top decoder stage):
    # the fused one when the weight-gradient kernel has the loader prologue too (network.StackedConvBlocks._can_fuse, "auto")
    with torch.enable_grad():
        train_fused = fms is not None and network.StackedConvBlocks._can_fuse(blk, blk2, x)
    bms = fms if train_fused else ums
    V = float(N * D * H * W)
    alg_bytes = (C + K) * V * 2.0
    flops = 2.0 * 27 * C * K * V
    gbs = alg_bytes / (bms * 1e-3) / 1e9
    cgbs = alg_bytes / (ms * 1e-3) / 1e9
    traffic, tsrc = measured_traffic("traffic_bytes_per_launch_bf16")
    fused_what = ("the producing block's finalize launch (per-workgroup sums -> scale / shift) + ONE conv launch on "
                  "v_mfma_f32_16x16x32_bf16 (k_fwd16y: z-marching 8x32 columns, weights resident in accumulator registers) with "
                  "the producing block's InstanceNorm-apply + LeakyReLU in its loader and its own statistics in the epilogue; the "
                  "activated tensor is never written (backward: the weight-gradient kernel k_wgrad16z applies the same prologue)")
    roof = {"kernel": f"fused block Conv3d 32->32 3x3x3 + InstanceNorm3d + LeakyReLU @{'x'.join(map(str, patch))} bf16, "
                      "batch 2, as the train step runs it: " +
                      (fused_what if train_fused else "conv with the statistics epilogue (k_fwd16y) + the finalize launch + the "
                                                      "apply pass"),
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "algorithmic_GB_per_launch": round(alg_bytes / 1e9, 4),
            "block_ms": round(bms, 4), "ms_per_launch": round(bms, 4), "train_step_form": "fused" if train_fused else "unfused",
            "traffic": traffic, "traffic_source": f"{tsrc} (conv launch only, committed PMC pass, not measured in this run)",
            "conv_only": {"ms_per_launch": round(ms, 4), "achieved": round(cgbs, 1), "frac": round(cgbs / HBM_PEAK_GBS, 4),
                          "note": "the plain conv launch alone against the block's byte model (round 2's `frac`)"},
            "instnorm_ms": round(bms - ms, 4),
            "mfma_view": {"tflops": round(flops / (ms * 1e-3) / 1e12, 1), "peak": BF16_MFMA_PEAK_TFLOPS,
                          "frac": round(flops / (ms * 1e-3) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 4), "of": "conv launch"}}
    ug = alg_bytes / (ums * 1e-3) / 1e9
    roof["unfused_form"] = {"block_ms": round(ums, 4), "achieved": round(ug, 1), "frac": round(ug / HBM_PEAK_GBS, 4),
                            "what": "conv with the statistics epilogue -> finalize launch -> apply pass writing the activated "
                                    "tensor: what blocks run whose consumer is not a z-marching 32-channel conv (and every "
                                    "block with MVD_FUSE_PROLOGUE_TRAIN=0)"}
    if fms is not None:
        fg = alg_bytes / (fms * 1e-3) / 1e9
        roof["inference_form"] = {"block_ms": round(fms, 4), "achieved": round(fg, 1), "frac": round(fg / HBM_PEAK_GBS, 4),
                                  "what": "the no-grad forward (sliding-window inference): " + fused_what}
    g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)

    def norm():
        ops.InstanceNormLeakyReLUFn.apply(y, g, b, 1e-5, 0.01)
    nms = time_kernel(norm, 10, torch)
    nbytes = 3.0 * K * V * 2
    ngb = nbytes / (nms * 1e-3) / 1e9
    roof["instnorm_lrelu_fwd"] = {"bound": "hbm", "achieved": round(ngb, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ngb / HBM_PEAK_GBS, 4), "ms_per_launch": round(nms, 4),
                                  "algorithmic_bytes": "3*C*N*2 (bf16 in, bf16 out): the stand-alone two-pass form"}
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
    time_kernel(conv, 3, torch)          # clocks / caches settle
    ms = time_kernel(conv, 10, torch)
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
            "traffic": measured_traffic()[0],
            "traffic_source": f"{measured_traffic()[1]} (committed PMC pass, not measured in this run)",
            "ms_per_launch": round(ms, 3),
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
    run_training.py:391-395: all usable host threads, no autocast) on a BOUNDED sample of the workload (>= 3
    timed steps at configs[1], and configs[0] = the reference's own CPU case in full).
    configs[1]: one batch-1 step on a 64^3 sub-patch first (warms oneDNN up and predicts the cost: every layer is
    convolutional, so time scales with voxels); if that predicts <= 12 s per full step, THREE batch-1 steps at the full
    4x128^3 patch are timed (value = 1 / median), otherwise three more sub-patch steps, scaled by 1/8.
    configs[0] (1 modality, 64^3, batch 2, five stages): one warm-up + three timed steps, reported under `cfg1`."""
    from oracle import loss_oracle as LO, step_oracle as SO, unet_oracle as UO
    import statistics
    cores = usable_cores()
    torch.set_num_threads(cores)

    def stepper(in_ch, n_stages, strides, sub, batch_size):
        net = UO.build_plainconv_unet(in_ch, NUM_CLASSES, n_stages, strides, seed=0)
        batch = SO.synthetic_batch(batch_size, in_ch, sub, strides, num_classes=NUM_CLASSES, seed=1234)
        loss_fn = LO.build_loss(len(batch["target"]))
        opt = SO.make_optimizer(net.parameters())

        def one():
            t0 = time.perf_counter()
            SO.train_step(net, loss_fn, opt, batch)
            return time.perf_counter() - t0
        return one

    sub = (64, 64, 64)
    dt_sub = stepper(IN_CH, 6, STRIDES, sub, 1)()
    frac = (sub[0] * sub[1] * sub[2]) / float(PATCH[0] * PATCH[1] * PATCH[2])
    if dt_sub / frac <= 12.0:
        one = stepper(IN_CH, 6, STRIDES, PATCH, 1)
        ts = [one() for _ in range(3)]
        dt = statistics.median(ts)
        out = {"value": round(1.0 / dt, 5), "unit": "samples/s", "cores": cores, "kind": "port",
               "sample": f"3 train steps (fwd+loss+bwd+clip+SGD), batch 1, the full 4x128^3 patch, torch "
                         f"{torch.__version__} CPU fp32, {cores} threads: {', '.join(f'{t:.1f}' for t in ts)} s wall, value = "
                         f"1 / median (after a {dt_sub:.1f} s batch-1 step on a 64^3 sub-patch that warmed oneDNN up)"}
    else:
        one = stepper(IN_CH, 6, STRIDES, sub, 1)
        ts = [one() for _ in range(3)]
        dt = statistics.median(ts)
        out = {"value": round(frac / dt, 5), "unit": "samples/s", "cores": cores, "kind": "port",
               "sample": f"3 train steps (fwd+loss+bwd+clip+SGD), batch 1, 4x64^3 sub-patch of the 4x128^3 workload, torch "
                         f"{torch.__version__} CPU fp32, {cores} threads: {', '.join(f'{t:.1f}' for t in ts)} s wall; value = "
                         f"(64^3/128^3) / median"}
    # BASELINE configs[0]: the reference's own CPU-runnable case, in full
    c1 = UO.CONFIGS["cfg1"]
    one = stepper(c1["input_channels"], c1["n_stages"], c1["strides"], c1["patch"], 2)
    one()
    ts = [one() for _ in range(3)]
    out["cfg1"] = {"value": round(2.0 / statistics.median(ts), 4), "unit": "samples/s", "cores": cores,
                   "sample": f"BASELINE configs[0]: 1 modality, 64^3, batch 2, 5 stages (16.55 M parameters): 3 timed steps "
                             f"after one warm-up: {', '.join(f'{t:.2f}' for t in ts)} s wall, value = 2 / median"}
    return out


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
    ap.add_argument("--no-secondary", action="store_true", help="skip the bf16 steps a default (configs[1], fp32) run appends")
    ap.add_argument("--no-graph", action="store_true", help="A/B aid: eager steps (one launch per kernel) instead of the "
                                                            "hipGraph replay")
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
    def build_trainer(config, precision):
        plans = trainer.make_plans(patch, STRIDES, batch_size=PER_GPU_BATCH * world)
        if config in ("cfg2", "cfg5"):
            t = trainer.nnUNetTrainerMI355Benchmark_noDataLoading(plans, "3d_fullres", 0, dataset_json(), device=dev)
        else:
            t = trainer.ContrastiveTrainerMI355(plans, "3d_fullres", 0, dataset_json(), device=dev)
            t.use_topo = config == "cfg4"
        if args.no_graph:
            t.use_hip_graph = False
        torch.manual_seed(0)
        t.precision = precision
        t.initialize()
        if config in ("cfg3", "cfg4"):
            t.dummy_batch = t.make_dummy_batch()
        assert t.batch_size == PER_GPU_BATCH
        t.on_train_epoch_start()
        return t

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(t, warmup, steps):
        """`warmup` untimed steps (the first ones eager, then the capture of the step as a hipGraph and its first
        replays), then EXACTLY `steps` timed steps between barrier + synchronize; returns (seconds, last step's output)."""
        batch = t.dummy_batch  # resident in HBM
        t.hip_graph_warmup = min(t.hip_graph_warmup, max(1, warmup - 1))  # capture inside the warm-up whenever W >= 2
        for _ in range(warmup):
            t.train_step(batch)
        sync()
        t0 = time.perf_counter()
        last = None
        for _ in range(steps):
            last = t.train_step(batch)
        sync()
        return time.perf_counter() - t0, last

    tr = build_trainer(args.config, args.precision)
    dt, last = timed_steps(tr, args.warmup, args.steps)
    graphed = tr._step_graph is not None and tr._step_graph.get("graph") is not None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the roofline kernel's launches INSIDE train steps, bracketed by HIP events on the stream they run on (two event
    # records per launch): fp32 = decoder stage 5 conv 0 forward (64 -> 32 at the patch), bf16 = the 32 -> 32 convs at the
    # patch (encoder stage 0 conv 1, decoder stage 5 conv 1).  Events cannot be read inside a graph replay, so these are a
    # few EAGER steps after the timed region (same kernels, same order).
    from multimodal_mvd_seg_amd import ops as _ops
    bf = args.precision == "bf16"
    timer = _ops.LaunchTimer((bf, PER_GPU_BATCH, 32, 0 if bf else 32, 32, *patch, (3, 3, 3), (1, 1, 1)))
    if not args.no_roofline and rank == 0 and world == 1:
        was = tr.use_hip_graph
        tr.use_hip_graph = False
        _ops.LAUNCH_TIMER = timer
        tr.train_step(tr.dummy_batch)
        timer.on = True
        for _ in range(5):
            tr.train_step(tr.dummy_batch)
        torch.cuda.synchronize()
        timer.on = False
        _ops.LAUNCH_TIMER = None
        tr.use_hip_graph = was

    out = None
    if rank == 0:
        samples = PER_GPU_BATCH * world * args.steps
        out = {"metric": metric_name(args.config, args.precision, patch), "value": round(samples / dt, 4), "unit": "samples/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.precision == "fp32" else "bf16",
               "data": "synthetic",
               "config": {"workload": {"cfg2": "BASELINE configs[1]", "cfg3": "BASELINE configs[2] (dual-branch MVD step)",
                                       "cfg4": "BASELINE configs[3] (dual-branch MVD step + soft-clDice + component count)",
                                       "cfg5": "BASELINE configs[4]"}[args.config] +
                                      ": PlainConvUNet 3d_fullres 6 stages 31.2M params, "
                                      f"{IN_CH}x{'x'.join(map(str, patch))} patch, {NUM_CLASSES} classes, "
                                      f"{'fp32' if args.precision == 'fp32' else 'bf16 mixed precision'}, "
                                      "DC+CE deep supervision, SGD-Nesterov+clip",
                          "per_gpu_batch": PER_GPU_BATCH, "global_batch": PER_GPU_BATCH * world,
                          "parallelism": f"dp{world}",
                          "step_launch": "one hipGraph replay per step (captured after 3 eager steps)" if graphed
                                         else "eager (one launch per kernel)"},
               "final_loss": float(last["loss"])}
        if not args.no_roofline:
            out["roofline"] = in_step_roofline(roofline(), timer)
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(torch)
    if rank == 0 and world == 1 and args.config == "cfg2" and args.precision == "fp32" and not args.no_secondary \
            and tuple(patch) == PATCH:
        # the same workload in bf16 mixed precision (the configuration north_star's 40 %-of-HBM block target is about):
        # 10 warm-up + 20 timed steps, reported beside the fp32 line, never in its `value`
        del tr
        torch.cuda.empty_cache()
        t16 = build_trainer("cfg2", "bf16")
        dt16, last16 = timed_steps(t16, 10, 20)
        sec = {"value": round(PER_GPU_BATCH * 20 / dt16, 3), "unit": "samples/s", "ms_per_step": round(dt16 / 20 * 1e3, 3),
               "steps": 20, "warmup": 10, "dtype": "bf16", "final_loss": float(last16["loss"]),
               "workload": "BASELINE configs[1] shape in bf16 mixed precision (bf16 activations on the bf16 MFMA engine, "
                           "fp32 master weights / statistics / losses / optimizer)"}
        if not args.no_roofline:
            sec["roofline"] = bf16_block_roofline(torch, dev, patch)
        out["secondary"] = {"bf16": sec}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
