"""Timing of the fused bf16 block's pieces at the headline shape [2, 32, 128^3] (HIP events, back-to-back launches):
the z-marching conv plain / + statistics epilogue / + loader prologue / + both, the finalize launch, the apply pass in
scale-shift form, and the stand-alone InstanceNorm+LeakyReLU (statistics + finalize + apply) it replaces.
usage: python tools/bench_fused_block.py [--iters 20] [--patch 128 128 128]"""
import argparse
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_mvd_seg_amd import ops  # noqa: E402
from multimodal_mvd_seg_amd._lib import call, i3, query  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--patch", type=int, nargs=3, default=[128, 128, 128])
args = ap.parse_args()
dev = torch.device("cuda:0")
N, C, K = 2, 32, 32
D, H, W = args.patch
V = D * H * W
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = ops.empty_cl3d((N, C, D, H, W), dev, torch.bfloat16).normal_()
w = torch.randn(K, C, 3, 3, 3, device=dev) * 0.05
b = torch.zeros(K, device=dev)
wf, _ = ops.pack_weight_bf16(w, False)
y = ops.empty_cl3d((N, K, D, H, W), dev, torch.bfloat16)
a = torch.empty_like(y)
ws = torch.empty(max(1 << 20, query("mvd_conv_fwd_workspace_bytes", N, V, K)), dtype=torch.uint8, device=dev)
nt = query("mvd_conv3d_fwd_bf16_stats_tiles", N, D, H, W, C, 0, K, i3((3, 3, 3)), i3((1, 1, 1)))
stats = torch.empty((N, max(nt, 1), K, 2), device=dev)
scale, shift = torch.rand(N, C, device=dev) + 0.5, torch.randn(N, C, device=dev) * 0.1
mean, rstd = torch.empty(N, C, device=dev), torch.empty(N, C, device=dev)
g, be = torch.ones(K, device=dev), torch.zeros(K, device=dev)
got = ctypes.c_int(0)


def conv(st, pro):
    def f():
        call("mvd_conv3d_fwd_bf16_fused", P(x), C, None, 0, P(wf), P(b), P(y), N, D, H, W, K, i3((3, 3, 3)), i3((1, 1, 1)),
             P(scale) if pro else None, P(shift) if pro else None, 0.01, P(stats) if st else None, ctypes.byref(got), P(ws),
             ws.numel(), s)
    return f


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


out = {"shape": [N, C, D, H, W], "stats_tiles": nt}
for rep in range(2):
    for name, fn in (("conv_plain", conv(False, False)), ("conv_stats", conv(True, False)), ("conv_prologue", conv(False, True)),
                     ("conv_stats_prologue", conv(True, True)),
                     ("finalize_tiles", lambda: call("mvd_instnorm_finalize_tiles", P(stats), max(nt, 1), P(g), P(be), P(mean),
                                                     P(rstd), P(scale), P(shift), N, V, K, 1e-5, s)),
                     ("apply_scale_shift", lambda: call("mvd_instnorm_lrelu_apply_bf16", P(y), P(scale), P(shift), P(a), N, V, K,
                                                        0.01, s)),
                     ("instnorm_standalone", lambda: ops.InstanceNormLeakyReLUFn.apply(x, g, be, 1e-5, 0.01))):
        out.setdefault(name, []).append(round(timeit(fn), 4))
alg = (C + K) * V * N * 2.0
best = lambda k: min(out[k])
out["block_training_ms"] = round(best("conv_stats") + best("finalize_tiles") + best("apply_scale_shift"), 4)
out["block_inference_ms"] = round(best("conv_stats_prologue") + best("finalize_tiles"), 4)
out["block_round2_ms"] = round(best("conv_plain") + best("instnorm_standalone"), 4)
for k in ("block_training_ms", "block_inference_ms", "block_round2_ms"):
    out[k.replace("_ms", "_frac_of_hbm")] = round(alg / (out[k] * 1e-3) / 8e12, 4)
print(json.dumps(out))
