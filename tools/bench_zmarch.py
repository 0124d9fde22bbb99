"""A/B of the two z-marching kernels (mvd_set_bf16_zmarch_kernel) on the 128^3 layers of the bf16 step: 32 -> 32 forward,
the 32 + 32 -> 32 two-pointer forward (k_fwd16y vs the generic k_fwd16) and the 32 -> 32 + 32 input gradient.
usage: python tools/bench_zmarch.py [--iters 20]"""
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
N = 2
D, H, W = args.patch
V = D * H * W
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
BF = torch.bfloat16
x1 = ops.empty_cl3d((N, 32, D, H, W), dev, BF).normal_()
x2 = ops.empty_cl3d((N, 32, D, H, W), dev, BF).normal_()
y = ops.empty_cl3d((N, 32, D, H, W), dev, BF)
y2 = ops.empty_cl3d((N, 32, D, H, W), dev, BF)
w32 = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.05
w64 = torch.randn(32, 64, 3, 3, 3, device=dev) * 0.04
b = torch.zeros(32, device=dev)
wf32, wb32 = ops.pack_weight_bf16(w32, False)
wf64, wb64 = ops.pack_weight_bf16(w64, False)
ws = torch.empty(max(1 << 20, query("mvd_conv_fwd_workspace_bytes", N, V, 64)), dtype=torch.uint8, device=dev)
ks, st = i3((3, 3, 3)), i3((1, 1, 1))


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / args.iters, 4)


cases = {
    "fwd_32_32": lambda: call("mvd_conv3d_fwd_bf16", P(x1), 32, None, 0, P(wf32), P(b), P(y), N, D, H, W, 32, ks, st, P(ws), ws.numel(), s),
    "fwd_32+32_32": lambda: call("mvd_conv3d_fwd_bf16", P(x1), 32, P(x2), 32, P(wf64), P(b), P(y), N, D, H, W, 32, ks, st, P(ws), ws.numel(), s),
    "dgrad_32_to_32+32": lambda: call("mvd_conv3d_dgrad_bf16", P(x1), P(wb64), P(y), 32, P(y2), 32, N, D, H, W, 32, ks, st, P(ws), ws.numel(), s),
    "dgrad_32_to_32": lambda: call("mvd_conv3d_dgrad_bf16", P(x1), P(wb32), P(y), 32, None, 0, N, D, H, W, 32, ks, st, P(ws), ws.numel(), s),
}
out = {}
for rep in range(2):
    for which in (1, 0):
        call("mvd_set_bf16_zmarch_kernel", which)
        for name, fn in cases.items():
            out.setdefault(f"{name}/{'y' if which else 'z_or_generic'}", []).append(timeit(fn))
call("mvd_set_bf16_zmarch_kernel", 1)
print(json.dumps(out))
