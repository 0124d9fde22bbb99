"""Micro-benchmark of the transposed-conv (2x2x2 up-sampling) forward / input-gradient kernels at the network's levels.
usage: python tools/bench_convT.py [--dtype fp32|bf16] [--iters 10]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodal_mvd_seg_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="fp32"); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
bf = a.dtype == "bf16"
for (C, K, S) in [(64, 32, 64), (128, 64, 32), (256, 128, 16), (320, 256, 8)]:
    x = torch.randn(2, C, S, S, S, device=dev).contiguous(memory_format=torch.channels_last_3d)
    if bf: x = x.bfloat16()
    x.requires_grad_()
    w = (torch.randn(C, K, 2, 2, 2, device=dev) / C ** 0.5).requires_grad_()
    b = torch.zeros(K, device=dev, requires_grad=True)
    y = ops.ConvTranspose3dFn.apply(x, w, b, (2, 2, 2))
    gy = torch.randn_like(y)
    def fwd(): return ops.ConvTranspose3dFn.apply(x, w, b, (2, 2, 2))
    def t(f):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters
    with torch.no_grad():
        tf = t(fwd)
    def fb():
        x.grad = None
        ops.ConvTranspose3dFn.apply(x, w.detach(), b.detach(), (2, 2, 2)).backward(gy)
    tb = t(fb) - tf  # forward + input gradient - forward
    gb = (x.numel() + y.numel()) * x.element_size() / 1e9
    print(f"convT {C}->{K} @{S}^3 {a.dtype}: fwd {tf:.3f} ms  {gb / tf:.2f} TB/s (x + y); dgrad {tb:.3f} ms  {gb / tb:.2f} TB/s")
