"""Diagnostic (GPU box): relative L2 error vs an fp64 evaluation of single ops in the small-volume / many-channel
regime, for the MFMA engine, the scalar engine and torch-CPU fp32."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_mvd_seg_amd import ops  # noqa: E402

DEV = torch.device("cuda:0")


def rel(a, b):
    return float((a.double().cpu() - b).norm() / (b.norm() + 1e-300))


def conv_case(C1, C2, K, sp, st, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.randn(N, C1, *sp, generator=g)
    x2 = torch.randn(N, C2, *sp, generator=g) if C2 else None
    w = torch.randn(K, C1 + C2, 3, 3, 3, generator=g) / np.sqrt(27 * (C1 + C2))
    b = torch.zeros(K)
    xs = [t.double().requires_grad_() for t in ([x1, x2] if C2 else [x1])]
    wr = w.double().requires_grad_()
    ref = F.conv3d(torch.cat(xs, 1), wr, b.double(), st, 1)
    gy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gy)
    # torch fp32
    xs32 = [t.clone().requires_grad_() for t in ([x1, x2] if C2 else [x1])]
    w32 = w.clone().requires_grad_()
    y32 = F.conv3d(torch.cat(xs32, 1), w32, b, st, 1)
    y32.backward(gy.float())
    out = {"cpu32": (rel(y32.detach(), ref.detach()), rel(xs32[0].grad, xs[0].grad), rel(w32.grad, wr.grad))}
    for eng in ("auto", "scalar"):
        ops.set_conv_engine(eng)
        g1 = x1.to(DEV).requires_grad_()
        g2 = x2.to(DEV).requires_grad_() if C2 else None
        gw = w.to(DEV).requires_grad_()
        y = ops.Conv3dFn.apply(g1, g2, gw, b.to(DEV), (st,) * 3)
        y.backward(gy.float().to(DEV))
        out[eng] = (rel(y.detach(), ref.detach()), rel(g1.grad, xs[0].grad), rel(gw.grad, wr.grad))
    ops.set_conv_engine("auto")
    print(f"conv C={C1}+{C2} K={K} sp={sp} st={st} N={N}")
    for k, v in out.items():
        print(f"   {k:7s} y {v[0]:.2e}  dx {v[1]:.2e}  dw {v[2]:.2e}")


def convT_case(C, K, sp, N, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, *sp, generator=g)
    w = torch.randn(C, K, 2, 2, 2, generator=g) / np.sqrt(C)
    b = torch.zeros(K)
    xr, wr = x.double().requires_grad_(), w.double().requires_grad_()
    ref = F.conv_transpose3d(xr, wr, b.double(), 2)
    gy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gy)
    x32, w32 = x.clone().requires_grad_(), w.clone().requires_grad_()
    y32 = F.conv_transpose3d(x32, w32, b, 2)
    y32.backward(gy.float())
    out = {"cpu32": (rel(y32.detach(), ref.detach()), rel(x32.grad, xr.grad), rel(w32.grad, wr.grad))}
    for eng in ("auto", "scalar"):
        ops.set_conv_engine(eng)
        gx, gw = x.to(DEV).requires_grad_(), w.to(DEV).requires_grad_()
        y = ops.ConvTranspose3dFn.apply(gx, gw, b.to(DEV), (2, 2, 2))
        y.backward(gy.float().to(DEV))
        out[eng] = (rel(y.detach(), ref.detach()), rel(gx.grad, xr.grad), rel(gw.grad, wr.grad))
    ops.set_conv_engine("auto")
    print(f"convT C={C} K={K} sp={sp} N={N}")
    for k, v in out.items():
        print(f"   {k:7s} y {v[0]:.2e}  dx {v[1]:.2e}  dw {v[2]:.2e}")


def norm_case(N, C, sp, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, C, *sp, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    gy = torch.randn(N, C, *sp, generator=g)
    xr, gr, br = x.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = F.leaky_relu(F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5), 0.01)
    ref.backward(gy.double())
    x32, g32, b32 = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    y32 = F.leaky_relu(F.instance_norm(x32, None, None, g32, b32, True, 0.1, 1e-5), 0.01)
    y32.backward(gy)
    gx, gg, gb = x.to(DEV).requires_grad_(), gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    y = ops.InstanceNormLeakyReLUFn.apply(gx, gg, gb, 1e-5, 0.01)
    y.backward(gy.to(DEV))
    print(f"instnorm N={N} C={C} sp={sp}")
    print(f"   cpu32   y {rel(y32.detach(), ref.detach()):.2e}  dx {rel(x32.grad, xr.grad):.2e}  dg {rel(g32.grad, gr.grad):.2e} db {rel(b32.grad, br.grad):.2e}")
    print(f"   hip     y {rel(y.detach(), ref.detach()):.2e}  dx {rel(gx.grad, xr.grad):.2e}  dg {rel(gg.grad, gr.grad):.2e} db {rel(gb.grad, br.grad):.2e}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    conv_case(320, 320, 320, (4, 4, 4), 1, 2)
    conv_case(320, 0, 320, (2, 2, 2), 1, 2)
    conv_case(256, 0, 320, (4, 4, 4), 2, 2)
    conv_case(128, 128, 128, (8, 8, 8), 1, 2)
    conv_case(32, 32, 32, (32, 32, 32), 1, 2)
    convT_case(320, 256, (2, 2, 2), 2)
    convT_case(64, 32, (16, 16, 16), 2)
    norm_case(2, 320, (2, 2, 2))
    norm_case(2, 256, (4, 4, 4))
    norm_case(2, 32, (32, 32, 32))
