"""Micro-benchmark (GPU box) of the conv engines on the layer shapes of BASELINE cfg 2 (batch 2):
   python tools/bench_conv.py [--layers enc0.conv1,dec5.conv0,...] [--iters 5] [--what fwd,dgrad,wgrad]
Prints ms and TFLOP/s per (layer, pass), HIP events on the launch stream."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_mvd_seg_amd import ops  # noqa: E402
from multimodal_mvd_seg_amd._lib import call, i3, query  # noqa: E402

# name: (C1, C2, K, spatial_in, stride)
LAYERS = {
    "enc0.conv0": (4, 0, 32, 128, 1), "enc0.conv1": (32, 0, 32, 128, 1), "enc1.conv0": (32, 0, 64, 128, 2),
    "enc1.conv1": (64, 0, 64, 64, 1), "enc2.conv0": (64, 0, 128, 64, 2), "enc2.conv1": (128, 0, 128, 32, 1),
    "enc3.conv1": (256, 0, 256, 16, 1), "enc4.conv1": (320, 0, 320, 8, 1), "enc5.conv1": (320, 0, 320, 4, 1),
    "dec1.conv0": (320, 320, 320, 8, 1), "dec3.conv0": (128, 128, 128, 32, 1), "dec4.conv0": (64, 64, 64, 64, 1),
    "dec5.conv0": (32, 32, 32, 128, 1), "dec5.conv1": (32, 0, 32, 128, 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", default="enc0.conv1,dec5.conv0,enc1.conv1,dec4.conv0,enc2.conv1,enc1.conv0,enc4.conv1,dec1.conv0")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--what", default="fwd,dgrad,wgrad")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--wino", type=int, default=1, help="fp32: 1 = Winograd entries (default, what ops.py calls), 0 = direct")
    args = ap.parse_args()
    bf = args.dtype == "bf16"
    dt = torch.bfloat16 if bf else torch.float32
    sfx = "_bf16" if bf else ""
    dev = torch.device("cuda:0")
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    N = args.batch
    for name in args.layers.split(","):
        C1, C2, K, S, st = LAYERS[name]
        So = S // st
        if bf and (C1 % 32 or K % 32):
            continue
        x1 = ops.empty_cl3d((N, C1, S, S, S), dev, dt).normal_()
        x2 = ops.empty_cl3d((N, C2, S, S, S), dev, dt).normal_() if C2 else None
        w = torch.randn(K, C1 + C2, 3, 3, 3, device=dev) * 0.03
        wf, wb = ops.pack_weight_bf16(w, False) if bf else ops.pack_weight(w, False)
        bias = torch.zeros(K, device=dev)
        y = ops.empty_cl3d((N, K, So, So, So), dev, dt)
        dy = ops.empty_cl3d((N, K, So, So, So), dev, dt).normal_()
        dx1 = ops.empty_cl3d((N, C1, S, S, S), dev, dt)
        dx2 = ops.empty_cl3d((N, C2, S, S, S), dev, dt) if C2 else None
        dw = torch.empty_like(w)
        db = torch.empty(K, device=dev)
        nb = max(query("mvd_conv3d_wgrad_workspace_bytes", C1 + C2, K, 27, N, So, So, So),
                 query("mvd_conv_fwd_workspace_bytes", N, S ** 3, max(K, C1 + C2)))
        ws = torch.empty(max(nb, 1024), dtype=torch.uint8, device=dev)
        ks, sd = i3((3, 3, 3)), i3((st, st, st))
        flops = 2.0 * 27 * (C1 + C2) * K * N * So ** 3
        uf = ub = None
        if not bf and args.wino and query("mvd_conv_wino_applicable", N, S, S, S, C1, C2, K, i3((3, 3, 3)), i3((st, st, st))):
            uf = torch.empty(query("mvd_wino_weight_elems", C1 + C2, K), device=dev)
            ub = torch.empty(query("mvd_wino_weight_elems", C1 + C2, K), device=dev)
            call("mvd_pack_weight_wino", P(w), P(uf), P(ub), K, C1 + C2, s)
        fns = {
            "fwd": (lambda: call("mvd_conv3d_fwd_wino", P(x1), C1, P(x2), C2, P(wf), P(uf), P(bias), P(y), N, S, S, S, K, ks,
                                 sd, P(ws), ws.numel(), s)) if not bf else
                   (lambda: call("mvd_conv3d_fwd_bf16", P(x1), C1, P(x2), C2, P(wf), P(bias), P(y), N, S, S, S, K, ks, sd,
                                 P(ws), ws.numel(), s)),
            "dgrad": (lambda: call("mvd_conv3d_dgrad_wino", P(dy), P(wb), P(ub), P(dx1), C1, P(dx2), C2, N, S, S, S, K, ks,
                                   sd, P(ws), ws.numel(), s)) if not bf else
                     (lambda: call("mvd_conv3d_dgrad_bf16", P(dy), P(wb), P(dx1), C1, P(dx2), C2, N, S, S, S, K, ks, sd,
                                   P(ws), ws.numel(), s)),
            "wgrad": lambda: call("mvd_conv3d_wgrad" + sfx, P(x1), C1, P(x2), C2, P(dy), P(dw), P(db), N, S, S, S, K, ks, sd,
                                  P(ws), ws.numel(), s),
        }
        for what in args.what.split(","):
            fn = fns[what]
            for _ in range(max(3, args.iters // 2)):  # ramps the shader clock up: the first timed entry used to read ~10 % slow
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            eb = 2 if bf else 4
            gb = N * (S ** 3 * (C1 + C2) + So ** 3 * K) * eb / 1e9  # algorithmic bytes: every activation once
            print(f"{name:12s} {what:6s} C={C1}+{C2} K={K} S={S} st={st}: {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s"
                  f"  {gb / ms * 1e3:7.1f} GB/s (algorithmic)", flush=True)
        del x1, x2, y, dy, dx1, dx2, ws


if __name__ == "__main__":
    main()
