"""Sliding-window inference throughput on the GPU box:
   python tools/bench_infer.py [--image 4 192 256 256] [--patch 128 128 128] [--no-mirror] [--precision fp32|bf16]
Reports tiles, network forwards, wall time and voxels/s for one volume (synthetic data, random-init cfg-2 network)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodal_mvd_seg_amd import trainer  # noqa: E402
from multimodal_mvd_seg_amd.inference import SlidingWindowPredictor  # noqa: E402
from multimodal_mvd_seg_amd.network import set_precision  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", type=int, nargs=4, default=[4, 192, 256, 256])
    ap.add_argument("--patch", type=int, nargs=3, default=[128, 128, 128])
    ap.add_argument("--no-mirror", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    strides = [[1, 1, 1]] + [[2, 2, 2]] * 5
    plans = trainer.make_plans(tuple(args.patch), strides, batch_size=2)
    ds = {"channel_names": {str(i): str(i) for i in range(args.image[0])},
          "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=dev)
    tr.precision = args.precision
    torch.manual_seed(0)
    tr.initialize()
    set_precision(tr.network, args.precision)
    pred = SlidingWindowPredictor(tr.network, args.patch, 5, 0.5, True, not args.no_mirror, (0, 1, 2), dev)
    img = torch.rand(*args.image)
    pred.predict_sliding_window_return_logits(img[:, :args.patch[0], :args.patch[1], :args.patch[2]])  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = pred.predict_sliding_window_return_logits(img)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tiles = len(pred._internal_get_sliding_window_slicers(tuple(max(i, p) for i, p in zip(args.image[1:], args.patch))))
    fw = tiles * (1 if args.no_mirror else 8)
    vox = args.image[1] * args.image[2] * args.image[3]
    print(f"image {tuple(args.image)} patch {tuple(args.patch)} {args.precision}: {tiles} tiles, {fw} forwards, "
          f"{dt * 1e3:.1f} ms, {vox / dt / 1e6:.1f} Mvoxel/s, {dt / fw * 1e3:.2f} ms per forward; out {tuple(out.shape)}")


if __name__ == "__main__":
    main()
