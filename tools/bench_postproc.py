"""Times the device post-processing chain on a full predicted volume and the scipy oracle beside it."""
import sys
import time

import numpy as np
import torch

from multimodal_mvd_seg_amd import ops, postprocessing as PP


def main():
    shape = (192, 256, 256)
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(5)
    f = torch.rand((1, 1, shape[0] // 4, shape[1] // 4, shape[2] // 4), generator=g)
    f = torch.nn.functional.interpolate(f, scale_factor=4, mode="trilinear").squeeze().to(dev)
    seg = (f > 0.62).to(torch.int32) + (f > 0.7).to(torch.int32)
    for _ in range(3):
        out = PP.remove_all_but_largest_component_from_segmentation(seg, [1, 2], 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 20
    for _ in range(iters):
        out = PP.remove_all_but_largest_component_from_segmentation(seg, [1, 2], 0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    n = seg.numel()
    _, count = ops.cc_label(ops.seg_label_mask(seg, [1, 2]), 26)
    print(f"postproc {shape}: {ms:.3f} ms/volume ({n / ms / 1e6:.2f} Gvoxel/s), components={int(count.item())}")
    if "--cpu" in sys.argv:
        sys.path.insert(0, ".")
        from oracle import postproc_oracle as PO
        s = seg.cpu().numpy()
        t0 = time.perf_counter()
        ref = PO.remove_all_but_largest_component_from_segmentation(s, [1, 2], 0)
        cpu_ms = (time.perf_counter() - t0) * 1e3
        print(f"scipy oracle: {cpu_ms:.1f} ms; identical={np.array_equal(ref, out.cpu().numpy())}")


if __name__ == "__main__":
    main()
