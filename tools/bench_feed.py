"""Times the on-device training feed (SURVEY 8f-2 slice) at the BASELINE cfg-2 batch shape and the numpy oracle
(the reference's per-batch CPU work without the batchgenerators intensity transforms) beside it."""
import sys
import time

import numpy as np
import torch

from multimodal_mvd_seg_amd.dataloading import DeviceDataLoader3D


class _DS:
    def __init__(self, n, shape, channels=4):
        rng = np.random.default_rng(0)
        self.cases = {}
        for i in range(n):
            data = rng.standard_normal((channels, *shape)).astype(np.float32)
            seg = (rng.random((1, *shape)) > 0.95).astype(np.int16) * rng.integers(1, 5, (1, *shape)).astype(np.int16)
            locs = {c: np.argwhere(seg == c)[:10000] for c in (1, 2, 3, 4)}
            self.cases[f"c{i}"] = (data, seg, {"class_locations": locs})

    def keys(self):
        return self.cases.keys()

    def load_case(self, k):
        return self.cases[k]


class _L:
    all_labels = [1, 2, 3, 4]
    has_ignore_label = False


def main():
    patch = (128, 128, 128)
    scales = [1, 0.5, 0.25, 0.125, 0.0625]
    ds = _DS(6, (160, 224, 192))
    dl = DeviceDataLoader3D(ds, 2, patch, patch, _L(), oversample_foreground_percent=0.33, mirror_axes=(0, 1, 2),
                            deep_supervision_scales=scales, device="cuda:0")
    np.random.seed(0)
    for _ in range(6):
        next(dl)  # uploads every case once
    torch.cuda.synchronize()
    plans = [dl.plan_batch() for _ in range(50)]
    t0 = time.perf_counter()
    for p in plans:
        b = dl.generate_train_batch(p)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / len(plans) * 1e3
    nbytes = 2 * (4 + 1) * 128 ** 3 * 4 * 2  # read + write of data and target, per batch
    print(f"device feed: {ms:.3f} ms per batch of 2 (4x128^3 + 5 DS targets) = {2 / ms * 1e3:.0f} samples/s, "
          f"{nbytes / ms / 1e6:.0f} GB/s of batch traffic")
    if "--cpu" in sys.argv:
        sys.path.insert(0, ".")
        from oracle import feed_oracle as FO
        cases = {k: (v[0], v[1]) for k, v in ds.cases.items()}
        t0 = time.perf_counter()
        for p in plans[:5]:
            FO.generate_train_batch(cases, p[0], p[1], p[2], patch, scales)
        cpu_ms = (time.perf_counter() - t0) / 5 * 1e3
        print(f"numpy oracle (one core): {cpu_ms:.1f} ms per batch")


if __name__ == "__main__":
    main()
