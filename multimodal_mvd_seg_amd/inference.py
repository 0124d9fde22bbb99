"""Sliding-window inference with Gaussian blending and mirror test-time augmentation on the MI355X.

Mirrors the prediction core of nnUNet/nnunetv2/inference/predict_from_raw_data.py::nnUNetPredictor
(`predict_sliding_window_return_logits` :643-714, `_internal_get_sliding_window_slicers` :528-560,
`_internal_maybe_mirror_and_predict` :562-588) and nnUNet/nnunetv2/inference/sliding_window_prediction.py:10-56.
The network forward is the HIP path of network.py; flips, Gaussian-weighted accumulation and normalisation are HIP
kernels behind the C ABI (mvd_flip_add, mvd_sw_accumulate, mvd_sw_normalize).  Host-side pieces (step placement, the
separable Gaussian importance map) are numpy.  Accumulators are fp32 (the reference keeps them in fp16 under autocast).
There is no CPU fallback: tensors are moved to the network's cuda device.
"""
import ctypes
from typing import List, Sequence, Tuple

import numpy as np
import torch

from ._lib import call


def compute_steps_for_sliding_window(image_size: Sequence[int], tile_size: Sequence[int], tile_step_size: float) \
        -> List[List[int]]:
    """Tile origins per axis (same name / arguments / result as sliding_window_prediction.py:32-56; pinned to the
    reference function by tests/golden/sw_steps.json): the fewest tiles whose spacing does not exceed
    tile_step_size * tile, spread evenly from 0 to image - tile."""
    image, tile = np.asarray(image_size, dtype=np.int64), np.asarray(tile_size, dtype=np.int64)
    if image.shape != tile.shape or np.any(image < tile):
        raise ValueError("image size must be as large or larger than patch_size")
    if not 0 < tile_step_size <= 1:
        raise ValueError("step_size must be larger than 0 and smaller or equal to 1")
    span = image - tile                                      # last admissible origin
    count = np.ceil(span / (tile * tile_step_size)).astype(np.int64) + 1
    origins = []
    for last, n in zip(span.tolist(), count.tolist()):
        pitch = last / (n - 1) if n > 1 else 0.0
        origins.append([int(np.round(pitch * i)) for i in range(n)])
    return origins


def _gaussian_kernel1d(sigma: float, radius: int) -> np.ndarray:
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def compute_gaussian(tile_size: Sequence[int], sigma_scale: float = 1. / 8, value_scaling_factor: float = 1.0) \
        -> np.ndarray:
    """sliding_window_prediction.py:10-29 without scipy: gaussian_filter of a unit impulse with mode='constant' is the
    outer product of the truncated (4 sigma), normalised 1-D kernels centred on the impulse."""
    axes = []
    for n in tile_size:
        sigma = n * sigma_scale
        radius = int(4.0 * sigma + 0.5)
        k = _gaussian_kernel1d(sigma, radius)
        line = np.zeros(n)
        c = n // 2
        for j, w in enumerate(k):
            p = c + j - radius
            if 0 <= p < n:
                line[p] = w
        axes.append(line)
    g = axes[0]
    for a in axes[1:]:
        g = np.multiply.outer(g, a)
    g = g.astype(np.float32)
    g = g / g.max() * value_scaling_factor
    g[g == 0] = g[g != 0].min()
    return g


class SlidingWindowPredictor:
    """`predict_sliding_window_return_logits(image [C,D,H,W]) -> logits [K,D,H,W]` as nnUNetPredictor's
    (`tile_step_size`, `use_gaussian`, `use_mirroring`, `allowed_mirroring_axes` have the reference's meaning)."""

    def __init__(self, network: torch.nn.Module, patch_size: Sequence[int], num_segmentation_heads: int,
                 tile_step_size: float = 0.5, use_gaussian: bool = True, use_mirroring: bool = True,
                 allowed_mirroring_axes: Tuple[int, ...] = (0, 1, 2), device: torch.device = torch.device('cuda')):
        if torch.device(device).type != 'cuda':
            raise RuntimeError("SlidingWindowPredictor needs an MI355X (device type 'cuda'); there is no CPU path")
        self.network = network
        self.patch_size = tuple(int(i) for i in patch_size)
        self.num_heads = int(num_segmentation_heads)
        self.tile_step_size = tile_step_size
        self.use_gaussian = use_gaussian
        self.use_mirroring = use_mirroring
        self.allowed_mirroring_axes = tuple(allowed_mirroring_axes)
        self.device = torch.device(device)
        self._gaussian = None

    # predict_from_raw_data.py:528-560 (3-D branch)
    def _internal_get_sliding_window_slicers(self, image_size):
        steps = compute_steps_for_sliding_window(image_size, self.patch_size, self.tile_step_size)
        return [(sx, sy, sz) for sx in steps[0] for sy in steps[1] for sz in steps[2]]

    def _mirror_masks(self):
        """non-empty subsets of the allowed axes as flip masks (bit 0: D, 1: H, 2: W), in the reference's order"""
        if not self.use_mirroring or not self.allowed_mirroring_axes:
            return []
        ax = self.allowed_mirroring_axes
        assert max(ax) <= 2, 'mirror_axes does not match the dimension of the input!'
        order = [(0,), (1,), (2,), (0, 1), (0, 2), (1, 2), (0, 1, 2)]
        return [sum(1 << a for a in c) for c in order if all(a in ax for a in c)]

    @staticmethod
    def _s():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _network_logits(self, x):
        out = self.network(x)
        if isinstance(out, (list, tuple)):
            out = out[0]
        return out.contiguous()  # planar [1,K,*patch]

    # predict_from_raw_data.py:562-588: sum of the prediction and the un-flipped predictions of every flipped input
    def _internal_maybe_mirror_and_predict(self, x):
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        pred = self._network_logits(x)
        masks = self._mirror_masks()
        if not masks:
            return pred, 1
        total = pred.clone()
        C = x.shape[1]
        K = pred.shape[1]
        d, h, w = self.patch_size
        xf = torch.empty_like(x)
        for m in masks:
            call("mvd_flip_add", P(x), P(xf), C, d, h, w, m, 0, self._s())
            pm = self._network_logits(xf)
            call("mvd_flip_add", P(pm), P(total), K, d, h, w, m, 1, self._s())
        return total, len(masks) + 1

    def predict_sliding_window_return_logits(self, input_image: torch.Tensor) -> torch.Tensor:
        assert isinstance(input_image, torch.Tensor)
        assert input_image.dim() == 4, 'input_image must be a 4D torch.Tensor (c, x, y, z)'
        P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        was_training = self.network.training
        ds = getattr(getattr(self.network, 'decoder', None), 'deep_supervision', None)
        self.network.eval()
        if ds is not None:
            self.network.decoder.deep_supervision = False  # inference returns the full-resolution head only
        try:
            with torch.no_grad():
                img = input_image.to(self.device, dtype=torch.float32)
                # pad to at least the patch size, centred (pad_nd_image(..., 'constant', value 0), :665-667)
                shape = tuple(img.shape[1:])
                new = [max(s, p) for s, p in zip(shape, self.patch_size)]
                below = [(n - s) // 2 for n, s in zip(new, shape)]
                data = torch.zeros((img.shape[0], *new), dtype=torch.float32, device=self.device)
                data[:, below[0]:below[0] + shape[0], below[1]:below[1] + shape[1], below[2]:below[2] + shape[2]] = img
                D, H, W = new
                K = self.num_heads
                logits = torch.zeros((K, D, H, W), dtype=torch.float32, device=self.device)
                npred = torch.zeros((D, H, W), dtype=torch.float32, device=self.device)
                if self.use_gaussian and self._gaussian is None:
                    self._gaussian = torch.from_numpy(compute_gaussian(self.patch_size, 1. / 8, 1000.0)).to(self.device)
                g = self._gaussian if self.use_gaussian else None
                pd, ph, pw = self.patch_size
                for (sx, sy, sz) in self._internal_get_sliding_window_slicers((D, H, W)):
                    workon = data[:, sx:sx + pd, sy:sy + ph, sz:sz + pw][None].contiguous()
                    total, npasses = self._internal_maybe_mirror_and_predict(workon)
                    call("mvd_sw_accumulate", P(total), P(g), 1.0 / npasses, P(logits), P(npred), K, pd, ph, pw, D, H, W,
                         sx, sy, sz, self._s())
                call("mvd_sw_normalize", P(logits), P(npred), K, D * H * W, self._s())
                return logits[:, below[0]:below[0] + shape[0], below[1]:below[1] + shape[1],
                              below[2]:below[2] + shape[2]]
        finally:
            if ds is not None:
                self.network.decoder.deep_supervision = ds
            self.network.train(was_training)
