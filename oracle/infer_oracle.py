"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's sliding-window inference
(SURVEY 8f-1).  Follows nnUNet/nnunetv2/inference/sliding_window_prediction.py:10-56 (Gaussian importance map, step
placement) and nnUNet/nnunetv2/inference/predict_from_raw_data.py:528-560 (slicers), :562-595 (mirror TTA),
:643-714 (accumulate / normalise).  Differences, on purpose: accumulators are fp32 (the reference keeps them in fp16
under autocast, :676-682), and `network` is any callable [1,C,*patch] -> logits [1,K,*patch].

Pinning: the reference module cannot be imported here (`acvl_utils` is absent, an ordinary ImportError); its own
comment at sliding_window_prediction.py:37-38 is a known-answer vector for the step placement (image 110, patch 64,
step 0.5 -> 0, 23, 46), checked in tests/test_oracle.py; the Gaussian map uses scipy.ndimage.gaussian_filter exactly
as the reference does."""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def compute_gaussian(tile_size, sigma_scale=1. / 8, value_scaling_factor=1.0):
    """sliding_window_prediction.py:10-29 (fp32 instead of fp16)."""
    tmp = np.zeros(tile_size)
    center_coords = [i // 2 for i in tile_size]
    sigmas = [i * sigma_scale for i in tile_size]
    tmp[tuple(center_coords)] = 1
    g = gaussian_filter(tmp, sigmas, 0, mode='constant', cval=0)
    g = torch.from_numpy(g).float()
    g = g / torch.max(g) * value_scaling_factor
    g[g == 0] = torch.min(g[g != 0])
    return g


def compute_steps_for_sliding_window(image_size, tile_size, tile_step_size):
    """sliding_window_prediction.py:32-56."""
    assert all(i >= j for i, j in zip(image_size, tile_size)), "image size must be as large or larger than patch_size"
    assert 0 < tile_step_size <= 1, 'step_size must be larger than 0 and smaller or equal to 1'
    target = [i * tile_step_size for i in tile_size]
    num_steps = [int(np.ceil((i - k) / j)) + 1 for i, j, k in zip(image_size, target, tile_size)]
    steps = []
    for dim in range(len(tile_size)):
        max_step_value = image_size[dim] - tile_size[dim]
        actual = max_step_value / (num_steps[dim] - 1) if num_steps[dim] > 1 else 99999999999
        steps.append([int(np.round(actual * i)) for i in range(num_steps[dim])])
    return steps


def pad_to_patch(image, patch_size):
    """acvl_utils pad_nd_image(image, patch_size, 'constant', {'value': 0}, True) semantics for [C,D,H,W]: centre the
    image in a zero volume of at least the patch size; returns (padded, slicer that undoes it)."""
    shape = image.shape[1:]
    new = [max(s, p) for s, p in zip(shape, patch_size)]
    diff = [n - s for n, s in zip(new, shape)]
    below = [d // 2 for d in diff]
    out = torch.zeros((image.shape[0], *new), dtype=image.dtype)
    sl = tuple(slice(b, b + s) for b, s in zip(below, shape))
    out[(slice(None), *sl)] = image
    return out, (slice(None), *sl)


def mirror_and_predict(network, x, mirror_axes):
    """predict_from_raw_data.py:562-588."""
    prediction = network(x).clone()
    if mirror_axes is not None and len(mirror_axes) > 0:
        combos = []
        axes = sorted(mirror_axes)
        for m in range(1, 2 ** len(axes)):
            combos.append(tuple(a + 2 for j, a in enumerate(axes) if (m >> j) & 1))
        for c in combos:
            prediction += torch.flip(network(torch.flip(x, c)), c)
        prediction /= (len(combos) + 1)
    return prediction


def predict_sliding_window_return_logits(network, input_image, patch_size, num_heads, tile_step_size=0.5,
                                         use_gaussian=True, mirror_axes=(0, 1, 2)):
    """predict_from_raw_data.py:643-714."""
    assert input_image.dim() == 4
    data, revert = pad_to_patch(input_image.float().cpu(), patch_size)
    steps = compute_steps_for_sliding_window(data.shape[1:], patch_size, tile_step_size)
    logits = torch.zeros((num_heads, *data.shape[1:]), dtype=torch.float32)
    n_pred = torch.zeros(data.shape[1:], dtype=torch.float32)
    gaussian = compute_gaussian(tuple(patch_size), 1. / 8, 1000.0) if use_gaussian else None
    for sx in steps[0]:
        for sy in steps[1]:
            for sz in steps[2]:
                sl = (slice(None), slice(sx, sx + patch_size[0]), slice(sy, sy + patch_size[1]),
                      slice(sz, sz + patch_size[2]))
                pred = mirror_and_predict(network, data[sl][None], mirror_axes)[0].float().cpu()
                logits[sl] += pred * gaussian if use_gaussian else pred
                n_pred[sl[1:]] += gaussian if use_gaussian else 1
    logits /= n_pred
    return logits[revert]
