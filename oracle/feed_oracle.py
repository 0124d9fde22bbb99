"""TEST INFRASTRUCTURE -- numpy restatement of the deterministic part of the reference's training feed.  Only tests/
may import this.

  crop_pad            nnUNetDataLoader3D.generate_train_batch (dataloading/data_loader_3d.py:31-46), literally:
                      clip the box to the volume, slice, np.pad with constant 0 (data) / -1 (seg)
  mirror              MirrorTransform as configured at nnUNetTrainer.py:738-739 (batchgenerators augment_mirroring:
                      a flagged axis is reversed)
  remove_label        RemoveLabelTransform(-1, 0) (nnUNetTrainer.py:745)
  downsample_seg      DownsampleSegForDSTransform2 order 0 (deep_supervision_donwsampling.py:33-53) ->
                      batchgenerators resize_segmentation(order=0) -> skimage.transform.resize(order=0, mode='edge',
                      anti_aliasing=False), which calls scipy.ndimage.zoom(order=0, mode='nearest', grid_mode=True)

Pinning: the crop/pad and the -1 -> 0 rule are the reference's own lines.  batchgenerators and skimage are absent
from /root/reference and not importable here, so the mirror convention and the order-0 resize are PARITY UNPINNED
against those packages; the resize is pinned to the scipy call skimage's published implementation makes
(tests/test_oracle.py checks the closed-form index used on the GPU against scipy.ndimage.zoom).
"""
import numpy as np
from scipy import ndimage


def crop_pad(arr, bbox_lbs, patch_size, pad_value):
    shape = arr.shape[1:]
    dim = len(shape)
    bbox_ubs = [bbox_lbs[i] + patch_size[i] for i in range(dim)]
    valid_lbs = [max(0, bbox_lbs[i]) for i in range(dim)]
    valid_ubs = [min(shape[i], bbox_ubs[i]) for i in range(dim)]
    sl = tuple([slice(0, arr.shape[0])] + [slice(i, j) for i, j in zip(valid_lbs, valid_ubs)])
    cropped = arr[sl]
    padding = [(-min(0, bbox_lbs[i]), max(bbox_ubs[i] - shape[i], 0)) for i in range(dim)]
    return np.pad(cropped, ((0, 0), *padding), 'constant', constant_values=pad_value)


def mirror(arr, flip_mask):
    for ax in range(3):
        if flip_mask & (1 << ax):
            arr = np.flip(arr, axis=1 + ax)
    return arr


def remove_label(seg, remove=-1, replace_with=0):
    seg = seg.copy()
    seg[seg == remove] = replace_with
    return seg


def downsample_seg(target, scale):
    """target [B,C,D,H,W] -> order-0 resized copy (scipy.ndimage.zoom as skimage.transform.resize calls it)."""
    if not isinstance(scale, (tuple, list)):
        scale = [scale] * 3
    if all(s == 1 for s in scale):
        return target
    new_shape = np.array(target.shape).astype(float)
    for i in range(3):
        new_shape[2 + i] *= scale[i]
    new_shape = np.round(new_shape).astype(int)
    out = np.zeros(new_shape, dtype=target.dtype)
    for b in range(target.shape[0]):
        for c in range(target.shape[1]):
            src = target[b, c].astype(float)
            zoom = [new_shape[2 + i] / src.shape[i] for i in range(3)]
            out[b, c] = ndimage.zoom(src, zoom, order=0, mode='nearest', grid_mode=True).astype(target.dtype)
    return out


def nn_index(o, n, m):
    """closed form of the order-0 source index used by the HIP kernel"""
    return np.minimum(((2 * np.asarray(o) + 1) * n) // (2 * m), n - 1)


def generate_train_batch(cases, keys, boxes, flips, patch_size, ds_scales=None):
    """cases: key -> (data [C,D,H,W] float32, seg [1,D,H,W] int).  Returns (data [B,C,*patch] float32, target list)."""
    data_all = np.zeros((len(keys), cases[keys[0]][0].shape[0], *patch_size), dtype=np.float32)
    seg_all = np.zeros((len(keys), cases[keys[0]][1].shape[0], *patch_size), dtype=np.int16)
    for j, k in enumerate(keys):
        data, seg = cases[k]
        data_all[j] = mirror(crop_pad(data, boxes[j], patch_size, 0), flips[j])
        seg_all[j] = mirror(crop_pad(seg.astype(np.int16), boxes[j], patch_size, -1), flips[j])
    target = remove_label(seg_all, -1, 0).astype(np.float32)
    if ds_scales is None:
        return data_all, target
    return data_all, [downsample_seg(target, s) for s in ds_scales]
