"""On-device training feed for case volumes resident in HBM (the deterministic part).

Host-side mirror of nnUNetDataLoader3D (nnunetv2/training/dataloading/data_loader_3d.py:6-48) and of its base class
(base_data_loader.py:10-139): same constructor arguments, same numpy RNG call order for the case selection
(batchgenerators DataLoader.get_indices: np.random.choice with replacement), the oversampling rule and get_bbox, so a
seeded run picks the boxes the reference loader picks.  What the reference then does on 12 CPU worker processes --
crop, pad (data 0 / seg -1), MirrorTransform, RemoveLabelTransform(-1, 0), DownsampleSegForDSTransform2 and
NumpyToTensor('float') (nnUNetTrainer.py:738-768) -- runs here as three HIP kernels through the C ABI
(csrc/feed.hip) on volumes that stay in HBM (288 GB holds a whole preprocessed dataset); there is no CPU fallback.
The intensity / spatial augmentations of the transform list (:703-737: rotation+scaling, noise, blur, brightness,
contrast, low-resolution simulation, gamma) live in batchgenerators, which is absent from the reference tree; they are
not part of this slice.
"""
import ctypes

import numpy as np
import torch

from ._lib import call


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def crop_pad_data(vol, out, bbox_lbs, flip_mask=0, pad=0.0):
    """out[C,pd,ph,pw] (float32, device) <- pad(vol[:, lb:lb+patch]) mirrored on the axes of flip_mask."""
    C, D, H, W = vol.shape
    pd, ph, pw = out.shape[1:]
    if not (vol.is_cuda and out.is_cuda and vol.dtype == torch.float32 and out.dtype == torch.float32
            and vol.is_contiguous() and out.is_contiguous() and out.shape[0] == C):
        raise RuntimeError("crop_pad_data: contiguous float32 device tensors [C,D,H,W] -> [C,pd,ph,pw]")
    call("mvd_feed_crop_pad_f32", _p(vol), _p(out), C, D, H, W, pd, ph, pw, int(bbox_lbs[0]), int(bbox_lbs[1]),
         int(bbox_lbs[2]), int(flip_mask), float(pad), _stream())


def crop_pad_seg(seg, out, bbox_lbs, flip_mask=0, pad=-1, replace=None):
    """int16 seg [C,D,H,W] -> float32 target; replace=(from, to) applies RemoveLabelTransform after the padding."""
    C, D, H, W = seg.shape
    pd, ph, pw = out.shape[1:]
    if not (seg.is_cuda and out.is_cuda and seg.dtype == torch.int16 and out.dtype == torch.float32
            and seg.is_contiguous() and out.is_contiguous() and out.shape[0] == C):
        raise RuntimeError("crop_pad_seg: contiguous int16 -> float32 device tensors [C,D,H,W] -> [C,pd,ph,pw]")
    rf, rt = (replace if replace is not None else (0, 0))
    call("mvd_feed_crop_pad_seg_i16", _p(seg), _p(out), C, D, H, W, pd, ph, pw, int(bbox_lbs[0]), int(bbox_lbs[1]),
         int(bbox_lbs[2]), int(flip_mask), int(pad), int(replace is not None), int(rf), int(rt), _stream())


def downsample_seg(target, scale):
    """DownsampleSegForDSTransform2 (deep_supervision_donwsampling.py:33-53), order 0: [B,C,D,H,W] float32 ->
    [B,C,round(D*s0),round(H*s1),round(W*s2)]."""
    if not isinstance(scale, (tuple, list)):
        scale = [scale] * 3
    if all(s == 1 for s in scale):
        return target
    B, C, D, H, W = target.shape
    new = np.round(np.array([D, H, W], dtype=float) * np.array(scale, dtype=float)).astype(int)
    out = torch.empty((B, C, int(new[0]), int(new[1]), int(new[2])), dtype=torch.float32, device=target.device)
    if not (target.is_cuda and target.dtype == torch.float32 and target.is_contiguous()):
        raise RuntimeError("downsample_seg: contiguous float32 device tensor")
    call("mvd_feed_downsample_seg", _p(target), _p(out), B * C, D, H, W, int(new[0]), int(new[1]), int(new[2]), _stream())
    return out


class DeviceDataLoader3D:
    """nnUNetDataLoader3D with the per-batch work on the GPU.  `data` is an nnUNetDataset-like object: `.keys()` and
    `.load_case(key) -> (data [C,D,H,W] float32, seg [1,D,H,W] integer, properties)` with
    properties['class_locations'] = {label: array of (c, z, y, x) rows} (nnunet_dataset.py:87-106).  Cases are
    uploaded once and stay resident."""

    def __init__(self, data, batch_size, patch_size, final_patch_size, label_manager, oversample_foreground_percent=0.0,
                 sampling_probabilities=None, pad_sides=None, probabilistic_oversampling=False, mirror_axes=None,
                 deep_supervision_scales=None, device="cuda:0"):
        self._data = data
        self.batch_size = int(batch_size)
        self.indices = list(data.keys())
        self.oversample_foreground_percent = oversample_foreground_percent
        self.final_patch_size = tuple(int(i) for i in final_patch_size)
        self.patch_size = tuple(int(i) for i in patch_size)
        if self.patch_size != self.final_patch_size:
            # the larger initial patch only exists to feed the rotation / scaling transform, which is not in this slice
            raise NotImplementedError("DeviceDataLoader3D: patch_size must equal final_patch_size (no SpatialTransform)")
        self.list_of_keys = list(data.keys())
        self.need_to_pad = (np.array(patch_size) - np.array(final_patch_size)).astype(int)  # base_data_loader.py:33
        if pad_sides is not None:
            self.need_to_pad += np.array(pad_sides)
        self.pad_sides = pad_sides
        self.sampling_probabilities = sampling_probabilities
        self.annotated_classes_key = tuple(label_manager.all_labels)
        self.has_ignore = label_manager.has_ignore_label
        self.get_do_oversample = self._oversample_last_XX_percent if not probabilistic_oversampling \
            else self._probabilistic_oversampling
        self.mirror_axes = tuple(mirror_axes) if mirror_axes else ()
        self.deep_supervision_scales = deep_supervision_scales
        self.device = torch.device(device)  # planning (plan_batch) is host logic; generate_train_batch needs a GPU
        self._resident = {}
        d0, s0, _ = self._case(self.indices[0])  # determine_shapes (:55-62)
        self.data_shape = (self.batch_size, d0.shape[0], *self.patch_size)
        self.seg_shape = (self.batch_size, s0.shape[0], *self.patch_size)

    # ------------------------------------------------------------------ residency
    def _case(self, key):
        hit = self._resident.get(key)
        if hit is None:
            data, seg, properties = self._data.load_case(key)
            data = torch.as_tensor(np.ascontiguousarray(data), dtype=torch.float32).to(self.device)
            seg = np.ascontiguousarray(seg)
            if seg.min() < -32768 or seg.max() > 32767:
                raise ValueError("segmentation labels must fit int16 (the reference batch buffer is int16)")
            seg = torch.as_tensor(seg.astype(np.int16)).to(self.device)
            hit = (data, seg, properties)
            self._resident[key] = hit
        return hit

    # ------------------------------------------------------------------ host logic (numpy RNG, reference call order)
    def get_indices(self):
        # batchgenerators DataLoader.get_indices, infinite=True
        return np.random.choice(self.indices, self.batch_size, replace=True, p=self.sampling_probabilities)

    def _oversample_last_XX_percent(self, sample_idx):
        return not sample_idx < round(self.batch_size * (1 - self.oversample_foreground_percent))

    def _probabilistic_oversampling(self, sample_idx):
        return np.random.uniform() < self.oversample_foreground_percent

    def _corner_range(self, data_shape):
        """(lowest, highest) admissible lower patch corner per axis.  A case smaller than the patch is padded on both
        sides (the odd voxel goes to the upper side); `need_to_pad` widens the range by the margin the reference keeps
        for its spatial transform."""
        shape = np.asarray(data_shape, dtype=np.int64)
        patch = np.asarray(self.patch_size, dtype=np.int64)
        margin = np.maximum(np.asarray(self.need_to_pad, dtype=np.int64), patch - shape)
        lowest = (-margin) // 2
        highest = shape + margin // 2 + margin % 2 - patch
        return lowest.tolist(), highest.tolist()

    def _centre_class(self, force_fg, class_locations, overwrite_class, verbose):
        """Which class (or region key) the patch must be centred on; None -> uniform corner.  One np.random.choice
        when a foreground class has to be drawn, none otherwise."""
        everything = self.annotated_classes_key
        if not force_fg:
            if not self.has_ignore:
                return None
            # ignore label present: stay inside the annotated area whenever the case has one
            if len(class_locations[everything]) == 0:
                print('Warning! No annotated pixels in image!')
                return None
            return everything
        if class_locations is None:
            raise AssertionError('if force_fg is set class_locations cannot be None')
        if overwrite_class is not None and overwrite_class not in class_locations.keys():
            raise AssertionError('desired class ("overwrite_class") does not have class_locations (missing key)')
        present = [k for k in class_locations.keys() if len(class_locations[k]) > 0]
        if len(present) > 1:
            # the all-annotated-classes key only serves cases that have nothing more specific
            for j, k in enumerate(present):
                if isinstance(k, tuple) and k == everything:
                    del present[j]
                    break
        if not present:
            if verbose:
                print('case does not contain any foreground classes')
            return None
        if overwrite_class is not None and overwrite_class in present:
            return overwrite_class
        return present[np.random.choice(len(present))]

    def get_bbox(self, data_shape, force_fg, class_locations, overwrite_class=None, verbose=False):
        """Lower / upper corner of the patch to cut from a case of spatial shape `data_shape` -- the sampler of
        nnUNetDataLoaderBase.get_bbox (base_data_loader.py:64-139) with the same arguments, result and numpy-RNG draw
        order (tests/golden/get_bbox.json holds boxes AND the RNG state after each call, generated by the reference
        method): centred on a random voxel of a random foreground class when `force_fg`, uniform otherwise."""
        lowest, highest = self._corner_range(data_shape)
        axes = range(len(data_shape))
        cls = self._centre_class(force_fg, class_locations, overwrite_class, verbose)
        voxels = class_locations[cls] if cls is not None else None
        if voxels is not None and len(voxels) > 0:
            centre = voxels[np.random.choice(len(voxels))]   # row = (channel, z, y, x)
            corner = [max(lowest[i], int(centre[i + 1]) - self.patch_size[i] // 2) for i in axes]
        else:
            corner = [int(np.random.randint(lowest[i], highest[i] + 1)) for i in axes]
        return corner, [corner[i] + self.patch_size[i] for i in axes]

    def draw_mirror(self):
        """MirrorTransform's per-sample draw (batchgenerators, absent from the reference tree -- restated from its
        published source: one uniform per listed axis, flip when < 0.5).  Returns the 3-bit axis mask."""
        mask = 0
        for ax in (0, 1, 2):
            if ax in self.mirror_axes and np.random.uniform() < 0.5:
                mask |= 1 << ax
        return mask

    # ------------------------------------------------------------------ the batch
    def plan_batch(self):
        """The host decisions of one batch, in the reference's RNG order: keys, then per sample (oversample?, bbox),
        then per sample the mirror draw."""
        keys = self.get_indices()
        boxes = []
        for j, k in enumerate(keys):
            force_fg = self.get_do_oversample(j)
            data, _, properties = self._case(k)
            lbs, _ = self.get_bbox(tuple(data.shape[1:]), force_fg, properties['class_locations'])
            boxes.append([int(v) for v in lbs])
        flips = [self.draw_mirror() for _ in keys]
        return list(keys), boxes, flips

    def generate_train_batch(self, plan=None):
        keys, boxes, flips = plan if plan is not None else self.plan_batch()
        data_all = torch.empty(self.data_shape, dtype=torch.float32, device=self.device)
        target = torch.empty(self.seg_shape, dtype=torch.float32, device=self.device)
        props = []
        for j, k in enumerate(keys):
            data, seg, properties = self._case(k)
            crop_pad_data(data, data_all[j], boxes[j], flips[j], 0.0)
            crop_pad_seg(seg, target[j], boxes[j], flips[j], -1, replace=(-1, 0))
            props.append(properties)
        if self.deep_supervision_scales is not None:
            target = [downsample_seg(target, s) for s in self.deep_supervision_scales]
        return {'data': data_all, 'target': target, 'properties': props, 'keys': keys}

    def __iter__(self):
        return self

    def __next__(self):
        return self.generate_train_batch()
