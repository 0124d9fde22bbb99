"""Connected-component post-processing of predicted segmentations on the MI355X.

Host-side mirror of nnunetv2/postprocessing/remove_connected_components.py:22-42 (same function names, argument
meaning and "do not modify the input" behaviour).  The reference builds a boolean mask on the CPU, calls
acvl_utils.morphology.morphology_helper.remove_all_but_two_largest_component (a dependency absent from
/root/reference; its published sibling remove_all_but_largest_component labels with skimage.measure.label at full
connectivity and keeps the component(s) with the largest voxel count) and rewrites the rest to the background label.
Here the whole chain runs in HIP kernels through the C-ABI (mvd_seg_label_mask -> mvd_cc_label(conn=26) ->
mvd_cc_keep_largest -> mvd_seg_remove_components); there is no CPU fallback.
"""
import numpy as np
import torch

from . import ops


def _flatten_labels(labels_or_regions):
    # region_or_label_to_mask (evaluate_predictions.py): an int selects one label, a tuple a union of labels
    if not isinstance(labels_or_regions, list):
        labels_or_regions = [labels_or_regions]
    flat = []
    for l_or_r in labels_or_regions:
        if isinstance(l_or_r, (tuple, list)):
            flat.extend(int(v) for v in l_or_r)
        else:
            flat.append(int(l_or_r))
    out = []
    for v in flat:
        if v not in out:
            out.append(v)
    if not 1 <= len(out) <= 16:
        raise ValueError("labels_or_regions must name between 1 and 16 distinct labels")
    return out


def remove_all_but_largest_component_from_segmentation(segmentation, labels_or_regions, background_label=0,
                                                       num_components=2, connectivity=26, device=None):
    """remove_connected_components.py:22-34.  segmentation: [D,H,W] integer volume (numpy array or torch tensor;
    a numpy input returns numpy, a device tensor returns a device tensor).  Keeps the `num_components` largest
    connected components (the fork's default is the TWO largest, :31) of the union of `labels_or_regions`."""
    is_np = isinstance(segmentation, np.ndarray)
    if is_np:
        dev = torch.device(device or "cuda:0")
        seg = torch.from_numpy(np.ascontiguousarray(segmentation)).to(dev)
    else:
        seg = segmentation
        if not seg.is_cuda:
            raise RuntimeError("postprocessing runs on the GPU: pass a device tensor or a numpy array")
    if seg.dim() != 3:
        raise RuntimeError("segmentation must be [D,H,W]")
    in_dtype = seg.dtype
    seg32 = seg.to(torch.int32).contiguous()
    mask = ops.seg_label_mask(seg32, _flatten_labels(labels_or_regions))
    cc, _ = ops.cc_label(mask, conn=connectivity)
    kept = ops.cc_keep_largest(cc, keep=num_components)
    out = ops.seg_remove_components(seg32, cc, kept, background_label).to(in_dtype)
    return out.cpu().numpy() if is_np else out


def apply_postprocessing(segmentation, pp_fns, pp_fn_kwargs):
    """remove_connected_components.py:37-42"""
    for fn, kwargs in zip(pp_fns, pp_fn_kwargs):
        segmentation = fn(segmentation, **kwargs)
    return segmentation
