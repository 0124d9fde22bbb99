"""TEST INFRASTRUCTURE -- CPU restatement of the connected-component post-processing
(nnunetv2/postprocessing/remove_connected_components.py:22-34).  Only tests/ may import this.

PARITY UNPINNED for the component-selection rule: the function the reference calls,
acvl_utils.morphology.morphology_helper.remove_all_but_two_largest_component, is a fork-local addition to a
dependency that is not vendored in /root/reference (acvl_utils is not importable here and its published releases only
hold remove_all_but_largest_component).  The restatement follows that published sibling: label the mask with full
connectivity (skimage.measure.label default == 26 in 3-D), count voxels per component, keep the largest two.  The
reference's own tests hold no fixture for this path.  The labelling itself IS pinned: scipy.ndimage.label (an
independent implementation) must produce the same partition as oracle/cc_oracle.c (tests/test_oracle.py).
Ties in size: the component that comes first in scan order wins (python's max()/sorted() stability over labels
numbered in scan order, which is what skimage produces).
"""
import numpy as np
from scipy import ndimage


def label_with_component_sizes(mask, connectivity=26):
    structure = ndimage.generate_binary_structure(3, {6: 1, 18: 2, 26: 3}[connectivity])
    labeled, n = ndimage.label(mask, structure=structure)  # numbered in scan order of each component's first voxel
    sizes = np.bincount(labeled.reshape(-1), minlength=n + 1)
    return labeled, {i: int(sizes[i]) for i in range(1, n + 1)}


def remove_all_but_n_largest_component(mask, n_keep=2, connectivity=26):
    labeled, sizes = label_with_component_sizes(mask, connectivity)
    order = sorted(sizes.keys(), key=lambda i: (-sizes[i], i))
    keep = order[:n_keep]
    return np.isin(labeled, keep) & (labeled > 0), [sizes[k] for k in keep]


def region_or_label_to_mask(segmentation, region_or_label):
    if np.isscalar(region_or_label):
        return segmentation == region_or_label
    mask = np.zeros_like(segmentation, dtype=bool)
    for r in region_or_label:
        mask |= segmentation == r
    return mask


def remove_all_but_largest_component_from_segmentation(segmentation, labels_or_regions, background_label=0,
                                                       num_components=2, connectivity=26):
    mask = np.zeros_like(segmentation, dtype=bool)
    if not isinstance(labels_or_regions, list):
        labels_or_regions = [labels_or_regions]
    for l_or_r in labels_or_regions:
        mask |= region_or_label_to_mask(segmentation, l_or_r)
    mask_keep, _ = remove_all_but_n_largest_component(mask, num_components, connectivity)
    ret = np.copy(segmentation)
    ret[mask & ~mask_keep] = background_label
    return ret
