"""Topology metrics of the reference's evaluation scripts on the device H0 machinery (csrc/persist.hip).

Reference: nnUNet/nnunetv2/training/metrics/betti_compute.py:8-53 (`compute_persistence_diagram(matrix, i=1)` through
gudhi's CubicalComplex with the pixels as top-dimensional cells, `betti_number`) and cal_betti.py:18-49 (`getBetti`).
gudhi is not available here and the reference holds no fixture for these functions: parity is against
oracle/cubical_oracle.py, a restatement of the published construction (PARITY UNPINNED, tests/test_gpu_metrics.py).

How a dimension-1 diagram comes out of an H0 computation (Alexander duality in the plane): in the sublevel filtration
of an image whose pixels are closed 2-cells, a 1-cycle at threshold a surrounds a bounded component of the complement,
and the complement of {pixels <= a} is {pixels > a} joined through shared EDGES (two pixels that touch only in a corner
are separated by the closed pixels around that corner) together with everything outside the image.  Walking a
downwards, the components of {pixels > a} are born at their maxima and merge at saddles, the younger one dying (elder
rule): that is H0 of the super-level filtration with 4-connectivity -- ops.h0_persistence(conn=6) on a [1, H, W] grid --
with a ring of +inf pixels standing for the outside.  A bar (born m, merged at s) of that diagram is the 1-cycle that
appears at a = s (the hole separates from an older hole or from the outside) and is filled at a = m: the
interval [s, m).  The outside component never dies: no interval.  The hard-skeleton clDice of clDice_metric.py:7-36
needs skimage's `skeletonize_3d` (Lee's sequential thinning; absent here and not restated): not provided."""
import torch
import torch.nn.functional as F

from . import ops


def persistence_intervals_dim1(matrix):
    """[n, 2] CPU tensor of (birth, death) rows: the dimension-1 intervals with death > birth of the sublevel cubical
    filtration of a 2-D image (`compute_persistence_diagram(matrix, i=1)`, betti_compute.py:8-40; row order unspecified
    there, sorted by (birth, death) here).  `matrix`: [H, W] tensor (any float dtype; moved to the GPU if needed)."""
    if matrix.dim() != 2:
        raise ValueError("persistence_intervals_dim1 takes a 2-D image")
    x = matrix.detach().to(dtype=torch.float32)
    if not x.is_cuda:
        x = x.cuda()
    f = F.pad(x[None, None], (1, 1, 1, 1), value=float("inf"))[0].contiguous()   # [1, H+2, W+2], the ring = the outside
    birth, death, _ = ops.h0_persistence(f, 6, False)                            # super-level, 4-connectivity
    keep = torch.isfinite(birth) & torch.isfinite(death) & (birth > death)
    iv = torch.stack([death[keep], birth[keep]], 1)
    if iv.shape[0] > 1:
        order = sorted(range(iv.shape[0]), key=lambda k: (float(iv[k, 0]), float(iv[k, 1])))
        iv = iv[order]
    return iv


def betti_number(imagely):
    """betti_compute.py:42-53: a copy of the 2-D image with its border rows and columns set to 0, then the number of
    dimension-1 intervals.  (For a binary mask this is the number of 4-connected foreground components of the cropped
    mask: every one of them is a hole of the background at threshold 0, filled at 1.)"""
    a = imagely.detach().clone().to(torch.float32)
    a[-1, :] = 0
    a[:, -1] = 0
    a[0, :] = 0
    a[:, 0] = 0
    return int(persistence_intervals_dim1(a).shape[0])


def get_betti_errors(binary_predict, masks, topo_size=65):
    """cal_betti.py:18-49 `getBetti`: |betti_number(prediction window) - betti_number(ground-truth window)| for every
    topo_size x topo_size window of a 2-D prediction / mask pair, in the reference's window order (rows of windows first)."""
    errs = []
    H, W = masks.shape[0], masks.shape[1]
    for y in range(0, H, topo_size):
        for x in range(0, W, topo_size):
            b = binary_predict[y:min(y + topo_size, H), x:min(x + topo_size, W)]
            g = masks[y:min(y + topo_size, H), x:min(x + topo_size, W)]
            errs.append(abs(betti_number(b) - betti_number(g)))
    return errs
