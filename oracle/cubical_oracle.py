"""TEST INFRASTRUCTURE -- CPU restatement of the cubical-complex persistence behind the reference's Betti metric.

What it follows in the reference: nnUNet/nnunetv2/training/metrics/betti_compute.py:8-40
(`compute_persistence_diagram(matrix, i=1)`: gudhi.CubicalComplex(dimensions=dims, top_dimensional_cells=matrix),
`persistence(homology_coeff_field=2, min_persistence=0)`, `persistence_intervals_in_dimension(1)`) and :42-53
(`betti_number`: zero the image border, count the dimension-1 intervals); cal_betti.py:18-49 (`getBetti`: per 65x65
window, |betti(prediction) - betti(ground truth)|).

PARITY UNPINNED: the algorithm lives in gudhi (third-party, not vendored, not importable in this image, no fixture in
the reference).  This file restates the published construction gudhi documents for `top_dimensional_cells`:
  * the complex of an h x w image has the pixels as 2-cells, their edges and corners as 1- and 0-cells;
  * a lower-dimensional cell enters the filtration with the SMALLEST value among the top-dimensional cells that contain
    it (lower-star filtration from the top cells);
  * persistence over Z/2 by the standard column reduction of the boundary matrix in filtration order (faces before
    cofaces at equal values); an interval (birth, death) is kept when death - birth > min_persistence = 0.
Pure Python loops: small images only (tests use <= 16 x 16)."""
import numpy as np


def _cells(h, w):
    """cells of the h x w pixel grid as (dim, i, j, orientation): vertices (h+1)(w+1); horizontal edges (h+1) x w
    (orientation 0: from vertex (i, j) to (i, j+1)); vertical edges h x (w+1) (orientation 1); squares h x w."""
    cells = []
    for i in range(h + 1):
        for j in range(w + 1):
            cells.append((0, i, j, 0))
    for i in range(h + 1):
        for j in range(w):
            cells.append((1, i, j, 0))
    for i in range(h):
        for j in range(w + 1):
            cells.append((1, i, j, 1))
    for i in range(h):
        for j in range(w):
            cells.append((2, i, j, 0))
    return cells


def _value(cell, img):
    """min over the pixels that contain the cell"""
    h, w = img.shape
    d, i, j, o = cell
    if d == 2:
        return float(img[i, j])
    if d == 0:
        px = [(i - 1, j - 1), (i - 1, j), (i, j - 1), (i, j)]
    elif o == 0:   # horizontal edge between rows i-1 and i, column j
        px = [(i - 1, j), (i, j)]
    else:          # vertical edge between columns j-1 and j, row i
        px = [(i, j - 1), (i, j)]
    return float(min(img[a, b] for a, b in px if 0 <= a < h and 0 <= b < w))


def _boundary(cell):
    d, i, j, o = cell
    if d == 0:
        return []
    if d == 1:
        return [(0, i, j, 0), (0, i, j + 1, 0)] if o == 0 else [(0, i, j, 0), (0, i + 1, j, 0)]
    return [(1, i, j, 0), (1, i + 1, j, 0), (1, i, j, 1), (1, i, j + 1, 1)]


def persistence_intervals(img, dim, min_persistence=0.0):
    """[(birth, death)] of the dimension-`dim` classes of the sublevel filtration of `img` (2-D array), essential
    classes with death = inf, intervals with death - birth <= min_persistence dropped (gudhi's convention)."""
    img = np.asarray(img, dtype=np.float64)
    h, w = img.shape
    cells = _cells(h, w)
    vals = [_value(c, img) for c in cells]
    order = sorted(range(len(cells)), key=lambda k: (vals[k], cells[k][0], k))
    pos = {cells[k]: r for r, k in enumerate(order)}
    low_to_col = {}
    paired = set()
    out = []
    cols = {}
    for r, k in enumerate(order):
        col = set(pos[b] for b in _boundary(cells[k]))
        while col:
            lo = max(col)
            if lo not in low_to_col:
                break
            col ^= cols[low_to_col[lo]]
        if col:
            lo = max(col)
            low_to_col[lo] = r
            cols[r] = col
            paired.add(lo)
            paired.add(r)
            kb = order[lo]
            if cells[kb][0] == dim and vals[k] - vals[kb] > min_persistence:
                out.append((vals[kb], vals[k]))
    for r, k in enumerate(order):
        if r not in paired and cells[k][0] == dim:
            out.append((vals[k], float("inf")))
    return sorted(out)


def betti_number(imagely):
    """betti_compute.py:42-53: zero the border, count the dimension-1 intervals"""
    a = np.array(imagely, dtype=np.float64, copy=True)
    a[-1, :] = 0
    a[:, -1] = 0
    a[0, :] = 0
    a[:, 0] = 0
    return len(persistence_intervals(a, 1))
