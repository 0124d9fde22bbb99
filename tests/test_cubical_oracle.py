"""CPU tests of oracle/cubical_oracle.py (the restated cubical persistence behind betti_compute.py) on cases whose
answers are known by hand -- the reference's library (gudhi) is absent, so these known answers and the structural
identity below are what the restatement is held to (PARITY UNPINNED against gudhi itself)."""
import numpy as np
import pytest
from scipy import ndimage

from oracle import cubical_oracle as co


def test_ring_has_one_loop_born_at_the_wall_and_filled_at_the_centre():
    img = np.zeros((7, 7))
    img[2:5, 2:5] = 1.0           # a plateau ...
    img[3, 3] = 3.0               # ... with a peak: sublevel loop around the plateau appears at 0, is filled at 3
    assert co.persistence_intervals(img, 1) == [(0.0, 3.0)]
    # dimension 0: one essential component born at 0
    assert co.persistence_intervals(img, 0) == [(0.0, float("inf"))]


def test_nested_and_separate_maxima():
    img = np.zeros((9, 13))
    img[2:7, 2:6] = 2.0
    img[4, 3] = 5.0
    img[3:6, 8:11] = 4.0
    iv = co.persistence_intervals(img, 1)
    assert iv == [(0.0, 4.0), (0.0, 5.0)]


def test_diagonal_pixels_are_separate_holes():
    """two foreground pixels touching in a corner: the background pixels around the corner are closed cells, so each
    foreground pixel is its own hole (4-connectivity of the foreground)"""
    img = np.zeros((6, 6))
    img[2, 2] = 1.0
    img[3, 3] = 1.0
    assert co.persistence_intervals(img, 1) == [(0.0, 1.0), (0.0, 1.0)]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_binary_masks_count_4_connected_foreground_components(seed):
    rng = np.random.default_rng(seed)
    m = (rng.random((14, 16)) > 0.6).astype(np.float64)
    crop = m.copy()
    crop[0, :] = crop[-1, :] = 0
    crop[:, 0] = crop[:, -1] = 0
    _, n = ndimage.label(crop, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    assert co.betti_number(m) == n


def test_euler_characteristic_identity_on_random_images():
    """at every threshold: #components - #loops of the sublevel set (from the intervals) = Euler characteristic of the
    closed-pixel complex (vertices - edges + squares), counted directly"""
    rng = np.random.default_rng(5)
    img = np.round(rng.random((8, 9)) * 6) / 6
    d0 = co.persistence_intervals(img, 0, min_persistence=-1)
    d1 = co.persistence_intervals(img, 1, min_persistence=-1)
    h, w = img.shape
    for a in sorted(set(img.ravel())):
        alive0 = sum(1 for b, d in d0 if b <= a < d)
        alive1 = sum(1 for b, d in d1 if b <= a < d)
        sq = img <= a
        V = np.zeros((h + 1, w + 1), bool)
        Eh = np.zeros((h + 1, w), bool)
        Ev = np.zeros((h, w + 1), bool)
        for i in range(h):
            for j in range(w):
                if sq[i, j]:
                    V[i:i + 2, j:j + 2] = True
                    Eh[i:i + 2, j] = True
                    Ev[i, j:j + 2] = True
        assert alive0 - alive1 == int(V.sum()) - int(Eh.sum()) - int(Ev.sum()) + int(sq.sum())
