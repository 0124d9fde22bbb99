"""GPU tests of multimodal_mvd_seg_amd.metrics (the Betti metric of betti_compute.py / cal_betti.py through the device
H0 pairing) against oracle/cubical_oracle.py -- PARITY UNPINNED: gudhi, the reference's library for these functions,
is absent and the reference holds no fixture; the oracle restates the published construction and is itself held to
hand-checked cases (tests/test_cubical_oracle.py)."""
import numpy as np
import pytest
import torch
from scipy import ndimage

from oracle import cubical_oracle as co

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.mark.parametrize("shape,levels,seed", [((9, 11), 5, 0), ((12, 12), 3, 1), ((7, 16), 50, 2), ((16, 9), 2, 3)])
def test_dim1_intervals_equal_the_cubical_oracle(shape, levels, seed):
    """random images quantised to a few levels (heavy ties): the multiset of (birth, death) with death > birth must be
    identical -- comparisons only, no arithmetic: exact"""
    from multimodal_mvd_seg_amd import metrics
    rng = np.random.default_rng(seed)
    img = (np.round(rng.random(shape) * levels) / levels).astype(np.float32)
    got = metrics.persistence_intervals_dim1(torch.from_numpy(img).to(DEV))
    ref = co.persistence_intervals(img, 1)
    assert [(float(b), float(d)) for b, d in got.tolist()] == [(float(np.float32(b)), float(np.float32(d))) for b, d in ref]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_betti_number_binary_windows(seed):
    from multimodal_mvd_seg_amd import metrics
    rng = np.random.default_rng(seed)
    m = (rng.random((65, 65)) > 0.55).astype(np.float32)
    crop = m.copy()
    crop[0, :] = crop[-1, :] = 0
    crop[:, 0] = crop[:, -1] = 0
    _, n = ndimage.label(crop, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    t = torch.from_numpy(m).to(DEV)
    assert metrics.betti_number(t) == n
    assert torch.equal(t.cpu(), torch.from_numpy(m))            # the input is not modified (the reference clones it)
    small = m[:13, :15]
    assert metrics.betti_number(torch.from_numpy(small.copy())) == co.betti_number(small)   # CPU tensor in: moved


def test_get_betti_errors_window_walk():
    from multimodal_mvd_seg_amd import metrics
    rng = np.random.default_rng(9)
    pred = (rng.random((100, 140)) > 0.5).astype(np.float32)
    gt = (rng.random((100, 140)) > 0.5).astype(np.float32)
    errs = metrics.get_betti_errors(torch.from_numpy(pred).to(DEV), torch.from_numpy(gt).to(DEV), 65)
    assert len(errs) == 2 * 3

    def count(a):
        c = a.copy()
        c[0, :] = c[-1, :] = 0
        c[:, 0] = c[:, -1] = 0
        return ndimage.label(c, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])[1]
    want = [abs(count(pred[y:y + 65, x:x + 65]) - count(gt[y:y + 65, x:x + 65])) for y in (0, 65) for x in (0, 65, 130)]
    assert errs == want
