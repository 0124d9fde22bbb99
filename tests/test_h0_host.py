"""CPU tests of the HOST half of the H0 pairing (mvd_h0_pair_host in libmvdseg_hip.so is plain C++; the device half --
edge keys + radix sort -- is exercised by the -m gpu tests).  The edge keys are built here in numpy with the formula of
csrc/persist.hip::k_h0_edge_keys and sorted with numpy, then paired by the library and compared with the reference's own
C++ diagrams (tests/golden/persistence_grid.json, multisets) and with oracle/cc_oracle.c (element for element)."""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from multimodal_mvd_seg_amd import _lib
from oracle import cc_oracle

OFFS = {6: [(0, 0, 1), (0, 1, 0), (1, 0, 0)],
        14: [(0, 0, 1), (0, 1, 0), (1, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1)],
        26: [(dz, dy, dx) for dz in (0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)
             if not (dz == 0 and (dy < 0 or (dy == 0 and dx <= 0)))]}


def _ord(v):
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    return np.where(u & 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)


def host_sorted_keys(f, conn, sublevel=True):
    D, H, W = f.shape
    g = (f if sublevel else -f).astype(np.float32)
    g = np.where(g == 0, np.float32(0), g)
    offs = OFFS[conn]
    z, y, x = np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij")
    keys = []
    for k, (dz, dy, dx) in enumerate(offs):
        zz, yy, xx = z + dz, y + dy, x + dx
        ok = (zz >= 0) & (zz < D) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = ((z * H + y) * W + x)[ok]
        u = ((zz * H + yy) * W + xx)[ok]
        val = np.maximum(g.reshape(-1)[v], g.reshape(-1)[u])
        keys.append((_ord(val) << np.uint64(32)) | (v.astype(np.uint64) * np.uint64(len(offs)) + np.uint64(k)))
    return np.sort(np.concatenate(keys))


def pair(f, conn, sublevel=True):
    lib = _lib.load()
    D, H, W = f.shape
    f = np.ascontiguousarray(f, dtype=np.float32)
    keys = np.ascontiguousarray(host_sorted_keys(f, conn, sublevel))
    assert lib.mvd_h0_num_edges(D, H, W, conn) == keys.size
    death = np.empty(f.size, dtype=np.float32)
    dv = np.empty(f.size, dtype=np.int64)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    n = lib.mvd_h0_pair_host(P(f), P(keys), keys.size, D, H, W, conn, int(sublevel), P(death), P(dv))
    assert n >= 0, lib.mvd_last_error()
    return f.reshape(-1).copy(), death, dv, n


def _sorted_pairs(b, d):
    a = np.stack([b, d], 1)
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def test_host_pairing_equals_reference_cpp_diagrams():
    d = json.load(open(os.path.join(GOLDEN, "persistence_grid.json")))
    assert "reference C++" in d["source"]
    for case in d["cases"]:
        f = np.asarray(case["f"], dtype=np.float32).reshape(case["shape"])
        b, de, dv, n = pair(f, case["conn"])
        want = np.array([[x, np.inf if y is None else y] for x, y in case["dgm0_sorted"]], dtype=np.float32)
        assert np.array_equal(_sorted_pairs(b, de), want), case["shape"]
        assert n == 1  # a grid is connected: one essential bar


@pytest.mark.parametrize("conn", [6, 14, 26])
@pytest.mark.parametrize("sublevel", [True, False])
def test_host_pairing_equals_c_oracle_elementwise(conn, sublevel):
    rng = np.random.default_rng(conn)
    for shape, ties in (((7, 9, 8), False), ((6, 5, 11), True), ((1, 12, 13), True), ((1, 1, 17), False)):
        f = rng.standard_normal(shape).astype(np.float32)
        if ties:
            f = np.round(f * 2) / 2  # many equal values, zeros of both signs after the super-level flip
        b, de, dv, n = pair(f, conn, sublevel)
        ob, ode, odv = cc_oracle.h0_persistence(f, conn, sublevel)
        assert np.array_equal(b, ob) and np.array_equal(de, ode) and np.array_equal(dv, odv), (shape, ties)
        assert n == 1 and int(np.isinf(de).sum()) == 1


def test_host_pairing_rejects_inconsistent_input():
    lib = _lib.load()
    f = np.zeros((2, 2, 2), dtype=np.float32)
    keys = np.zeros(5, dtype=np.uint64)
    death, dv = np.empty(8, dtype=np.float32), np.empty(8, dtype=np.int64)
    P = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert lib.mvd_h0_pair_host(P(f), P(keys), 5, 2, 2, 2, 6, 1, P(death), P(dv)) < 0   # 12 edges expected
    assert lib.mvd_h0_num_edges(2, 2, 2, 7) < 0
