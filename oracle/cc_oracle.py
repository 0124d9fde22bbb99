"""TEST INFRASTRUCTURE -- ctypes front-end of oracle/cc_oracle.c (built by oracle/Makefile)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_cc.so")


def build():
    src = os.path.join(_HERE, "cc_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", _SO, src, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.mvd_oracle_cc_label.restype = ctypes.c_int64
        _lib.mvd_oracle_h0_persistence.restype = ctypes.c_int64
    return _lib


def cc_label(mask, conn=6):
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    assert mask.ndim == 3
    labels = np.empty(mask.shape, dtype=np.int32)
    D, H, W = mask.shape
    n = lib().mvd_oracle_cc_label(mask.ctypes.data_as(ctypes.c_void_p), D, H, W, conn,
                                  labels.ctypes.data_as(ctypes.c_void_p))
    return labels, int(n)


def h0_persistence(f, conn=6, sublevel=True):
    """Returns (birth[N], death[N], death_vertex[N]) for every vertex of the D x H x W grid."""
    f = np.ascontiguousarray(f, dtype=np.float32)
    assert f.ndim == 3
    g = f if sublevel else -f
    death = np.empty(f.size, dtype=np.float32)
    dv = np.empty(f.size, dtype=np.int64)
    D, H, W = f.shape
    lib().mvd_oracle_h0_persistence(g.ctypes.data_as(ctypes.c_void_p), D, H, W, conn,
                                    death.ctypes.data_as(ctypes.c_void_p), dv.ctypes.data_as(ctypes.c_void_p))
    birth = f.reshape(-1).copy()
    if not sublevel:
        death = -death
    return birth, death, dv
