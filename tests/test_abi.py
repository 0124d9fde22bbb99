"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol include/mvdseg_hip.h
declares, with the arity the ctypes binding assumes.  No compute call is made (no GPU here)."""
import os
import re

import pytest

from conftest import ROOT
from multimodal_mvd_seg_amd import _lib

HEADER = os.path.join(ROOT, "include", "mvdseg_hip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)  # strip comments
    decls = {}
    for m in re.finditer(r"\b(?:int|long|size_t|const char \*)\s*\*?\s*(mvd_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[name] = n
    return decls


def test_header_declares_the_whole_path():
    d = _declared()
    for must in ("mvd_conv3d_fwd", "mvd_conv3d_dgrad", "mvd_conv3d_wgrad", "mvd_convT3d_fwd", "mvd_convT3d_dgrad",
                 "mvd_convT3d_wgrad", "mvd_instnorm_lrelu_fwd", "mvd_instnorm_lrelu_bwd", "mvd_seghead_fwd",
                 "mvd_seghead_bwd", "mvd_dcce_fwd", "mvd_dcce_bwd", "mvd_kl_fwd", "mvd_kl_bwd", "mvd_soft_erode_fwd",
                 "mvd_soft_dilate_fwd", "mvd_skel_update_fwd", "mvd_cc_label", "mvd_sgd_nesterov_step",
                 "mvd_h0_sorted_edges", "mvd_h0_pair_host", "mvd_mse_fwd", "mvd_last_error", "mvd_version"):
        assert must in d, must


def test_binding_matches_header():
    d = _declared()
    assert set(d) == set(_lib.SIGNATURES), (set(d) ^ set(_lib.SIGNATURES))
    for name, n in d.items():
        assert len(_lib.SIGNATURES[name][1]) == n, f"{name}: header has {n} args, binding {len(_lib.SIGNATURES[name][1])}"


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libmvdseg_hip.so not built (run __graft_entry__.build())")
def test_library_loads_and_exports_every_symbol():
    lib = _lib.load()  # resolves every symbol of SIGNATURES (AttributeError otherwise)
    assert lib.mvd_version() == 100
    assert lib.mvd_has_mfma() == 1
    assert isinstance(lib.mvd_last_error(), bytes)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libmvdseg_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from multimodal_mvd_seg_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.to_ndhwc(torch.zeros(1, 2, 3, 3, 3))


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libmvdseg_hip.so not built (run __graft_entry__.build())")
def test_shape_queries_and_selector_errors_are_host_logic():
    """The eligibility queries of the fused block (round 3) are pure host arithmetic -- what network.StackedConvBlocks asks
    before anything runs -- and the kernel selectors reject bad values with an error message (no compute call here)."""
    from multimodal_mvd_seg_amd._lib import i3
    lib = _lib.load()
    k3, s1, s2 = i3((3, 3, 3)), i3((1, 1, 1)), i3((2, 2, 2))
    # weight-gradient loader prologue (k_wgrad16z): plain 3x3x3 stride 1, ONE producer tensor, channels in blocks of 32, W >= 32
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(2, 128, 128, 128, 32, 0, 32, k3, s1) == 1
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(2, 64, 64, 64, 64, 0, 64, k3, s1) == 1
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(1, 9, 8, 32, 32, 0, 64, k3, s1) == 1
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(2, 128, 128, 16, 32, 0, 32, k3, s1) == 0     # narrower than a column
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(2, 128, 128, 128, 32, 32, 32, k3, s1) == 0   # two producer tensors
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(2, 128, 128, 128, 32, 0, 64, k3, s2) == 0    # strided
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(2, 128, 128, 128, 24, 0, 32, k3, s1) == 0    # not a multiple of 32 channels
    assert lib.mvd_conv3d_wgrad_bf16_prologue_ok(0, 128, 128, 128, 32, 0, 32, k3, s1) == 0    # empty batch
    # forward loader prologue (k_fwd16y): one producer tensor of 32 or 64 channels on a large volume
    assert lib.mvd_conv3d_fwd_bf16_prologue_ok(2, 128, 128, 128, 32, 0, 32, k3, s1) == 1
    assert lib.mvd_conv3d_fwd_bf16_prologue_ok(2, 64, 64, 64, 64, 0, 64, k3, s1) == 1
    assert lib.mvd_conv3d_fwd_bf16_prologue_ok(2, 128, 128, 128, 32, 32, 32, k3, s1) == 0
    assert lib.mvd_conv3d_fwd_bf16_prologue_ok(2, 128, 128, 128, 32, 0, 64, k3, s2) == 0
    assert lib.mvd_conv3d_fwd_bf16_prologue_ok(2, 8, 8, 8, 32, 0, 32, k3, s1) == 0            # too few tiles for the kernel
    # statistics epilogue: tiles per sample of the kernel that would run, 0 when it has none
    assert lib.mvd_conv3d_fwd_bf16_stats_tiles(2, 128, 128, 128, 32, 0, 32, k3, s1) > 0
    assert lib.mvd_conv3d_fwd_bf16_stats_tiles(2, 8, 8, 8, 320, 0, 320, k3, s1) == 0
    # selectors
    for fn, bad in ((lib.mvd_set_bf16_wgrad_kernel, 5), (lib.mvd_set_bf16_zmarch_kernel, 7)):
        assert fn(bad) == 2 and len(lib.mvd_last_error()) > 10
        assert fn(1) == 0
