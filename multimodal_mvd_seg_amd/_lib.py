"""ctypes binding of libmvdseg_hip.so (the C ABI declared in include/mvdseg_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised
(the reference treats RuntimeError as "OOM-like", nnUNetTrainer.py:1187, nnUNetTrainerBenchmark_5epochs.py:28).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvdseg_hip.so")

_P = c_void_p
_I3 = ctypes.c_int * 3

# name -> (restype, argtypes); must list every symbol of include/mvdseg_hip.h (tests/test_abi.py checks it)
SIGNATURES = {
    "mvd_version": (c_int, []),
    "mvd_last_error": (ctypes.c_char_p, []),
    "mvd_has_mfma": (c_int, []),
    "mvd_set_conv_engine": (c_int, [c_int]),
    "mvd_set_wino_min_items": (c_int, [c_long]),
    "mvd_pack_weight": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mvd_conv_fwd_workspace_bytes": (c_size_t, [c_int, c_long, c_int]),
    "mvd_conv3d_fwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                               c_size_t, _P]),
    "mvd_conv3d_dgrad": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                 c_size_t, _P]),
    "mvd_conv3d_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mvd_conv3d_wgrad": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                 c_size_t, _P]),
    "mvd_convT3d_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _P, c_size_t, _P]),
    "mvd_convT3d_dgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _P, c_size_t, _P]),
    "mvd_convT3d_wgrad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]),
    "mvd_convT3d_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _P, c_size_t, _P]),
    "mvd_conv_wino_applicable": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3]),
    "mvd_wino_mode": (c_int, []),
    "mvd_wino_weight_elems": (c_size_t, [c_int, c_int]),
    "mvd_pack_weight_wino": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "mvd_pack_weights_batch": (c_int, [c_int] + [_P] * 9 + [_P]),
    "mvd_conv3d_fwd_wino": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                    c_size_t, _P]),
    "mvd_conv_stats_tiles": (c_size_t, [c_int, c_int, c_int]),
    "mvd_conv3d_fwd_wino_stats": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                          _I3, _I3, _P, c_size_t, _P]),
    "mvd_conv3d_dgrad_wino": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                      c_size_t, _P]),
    "mvd_conv3d_wgrad_bf16": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                      c_size_t, _P]),
    "mvd_convT3d_wgrad_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _P, c_size_t,
                                       _P]),
    "mvd_instnorm_lrelu_fwd_prestats": (c_int, [_P, _P, c_long, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_float, c_float,
                                                _P, c_size_t, _P]),
    "mvd_conv3d_fwd_bf16_stats_tiles": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3]),
    "mvd_set_bf16_zmarch_kernel": (c_int, [c_int]),
    "mvd_set_bf16_wgrad_kernel": (c_int, [c_int]),
    "mvd_conv3d_dgrad_acc_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3]),
    "mvd_conv3d_dgrad_acc": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P]),
    "mvd_conv3d_dgrad_bf16_acc": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P, c_size_t, _P]),
    "mvd_conv3d_fwd_bf16_prologue_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3]),
    "mvd_conv3d_wgrad_bf16_prologue_ok": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3]),
    "mvd_conv3d_wgrad_bf16_fused": (c_int, [_P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P, _P, c_float,
                                            _P, c_size_t, _P]),
    "mvd_instnorm_stats_bf16": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_float, _P, c_size_t, _P]),
    "mvd_conv3d_fwd_bf16_fused": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                          _P, c_float, _P, _P, _P, c_size_t, _P]),
    "mvd_instnorm_finalize_tiles": (c_int, [_P, c_long, _P, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_float, _P]),
    "mvd_instnorm_lrelu_apply_bf16": (c_int, [_P, _P, _P, _P, c_int, c_long, c_int, c_float, _P]),
    "mvd_instnorm_lrelu_fwd_bf16": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_float, c_float, _P,
                                            c_size_t, _P]),
    "mvd_instnorm_lrelu_bwd_bf16": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_float, _P,
                                            c_size_t, _P]),
    "mvd_seghead_fwd_bf16": (c_int, [_P, _P, _P, _P, c_int, c_long, c_int, c_int, _P]),
    "mvd_seghead_bwd_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mvd_seghead_bf16_fused_ok": (c_int, [c_int, c_long, c_int, c_int]),
    "mvd_seghead_fwd_fused": (c_int, [_P, _P, _P, _P, _P, c_float, _P, _P, _P, c_int, c_long, c_int, c_int, _P]),
    "mvd_seghead_bwd_fused": (c_int, [_P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_int, c_int, _P,
                                      c_size_t, _P]),
    "mvd_instnorm_stats_from_tiles": (c_int, [_P, c_long, _P, _P, c_int, c_long, c_int, c_float, _P, c_size_t, _P]),
    "mvd_seghead_fwd_bf16_fused": (c_int, [_P, _P, _P, c_float, _P, _P, _P, c_int, c_long, c_int, c_int, _P]),
    "mvd_seghead_bwd_bf16_fused": (c_int, [_P, _P, _P, c_float, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_int, c_int, _P,
                                           c_size_t, _P]),
    "mvd_cast_f32_to_bf16": (c_int, [_P, _P, c_long, _P]),
    "mvd_cast_bf16_to_f32": (c_int, [_P, _P, c_long, _P]),
    "mvd_pack_weight_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mvd_pack_weights_bf16_batch": (c_int, [c_int] + [_P] * 7 + [_P]),
    "mvd_conv3d_fwd_bf16": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                    c_size_t, _P]),
    "mvd_conv3d_dgrad_bf16": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _I3, _P,
                                      c_size_t, _P]),
    "mvd_convT3d_fwd_bf16": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _P, c_size_t, _P]),
    "mvd_convT3d_dgrad_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _I3, _P, c_size_t, _P]),
    "mvd_instnorm_nblk": (c_int, [c_int, c_long, c_int]),
    "mvd_instnorm_workspace_bytes": (c_size_t, [c_int, c_long, c_int]),
    "mvd_instnorm_lrelu_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_float, c_float, _P, c_size_t, _P]),
    "mvd_instnorm_lrelu_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_float, _P, c_size_t,
                                       _P]),
    "mvd_seghead_fwd": (c_int, [_P, _P, _P, _P, c_int, c_long, c_int, c_int, _P]),
    "mvd_seghead_bwd_workspace_bytes": (c_size_t, [c_int, c_long, c_int, c_int]),
    "mvd_seghead_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_long, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mvd_dcce_workspace_bytes": (c_size_t, [c_int, c_long, c_int]),
    "mvd_dcce_fwd": (c_int, [_P, _P, _P, c_int, c_long, c_int, _P, c_size_t, _P]),
    "mvd_dcce_finalize": (c_int, [_P, c_int, _P, c_int, _P, _P, c_long, c_int, c_int, c_int, c_float, c_float, c_float,
                                  _P]),
    "mvd_dcce_bwd": (c_int, [_P, _P, _P, _P, c_float, _P, c_int, c_long, c_int, c_float, _P]),
    "mvd_argmax_counts": (c_int, [_P, _P, _P, c_int, c_long, c_int, _P]),
    "mvd_softmax_select_fwd": (c_int, [_P, _P, c_int, c_long, c_int, c_int, _P]),
    "mvd_softmax_select_bwd": (c_int, [_P, _P, _P, c_int, c_long, c_int, c_int, _P]),
    "mvd_label_mask": (c_int, [_P, _P, c_long, c_float, _P]),
    "mvd_cldice_combine": (c_int, [_P, _P, c_float, _P]),
    "mvd_kl_workspace_bytes": (c_size_t, [c_int, c_long]),
    "mvd_kl_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_long, c_long, c_long, c_long, c_float, c_float, c_int, _P,
                           c_size_t, _P]),
    "mvd_kl_bwd": (c_int, [_P, _P, _P, c_float, _P, _P, c_int, c_int, c_long, c_long, c_long, c_long, c_float, c_float,
                           c_int, _P]),
    "mvd_kl_fwd_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_long, c_float, c_float, _P, c_size_t, _P]),
    "mvd_kl_bwd_bf16": (c_int, [_P, _P, _P, c_float, _P, _P, c_int, c_int, c_long, c_float, c_float, _P]),
    "mvd_mse_workspace_bytes": (c_size_t, [c_long]),
    "mvd_mse_fwd": (c_int, [_P, _P, _P, c_long, _P, c_size_t, _P]),
    "mvd_mse_bwd": (c_int, [_P, _P, _P, _P, _P, c_long, _P]),
    "mvd_soft_erode_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mvd_soft_erode_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mvd_soft_dilate_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mvd_soft_dilate_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mvd_skel_update_fwd": (c_int, [_P, _P, _P, _P, c_long, c_int, _P]),
    "mvd_skel_update_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_long, c_int, _P]),
    "mvd_skel_iter_fwd": (c_int, [_P] * 8 + [c_int] * 5 + [_P]),
    "mvd_dot_sum": (c_int, [_P, _P, _P, c_long, _P, c_size_t, _P]),
    "mvd_dot_sum_workspace_bytes": (c_size_t, [c_long]),
    "mvd_cc_label": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mvd_threshold_mask": (c_int, [_P, _P, c_long, c_float, c_int, _P]),
    "mvd_h0_num_edges": (c_long, [c_int, c_int, c_int, c_int]),
    "mvd_h0_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "mvd_h0_sorted_edges": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "mvd_h0_pair_host": (c_long, [_P, _P, c_long, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "mvd_seg_label_mask": (c_int, [_P, _P, c_long, _P, c_int, _P]),
    "mvd_cc_keep_workspace_bytes": (c_size_t, [c_long]),
    "mvd_cc_keep_largest": (c_int, [_P, c_long, c_int, _P, _P, _P]),
    "mvd_seg_remove_components": (c_int, [_P, _P, _P, _P, c_long, c_int, _P]),
    "mvd_sumsq_workspace_bytes": (c_size_t, [c_long]),
    "mvd_grad_sumsq": (c_int, [_P, _P, c_long, _P, c_size_t, _P]),
    "mvd_sgd_nesterov_step": (c_int, [_P, _P, _P, _P, c_long, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "mvd_sgd_nesterov_step_dev": (c_int, [_P, _P, _P, _P, c_long, _P, _P]),
    "mvd_nchw_to_ndhwc": (c_int, [_P, _P, c_int, c_int, c_long, _P]),
    "mvd_pad_channels_bf16": (c_int, [_P, _P, c_int, c_int, c_int, c_long, c_int, _P]),
    "mvd_ndhwc_to_nchw": (c_int, [_P, _P, c_int, c_int, c_long, _P]),
    "mvd_axpy": (c_int, [_P, _P, c_float, c_long, _P]),
    "mvd_flip_add": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "mvd_sw_accumulate": (c_int, [_P, _P, c_float, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, _P]),
    "mvd_sw_normalize": (c_int, [_P, _P, c_int, c_long, _P]),
    "mvd_feed_crop_pad_f32": (c_int, [_P, _P] + [c_int] * 11 + [c_float, _P]),
    "mvd_feed_crop_pad_seg_i16": (c_int, [_P, _P] + [c_int] * 15 + [_P]),
    "mvd_feed_downsample_seg": (c_int, [_P, _P, c_long] + [c_int] * 6 + [_P]),
}

_lib = None


def load():
    """Load the shared library (built by __graft_entry__.build() / csrc/Makefile).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP/HSA runtime (torch/lib/libamdhip64.so, SONAME libamdhip64.so.7).  It must be
    # in the process BEFORE our library is opened so that the dynamic loader binds our DT_NEEDED libamdhip64.so.7 to
    # that same copy: two HIP runtimes in one process cannot share streams or allocations (the second one reports
    # "no ROCm-capable device").
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built (run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C multimodal_mvd_seg_amd/csrc`).  There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def i3(v):
    return _I3(int(v[0]), int(v[1]), int(v[2]))


def call(name, *args):
    """Call an `int`-returning entry point; raise RuntimeError(mvd_last_error()) on a non-zero status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name} failed (status {rc}): {lib.mvd_last_error().decode()}")


def query(name, *args):
    """Call a value-returning entry point (workspace sizes etc.)."""
    return getattr(load(), name)(*args)
