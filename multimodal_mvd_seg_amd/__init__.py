"""multimodal_mvd_seg_amd -- MI355X-native (gfx950) train-step hot path of JaronTu/Multimodal_MVD_Seg.

Only what the path needs: csrc/ (hand-written HIP kernels + the C ABI of
include/mvdseg_hip.h), ops.py (autograd glue over the ABI), network.py / losses.py / optim.py / trainer.py (host-side
mirrors of the reference's plugin interface), parallel.py (RCCL data parallelism) and inference.py (sliding-window
prediction).  Importing the package does
not need a GPU; running any op does, and fails loudly when libmvdseg_hip.so is missing (there is no CPU fallback).
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
