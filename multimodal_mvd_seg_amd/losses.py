"""Host-side mirrors of the reference's loss classes for the hot path; every evaluation is a HIP kernel.

Class / function names and constructor arguments follow the reference so that a trainer's `_build_loss`
(nnUNet/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:351-375) reads the same:
RobustCrossEntropyLoss (training/loss/robust_ce_loss.py:6-16), MemoryEfficientSoftDiceLoss / DC_and_CE_loss /
DeepSupervisionWrapper (files missing from the fork: upstream nnU-Net 2.1.1 semantics),
distill_kl / l2_loss (training/loss/other_loss.py:51-78), soft_erode / soft_dilate / soft_open / soft_skel
(training/loss/soft_skeleton.py:6-37), clDice formula (training/metrics/clDice_metric.py:7-36).
"""
import numpy as np
import torch
from torch import nn

from . import ops


class _FusedDCCE(nn.Module):
    """One level of w_ce*CE + w_dice*Dice, evaluated by mvd_dcce_{fwd,finalize,bwd}."""

    def __init__(self, batch_dice=False, do_bg=False, smooth=1e-5, ddp=False, weight_ce=1.0, weight_dice=1.0):
        super().__init__()
        self.batch_dice, self.do_bg, self.smooth, self.ddp = batch_dice, do_bg, smooth, ddp
        self.weight_ce, self.weight_dice = weight_ce, weight_dice

    def _cfg(self):
        return (self.batch_dice, self.do_bg, self.smooth, self.weight_ce, self.weight_dice)

    def _gather(self):
        if not (self.ddp and self.batch_dice):
            return None
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return None
        from .parallel import gather_dice_stats
        return gather_dice_stats

    def forward(self, net_output, target):
        return ops.DeepSupervisedDCCEFn.apply([1.0], self._cfg(), self._gather(), [target], net_output)


class RobustCrossEntropyLoss(_FusedDCCE):
    """robust_ce_loss.py:6-16: float target with a singleton channel, mean over voxels."""

    def __init__(self):
        super().__init__(weight_ce=1.0, weight_dice=0.0)


class MemoryEfficientSoftDiceLoss(_FusedDCCE):
    """App. B; apply_nonlin is always softmax over dim 1 on this path (nnUNetTrainer.py:359-361)."""

    def __init__(self, apply_nonlin=None, batch_dice: bool = False, do_bg: bool = True, smooth: float = 1.,
                 ddp: bool = True):
        super().__init__(batch_dice, do_bg, smooth, ddp, weight_ce=0.0, weight_dice=1.0)


class DC_and_CE_loss(_FusedDCCE):
    def __init__(self, soft_dice_kwargs, ce_kwargs, weight_ce=1, weight_dice=1, ignore_label=None,
                 dice_class=MemoryEfficientSoftDiceLoss):
        if ignore_label is not None:
            raise NotImplementedError("ignore_label is not on the benchmarked path (SURVEY 8 a-6)")
        if ce_kwargs:
            raise NotImplementedError("ce_kwargs is {} at the reference call site (nnUNetTrainer.py:360)")
        kw = dict(batch_dice=False, do_bg=True, smooth=1., ddp=True)
        kw.update(soft_dice_kwargs)
        super().__init__(kw['batch_dice'], kw['do_bg'], kw['smooth'], kw['ddp'], float(weight_ce), float(weight_dice))


class DeepSupervisionWrapper(nn.Module):
    """sum_i w_i * loss(x_i, t_i) (nnUNetTrainer.py:374) as ONE autograd node over all levels."""

    def __init__(self, loss: _FusedDCCE, weight_factors=None):
        super().__init__()
        if not isinstance(loss, _FusedDCCE):
            raise TypeError("DeepSupervisionWrapper wraps the fused DC/CE losses of this package")
        self.loss = loss
        self.weight_factors = weight_factors

    def forward(self, net_output, target):
        assert isinstance(net_output, (tuple, list)) and isinstance(target, (tuple, list))
        w = [1.0] * len(net_output) if self.weight_factors is None else [float(i) for i in self.weight_factors]
        return ops.DeepSupervisedDCCEFn.apply(w, self.loss._cfg(), self.loss._gather(), list(target), *net_output)


def ds_weights(n_scales):
    """nnUNetTrainer.py:366-372."""
    w = np.array([1 / (2 ** i) for i in range(n_scales)])
    w[-1] = 0
    return w / w.sum()


# ------------------------------------------------------------------------------------------------ distillation
def distill_kl(y_s, y_t, T=1):
    """other_loss.py:51-64."""
    pad = y_s.shape[1] == 1
    return ops.DistillKLFn.apply(y_s, y_t, T, 1e-40, pad)


def l2_loss(input, target, channel_wise=False, T=1):
    """other_loss.py:67-78: channel_wise=True is the feature-distillation KL used on the path, channel_wise=False the
    plain mean squared difference (:77-78)."""
    if not channel_wise:
        return ops.MseFn.apply(ops.widen(input), ops.widen(target))
    return ops.DistillKLFn.apply(input, target, T, 0.0, False)


def kl_loss_compute1(vessel1, vessel2, T=1):
    """Unpinned wrapper (imported at MVDTrainer.py:74, defined nowhere): KL between the branches' vessel maps."""
    return distill_kl(vessel1[:, None], vessel2[:, None], T)


# ------------------------------------------------------------------------------------------------ soft skeleton
def soft_erode(img):
    return ops.SoftErodeFn.apply(img)


def soft_dilate(img):
    return ops.SoftDilateFn.apply(img)


def soft_open(img):
    return soft_dilate(soft_erode(img))


def soft_skel(img, iter_):
    """soft_skeleton.py:29-37: one fused launch for the step in front of the loop and one per iteration
    (erode -> erode -> dilate -> skeleton update walk an LDS tile; csrc/topo.hip::k_skel_iter_fwd)."""
    skel = ops.SkelInitFn.apply(img)
    for _ in range(iter_):
        img, skel = ops.SkelIterFn.apply(img, skel)
    return skel


def soft_skel_unfused(img, iter_):
    """The same chain one primitive per launch (the round-1 path; kept as the cross-check of the fused kernels)."""
    img1 = soft_open(img)
    skel = ops.SkelUpdateFn.apply(img, img1, None)
    for _ in range(iter_):
        img = soft_erode(img)
        img1 = soft_open(img)
        skel = ops.SkelUpdateFn.apply(img, img1, skel)
    return skel


def soft_cldice(pred, target, iter_=3, smooth=1.0):
    """1 - clDice on soft skeletons (see oracle/loss_oracle.py::soft_cldice for the unpinned choices)."""
    skel_pred = soft_skel(pred, iter_)
    with torch.no_grad():
        skel_true = soft_skel(target, iter_)
    return ops.ClDiceFn.apply(skel_pred, target, skel_true, pred, smooth)
