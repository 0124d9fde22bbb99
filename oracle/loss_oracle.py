"""TEST INFRASTRUCTURE -- CPU (torch fp32) restatement of the losses on the reference's hot path.

Reference-pinned (module importable by file path; cross-checked in tests/test_oracle_vs_reference.py and
frozen in tests/golden/):
* RobustCrossEntropyLoss        nnUNet/nnunetv2/training/loss/robust_ce_loss.py:6-16
* soft_erode/dilate/open/skel   nnUNet/nnunetv2/training/loss/soft_skeleton.py:6-37
Reference-pinned through the function body (the module itself fails at `import lightly`; tools/make_golden.py compiles
the two function definitions out of the file, runs them and freezes inputs/outputs in tests/golden/; the restatements
below reproduce those fixtures bit for bit, tests/test_oracle.py):
* distill_kl, l2_loss           nnUNet/nnunetv2/training/loss/other_loss.py:51-64, :67-78
"parity unpinned" (files missing from the fork; semantics = upstream nnU-Net 2.1.1 constrained by the call
site nnUNet/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:351-375):
* MemoryEfficientSoftDiceLoss, DC_and_CE_loss, DeepSupervisionWrapper, get_tp_fp_fn_tn
* kl_loss_compute1 (imported at MVDTrainer.py:74, defined nowhere) := distill_kl on [B,1,...] vessel maps
* soft_cldice: clDice formula of nnUNet/nnunetv2/training/metrics/clDice_metric.py:7-36 on soft skeletons
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------- CE (robust_ce_loss.py:6-16)
class RobustCrossEntropyLoss(nn.CrossEntropyLoss):
    def forward(self, input, target):
        if len(target.shape) == len(input.shape):
            assert target.shape[1] == 1
            target = target[:, 0]
        return super().forward(input, target.long())


# ----------------------------------------------------------------------------- Dice (App. B)
class MemoryEfficientSoftDiceLoss(nn.Module):
    def __init__(self, apply_nonlin=None, batch_dice=False, do_bg=True, smooth=1., ddp=False):
        super().__init__()
        self.apply_nonlin, self.batch_dice, self.do_bg, self.smooth, self.ddp = \
            apply_nonlin, batch_dice, do_bg, smooth, ddp

    def forward(self, x, y, loss_mask=None):
        shp_x, shp_y = x.shape, y.shape
        if self.apply_nonlin is not None:
            x = self.apply_nonlin(x)
        if not self.do_bg:
            x = x[:, 1:]
        axes = list(range(2, len(shp_x)))
        with torch.no_grad():
            if len(shp_x) != len(shp_y):
                y = y.view((shp_y[0], 1, *shp_y[1:]))
            if all(i == j for i, j in zip(shp_x, shp_y)):
                y_onehot = y
            else:
                y_onehot = torch.zeros(shp_x, device=x.device, dtype=torch.bool)
                y_onehot.scatter_(1, y.long(), 1)
            if not self.do_bg:
                y_onehot = y_onehot[:, 1:]
            sum_gt = y_onehot.sum(axes) if loss_mask is None else (y_onehot * loss_mask).sum(axes)
        intersect = (x * y_onehot).sum(axes) if loss_mask is None else (x * y_onehot * loss_mask).sum(axes)
        sum_pred = x.sum(axes) if loss_mask is None else (x * loss_mask).sum(axes)
        if self.ddp and self.batch_dice:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                from .step_oracle import AllGatherGrad
                intersect = AllGatherGrad.apply(intersect).sum(0)
                sum_pred = AllGatherGrad.apply(sum_pred).sum(0)
                sum_gt = AllGatherGrad.apply(sum_gt.float()).sum(0)
        if self.batch_dice:
            intersect, sum_pred, sum_gt = intersect.sum(0), sum_pred.sum(0), sum_gt.sum(0)
        dc = (2 * intersect + self.smooth) / (torch.clip(sum_gt + sum_pred + self.smooth, 1e-8))
        return -dc.mean()


def softmax_helper_dim1(x):
    return torch.softmax(x, 1)


class DC_and_CE_loss(nn.Module):
    def __init__(self, soft_dice_kwargs, ce_kwargs, weight_ce=1, weight_dice=1, ignore_label=None,
                 dice_class=MemoryEfficientSoftDiceLoss):
        super().__init__()
        if ignore_label is not None:
            ce_kwargs['ignore_index'] = ignore_label
        self.weight_dice, self.weight_ce, self.ignore_label = weight_dice, weight_ce, ignore_label
        self.ce = RobustCrossEntropyLoss(**ce_kwargs)
        self.dc = dice_class(apply_nonlin=softmax_helper_dim1, **soft_dice_kwargs)

    def forward(self, net_output, target):
        if self.ignore_label is not None:
            mask = (target != self.ignore_label).bool()
            target_dice = torch.clone(target)
            target_dice[target == self.ignore_label] = 0
            num_fg = mask.sum()
        else:
            target_dice, mask = target, None
        dc_loss = self.dc(net_output, target_dice, loss_mask=mask) if self.weight_dice != 0 else 0
        ce_loss = self.ce(net_output, target[:, 0].long()) \
            if self.weight_ce != 0 and (self.ignore_label is None or num_fg > 0) else 0
        return self.weight_ce * ce_loss + self.weight_dice * dc_loss


class DeepSupervisionWrapper(nn.Module):
    """Sum_i w_i * loss(x_i, t_i) (nnUNetTrainer.py:374).  Upstream 2.1.1 evaluates every term, so the
    zero-weighted lowest-resolution head receives an exact-zero gradient rather than None."""

    def __init__(self, loss, weight_factors=None):
        super().__init__()
        self.weight_factors, self.loss = weight_factors, loss

    def forward(self, *args):
        weights = [1] * len(args[0]) if self.weight_factors is None else self.weight_factors
        l = weights[0] * self.loss(*[j[0] for j in args])
        for i, inputs in enumerate(zip(*args)):
            if i == 0:
                continue
            l = l + weights[i] * self.loss(*inputs)
        return l


def ds_weights(n_scales):
    """nnUNetTrainer.py:366-372."""
    w = np.array([1 / (2 ** i) for i in range(n_scales)])
    w[-1] = 0
    return w / w.sum()


def build_loss(n_scales, batch_dice=False, ddp=False, deep_supervision=True):
    """nnUNetTrainer._build_loss (:351-375), label (non-region) branch, no ignore label."""
    loss = DC_and_CE_loss({'batch_dice': batch_dice, 'smooth': 1e-5, 'do_bg': False, 'ddp': ddp}, {},
                          weight_ce=1, weight_dice=1, ignore_label=None, dice_class=MemoryEfficientSoftDiceLoss)
    if deep_supervision:
        loss = DeepSupervisionWrapper(loss, ds_weights(n_scales))
    return loss


def get_tp_fp_fn_tn(net_output, gt, axes=None, mask=None):
    """App. B: tp = sum pred*y, fp = sum pred*(1-y), fn = sum (1-pred)*y over `axes`."""
    if axes is None:
        axes = tuple(range(2, net_output.ndim))
    with torch.no_grad():
        if net_output.ndim != gt.ndim:
            gt = gt.view((gt.shape[0], 1, *gt.shape[1:]))
        if net_output.shape == gt.shape:
            y_onehot = gt
        else:
            y_onehot = torch.zeros(net_output.shape, device=net_output.device)
            y_onehot.scatter_(1, gt.long(), 1)
    tp = net_output * y_onehot
    fp = net_output * (1 - y_onehot)
    fn = (1 - net_output) * y_onehot
    tn = (1 - net_output) * (1 - y_onehot)
    if mask is not None:
        tp, fp, fn, tn = tp * mask, fp * mask, fn * mask, tn * mask
    if len(axes) > 0:
        tp, fp, fn, tn = (t.sum(dim=axes, keepdim=False) for t in (tp, fp, fn, tn))
    return tp, fp, fn, tn


def validation_counts(logits, target):
    """nnUNetTrainer.validation_step :966-1003 (label branch, no ignore label): argmax -> one-hot ->
    tp/fp/fn over axes [0,2,3,4]; background dropped."""
    axes = [0] + list(range(2, logits.ndim))
    seg = logits.argmax(1)[:, None]
    onehot = torch.zeros(logits.shape, dtype=torch.float32)
    onehot.scatter_(1, seg, 1)
    tp, fp, fn, _ = get_tp_fp_fn_tn(onehot, target, axes=axes)
    return tp.numpy()[1:], fp.numpy()[1:], fn.numpy()[1:]


def dice_from_counts(tp, fp, fn):
    """nnUNetTrainer.on_validation_epoch_end :1033-1034."""
    with np.errstate(divide='ignore', invalid='ignore'):
        per_class = [2 * i / (2 * i + j + k) for i, j, k in zip(tp, fp, fn)]
    return per_class, float(np.nanmean(per_class))


# ----------------------------------------------------------------------------- distillation (other_loss.py)
def distill_kl(y_s, y_t, T=1):
    """other_loss.py:51-64 (the stray `self` argument dropped)."""
    if y_s.shape[1] == 1:
        y_s = torch.cat([y_s, torch.zeros_like(y_s)], 1)
        y_t = torch.cat([y_t, torch.zeros_like(y_t)], 1)
    p_s = F.log_softmax(y_s / T + 1e-40, dim=1)
    p_t = F.softmax(y_t / T, dim=1)
    return F.kl_div(p_s, p_t, reduction='mean') * (T ** 2)


def l2_loss(input, target, channel_wise=False, T=1):
    """other_loss.py:67-78."""
    if channel_wise:
        return F.kl_div(F.log_softmax(input / T, dim=1), F.softmax(target / T, dim=1), reduction='mean') * (T ** 2)
    return torch.mean(torch.abs(input - target).pow(2))


def kl_loss_compute1(vessel1, vessel2, T=1):
    """UNPINNED wrapper (MVDTrainer.py:74,899): KL between the two branches' vessel maps [B,D,H,W] -> [B,1,...]
    (exercises the zero-channel padding of distill_kl)."""
    return distill_kl(vessel1[:, None], vessel2[:, None], T)


# ----------------------------------------------------------------------------- soft skeleton (soft_skeleton.py)
def soft_erode(img):
    p1 = -F.max_pool3d(-img, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    p2 = -F.max_pool3d(-img, (1, 3, 1), (1, 1, 1), (0, 1, 0))
    p3 = -F.max_pool3d(-img, (1, 1, 3), (1, 1, 1), (0, 0, 1))
    return torch.min(torch.min(p1, p2), p3)


def soft_dilate(img):
    return F.max_pool3d(img, (3, 3, 3), (1, 1, 1), (1, 1, 1))


def soft_open(img):
    return soft_dilate(soft_erode(img))


def soft_skel(img, iter_):
    img1 = soft_open(img)
    skel = F.relu(img - img1)
    for _ in range(iter_):
        img = soft_erode(img)
        img1 = soft_open(img)
        delta = F.relu(img - img1)
        skel = skel + F.relu(delta - skel * delta)
    return skel


def soft_cldice(pred, target, iter_=3, smooth=1.0):
    """clDice formula (clDice_metric.py:7-36: cl_score(v,s)=sum(v*s)/sum(s); 2*tprec*tsens/(tprec+tsens)) on
    soft skeletons; returned as a loss 1 - clDice.  pred/target: [B,1,D,H,W] in [0,1].  `smooth` keeps the
    ratios finite for empty skeletons (unpinned choice, recorded in DESIGN.md)."""
    skel_pred = soft_skel(pred, iter_)
    skel_true = soft_skel(target, iter_)
    tprec = (torch.sum(skel_pred * target) + smooth) / (torch.sum(skel_pred) + smooth)
    tsens = (torch.sum(skel_true * pred) + smooth) / (torch.sum(skel_true) + smooth)
    return 1.0 - 2.0 * (tprec * tsens) / (tprec + tsens)
