"""TEST INFRASTRUCTURE -- fp64 "truth" evaluations of the hot path's ops, fast enough for the real cfg-2 layer shapes.

torch's own CPU fp64 conv3d (slow_conv3d) runs at 2-3 GFLOP/s; the parity tests at BASELINE configs[1] shapes need
~6 TFLOP of fp64 convolution.  These restatements compute the SAME sums (nn.Conv3d / nn.ConvTranspose3d /
InstanceNorm3d + LeakyReLU as parameterised by nnUNet/nnunetv2/utilities/get_network_from_plans.py:38-45) as
slab-wise im2col + one dgemm per slab (MKL, ~10 GFLOP/s per core).  tests/test_fp64_ops.py pins them against
torch.nn.functional.{conv3d, conv_transpose3d, instance_norm, leaky_relu} in fp64 (values and autograd gradients).

LeakyReLU's derivative is discontinuous at 0: an fp32 evaluation whose pre-activation z differs from the fp64 one by
1e-7 picks the other branch for the ~1e-6 fraction of voxels with |z| < 1e-6 and its gradient then differs by 0.99*dy
at those voxels -- a relative L2 deviation of ~1/sqrt(#voxels) per flip that every later layer of the backward pass
inherits.  `MaskedLeakyReLU` lets a test evaluate the fp64 truth *for a given branch pattern* (the one the fp32
implementation took), which separates that effect from genuine arithmetic error.
"""
import numpy as np
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function


def _tup3(v):
    return (int(v),) * 3 if isinstance(v, (int, np.integer)) else tuple(int(i) for i in v)


def _slab_planes(C, T, Ho, Wo, budget_bytes=1 << 30):
    per_plane = T * C * Ho * Wo * 8
    return max(1, int(budget_bytes // max(per_plane, 1)))


def _im2col_slab(xp, n, d0, d1, ks, st, Ho, Wo):
    """[T*C, (d1-d0)*Ho*Wo] patch matrix of output planes d0..d1 of sample n (xp is the zero-padded input)."""
    C = xp.shape[1]
    kd, kh, kw = ks
    sd, sh, sw = st
    nd = d1 - d0
    cols = torch.empty((kd * kh * kw, C, nd, Ho, Wo), dtype=xp.dtype)
    t = 0
    for a in range(kd):
        for b in range(kh):
            for c in range(kw):
                cols[t] = xp[n, :, d0 * sd + a: (d1 - 1) * sd + a + 1: sd, b: (Ho - 1) * sh + b + 1: sh,
                             c: (Wo - 1) * sw + c + 1: sw]
                t += 1
    return cols.view(kd * kh * kw * C, nd * Ho * Wo)


def conv3d_fwd(x, w, b, stride, pad=None):
    """nn.Conv3d(padding=(k-1)//2) forward; x [N,C,D,H,W], w [K,C,kd,kh,kw].  `pad` overrides the per-axis padding
    (slab evaluation of a big layer: the caller pads D itself and passes pad=(0, 1, 1))."""
    st, ks = _tup3(stride), tuple(w.shape[2:])
    pd = tuple((k - 1) // 2 for k in ks) if pad is None else tuple(pad)
    N, C, D, H, W = x.shape
    K = w.shape[0]
    Do, Ho, Wo = [(i + 2 * p - k) // s + 1 for i, p, k, s in zip((D, H, W), pd, ks, st)]
    xp = F.pad(x, (pd[2], pd[2], pd[1], pd[1], pd[0], pd[0]))
    w2 = w.permute(0, 2, 3, 4, 1).reshape(K, -1)  # [K, T*C] tap-major, matching _im2col_slab
    y = torch.empty((N, K, Do, Ho, Wo), dtype=x.dtype)
    step = _slab_planes(C, ks[0] * ks[1] * ks[2], Ho, Wo)
    for n in range(N):
        for d0 in range(0, Do, step):
            d1 = min(Do, d0 + step)
            cols = _im2col_slab(xp, n, d0, d1, ks, st, Ho, Wo)
            y[n, :, d0:d1] = (w2 @ cols).view(K, d1 - d0, Ho, Wo)
    if b is not None:
        y += b.view(1, K, 1, 1, 1)
    return y


def conv3d_bwd(x, w, dy, stride, need_dx=True, pad=None):
    """(dx, dw, db) of conv3d_fwd."""
    st, ks = _tup3(stride), tuple(w.shape[2:])
    pd = tuple((k - 1) // 2 for k in ks) if pad is None else tuple(pad)
    N, C, D, H, W = x.shape
    K = w.shape[0]
    Do, Ho, Wo = dy.shape[2:]
    T = ks[0] * ks[1] * ks[2]
    xp = F.pad(x, (pd[2], pd[2], pd[1], pd[1], pd[0], pd[0]))
    w2 = w.permute(0, 2, 3, 4, 1).reshape(K, -1)
    dw2 = torch.zeros_like(w2)
    dxp = torch.zeros_like(xp) if need_dx else None
    step = _slab_planes(C, T, Ho, Wo)
    sd, sh, sw = st
    for n in range(N):
        for d0 in range(0, Do, step):
            d1 = min(Do, d0 + step)
            nd = d1 - d0
            cols = _im2col_slab(xp, n, d0, d1, ks, st, Ho, Wo)
            g = dy[n, :, d0:d1].reshape(K, -1)
            dw2 += g @ cols.t()
            if need_dx:
                dcols = (w2.t() @ g).view(T, C, nd, Ho, Wo)
                t = 0
                for a in range(ks[0]):
                    for b in range(ks[1]):
                        for c in range(ks[2]):
                            dxp[n, :, d0 * sd + a: (d1 - 1) * sd + a + 1: sd, b: (Ho - 1) * sh + b + 1: sh,
                                c: (Wo - 1) * sw + c + 1: sw] += dcols[t]
                            t += 1
    dw = dw2.view(K, *ks, C).permute(0, 4, 1, 2, 3).contiguous()
    db = dy.sum((0, 2, 3, 4))
    dx = dxp[:, :, pd[0]: pd[0] + D, pd[1]: pd[1] + H, pd[2]: pd[2] + W].contiguous() if need_dx else None
    return dx, dw, db


def conv3d_fwd_planes(x, w, b, stride, d0, d1):
    """Output planes [d0, d1) (D axis) of conv3d_fwd(x, w, b, stride) without evaluating the rest: the exact same
    sums, used to check a 128^3 layer on a few slabs.  3x3x3 kernels (pad 1)."""
    st = _tup3(stride)
    xpd = F.pad(x, (0, 0, 0, 0, 1, 1))
    xs = xpd[:, :, st[0] * d0: st[0] * (d1 - 1) + 3]
    return conv3d_fwd(xs, w, b, stride, pad=(0, 1, 1))


def conv3d_dx_planes(w, dy, stride, in_dhw, i0, i1):
    """Input-gradient planes [i0, i1) of conv3d_bwd(., w, dy, stride) (3x3x3, pad 1), from the dy planes that reach
    them only."""
    st = _tup3(stride)
    s = st[0]
    Do = dy.shape[2]
    o0 = max(0, -(-(i0 - 1) // s))          # ceil((i0-1)/s)
    o1 = min(Do, i1 // s + 1)               # outputs o with s*o - 1 <= i1 - 1
    N, K = dy.shape[:2]
    C = w.shape[1]
    nd = s * (o1 - 1 - o0) + 3               # planes of the D-padded input the dy slab touches
    xs = torch.zeros((N, C, nd, in_dhw[1], in_dhw[2]), dtype=dy.dtype)
    dxs, _, _ = conv3d_bwd(xs, w, dy[:, :, o0:o1].contiguous(), stride, True, pad=(0, 1, 1))
    # sub plane p <-> D-padded plane s*o0 + p <-> input plane s*o0 + p - 1
    lo = i0 + 1 - s * o0
    out = torch.zeros((N, C, i1 - i0, in_dhw[1], in_dhw[2]), dtype=dy.dtype)
    a, b_ = max(lo, 0), min(lo + (i1 - i0), nd)
    out[:, :, a - lo: b_ - lo] = dxs[:, :, a:b_]
    return out


class Conv3dF64(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.has_b = stride, b is not None
        return conv3d_fwd(x, w, b, stride)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = conv3d_bwd(x, w, dy.contiguous(), ctx.stride, ctx.needs_input_grad[0])
        return dx, dw, (db if ctx.has_b else None), None


def convT3d_fwd(x, w, b, stride):
    """nn.ConvTranspose3d with kernel_size == stride (UNetDecoder.py:56-59); w [C,K,sd,sh,sw]."""
    st = _tup3(stride)
    N, C, D, H, W = x.shape
    K = w.shape[1]
    assert tuple(w.shape[2:]) == st
    w2 = w.reshape(C, -1).t()  # [K*S, C]
    y = (w2 @ x.reshape(N, C, -1)).view(N, K, *st, D, H, W)
    y = y.permute(0, 1, 5, 2, 6, 3, 7, 4).reshape(N, K, D * st[0], H * st[1], W * st[2])
    if b is not None:
        y = y + b.view(1, K, 1, 1, 1)
    return y


def convT3d_bwd(x, w, dy, stride):
    st = _tup3(stride)
    N, C, D, H, W = x.shape
    K = w.shape[1]
    g = dy.view(N, K, D, st[0], H, st[1], W, st[2]).permute(0, 1, 3, 5, 7, 2, 4, 6).reshape(N, -1, D * H * W)
    w2 = w.reshape(C, -1)  # [C, K*S]
    dx = (w2 @ g).view(N, C, D, H, W)
    dw = torch.einsum('ncv,nkv->ck', x.reshape(N, C, -1), g).view_as(w)
    db = dy.sum((0, 2, 3, 4))
    return dx, dw, db


class ConvT3dF64(Function):
    @staticmethod
    def forward(ctx, x, w, b, stride):
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.has_b = stride, b is not None
        return convT3d_fwd(x, w, b, stride)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = convT3d_bwd(x, w, dy.contiguous(), ctx.stride)
        return dx, dw, (db if ctx.has_b else None), None


# ----------------------------------------------------------------------------------------- InstanceNorm + LeakyReLU
def instnorm_lrelu_fwd(x, gamma, beta, eps=1e-5, slope=0.01):
    """(y, z, xhat, rstd): InstanceNorm3d(eps, affine, biased variance) then LeakyReLU(slope)."""
    N, C = x.shape[:2]
    xf = x.reshape(N, C, -1)
    mean = xf.mean(2, keepdim=True)
    var = ((xf - mean) ** 2).mean(2, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    xhat = ((xf - mean) * rstd).view_as(x)
    z = xhat * gamma.view(1, C, 1, 1, 1) + beta.view(1, C, 1, 1, 1)
    y = torch.where(z > 0, z, z * slope)
    return y, z, xhat, rstd.view(N, C)


def instnorm_lrelu_bwd(dy, xhat, rstd, gamma, mask, slope=0.01):
    """(dx, dgamma, dbeta) with the LeakyReLU branch given by `mask` (True: the z > 0 branch was taken)."""
    N, C = xhat.shape[:2]
    dz = torch.where(mask, dy, dy * slope)
    dzf, xh = dz.reshape(N, C, -1), xhat.reshape(N, C, -1)
    dbeta_nc = dzf.sum(2)
    dgamma_nc = (dzf * xh).sum(2)
    V = xh.shape[2]
    g = gamma.view(1, C, 1)
    dx = (rstd.view(N, C, 1) * g) * (dzf - dbeta_nc.unsqueeze(2) / V - xh * (dgamma_nc.unsqueeze(2) / V))
    return dx.view_as(xhat), dgamma_nc.sum(0), dbeta_nc.sum(0)


class MaskedLeakyReLU(Function):
    """LeakyReLU whose backward takes the branch from `mask` (bool, True = slope 1) instead of sign(z)."""

    @staticmethod
    def forward(ctx, z, mask, slope):
        ctx.save_for_backward(mask)
        ctx.slope = slope
        return torch.where(z > 0, z, z * slope)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return torch.where(mask, dy, dy * ctx.slope), None, None


# ----------------------------------------------------------------------------------------- fp64 network evaluation
class _FastConv(nn.Module):
    def __init__(self, conv: nn.Conv3d):
        super().__init__()
        self.weight, self.bias, self.stride = conv.weight, conv.bias, tuple(conv.stride)

    def forward(self, x):
        return Conv3dF64.apply(x, self.weight, self.bias, self.stride)


class _FastConvT(nn.Module):
    def __init__(self, conv: nn.ConvTranspose3d):
        super().__init__()
        self.weight, self.bias, self.stride = conv.weight, conv.bias, tuple(conv.stride)

    def forward(self, x):
        return ConvT3dF64.apply(x, self.weight, self.bias, self.stride)


class _MaskedAct(nn.Module):
    """Stands in for the block's LeakyReLU: records z, applies the branch pattern in `masks[name]` when present."""

    def __init__(self, name, slope, masks, record):
        super().__init__()
        self.name, self.slope, self.masks, self.record = name, slope, masks, record

    def forward(self, z):
        if self.record is not None:
            self.record[self.name] = z.detach()
        m = self.masks.get(self.name) if self.masks is not None else None
        if m is None:
            return F.leaky_relu(z, self.slope)
        return MaskedLeakyReLU.apply(z, m, self.slope)


def fp64_twin(oracle_net, masks=None, record=None):
    """A double-precision deep copy of an oracle PlainConvUNet (oracle/unet_oracle.py) whose 3x3x3 / transposed convs
    run through the dgemm restatements above and whose LeakyReLUs take their backward branch from `masks`
    ({block name: bool tensor [N,C,D,H,W]}; block name = module path of the ConvDropoutNormReLU, e.g.
    'encoder.stages.0.0.convs.1').  `record` (dict) receives every block's pre-activation z.  Parameters keep the
    oracle's names, so gradients are compared by name."""
    import copy
    net = copy.deepcopy(oracle_net).double()
    for p in net.parameters():
        p.grad = None
    from .unet_oracle import ConvDropoutNormReLU
    for name, m in list(net.named_modules()):
        if isinstance(m, ConvDropoutNormReLU):
            if name.startswith("decoder.encoder."):
                continue  # alias of encoder.* (same module objects; visited once under its primary name)
            fast = _FastConv(m.conv)
            act = _MaskedAct(name, m.nonlin.negative_slope, masks, record)
            norm = m.norm
            m.forward = (lambda x, fast=fast, norm=norm, act=act: act(norm(fast(x))))
    dec = net.decoder
    for i, t in enumerate(dec.transpconvs):
        fast = _FastConvT(t)
        t.forward = (lambda x, fast=fast: fast(x))
    return net
