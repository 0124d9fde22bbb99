"""CPU tests: oracle/fp64_ops.py (the dgemm restatements the full-size GPU parity tests use as fp64 truth) against
torch.nn.functional in fp64 -- values and gradients, strides 1 / 2 / anisotropic, odd sizes, the network twin."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import fp64_ops as O, loss_oracle as LO, step_oracle as SO, unet_oracle as UO

torch.set_num_threads(4)
D64 = torch.float64


@pytest.mark.parametrize("C,K,sp,stride,ks", [
    (3, 5, (6, 7, 5), 1, (3, 3, 3)), (4, 6, (8, 7, 9), 2, (3, 3, 3)), (5, 4, (6, 8, 10), (1, 2, 2), (3, 3, 3)),
    (6, 3, (5, 4, 7), 1, (1, 3, 3)), (7, 2, (4, 5, 6), 1, (1, 1, 1))])
def test_conv3d_restatement_equals_torch_fp64(C, K, sp, stride, ks):
    g = torch.Generator().manual_seed(C * 31 + K)
    x = torch.randn(2, C, *sp, generator=g, dtype=D64)
    w = torch.randn(K, C, *ks, generator=g, dtype=D64)
    b = torch.randn(K, generator=g, dtype=D64)
    pad = tuple((k - 1) // 2 for k in ks)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv3d(xr, wr, br, stride, pad)
    gy = torch.randn(ref.shape, generator=g, dtype=D64)
    ref.backward(gy)
    xo, wo, bo = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = O.Conv3dF64.apply(xo, wo, bo, stride)
    y.backward(gy)
    for a, r in ((y, ref), (xo.grad, xr.grad), (wo.grad, wr.grad), (bo.grad, br.grad)):
        assert float((a.detach() - r.detach()).abs().max()) <= 1e-12 * max(1.0, float(r.abs().max()))


def test_conv3d_restatement_slabs(monkeypatch):
    """several slabs per sample (the path the 128^3 layers take)"""
    monkeypatch.setattr(O, "_slab_planes", lambda *a, **k: 3)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1, 4, 10, 6, 7, generator=g, dtype=D64)
    w = torch.randn(3, 4, 3, 3, 3, generator=g, dtype=D64)
    for st in (1, 2):
        xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
        ref = F.conv3d(xr, wr, None, st, 1)
        gy = torch.randn(ref.shape, generator=g, dtype=D64)
        ref.backward(gy)
        xo, wo = x.clone().requires_grad_(), w.clone().requires_grad_()
        y = O.Conv3dF64.apply(xo, wo, None, st)
        y.backward(gy)
        assert torch.allclose(y, ref, rtol=0, atol=1e-12)
        assert torch.allclose(xo.grad, xr.grad, rtol=0, atol=1e-12)
        assert torch.allclose(wo.grad, wr.grad, rtol=0, atol=1e-11)


@pytest.mark.parametrize("stride", [(2, 2, 2), (1, 2, 2)])
def test_convT3d_restatement_equals_torch_fp64(stride):
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 6, 3, 4, 5, generator=g, dtype=D64)
    w = torch.randn(6, 4, *stride, generator=g, dtype=D64)
    b = torch.randn(4, generator=g, dtype=D64)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv_transpose3d(xr, wr, br, stride)
    gy = torch.randn(ref.shape, generator=g, dtype=D64)
    ref.backward(gy)
    xo, wo, bo = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = O.ConvT3dF64.apply(xo, wo, bo, stride)
    y.backward(gy)
    for a, r in ((y, ref), (xo.grad, xr.grad), (wo.grad, wr.grad), (bo.grad, br.grad)):
        assert float((a.detach() - r.detach()).abs().max()) <= 1e-12 * max(1.0, float(r.abs().max()))


def test_instnorm_lrelu_restatement_equals_torch_fp64():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 5, 4, 6, 3, generator=g, dtype=D64) * 1.3 + 0.4
    ga = torch.rand(5, generator=g, dtype=D64) + 0.5
    be = torch.randn(5, generator=g, dtype=D64) * 0.2
    gy = torch.randn(x.shape, generator=g, dtype=D64)
    xr, gr, br = x.clone().requires_grad_(), ga.clone().requires_grad_(), be.clone().requires_grad_()
    ref = F.leaky_relu(F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5), 0.01)
    ref.backward(gy)
    y, z, xhat, rstd = O.instnorm_lrelu_fwd(x, ga, be)
    dx, dg, db = O.instnorm_lrelu_bwd(gy, xhat, rstd, ga, z > 0)
    assert torch.allclose(y, ref, rtol=0, atol=1e-13)
    assert torch.allclose(dx, xr.grad, rtol=0, atol=1e-12)
    assert torch.allclose(dg, gr.grad, rtol=0, atol=1e-12)
    assert torch.allclose(db, br.grad, rtol=0, atol=1e-12)
    # a flipped branch at one voxel changes dbeta by 0.99*dy there and leaves every other voxel's dz alone
    m = (z > 0).clone()
    m[0, 1, 2, 3, 1] = ~m[0, 1, 2, 3, 1]
    _, _, db2 = O.instnorm_lrelu_bwd(gy, xhat, rstd, ga, m)
    d = (db2 - db)
    assert abs(abs(float(d[1])) - 0.99 * abs(float(gy[0, 1, 2, 3, 1]))) < 1e-12 and float(d.abs().sum()) == abs(float(d[1]))


def test_fp64_twin_equals_plain_double_network():
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    ora = UO.build_plainconv_unet(2, 3, 3, strides, seed=0, features_per_stage=[4, 8, 12])
    batch = SO.synthetic_batch(2, 2, (8, 8, 8), strides, num_classes=3, seed=7)
    loss_fn = LO.build_loss(len(batch["target"]))
    import copy
    plain = copy.deepcopy(ora).double()
    loss_fn(plain(batch["data"].double()), [t.double() for t in batch["target"]]).backward()
    rec = {}
    twin = O.fp64_twin(ora, masks=None, record=rec)
    out = twin(batch["data"].double())
    loss_fn(out, [t.double() for t in batch["target"]]).backward()
    gp = dict(plain.named_parameters())
    for n, p in twin.named_parameters():
        assert p.grad is not None, n
        assert float((p.grad - gp[n].grad).abs().max()) <= 1e-11 * max(1.0, float(gp[n].grad.abs().max())), n
    assert len(rec) == 10  # 3 encoder stages x 2 + 2 decoder stages x 2 blocks
    # the same masks the network took on its own -> identical gradients through MaskedLeakyReLU
    masks = {k: v > 0 for k, v in rec.items()}
    twin2 = O.fp64_twin(ora, masks=masks)
    loss_fn(twin2(batch["data"].double()), [t.double() for t in batch["target"]]).backward()
    for n, p in twin2.named_parameters():
        assert float((p.grad - gp[n].grad).abs().max()) <= 1e-11 * max(1.0, float(gp[n].grad.abs().max())), n


@pytest.mark.parametrize("stride,D", [(1, 12), (2, 12), (2, 11)])
def test_plane_restricted_evaluations_equal_the_full_ones(stride, D):
    g = torch.Generator().manual_seed(stride * 100 + D)
    x = torch.randn(2, 3, D, 5, 6, generator=g, dtype=D64)
    w = torch.randn(4, 3, 3, 3, 3, generator=g, dtype=D64)
    b = torch.randn(4, generator=g, dtype=D64)
    y = O.conv3d_fwd(x, w, b, stride)
    gy = torch.randn(y.shape, generator=g, dtype=D64)
    dx, _, _ = O.conv3d_bwd(x, w, gy, stride)
    Do = y.shape[2]
    for d0, d1 in ((0, 2), (1, 4), (Do - 2, Do), (0, Do)):
        assert torch.allclose(O.conv3d_fwd_planes(x, w, b, stride, d0, d1), y[:, :, d0:d1], rtol=0, atol=1e-12)
    for i0, i1 in ((0, 3), (2, 5), (D - 3, D), (0, D), (D - 1, D)):
        got = O.conv3d_dx_planes(w, gy, stride, x.shape[2:], i0, i1)
        assert torch.allclose(got, dx[:, :, i0:i1], rtol=0, atol=1e-12), (i0, i1)
