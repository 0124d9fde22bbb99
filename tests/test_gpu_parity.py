"""GPU parity tests (run with `-m gpu` on an MI355X): every HIP kernel, called through the C ABI
(multimodal_mvd_seg_amd.ops -> ctypes -> libmvdseg_hip.so), against the golden fixtures of tests/golden/ and, for
larger seeded inputs, against the CPU oracle (oracle/) evaluated on the box's host cores.

Tolerances: fp32 results within 1e-4 abs of the oracle (north_star) -- most checks are tighter and say so;
integer / min-max work (soft skeleton forward, connected components, argmax counts) must be bit-exact."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, load_npz

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from multimodal_mvd_seg_amd import _lib
    _lib.load()  # fails loudly if the HIP extension is missing
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // 2))


def G(a, requires_grad=False):
    t = torch.from_numpy(np.asarray(a)).to(DEV)
    if requires_grad:
        t.requires_grad_()
    return t


def close(a, b, atol, rtol=0.0, what=""):
    a, b = a.detach().float().cpu(), torch.as_tensor(np.asarray(b)).float()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f"{what}: max abs err {float(err.max()):.3e} (tol {atol:g}+{rtol:g}*|ref|)"


# ================================================================================================ conv
CONV_FIXT = sorted(f for f in os.listdir(GOLDEN) if f.startswith("conv3d_"))


@pytest.mark.parametrize("engine", ["auto", "scalar"])
@pytest.mark.parametrize("name", CONV_FIXT)
def test_conv3d_fwd_dgrad_wgrad(name, engine):
    from multimodal_mvd_seg_amd import ops
    z = load_npz(name)
    ops.set_conv_engine(engine)
    try:
        x, w, b = G(z["x"], True), G(z["w"], True), G(z["b"], True)
        y = ops.Conv3dFn.apply(x, None, w, b, tuple(int(i) for i in z["stride"]))
        close(y, z["y"], 2e-5, 1e-5, "y")
        y.backward(G(z["gy"]))
        close(x.grad, z["gx"], 2e-5, 1e-5, "dx")
        close(w.grad, z["gw"], 1e-4, 1e-5, "dw")
        close(b.grad, z["gb"], 1e-4, 1e-5, "db")
    finally:
        ops.set_conv_engine("auto")


def test_conv3d_two_pointer_concat_equals_cat():
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(3)
    a = torch.randn(2, 32, 6, 8, 10, generator=g)
    b = torch.randn(2, 32, 6, 8, 10, generator=g)
    w = torch.randn(32, 64, 3, 3, 3, generator=g) * 0.05
    bias = torch.randn(32, generator=g) * 0.1
    xa, xb = a.clone().requires_grad_(), b.clone().requires_grad_()
    wr = w.clone().requires_grad_()
    ref = F.conv3d(torch.cat((xa, xb), 1), wr, bias, 1, 1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    ga, gb, gw = G(a, True), G(b, True), G(w, True)
    y = ops.Conv3dFn.apply(ga, gb, gw, G(bias), (1, 1, 1))
    y.backward(G(gy))
    close(y, ref.detach(), 5e-5, 1e-5, "y")
    close(ga.grad, xa.grad, 5e-5, 1e-5, "dx1")
    close(gb.grad, xb.grad, 5e-5, 1e-5, "dx2")
    close(gw.grad, wr.grad, 2e-4, 1e-5, "dw")


@pytest.mark.parametrize("C1,C2,K,sp,stride,N", [
    (128, 0, 128, (8, 8, 8), 1, 2),      # split-reduce path (few tiles, many channels), CK=32 persistent kernel
    (320, 320, 320, (4, 4, 4), 1, 2),    # bottleneck decoder conv: two pointers + split
    (64, 64, 64, (9, 10, 11), 1, 1),     # ragged tiles, NT=2
    (32, 0, 32, (20, 17, 13), 1, 2),     # NT=1, several tiles per axis, ragged
    (32, 0, 64, (16, 12, 20), 2, 1),     # stride 2 (8-channel sub-chunks of the 32-layout); dgrad parity classes
    (32, 0, 64, (32, 32, 64), 2, 2),     # stride 2 through the CK=32 strided kernel (>= 256 work items)
    (64, 0, 128, (30, 34, 66), 2, 1),    # same, two chunks, two k-blocks, ragged tiles and odd output sizes
    (96, 0, 32, (6, 7, 8), (1, 2, 2), 1),  # anisotropic stride
    (4, 0, 32, (10, 9, 8), 1, 2),        # the 4-channel input layer (CK=4)
    (8, 0, 32, (7, 6, 5), 1, 1),         # CK=8 layout
    (32, 0, 32, (18, 36, 44), 1, 2),     # Winograd wgrad tile loop: 540 ragged tiles over 256 workgroups (two LDS images,
                                         # buffer loads with out-of-range lanes, zero-record descriptors after the last tile)
    (32, 32, 32, (20, 40, 24), 1, 2),    # the same with two input pointers (128 workgroups per channel block)
    (32, 0, 64, (40, 44, 68), 2, 2),     # generic wgrad tile loop (stride 2): several ragged tiles per workgroup
])
def test_conv3d_engine_shapes_vs_torch_fp64(C1, C2, K, sp, stride, N):
    from multimodal_mvd_seg_amd import ops
    st = (stride,) * 3 if isinstance(stride, int) else stride
    g = torch.Generator().manual_seed(C1 + K + sp[0])
    x1 = torch.randn(N, C1, *sp, generator=g)
    x2 = torch.randn(N, C2, *sp, generator=g) if C2 else None
    w = torch.randn(K, C1 + C2, 3, 3, 3, generator=g) * (1.0 / np.sqrt(27 * (C1 + C2)))
    b = torch.randn(K, generator=g) * 0.1
    xs = [t.double().requires_grad_() for t in ([x1, x2] if C2 else [x1])]
    wr = w.double().requires_grad_()
    br = b.double().requires_grad_()
    ref = F.conv3d(torch.cat(xs, 1), wr, br, st, 1)
    gy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gy)
    g1, g2 = G(x1, True), (G(x2, True) if C2 else None)
    gw, gb = G(w, True), G(b, True)
    y = ops.Conv3dFn.apply(g1, g2, gw, gb, st)
    y.backward(G(gy.float()))
    close(y, ref.detach(), 2e-5, 1e-5, "y")
    close(g1.grad, xs[0].grad, 2e-5, 1e-5, "dx1")
    if C2:
        close(g2.grad, xs[1].grad, 2e-5, 1e-5, "dx2")
    scale = float(wr.grad.abs().max())
    close(gw.grad, wr.grad, 2e-5 * scale, 1e-5, "dw")
    close(gb.grad, br.grad, 1e-4, 1e-5, "db")


@pytest.mark.parametrize("C1,C2,K,sp,N", [
    (32, 0, 32, (8, 8, 16), 1),          # whole tiles
    (32, 0, 32, (9, 7, 13), 2),          # ragged in every axis, odd W (a pair straddles the border)
    (32, 32, 64, (6, 10, 11), 1),        # two input pointers, two k-blocks
    (64, 0, 32, (5, 4, 8), 2),           # two chunks
    (96, 0, 96, (4, 4, 8), 1),           # three chunks, three k-blocks
])
def test_conv3d_winograd_engine_vs_torch_fp64(C1, C2, K, sp, N):
    """fwd and dgrad through the Winograd F(2,3) kernel (forced on for small problems) against fp64; also equality
    with the direct MFMA engine to fp32 round-off.  dgrad of the two-pointer case writes dx1/dx2 (split outputs)."""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call, query, i3
    if query("mvd_wino_mode") == 0:
        pytest.skip("MVD_WINO=0: the Winograd engines are switched off for this run")
    g = torch.Generator().manual_seed(C1 + K + sp[2])
    x1 = torch.randn(N, C1, *sp, generator=g)
    x2 = torch.randn(N, C2, *sp, generator=g) if C2 else None
    w = torch.randn(K, C1 + C2, 3, 3, 3, generator=g) * (1.0 / np.sqrt(27 * (C1 + C2)))
    b = torch.randn(K, generator=g) * 0.1
    xs = [t.double().requires_grad_() for t in ([x1, x2] if C2 else [x1])]
    ref = F.conv3d(torch.cat(xs, 1), w.double(), b.double(), 1, 1)
    gy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gy)
    outs = {}
    try:
        for mode, min_items in (("wino", 1), ("direct", 1 << 40)):
            call("mvd_set_wino_min_items", min_items)
            assert bool(query("mvd_conv_wino_applicable", N, *sp, C1, C2, K, i3((3, 3, 3)), i3((1, 1, 1)))) == (mode == "wino")
            g1, g2 = G(x1, True), (G(x2, True) if C2 else None)
            y = ops.Conv3dFn.apply(g1, g2, G(w, True), G(b, True), (1, 1, 1))
            y.backward(G(gy.float()))
            outs[mode] = (y.detach(), g1.grad, g2.grad if C2 else None)
    finally:
        call("mvd_set_wino_min_items", -1)
    y, d1, d2 = outs["wino"]
    close(y, ref.detach(), 1e-5, 1e-5, "y")
    close(d1, xs[0].grad, 1e-5, 1e-5, "dx1")
    if C2:
        close(d2, xs[1].grad, 1e-5, 1e-5, "dx2")
    close(y, outs["direct"][0].cpu(), 1e-5, 1e-5, "y vs direct engine")
    close(d1, outs["direct"][1].cpu(), 1e-5, 1e-5, "dx1 vs direct engine")


def test_conv_instnorm_statistics_epilogue_vs_fp64():
    """The Winograd conv kernel hands per-tile (sum, sum of squares) of its output to the following fused
    InstanceNorm+LeakyReLU (no statistics pass over the activation).  Ragged tiles; conv -> norm against fp64,
    forward and backward, and bit-level agreement of the activation's statistics with the plain two-pass norm."""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call, query
    g = torch.Generator().manual_seed(21)
    N, C, K, sp = 2, 32, 64, (9, 10, 13)
    x = torch.randn(N, C, *sp, generator=g)
    w = torch.randn(K, C, 3, 3, 3, generator=g) / np.sqrt(27 * C)
    b = torch.randn(K, generator=g) * 0.1
    gamma = torch.rand(K, generator=g) + 0.5
    beta = torch.randn(K, generator=g) * 0.1
    gy = torch.randn(N, K, *sp, generator=g)
    xr, wr, br, gr, ber = [t.double().requires_grad_() for t in (x, w, b, gamma, beta)]
    ref = F.leaky_relu(F.instance_norm(F.conv3d(xr, wr, br, 1, 1), None, None, gr, ber, True, 0.1, 1e-5), 0.01)
    ref.backward(gy.double())
    try:
        call("mvd_set_wino_min_items", 1)
        gx, gw, gb, gg, gbe = G(x, True), G(w, True), G(b, True), G(gamma, True), G(beta, True)
        y = ops.Conv3dFn.apply(gx, None, gw, gb, (1, 1, 1))
        if query("mvd_wino_mode") == 2:  # the epilogue belongs to the F(2x2,3x3) kernel
            assert getattr(y, "_mvd_tile_stats", None) is not None, "the statistics epilogue did not run"
        z = ops.InstanceNormLeakyReLUFn.apply(y, gg, gbe, 1e-5, 0.01)
        z.backward(G(gy))
        y2 = y.detach().clone()  # no statistics attached: plain two-pass norm
        z2 = ops.InstanceNormLeakyReLUFn.apply(y2, gg.detach(), gbe.detach(), 1e-5, 0.01)
    finally:
        call("mvd_set_wino_min_items", -1)
    close(z, ref.detach(), 2e-5, 1e-5, "conv+norm output")
    close(z, z2.cpu(), 2e-6, 1e-6, "epilogue statistics vs two-pass statistics")
    close(gx.grad, xr.grad, 2e-5 * float(xr.grad.abs().max()), 1e-5, "dx")
    close(gg.grad, gr.grad, 2e-5 * float(gr.grad.abs().max()), 1e-5, "dgamma")


@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.startswith("convT3d_")))
def test_convT3d(name):
    from multimodal_mvd_seg_amd import ops
    z = load_npz(name)
    x, w, b = G(z["x"], True), G(z["w"], True), G(z["b"], True)
    y = ops.ConvTranspose3dFn.apply(x, w, b, tuple(int(i) for i in z["stride"]))
    close(y, z["y"], 2e-5, 1e-5, "y")
    y.backward(G(z["gy"]))
    close(x.grad, z["gx"], 2e-5, 1e-5, "dx")
    close(w.grad, z["gw"], 1e-4, 1e-5, "dw")
    close(b.grad, z["gb"], 1e-4, 1e-5, "db")


@pytest.mark.parametrize("C,K,sp,stride,N", [
    (64, 32, (4, 4, 4), (2, 2, 2), 2),      # dgrad reduces over K=32 -> CK=32 layout, stride-2 gather (sub-chunk path)
    (320, 320, (2, 2, 2), (2, 2, 2), 2),    # bottleneck up-sampling, split-reduce
    (128, 64, (5, 6, 7), (2, 2, 2), 1),     # ragged
    (64, 32, (4, 6, 5), (1, 2, 2), 1),      # anisotropic
    (64, 32, (33, 41, 53), (2, 2, 2), 2),   # the persistent forward kernel of the top level (>= 4096 blocks), ragged tail
])
def test_convT3d_engine_shapes_vs_torch_fp64(C, K, sp, stride, N):
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(C + K)
    x = torch.randn(N, C, *sp, generator=g)
    w = torch.randn(C, K, *stride, generator=g) * (1.0 / np.sqrt(C))
    b = torch.randn(K, generator=g) * 0.1
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = F.conv_transpose3d(xr, wr, br, stride)
    gy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gy)
    gx, gw, gb = G(x, True), G(w, True), G(b, True)
    y = ops.ConvTranspose3dFn.apply(gx, gw, gb, stride)
    y.backward(G(gy.float()))
    close(y, ref.detach(), 2e-5, 1e-5, "y")
    close(gx.grad, xr.grad, 2e-5, 1e-5, "dx")
    close(gw.grad, wr.grad, 2e-5 * float(wr.grad.abs().max()), 1e-5, "dw")
    close(gb.grad, br.grad, 1e-4, 1e-5, "db")


def test_seghead_conv1x1():
    from multimodal_mvd_seg_amd import ops
    z = load_npz("conv1x1.npz")
    x, w, b = G(z["x"], True), G(z["w"], True), G(z["b"], True)
    y = ops.SegHeadFn.apply(x, w, b)
    assert y.is_contiguous()  # planar logits
    close(y, z["y"], 1e-5, 1e-5, "y")
    y.backward(G(z["gy"]))
    close(x.grad, z["gx"], 1e-5, 1e-5, "dx")
    close(w.grad, z["gw"], 1e-4, 1e-5, "dw")
    close(b.grad, z["gb"], 1e-4, 1e-5, "db")


# ================================================================================================ norm
def test_instnorm_lrelu_fixture():
    from multimodal_mvd_seg_amd import ops
    z = load_npz("instnorm_lrelu.npz")
    x, ga, be = G(z["x"], True), G(z["gamma"], True), G(z["beta"], True)
    y = ops.InstanceNormLeakyReLUFn.apply(x, ga, be, 1e-5, 0.01)
    close(y, z["y"], 1e-5, 1e-5, "y")
    y.backward(G(z["gy"]))
    close(x.grad, z["gx"], 1e-5, 1e-5, "dx")
    close(ga.grad, z["ggamma"], 1e-4, 1e-5, "dgamma")
    close(be.grad, z["gbeta"], 1e-4, 1e-5, "dbeta")


@pytest.mark.parametrize("N,C,sp", [(2, 320, (2, 2, 2)), (2, 256, (4, 4, 4)), (1, 32, (8, 8, 8)), (3, 5, (3, 3, 3))])
def test_instnorm_lrelu_small_volumes_vs_fp64(N, C, sp):
    """few voxels per instance (the 2^3 / 4^3 stages): accuracy relative to an fp64 evaluation must be fp32-level"""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(N * C)
    x = torch.randn(N, C, *sp, generator=g) * 0.7 + 0.2
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    gy = torch.randn(N, C, *sp, generator=g)
    xr, gr, br = x.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = F.leaky_relu(F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5), 0.01)
    ref.backward(gy.double())
    gx, gg, gb = G(x, True), G(gamma, True), G(beta, True)
    y = ops.InstanceNormLeakyReLUFn.apply(gx, gg, gb, 1e-5, 0.01)
    y.backward(G(gy))
    close(y, ref.detach(), 2e-6, 2e-6, "y")
    close(gx.grad, xr.grad, 5e-6 * float(xr.grad.abs().max()), 1e-5, "dx")
    close(gg.grad, gr.grad, 1e-5 * float(gr.grad.abs().max()), 1e-5, "dgamma")
    close(gb.grad, br.grad, 1e-5 * float(br.grad.abs().max()), 1e-5, "dbeta")


def test_instnorm_lrelu_full_size_vs_oracle():
    """largest instance on the path: [1,32,128^3] (268 MB); statistics over 2 097 152 voxels."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 32, 128, 128, 128, generator=g) * 1.7 + 0.3
    gamma = torch.rand(32, generator=g) + 0.5
    beta = torch.randn(32, generator=g) * 0.1
    ref = F.leaky_relu(F.instance_norm(x, None, None, gamma, beta, True, 0.1, 1e-5), 0.01)
    y = ops.InstanceNormLeakyReLUFn.apply(G(x), G(gamma), G(beta), 1e-5, 0.01)
    close(y, ref, 1e-5, 1e-5, "y")
    # run-to-run determinism (fixed-order reductions, no float atomics)
    y2 = ops.InstanceNormLeakyReLUFn.apply(G(x), G(gamma), G(beta), 1e-5, 0.01)
    assert torch.equal(y, y2)


# ================================================================================================ losses
def test_robust_ce_reference_fixture():
    from multimodal_mvd_seg_amd import losses
    z = load_npz("robust_ce.npz")
    logits = G(z["logits"], True)
    l = losses.RobustCrossEntropyLoss()(logits, G(z["target"]))
    l.backward()
    close(l, z["loss"], 1e-6, 1e-6, "loss")
    close(logits.grad, z["glogits"], 1e-7, 1e-5, "dlogits")


@pytest.mark.parametrize("bd", [0, 1])
def test_dc_and_ce_fixture(bd):
    from multimodal_mvd_seg_amd import losses, ops
    z = load_npz(f"dc_ce_batchdice{bd}.npz")
    logits = G(z["logits"], True)
    lf = losses.DC_and_CE_loss({'batch_dice': bool(bd), 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {},
                               weight_ce=1, weight_dice=1, ignore_label=None,
                               dice_class=losses.MemoryEfficientSoftDiceLoss)
    l = lf(logits, G(z["target"]))
    l.backward()
    close(l, z["loss"], 2e-6, 1e-6, "loss")
    close(logits.grad, z["glogits"], 1e-7, 1e-4, "dlogits")
    counts = ops.argmax_counts(logits.detach(), G(z["target"])).cpu().numpy()
    assert np.array_equal(counts[1:, 0], z["tp"].astype(np.int64))  # integer counts: bit-exact
    assert np.array_equal(counts[1:, 1], z["fp"].astype(np.int64))
    assert np.array_equal(counts[1:, 2], z["fn"].astype(np.int64))


@pytest.mark.parametrize("name", ["distill_kl_c5_T1.npz", "distill_kl_c5_T4.npz", "distill_kl_c1_T1.npz",
                                  "distill_kl_c1_T4.npz"])
def test_distill_kl(name):
    from multimodal_mvd_seg_amd import losses
    z = load_npz(name)
    ys, yt = G(z["ys"], True), G(z["yt"], True)
    l = losses.distill_kl(ys, yt, int(z["T"]))
    l.backward()
    close(l, z["loss"], 1e-7, 1e-5, "loss")
    close(ys.grad, z["gys"], 1e-8, 1e-4, "gys")
    close(yt.grad, z["gyt"], 1e-8, 1e-4, "gyt")


@pytest.mark.parametrize("name", ["feat_kl_T1.npz", "feat_kl_T4.npz"])
def test_feature_kl_ndhwc(name):
    from multimodal_mvd_seg_amd import losses, ops
    z = load_npz(name)
    a, b = G(z["a"]), G(z["b"])
    a, b = ops.to_ndhwc(a).requires_grad_(), ops.to_ndhwc(b).requires_grad_()  # feature maps live in NDHWC
    l = losses.l2_loss(a, b, channel_wise=True, T=int(z["T"]))
    l.backward()
    close(l, z["loss"], 1e-7, 1e-5, "loss")
    close(a.grad, z["ga"], 1e-8, 1e-4, "ga")
    close(b.grad, z["gb"], 1e-8, 1e-4, "gb")


@pytest.mark.parametrize("layout", ["planar", "ndhwc", "mixed"])
def test_l2_loss_plain_mse_reference_fixture(layout):
    """l2_loss(channel_wise=False) (other_loss.py:77-78) against the fixture the reference function produced; feature
    maps arrive NDHWC on the path, planar from outside, and the two operands may differ in layout."""
    from multimodal_mvd_seg_amd import losses, ops
    z = load_npz("l2_loss_plain.npz")
    assert str(z["source"]).startswith("reference ")
    a, b = G(z["a"]), G(z["b"])
    if layout in ("ndhwc", "mixed"):
        a = ops.to_ndhwc(a)
    if layout == "ndhwc":
        b = ops.to_ndhwc(b)
    a.requires_grad_()
    b.requires_grad_()
    l = losses.l2_loss(a, b, channel_wise=False)
    l.backward()
    close(l, z["loss"], 1e-7, 1e-6, "loss")
    close(a.grad, z["ga"], 1e-9, 1e-5, "ga")
    close(b.grad, z["gb"], 1e-9, 1e-5, "gb")
    # odd element count (scalar tail of the float4 loop)
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(3, 5, 7, generator=g), torch.randn(3, 5, 7, generator=g)
    gx = G(x, True)
    lo = ops.MseFn.apply(gx, G(y))
    lo.backward()
    close(lo, ((x - y) ** 2).mean(), 1e-7, 1e-6, "loss (odd n)")
    close(gx.grad, 2 * (x - y) / x.numel(), 1e-9, 1e-5, "grad (odd n)")


# ================================================================================================ soft skeleton
@pytest.mark.parametrize("name", sorted(f for f in os.listdir(GOLDEN) if f.startswith("soft_skel_")))
def test_soft_skel_reference_fixture(name):
    from multimodal_mvd_seg_amd import losses
    z = load_npz(name)
    x = G(z["x"], True)
    assert torch.equal(losses.soft_erode(x.detach()).cpu(), torch.from_numpy(z["erode"]))
    assert torch.equal(losses.soft_dilate(x.detach()).cpu(), torch.from_numpy(z["dilate"]))
    y = losses.soft_skel(x, int(z["iter"]))
    assert torch.equal(y.detach().cpu(), torch.from_numpy(z["skel"])), "soft_skel forward must be bit-exact"
    y.backward(G(z["gy"]))
    close(x.grad, z["gx"], 1e-6, 1e-5, "gx")


@pytest.mark.parametrize("shape,iters", [((2, 1, 19, 21, 45), 3), ((1, 2, 8, 8, 32), 2), ((1, 1, 30, 9, 70), 10), ((2, 1, 128, 128, 128), 3)])
def test_soft_skel_fused_equals_primitive_chain(shape, iters):
    """The LDS-tiled fused step against the one-primitive-per-launch chain (itself pinned to the reference fixtures):
    forward bit for bit -- ragged tiles on every axis, tie-heavy and continuous inputs, the full 2x128^3 vessel map --
    and input gradients bit for bit (same routing codes, same accumulation order)."""
    from multimodal_mvd_seg_amd import losses
    g = torch.Generator().manual_seed(shape[2] + iters)
    for binary in (False, True):
        x = torch.rand(shape, generator=g)
        if binary:
            x = (F.avg_pool3d(x, 3, 1, 1) > 0.5).float()
        gy = torch.randn(shape, generator=g)
        a, b = G(x, True), G(x, True)
        ya = losses.soft_skel(a, iters)
        yb = losses.soft_skel_unfused(b, iters)
        assert torch.equal(ya, yb), f"fused forward differs (binary={binary})"
        ya.backward(G(gy))
        yb.backward(G(gy))
        assert torch.equal(a.grad, b.grad), f"fused backward differs (binary={binary}): max {float((a.grad - b.grad).abs().max()):.3e}"
        with torch.no_grad():  # the no-grad path (target skeleton of soft-clDice) skips the routing codes
            assert torch.equal(losses.soft_skel(G(x), iters), yb.detach())


def test_soft_cldice_fixture():
    from multimodal_mvd_seg_amd import losses
    z = load_npz("soft_cldice.npz")
    p = G(z["pred"], True)
    l = losses.soft_cldice(p, G(z["target"]), int(z["iter"]), float(z["smooth"]))
    l.backward()
    close(l, z["loss"], 1e-6, 1e-5, "loss")
    close(p.grad, z["gpred"], 1e-8, 1e-4, "gpred")


# ================================================================================================ connected components
def test_cc_label_fixture_bit_exact():
    from multimodal_mvd_seg_amd import ops
    d = json.load(open(os.path.join(GOLDEN, "cc_label.json")))
    for case in d["cases"]:
        mask = np.asarray(case["mask"], dtype=np.uint8).reshape(case["shape"])
        labels, count = ops.cc_label(G(mask), case["conn"])
        assert int(count.item()) == case["count"]
        assert np.array_equal(labels.cpu().numpy().reshape(-1), np.asarray(case["labels"], dtype=np.int32))


@pytest.mark.parametrize("conn", [6, 14, 26])
def test_cc_label_large_vs_c_oracle(conn):
    from multimodal_mvd_seg_amd import ops
    from oracle import cc_oracle
    rng = np.random.default_rng(5)
    f = rng.random((48, 64, 56)).astype(np.float32)
    for thr in (0.3, 0.7):  # near / far from the percolation threshold: few huge vs many small components
        m = ops.threshold_mask(G(f), thr)
        assert np.array_equal(m.cpu().numpy(), (f > thr).astype(np.uint8))
        labels, count = ops.cc_label(m, conn)
        ref_labels, ref_n = cc_oracle.cc_label(f > thr, conn)
        assert int(count.item()) == ref_n
        assert np.array_equal(labels.cpu().numpy(), ref_labels)
    for fill in (0, 1):  # empty / full
        labels, count = ops.cc_label(G(np.full((3, 4, 5), fill, dtype=np.uint8)), conn)
        assert int(count.item()) == fill and int(labels.max().item()) == fill


def test_h0_persistence_equals_reference_cpp_fixture():
    """H0 birth/death pairing in the product (device edge keys + radix sort, host elder-rule sweep) against the diagrams
    the reference's own C++ produced (hom.cpp / cohom.cpp compiled from /root/reference): equal as multisets."""
    from multimodal_mvd_seg_amd import ops
    d = json.load(open(os.path.join(GOLDEN, "persistence_grid.json")))
    for case in d["cases"]:
        f = np.asarray(case["f"], dtype=np.float32).reshape(case["shape"])
        b, de, dv = ops.h0_persistence(G(f), case["conn"])
        got = np.stack([b.numpy(), de.numpy()], 1)
        got = got[np.lexsort((got[:, 1], got[:, 0]))]
        want = np.array([[x, np.inf if y is None else y] for x, y in case["dgm0_sorted"]], dtype=np.float32)
        assert np.array_equal(got, want), case["shape"]


@pytest.mark.parametrize("conn", [6, 14, 26])
@pytest.mark.parametrize("sublevel", [True, False])
def test_h0_persistence_large_vs_c_oracle(conn, sublevel):
    """bit-exact, bar for bar (same tie-breaking), incl. the critical vertices used for back-propagation; tie-heavy and
    tie-free fields; the essential bars count the connected components"""
    from multimodal_mvd_seg_amd import ops
    from oracle import cc_oracle
    rng = np.random.default_rng(17 + conn)
    for shape, ties in (((40, 48, 36), False), ((24, 30, 28), True), ((1, 64, 80), True)):
        f = rng.standard_normal(shape).astype(np.float32)
        if ties:
            f = np.round(f * 3) / 3
        b, de, dv = ops.h0_persistence(G(f), conn, sublevel)
        ob, ode, odv = cc_oracle.h0_persistence(f, conn, sublevel)
        assert np.array_equal(b.numpy(), ob) and np.array_equal(de.numpy(), ode) and np.array_equal(dv.numpy(), odv)
        assert int(torch.isinf(de).sum()) == 1


def test_h0_diagram_backward_is_the_reference_scatter():
    """persistence_backward (cohom.cpp:148-196): d birth -> the bar's own vertex, d death -> the critical vertex of the
    killing edge, essential bars contribute their birth only"""
    from multimodal_mvd_seg_amd import ops
    rng = np.random.default_rng(3)
    f = rng.standard_normal((6, 7, 8)).astype(np.float32)
    gf = G(f, True)
    dgm = ops.H0DiagramFn.apply(gf, 6, True)
    assert tuple(dgm.shape) == (f.size, 2) and not dgm.is_cuda
    finite = torch.isfinite(dgm[:, 1])
    w = torch.from_numpy(rng.standard_normal((f.size, 2)).astype(np.float32))
    (torch.where(finite[:, None], dgm, torch.zeros(())) * w).sum().backward()
    _, _, dv = ops.h0_persistence(G(f), 6, True)
    want = torch.where(finite, w[:, 0], torch.zeros(())).clone()
    want.index_add_(0, dv[finite], w[:, 1][finite])
    # the essential bar's birth still receives its gradient when the loss uses it
    assert torch.equal(gf.grad.cpu().reshape(-1), want)
    # total persistence of the finite bars: d/df via the layer == analytic (+1 on death vertices, -1 on births)
    gf2 = G(f, True)
    dgm2 = ops.H0DiagramFn.apply(gf2, 6, True)
    fin = torch.isfinite(dgm2[:, 1])
    (dgm2[fin, 1] - dgm2[fin, 0]).sum().backward()
    want2 = torch.zeros(f.size)
    want2[fin] -= 1
    want2.index_add_(0, dv[fin], torch.ones(int(fin.sum())))
    assert torch.equal(gf2.grad.cpu().reshape(-1), want2)


# ================================================================================================ optimizer
def test_fused_sgd_matches_torch_sgd_with_clipping():
    from multimodal_mvd_seg_amd import optim
    torch.manual_seed(0)
    shapes = [(33, 7, 3), (5,), (1023,), (64, 64)]
    ref = [torch.nn.Parameter(torch.randn(s)) for s in shapes]
    mine = [torch.nn.Parameter(p.detach().clone().to(DEV)) for p in ref]
    o_ref = torch.optim.SGD(ref, 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True)
    o_mine = optim.FusedSGDNesterov(mine, 1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True, max_grad_norm=12)
    for step in range(4):
        o_mine.zero_grad()
        scale = 30.0 if step % 2 == 0 else 0.01  # clipped and un-clipped steps
        for p, q in zip(ref, mine):
            g = torch.randn(p.shape) * scale
            p.grad = g.clone()
            q.grad.copy_(g.to(DEV))
        gn = torch.nn.utils.clip_grad_norm_(ref, 12)
        o_ref.step()
        o_mine.step()
        assert abs(float(o_mine.grad_norm()) - float(gn)) <= 1e-4 * float(gn)
        for p, q in zip(ref, mine):
            close(q, p.detach(), 1e-6, 1e-6, f"param step {step}")


def test_fused_sgd_grad_scale_is_the_data_parallel_mean():
    """grad_scale = 1/world inside the optimizer kernel == `grad *= 1/world` followed by the unscaled step (what DDP's
    SUM all-reduce + division does, nnUNetTrainer.py:220-222), including the clipping decision."""
    from multimodal_mvd_seg_amd import optim
    torch.manual_seed(1)
    base = [torch.randn(s) for s in [(40, 9, 3), (7,), (515,)]]
    for gmag in (40.0, 0.02):
        a = [torch.nn.Parameter(p.clone().to(DEV)) for p in base]
        b = [torch.nn.Parameter(p.clone().to(DEV)) for p in base]
        oa = optim.FusedSGDNesterov(a, 1e-2, weight_decay=3e-5, momentum=0.99, max_grad_norm=12)
        ob = optim.FusedSGDNesterov(b, 1e-2, weight_decay=3e-5, momentum=0.99, max_grad_norm=12)
        oa.grad_scale = 0.25
        for step in range(3):
            oa.zero_grad()
            ob.zero_grad()
            for p, q in zip(a, b):
                g = (torch.randn(p.shape) * gmag).to(DEV)
                p.grad.copy_(g)            # the SUM over 4 ranks
                q.grad.copy_(g * 0.25)     # the mean, taken by a separate pass
            oa.step()
            ob.step()
            assert abs(float(oa.grad_norm()) - float(ob.grad_norm())) <= 1e-6 * float(ob.grad_norm())
            for p, q in zip(a, b):
                close(p, q.detach().cpu(), 1e-7, 1e-6, f"param step {step}")


def test_packed_weight_cache_sees_every_torch_visible_write_and_guards_saved_packs():
    """the packed copies the conv kernels read must follow writes through the flat buffer (dist.broadcast,
    `flat -= ...`), through the parameter, and declared raw writes; a graph kept across optimizer.step() must not
    back-propagate through rebuilt packs silently."""
    from multimodal_mvd_seg_amd import network, ops, optim
    from torch import nn
    torch.manual_seed(0)
    conv = network.HipConv3d(32, 32, 3, 1, padding=1, bias=True).to(DEV)
    tconv = network.HipConvTranspose3d(32, 32, 2, 2, bias=True).to(DEV)
    fp = optim.FlatParams(list(conv.parameters()) + list(tconv.parameters()))
    x = torch.randn(1, 32, 8, 8, 16, device=DEV)

    def fresh():
        c2 = network.HipConv3d(32, 32, 3, 1, padding=1, bias=True).to(DEV)
        t2 = network.HipConvTranspose3d(32, 32, 2, 2, bias=True).to(DEV)
        c2.load_state_dict(conv.state_dict())
        t2.load_state_dict(tconv.state_dict())
        with torch.no_grad():
            return c2(x), t2(x)

    def check(what):
        with torch.no_grad():
            y, t = conv(x), tconv(x)
        fy, ft = fresh()
        assert torch.equal(y, fy) and torch.equal(t, ft), f"stale packed weights after {what}"

    check("construction")
    with torch.no_grad():
        fp.flat.mul_(1.5)                      # in-place write through the flat buffer (what dist.broadcast does)
    check("an in-place write to FlatParams.flat")
    with torch.no_grad():
        conv.weight.add_(0.01)                 # write through the parameter
        tconv.weight.mul_(0.5)
    check("an in-place write to the parameter")
    conv.weight.data.copy_(conv.weight.data * 2)   # invisible to torch's version counters ...
    fp.invalidate_packs()                          # ... so the writer declares it
    check("p.data.copy_ + invalidate_packs()")
    # a graph kept across an optimizer step
    opt = optim.FusedSGDNesterov(fp, 1e-2)
    xg = x.clone().requires_grad_()
    y = conv(xg)
    opt.zero_grad()
    fp.grad.normal_()
    opt.step()                                  # rebuilds every cached pack in place
    with pytest.raises(RuntimeError, match="weights were updated"):
        y.sum().backward()
    check("optimizer.step()")
    y = conv(xg)                                # a fresh graph is fine
    y.sum().backward()
    assert xg.grad is not None


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_pack_cache_is_scoped_per_optimizer(precision):
    """another optimizer's step() must neither re-pack nor invalidate THIS optimizer's weights -- an autograd
    graph kept across it back-propagates (its packs are untouched), its packed copies stay valid, and the stepping
    optimizer's own graph is still guarded."""
    from multimodal_mvd_seg_amd import network, ops, optim
    torch.manual_seed(1)
    ca = network.HipConv3d(32, 32, 3, 1, padding=1, bias=True).to(DEV)
    cb = network.HipConv3d(32, 32, 3, 1, padding=1, bias=True).to(DEV)
    fa, fb = optim.FlatParams(list(ca.parameters())), optim.FlatParams(list(cb.parameters()))
    oa, ob = optim.FusedSGDNesterov(fa, 1e-2), optim.FusedSGDNesterov(fb, 1e-2)
    x = torch.randn(1, 32, 8, 8, 16, device=DEV)
    if precision == "bf16":
        ca.precision = cb.precision = "bf16"
        x = x.to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
    ya, yb = ca(xa), cb(xb)
    ref = torch.autograd.grad(ca(xa), xa, torch.ones_like(ya))[0]
    ob.zero_grad()
    fb.grad.normal_()
    ob.step()                                  # B steps while A's graph is alive
    ya.backward(torch.ones_like(ya))           # ... A's graph is untouched: no "weights were updated", same gradient
    assert torch.equal(xa.grad, ref)
    if precision == "fp32":                    # B's own graph is guarded as before (fp32 packs are rebuilt in place; the bf16
        with pytest.raises(RuntimeError, match="weights were updated"):   # packs of an eager run are fresh buffers per step)
            yb.backward(torch.ones_like(yb))
    with torch.no_grad():                      # and A's packed copies are still the ones of A's (unchanged) weights
        c2 = network.HipConv3d(32, 32, 3, 1, padding=1, bias=True).to(DEV)
        c2.load_state_dict(ca.state_dict())
        if precision == "bf16":
            c2.precision = "bf16"
        assert torch.equal(ca(x), c2(x))


# ================================================================================================ end to end
def _mi355_net_from_fixture(z, in_ch, n_stages):
    from multimodal_mvd_seg_amd import network
    from torch import nn
    net = network.MI355PlainConvUNet(in_ch, n_stages, z["features"].tolist(), nn.Conv3d, 3, z["strides"].tolist(), 2,
                                     int(z["num_classes"]), 2, True, nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True},
                                     None, None, nn.LeakyReLU, {'inplace': True}, deep_supervision=True)
    return net


def test_unet_tiny_train_steps_match_oracle_fixture():
    """inputs, initial state_dict -> logits list, loss, every gradient, parameters after 1 and 3 SGD steps."""
    from multimodal_mvd_seg_amd import losses, optim
    z = load_npz("unet_tiny_step.npz")
    net = _mi355_net_from_fixture(z, 2, 3)
    net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")})
    net.to(DEV)
    opt = optim.FusedSGDNesterov(list(net.parameters()), 1e-2, weight_decay=3e-5, momentum=0.99, max_grad_norm=12)
    loss_fn = losses.DeepSupervisionWrapper(
        losses.DC_and_CE_loss({'batch_dice': False, 'smooth': 1e-5, 'do_bg': False, 'ddp': False}, {}),
        losses.ds_weights(2))
    data, target = G(z["data"]), [G(z["target0"]), G(z["target1"])]
    names = dict(net.named_parameters())
    for step in range(3):
        opt.zero_grad()
        out = net(data)
        l = loss_fn(out, target)
        l.backward()
        if step == 0:
            for i, o in enumerate(out):
                close(o, z[f"logits{i}"], 1e-4, 0, f"logits{i}")
                assert o.is_contiguous() and tuple(o.shape) == tuple(z[f"logits{i}"].shape)
            for n, p in names.items():
                ref = z["grad0/" + n]
                # conv bias before InstanceNorm has an analytically zero gradient (fp noise on both sides)
                tol = 2e-5 if (n.endswith("conv.bias") or "all_modules.0.bias" in n) else 1e-4
                close(p.grad, ref, tol * max(1.0, float(np.abs(ref).max())), 0, "grad " + n)
        close(l, z[f"loss{step}"], 1e-5, 1e-5, f"loss{step}")
        opt.step()
        assert abs(float(opt.grad_norm()) - float(z[f"gradnorm{step}"])) < 1e-3 * float(z[f"gradnorm{step}"])
        if step in (0, 2):
            for n, p in names.items():
                close(p, z[f"sd{step + 1}/" + n], 2e-5 if step == 0 else 1e-4, 0, f"param after step {step + 1}: {n}")
    # Dice parity: identical argmax counts after the same three steps
    from multimodal_mvd_seg_amd import ops
    with torch.no_grad():
        counts = ops.argmax_counts(net(data)[0], target[0]).cpu().numpy()
    assert np.array_equal(counts[1:, 0], z["val_tp"].astype(np.int64))
    assert np.array_equal(counts[1:, 1], z["val_fp"].astype(np.int64))
    assert np.array_equal(counts[1:, 2], z["val_fn"].astype(np.int64))


def test_unet_anisotropic_strides_forward():
    z = load_npz("unet_aniso_fwd.npz")
    net = _mi355_net_from_fixture(z, 1, 3)
    net.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")})
    net.to(DEV)
    with torch.no_grad():
        out = net(G(z["data"]))
    for i, o in enumerate(out):
        close(o, z[f"logits{i}"], 1e-4, 0, f"logits{i}")
    net.decoder.deep_supervision = False  # bare tensor without DS (UNetDecoder.py:117-118)
    with torch.no_grad():
        o = net(G(z["data"]))
    assert torch.is_tensor(o)
    close(o, z["logits0"], 1e-4, 0, "logits (no DS)")


def test_mvd_dual_branch_step_matches_oracle_fixture():
    from multimodal_mvd_seg_amd import trainer
    z = load_npz("mvd_tiny_step.npz")
    plans = trainer.make_plans((16, 16, 16), z["strides"].tolist(), batch_size=2, base_features=8, max_features=32)
    ds = {"channel_names": {"0": "a", "1": "b"}, "labels": {"background": 0, "l1": 1, "l2": 2, "l3": 3}}
    tr = trainer.ContrastiveTrainerMI355(plans, "3d_fullres", 0, ds, device=DEV)
    tr.initialize()
    sd0 = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0/")}
    tr.network.load_state_dict(sd0)
    batch = {"data": G(z["data"]), "target": [G(z["target0"]), G(z["target1"])]}
    tr.skel_iter = int(z["skel_iter"])
    tr.on_train_epoch_start()
    res = tr.train_step(batch)
    assert abs(float(res["loss"]) - float(z["loss0"])) < 2e-5 * max(1.0, abs(float(z["loss0"])))
    assert abs(float(tr.optimizer.grad_norm()) - float(z["gradnorm0"])) < 1e-3 * float(z["gradnorm0"])
    for n, p in tr.network.named_parameters():
        close(p, z["sd1/" + n], 2e-5, 0, "param after step 1: " + n)


def _cfg2_pair(P, batch_size=1, n_stages=6):
    from multimodal_mvd_seg_amd import trainer
    from oracle import loss_oracle as LO, step_oracle as SO, unet_oracle as UO
    strides = UO.CONFIGS["cfg2"]["strides"][:n_stages]
    ora = UO.build_plainconv_unet(4, 5, n_stages, strides, seed=0)
    batch = SO.synthetic_batch(batch_size, 4, (P, P, P), strides, num_classes=5, seed=1234)
    loss_fn = LO.build_loss(len(batch["target"]))
    plans = trainer.make_plans((P, P, P), strides, batch_size=batch_size)
    ds = {"channel_names": {str(i): str(i) for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=DEV)
    tr.initialize()
    tr.network.load_state_dict(ora.state_dict())
    return ora, loss_fn, batch, tr


def test_cfg2_network_4x64cube_train_step_vs_oracle():
    """The cfg-2 network (6 stages, 31.2 M parameters, 4 modalities) on a 64^3 patch against the torch-CPU fp32 oracle
    evaluated on this box: logits within 1e-4, loss within 1e-5, global grad norm within 1e-3 relative, every parameter
    after one clip+SGD-Nesterov step within 1e-5."""
    from oracle import step_oracle as SO
    ora, loss_fn, batch, tr = _cfg2_pair(64)
    opt = SO.make_optimizer(ora.parameters())
    l_ref, out_ref, gn_ref = SO.train_step(ora, loss_fn, opt, batch)
    gbatch = {"data": batch["data"].to(DEV), "target": [t.to(DEV) for t in batch["target"]]}
    with torch.no_grad():
        out = tr.network(gbatch["data"])
    for i, (o, r) in enumerate(zip(out, out_ref)):
        close(o, r.detach(), 1e-4, 0, f"logits{i}")
    tr.on_train_epoch_start()
    res = tr.train_step(gbatch)
    assert abs(float(res["loss"]) - float(l_ref)) < 1e-5 * max(1.0, abs(float(l_ref)))
    assert abs(float(tr.optimizer.grad_norm()) - gn_ref) < 1e-3 * gn_ref
    ref_params = dict(ora.named_parameters())
    for n, p in tr.network.named_parameters():
        close(p, ref_params[n].detach(), 1e-5, 0, "param after step: " + n)


# The gradient parity of the cfg-2 network (all six stages, against an fp64 evaluation with the LeakyReLU branch pattern of
# the HIP forward), the per-block backward checks at the real cfg-2 layer shapes and the full 4x128^3 step live in
# tests/test_gpu_cfg2.py.


# ================================================================================================ inference (8f-1)
@pytest.mark.parametrize("mask", [1, 2, 4, 3, 5, 6, 7])
def test_flip_add_kernel_equals_torch_flip(mask):
    from multimodal_mvd_seg_amd._lib import call
    import ctypes
    g = torch.Generator().manual_seed(mask)
    x = torch.randn(3, 5, 6, 7, generator=g)
    y = torch.randn(3, 5, 6, 7, generator=g)
    dims = tuple(a + 1 for a in range(3) if mask & (1 << a))
    xd, yd = x.to(DEV), y.to(DEV)
    out = torch.empty_like(xd)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    call("mvd_flip_add", P(xd), P(out), 3, 5, 6, 7, mask, 0, s)
    assert torch.equal(out.cpu(), torch.flip(x, dims))
    call("mvd_flip_add", P(xd), P(yd), 3, 5, 6, 7, mask, 1, s)
    assert torch.equal(yd.cpu(), y + torch.flip(x, dims))


@pytest.mark.parametrize("mirror,gauss", [((0, 1, 2), True), ((), True), ((1, 2), False)])
def test_sliding_window_prediction_vs_oracle(mirror, gauss):
    """SlidingWindowPredictor against the CPU restatement of predict_sliding_window_return_logits, both driving the SAME
    HIP network (the tile logits are identical; what is compared is padding, step placement, mirror TTA, Gaussian
    blending and normalisation).  Image smaller than the patch on one axis (padding path), several tiles on the others."""
    from torch import nn
    from multimodal_mvd_seg_amd.inference import SlidingWindowPredictor
    from multimodal_mvd_seg_amd.network import MI355PlainConvUNet, InitWeights_He
    from oracle import infer_oracle as IO
    torch.manual_seed(3)
    net = MI355PlainConvUNet(4, 3, [32, 64, 128], nn.Conv3d, 3, [[1, 1, 1], [2, 2, 2], [2, 2, 2]], 2, 5, 2, True,
                             nn.InstanceNorm3d, {'eps': 1e-5, 'affine': True}, None, None, nn.LeakyReLU,
                             {'inplace': True}, deep_supervision=True).to(DEV)
    net.apply(InitWeights_He(1e-2))
    patch = (32, 32, 32)
    img = torch.randn(4, 28, 44, 50, generator=torch.Generator().manual_seed(5))
    pred = SlidingWindowPredictor(net, patch, 5, 0.5, gauss, len(mirror) > 0, mirror, DEV)
    out = pred.predict_sliding_window_return_logits(img)
    assert tuple(out.shape) == (5, 28, 44, 50)
    assert net.decoder.deep_supervision is True  # restored

    def oracle_net(x):
        net.eval()
        net.decoder.deep_supervision = False
        with torch.no_grad():
            y = net(x.to(DEV)).cpu()
        net.decoder.deep_supervision = True
        return y

    ref = IO.predict_sliding_window_return_logits(oracle_net, img, patch, 5, 0.5, gauss, mirror if mirror else None)
    close(out, ref, 2e-5, 1e-5, "sliding-window logits")


# ------------------------------------------------------------------------------------------ post-processing (8f-3)
def _pp_seg(rng, shape, thr=0.55):
    from scipy import ndimage
    f = ndimage.gaussian_filter(rng.random(shape), 1.5)
    f = (f - f.min()) / (f.max() - f.min())
    seg = np.zeros(shape, dtype=np.int32)
    seg[f > thr] = 1
    seg[f > thr + 0.12] = 2
    seg[f < 0.25] = 3
    return seg


@pytest.mark.gpu
@pytest.mark.parametrize("conn", [6, 26])
@pytest.mark.parametrize("keep", [1, 2])
def test_postproc_keep_largest_bit_exact(conn, keep):
    from multimodal_mvd_seg_amd import postprocessing as PP
    from oracle import postproc_oracle as PO
    rng = np.random.default_rng(100 + conn + keep)
    for shape, thr in (((24, 40, 56), 0.55), ((33, 17, 70), 0.6), ((5, 7, 300), 0.5)):
        seg = _pp_seg(rng, shape, thr)
        for labels in (1, [1, 2], [(1, 2), 3], (2, 3)):
            ref = PO.remove_all_but_largest_component_from_segmentation(seg, labels, 0, keep, conn)
            got = PP.remove_all_but_largest_component_from_segmentation(seg, labels, 0, num_components=keep,
                                                                        connectivity=conn)
            assert got.dtype == seg.dtype and np.array_equal(got, ref), (shape, labels)
            # idempotent: a second pass changes nothing
            again = PP.remove_all_but_largest_component_from_segmentation(got, labels, 0, num_components=keep,
                                                                          connectivity=conn)
            assert np.array_equal(again, got)


@pytest.mark.gpu
def test_postproc_sizes_ties_and_edge_cases():
    from multimodal_mvd_seg_amd import ops, postprocessing as PP
    from oracle import postproc_oracle as PO
    # kept = {label0, label1, size0, size1}; ties go to the first component in scan order
    seg = np.zeros((4, 5, 40), dtype=np.int32)
    seg[0, 0, 0:3] = 1
    seg[1, 2, 10:13] = 1
    seg[3, 4, 30:33] = 1
    seg[3, 0, 0:37] = 1       # 37 voxels: crosses several 16-voxel runs
    m = ops.seg_label_mask(G(seg), [1])
    cc, count = ops.cc_label(m, 26)
    kept = ops.cc_keep_largest(cc, 2).cpu().numpy()
    assert int(count.item()) == 4
    assert kept.tolist() == [1 + (3 * 5 + 0) * 40, 1, 37, 3]
    ref = PO.remove_all_but_largest_component_from_segmentation(seg, 1, 9)
    got = PP.remove_all_but_largest_component_from_segmentation(seg, 1, background_label=9)
    assert np.array_equal(got, ref)
    # empty mask / one component / label not present: no-ops; device tensors stay on the device; input untouched
    z = torch.zeros((3, 4, 5), dtype=torch.int32, device=DEV)
    out = PP.remove_all_but_largest_component_from_segmentation(z, 1)
    assert out.is_cuda and torch.equal(out, z)
    one = np.ones((3, 4, 5), dtype=np.int64)
    got = PP.remove_all_but_largest_component_from_segmentation(one, [1], 0)
    assert got.dtype == np.int64 and np.array_equal(got, one)
    with pytest.raises(ValueError):
        PP.remove_all_but_largest_component_from_segmentation(one, list(range(17)))
    with pytest.raises(RuntimeError):
        ops.cc_keep_largest(cc, 3)


@pytest.mark.gpu
def test_postproc_full_volume_properties():
    # full-size predicted volume (192 x 256 x 256): size-independent properties instead of the scipy oracle
    from multimodal_mvd_seg_amd import ops, postprocessing as PP
    g = torch.Generator(device="cpu").manual_seed(5)
    f = torch.rand((1, 1, 48, 64, 64), generator=g)
    f = torch.nn.functional.interpolate(f, scale_factor=4, mode="trilinear").squeeze().to(DEV)
    seg = (f > 0.62).to(torch.int32) + (f > 0.7).to(torch.int32)
    out = PP.remove_all_but_largest_component_from_segmentation(seg, [1, 2], 0)
    assert torch.equal(out[out != 0], seg[out != 0])            # only removals, never relabels a kept voxel
    m = ops.seg_label_mask(out, [1, 2])
    cc, count = ops.cc_label(m, 26)
    assert int(count.item()) <= 2
    kept0 = ops.cc_keep_largest(ops.cc_label(ops.seg_label_mask(seg, [1, 2]), 26)[0], 2).cpu()
    assert int(m.sum().item()) == int(kept0[2] + kept0[3])      # surviving voxels == the two component sizes
    kept1 = ops.cc_keep_largest(cc, 2).cpu()
    assert kept1[2:].tolist() == kept0[2:].tolist()
    assert torch.equal(PP.remove_all_but_largest_component_from_segmentation(out, [1, 2], 0), out)


# ------------------------------------------------------------------------------------------ device feed (8f-2)
class _FeedDataset:
    def __init__(self, shapes, channels, seed):
        rng = np.random.default_rng(seed)
        self.cases = {}
        for i, shp in enumerate(shapes):
            data = rng.standard_normal((channels, *shp)).astype(np.float32)
            seg = (rng.random((1, *shp)) > 0.9).astype(np.int16) * rng.integers(1, 5, (1, *shp)).astype(np.int16)
            seg[0, 0, 0, 0] = -1  # an ignore-style label inside the volume is removed as well
            self.cases[f"c{i}"] = (data, seg, {"class_locations": {c: np.argwhere(seg == c) for c in (1, 2, 3, 4)}})

    def keys(self):
        return self.cases.keys()

    def load_case(self, k):
        return self.cases[k]


class _FeedLabels:
    all_labels = [1, 2, 3, 4]
    has_ignore_label = False


@pytest.mark.gpu
def test_device_feed_batches_bit_exact_vs_oracle():
    from multimodal_mvd_seg_amd.dataloading import DeviceDataLoader3D
    from oracle import feed_oracle as FO
    ds = _FeedDataset([(40, 44, 52), (17, 60, 30), (33, 20, 70)], 4, 1)
    cases = {k: (v[0], v[1]) for k, v in ds.cases.items()}
    for patch, scales in (((32, 32, 32), [1, 0.5, 0.25, 0.125]), ((24, 40, 16), [(1, 1, 1), (1, 0.5, 0.5), (0.5, 0.25, 0.25)])):
        dl = DeviceDataLoader3D(ds, 6, patch, patch, _FeedLabels(), oversample_foreground_percent=0.33,
                                mirror_axes=(0, 1, 2), deep_supervision_scales=scales, device=DEV)
        np.random.seed(11)
        for _ in range(3):
            plan = dl.plan_batch()
            got = dl.generate_train_batch(plan)
            ref_data, ref_t = FO.generate_train_batch(cases, plan[0], plan[1], plan[2], patch, scales)
            assert got["data"].dtype == torch.float32 and np.array_equal(got["data"].cpu().numpy(), ref_data)
            assert len(got["target"]) == len(scales)
            for g, r in zip(got["target"], ref_t):
                assert g.dtype == torch.float32 and tuple(g.shape) == r.shape and np.array_equal(g.cpu().numpy(), r)
                assert float(g.min()) >= 0  # the -1 padding / ignore label never reaches the loss
    # no deep supervision: a bare tensor
    dl = DeviceDataLoader3D(ds, 2, (16, 16, 16), (16, 16, 16), _FeedLabels(), device=DEV)
    b = next(dl)
    assert torch.is_tensor(b["target"]) and tuple(b["target"].shape) == (2, 1, 16, 16, 16)


@pytest.mark.gpu
def test_device_feed_kernels_edges_and_full_size():
    from multimodal_mvd_seg_amd import dataloading as DLD
    from oracle import feed_oracle as FO
    rng = np.random.default_rng(2)
    vol = rng.standard_normal((3, 9, 10, 11)).astype(np.float32)
    seg = rng.integers(-1, 4, (1, 9, 10, 11)).astype(np.int16)
    gv, gs = G(vol), G(seg)
    patch = (8, 12, 6)
    for lbs in ([-3, -1, 7], [5, 4, -5], [-7, 0, 0], [8, 9, 10], [0, -2, 3]):  # overhangs down to a one-voxel overlap
        for mask in (0, 1, 2, 4, 7):
            out = torch.empty((3, *patch), dtype=torch.float32, device=DEV)
            DLD.crop_pad_data(gv, out, lbs, mask, 0.0)
            assert np.array_equal(out.cpu().numpy(), FO.mirror(FO.crop_pad(vol, lbs, patch, 0), mask))
            t = torch.empty((1, *patch), dtype=torch.float32, device=DEV)
            DLD.crop_pad_seg(gs, t, lbs, mask, -1, replace=(-1, 0))
            assert np.array_equal(t.cpu().numpy(),
                                  FO.remove_label(FO.mirror(FO.crop_pad(seg, lbs, patch, -1), mask)).astype(np.float32))
            DLD.crop_pad_seg(gs, t, lbs, mask, -1)  # without RemoveLabel the padding stays -1
            assert np.array_equal(t.cpu().numpy(), FO.mirror(FO.crop_pad(seg, lbs, patch, -1), mask).astype(np.float32))
    # a box that misses the volume (get_bbox never produces one; the reference's slicing would mis-shape): all padding
    out = torch.empty((3, *patch), dtype=torch.float32, device=DEV)
    DLD.crop_pad_data(gv, out, [-20, 0, 0], 0, 5.0)
    assert bool((out == 5.0).all())
    with pytest.raises(RuntimeError):
        DLD.crop_pad_data(gv, torch.empty((2, *patch), dtype=torch.float32, device=DEV), [0, 0, 0])
    # BASELINE patch from a full-size case: an in-bounds box is a plain slice, DS targets are strided picks
    g = torch.Generator(device="cpu").manual_seed(0)
    big = torch.randn((4, 150, 200, 170), generator=g).to(DEV)
    bseg = torch.randint(0, 5, (1, 150, 200, 170), generator=g).to(torch.int16).to(DEV)
    out = torch.empty((4, 128, 128, 128), dtype=torch.float32, device=DEV)
    DLD.crop_pad_data(big, out, [10, 50, 30])
    assert torch.equal(out, big[:, 10:138, 50:178, 30:158])
    t = torch.empty((1, 1, 128, 128, 128), dtype=torch.float32, device=DEV)
    DLD.crop_pad_seg(bseg, t[0], [10, 50, 30], 0, -1, replace=(-1, 0))
    assert torch.equal(t[0], bseg[:, 10:138, 50:178, 30:158].float())
    for k, s in enumerate((0.5, 0.25, 0.125, 0.0625)):
        f = 2 << k
        d = DLD.downsample_seg(t, s)
        assert tuple(d.shape) == (1, 1, 128 // f, 128 // f, 128 // f)
        assert torch.equal(d, t[:, :, f // 2::f, f // 2::f, f // 2::f])


@pytest.mark.gpu
def test_packed_weight_cache_follows_every_kind_of_update():
    """The packed copies are cached on the weight tensor: a torch in-place write (version bump), the fused optimizer
    (raw-pointer update -> epoch bump + one batched re-pack) and a brand-new tensor at a recycled address must all be
    seen.  Conv output after each kind of change == conv with freshly packed weights (bit-exact) and == torch."""
    from multimodal_mvd_seg_amd import ops, optim
    g = torch.Generator().manual_seed(3)
    conv = torch.nn.Conv3d(32, 32, 3, padding=1).to(DEV)
    convT = torch.nn.ConvTranspose3d(32, 32, 2, stride=2).to(DEV)
    x = torch.randn(1, 32, 8, 8, 16, generator=g).to(DEV)
    params = list(conv.parameters()) + list(convT.parameters())
    opt = optim.FusedSGDNesterov(params, 0.1, weight_decay=0.0, momentum=0.9, max_grad_norm=12)

    def run():
        y = ops.Conv3dFn.apply(x, None, conv.weight, conv.bias, (1, 1, 1))
        z = ops.ConvTranspose3dFn.apply(ops.widen(y) if hasattr(ops, "widen") and y.dtype != torch.float32 else y,
                                        convT.weight, convT.bias, (2, 2, 2))
        return y, z

    def check(tag):
        y, z = run()
        ref_y = F.conv3d(x, conv.weight, conv.bias, 1, 1)
        ref_z = F.conv_transpose3d(ref_y, convT.weight, convT.bias, 2)
        close(y, ref_y.detach().cpu(), 2e-5, 1e-5, tag + " conv")
        close(z, ref_z.detach().cpu(), 5e-5, 1e-5, tag + " convT")
        return y, z

    from multimodal_mvd_seg_amd._lib import call, query
    call("mvd_set_wino_min_items", 1)  # Winograd tables in play
    try:
        check("initial")
        wino_on = query("mvd_wino_mode") != 0
        assert getattr(conv.weight, "_mvd_pack", None) is not None and (conv.weight._mvd_pack.uf is not None) == wino_on
        with torch.no_grad():
            conv.weight.mul_(1.5)               # version bump
            convT.weight.add_(0.01)
        check("after in-place write")
        y, z = run()
        (y.sum() + z.sum()).backward()
        opt.step()                              # raw-pointer update + repack_all
        e = conv.weight._mvd_pack
        if query("mvd_wino_mode") != 1:  # F(2,3) tables are not in the batch entry: those entries go stale and are
            assert e.stamp == ops._pack_stamp(conv.weight.detach(), conv.weight)  # re-packed per layer by the next forward
        y1, z1 = check("after optimizer step")
        assert e.stamp == ops._pack_stamp(conv.weight.detach(), conv.weight)
        # the batched pack == the per-layer pack, bit for bit
        wf, wb = ops.pack_weight(conv.weight, False)
        assert torch.equal(wf, e.wf) and torch.equal(wb, e.wb)
        if wino_on:
            uf = torch.empty_like(e.uf)
            call("mvd_pack_weight_wino", ctypes.c_void_p(conv.weight.data_ptr()), ctypes.c_void_p(uf.data_ptr()), None, 32, 32,
                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert torch.equal(uf, e.uf)
        wfT, wbT = ops.pack_weight(convT.weight, True)
        assert torch.equal(wfT, convT.weight._mvd_pack.wf) and torch.equal(wbT, convT.weight._mvd_pack.wb)
        # a fresh tensor (possibly at a recycled address) never inherits an entry
        w2 = (conv.weight.detach() * 0.5).clone()
        y2 = ops.Conv3dFn.apply(x, None, w2, conv.bias.detach(), (1, 1, 1))
        close(y2, F.conv3d(x, w2, conv.bias, 1, 1).detach().cpu(), 2e-5, 1e-5, "fresh tensor")
    finally:
        call("mvd_set_wino_min_items", -1)


@pytest.mark.gpu
def test_trainer_fed_by_device_loader_learns_and_postprocesses():
    """The widened rows together: DeviceDataLoader3D (8f-2) feeds nnUNetTrainerMI355.train_step (the hot path), the
    trained net predicts through the sliding-window predictor (8f-1) and the result goes through the connected-component
    post-processing (8f-3).  Synthetic task: label = which of two intensity blobs a voxel belongs to."""
    from multimodal_mvd_seg_amd import trainer, inference, postprocessing as PP
    from multimodal_mvd_seg_amd.dataloading import DeviceDataLoader3D
    rng = np.random.default_rng(0)

    class DS:
        def __init__(self):
            self.cases = {}
            for i in range(4):
                shp = (40, 44, 48)
                zz, yy, xx = np.meshgrid(*[np.arange(s) for s in shp], indexing="ij")
                seg = np.zeros((1, *shp), dtype=np.int16)
                c1 = rng.integers(10, 30, 3)
                c2 = rng.integers(10, 30, 3)
                seg[0][(zz - c1[0]) ** 2 + (yy - c1[1]) ** 2 + (xx - c1[2]) ** 2 < 64] = 1
                seg[0][(zz - c2[0] - 8) ** 2 + (yy - c2[1] - 8) ** 2 + (xx - c2[2] - 12) ** 2 < 49] = 2
                data = np.stack([(seg[0] == 1) * 1.0, (seg[0] == 2) * 1.0], 0).astype(np.float32)
                data += 0.3 * rng.standard_normal(data.shape).astype(np.float32)
                locs = {c: np.argwhere(seg == c) for c in (1, 2)}
                self.cases[f"c{i}"] = (data, seg, {"class_locations": locs})

        def keys(self):
            return self.cases.keys()

        def load_case(self, k):
            return self.cases[k]

    ds = DS()
    patch = (32, 32, 32)
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    plans = trainer.make_plans(patch, strides, batch_size=2, base_features=32, max_features=64)
    dj = {"channel_names": {"0": "a", "1": "b"}, "labels": {"background": 0, "one": 1, "two": 2}}
    torch.manual_seed(0)
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, dj, device=DEV)
    tr.initialize()
    tr.on_train_epoch_start()
    dl = DeviceDataLoader3D(ds, tr.batch_size, patch, patch, tr.label_manager, oversample_foreground_percent=0.33,
                            mirror_axes=(0, 1, 2), deep_supervision_scales=tr._get_deep_supervision_scales(), device=DEV)
    np.random.seed(0)
    losses = []
    for _ in range(60):
        b = next(dl)
        assert b["data"].is_cuda and len(b["target"]) == len(tr._get_deep_supervision_scales())
        losses.append(float(tr.train_step(b)["loss"]))
    print('feed-trained losses', np.mean(losses[:5]), np.mean(losses[-10:]))
    assert np.mean(losses[-10:]) < np.mean(losses[:5]) - 0.3, (losses[:5], losses[-10:])
    # predict one whole case tile by tile, argmax, keep the two largest components of the foreground
    tr.set_deep_supervision_enabled(False)
    tr.network.eval()
    data, seg, _ = ds.load_case("c0")
    pred = inference.SlidingWindowPredictor(tr.network, patch, 3, tile_step_size=0.5, use_gaussian=True,
                                            use_mirroring=False, device=DEV)
    with torch.no_grad():
        logits = pred.predict_sliding_window_return_logits(torch.from_numpy(data).to(DEV))
    hard = logits.argmax(0).to(torch.int32)
    fg = (hard > 0)
    ref_fg = torch.from_numpy(seg[0] > 0).to(DEV)
    dice = 2.0 * float((fg & ref_fg).sum()) / max(1.0, float(fg.sum() + ref_fg.sum()))
    print('feed-trained foreground dice', dice)
    assert dice > 0.6, dice
    cleaned = PP.remove_all_but_largest_component_from_segmentation(hard, [1, 2], 0)
    from multimodal_mvd_seg_amd import ops
    _, count = ops.cc_label(ops.seg_label_mask(cleaned, [1, 2]), 26)
    assert int(count.item()) <= 2 and bool(((cleaned == hard) | (cleaned == 0)).all())


@pytest.mark.gpu
@pytest.mark.parametrize("C,K,sp,N", [
    (32, 64, (8, 8, 16), 1),      # whole tiles
    (32, 32, (9, 7, 13), 2),      # odd sizes: the last odd-parity voxel has no o = j + 1 neighbour, ragged tiles
    (64, 96, (6, 10, 18), 1),     # two input-gradient channel blocks, three reduce chunks
])
def test_stride2_dgrad_fused_parity_classes_vs_fp64(C, K, sp, N):
    """Input gradient of the 3x3x3 stride-2 conv through the fused eight-class kernel (forced on for small problems)
    against fp64 and against the per-class gather launches."""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call
    g = torch.Generator().manual_seed(C + K + sp[1])
    x = torch.randn(N, C, *sp, generator=g)
    w = torch.randn(K, C, 3, 3, 3, generator=g) * (1.0 / np.sqrt(27 * C))
    b = torch.randn(K, generator=g) * 0.1
    xr = x.double().requires_grad_()
    ref = F.conv3d(xr, w.double(), b.double(), 2, 1)
    gy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(gy)
    outs = {}
    try:
        for mode, mi in (("fused", 1), ("classes", -1)):
            call("mvd_set_wino_min_items", mi)
            gx = G(x, True)
            y = ops.Conv3dFn.apply(gx, None, G(w, True), G(b, True), (2, 2, 2))
            y.backward(G(gy.float()))
            outs[mode] = gx.grad
            close(y, ref.detach(), 1e-5, 1e-5, "y")
    finally:
        call("mvd_set_wino_min_items", -1)
    close(outs["fused"], xr.grad, 1e-5, 1e-5, "dx fused")
    close(outs["classes"], xr.grad, 1e-5, 1e-5, "dx per class")
    close(outs["fused"], outs["classes"].cpu(), 1e-5, 1e-5, "fused vs per class")
