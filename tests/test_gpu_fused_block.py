"""The fused bf16 block of the north_star: Conv3d 3x3x3 -> InstanceNorm3d(affine) -> LeakyReLU
(get_network_from_plans.py:41-44) with (a) the InstanceNorm statistics out of the conv's epilogue and (b) the
InstanceNorm-apply + LeakyReLU of the producing block inside the consumer conv's loader (k_fwd16z<.., FUSE>,
mvd_conv3d_fwd_bf16_fused / mvd_instnorm_finalize_tiles / mvd_instnorm_lrelu_apply_bf16, ops.NormActConv3dFn).

Checks, all through the C ABI:
* epilogue statistics == sums over the stored bf16 tensor (fp64 on the host), mean / rstd / scale / shift from them;
* the apply pass in scale / shift form against fp64;
* conv with the loader prologue == conv over the materialised activation, BIT FOR BIT (same arithmetic, by construction),
  on ragged volumes too (halo voxels outside the volume must stay zero AFTER the activation) and on exact-integer data
  against torch;
* the fused block against an fp64 evaluation ("an fp64 block test for the fused path");
* autograd: NormActConv3dFn (forward + backward) bit-identical to the un-fused chain; the network's inference forward
  with and without the fusion bit-identical; the statistics epilogue against the stand-alone statistics pass in a train step.
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
CL, BF = torch.channels_last_3d, torch.bfloat16


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from multimodal_mvd_seg_amd import _lib
    _lib.load()


def _conv_fused(x, w, b, scale=None, shift=None, slope=0.01, want_stats=True):
    """raw C-ABI call: returns (y, stats or None, ntiles)"""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call, i3, query
    N, C, D, H, W = x.shape
    K = w.shape[0]
    wf, _ = ops.pack_weight_bf16(w, False)
    y = ops.empty_cl3d((N, K, D, H, W), DEV, BF)
    ws = torch.empty(max(1024, query("mvd_conv_fwd_workspace_bytes", N, D * H * W, K)), dtype=torch.uint8, device=DEV)
    nt = query("mvd_conv3d_fwd_bf16_stats_tiles", N, D, H, W, C, 0, K, i3((3, 3, 3)), i3((1, 1, 1)))
    stats = torch.full((N, max(nt, 1), K, 2), float("nan"), dtype=torch.float32, device=DEV) if want_stats else None
    got = ctypes.c_int(-1)
    call("mvd_conv3d_fwd_bf16_fused", _p(x), C, None, 0, _p(wf), _p(b), _p(y), N, D, H, W, K, i3((3, 3, 3)), i3((1, 1, 1)),
         _p(scale), _p(shift), float(slope), _p(stats), ctypes.byref(got), _p(ws), ws.numel(), _stream())
    return y, stats, got.value, nt


def _sum_tolerances(yd):
    """The epilogue sums the fp32 accumulators, the stored tensor holds them rounded to bf16 (relative error uniform in
    +-2^-9): the two sums differ by the accumulated rounding noise.  Six standard deviations of it, per (n, c):
    sum: 2^-9 sqrt(sum y^2 / 3); sum of squares: 2 * 2^-9 sqrt(sum y^4 / 3)."""
    e = 2.0 ** -9
    return 6 * e * torch.sqrt((yd ** 2).sum(1) / 3) + 1e-6 * yd.abs().sum(1), \
        6 * 2 * e * torch.sqrt((yd ** 4).sum(1) / 3) + 1e-6 * (yd ** 2).sum(1)


def _rand_block(N, D, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(N, 32, D, H, W, generator=g) * 1.5 + 0.3).to(BF).to(DEV).contiguous(memory_format=CL)
    w = (torch.randn(32, 32, 3, 3, 3, generator=g) / np.sqrt(27 * 32)).to(DEV)
    b = (torch.randn(32, generator=g) * 0.1).to(DEV)
    gamma = (torch.rand(32, generator=g) + 0.5).to(DEV)
    beta = (torch.randn(32, generator=g) * 0.2).to(DEV)
    return x, w, b, gamma, beta


@pytest.mark.parametrize("N,D,H,W", [(2, 128, 128, 128), (1, 64, 64, 64), (2, 44, 48, 64), (1, 72, 64, 96)])
def test_epilogue_statistics_equal_the_sums_over_the_stored_tensor(N, D, H, W):
    from multimodal_mvd_seg_amd._lib import call
    x, w, b, gamma, beta = _rand_block(N, D, H, W, D + W)
    y, stats, got, nt = _conv_fused(x, w, b)
    if nt == 0:
        pytest.skip("no kernel with the statistics epilogue takes this shape in this build")
    assert got == nt, (nt, got)
    y_plain, _s, _g, _n = _conv_fused(x, w, b, want_stats=False)
    assert torch.equal(y, y_plain), "the statistics epilogue must not change the conv's output"
    assert bool(torch.isfinite(stats).all()), "a tile wrote no statistics"
    yd = y.permute(0, 2, 3, 4, 1).reshape(N, -1, 32).double()      # [N, V, C] of the STORED (bf16-rounded) values
    ref_s, ref_q = yd.sum(1), (yd * yd).sum(1)
    tot = stats.double().sum(1)                                     # [N, C, 2]
    tol_s, tol_q = _sum_tolerances(yd)
    assert bool(((tot[..., 0] - ref_s).abs() <= tol_s).all()), float(((tot[..., 0] - ref_s).abs() / tol_s).max())
    assert bool(((tot[..., 1] - ref_q).abs() <= tol_q).all()), float(((tot[..., 1] - ref_q).abs() / tol_q).max())
    V = D * H * W
    mean, rstd, scale, shift = (torch.empty((N, 32), dtype=torch.float32, device=DEV) for _ in range(4))
    call("mvd_instnorm_finalize_tiles", _p(stats), nt, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(scale), _p(shift), N, V, 32,
         1e-5, _stream())
    m64 = ref_s / V
    v64 = ref_q / V - m64 * m64
    r64 = 1.0 / torch.sqrt(v64 + 1e-5)
    assert bool(((mean.double() - m64).abs() <= tol_s / V + 1e-7 * yd.abs().amax(1)).all())
    assert float(((rstd.double() - r64) / r64).abs().max()) < 2e-5
    assert torch.equal(scale, gamma[None] * rstd)
    assert torch.equal(shift, torch.addcmul(beta[None].expand(N, 32), -mean, scale)) or \
        float((shift.double() - (beta.double()[None] - mean.double() * scale.double())).abs().max()) < 1e-6
    # the apply pass in scale / shift form against fp64 on the same bf16 input: one rounding to bf16
    a = torch.empty_like(y)
    call("mvd_instnorm_lrelu_apply_bf16", _p(y), _p(scale), _p(shift), _p(a), N, V, 32, 0.01, _stream())
    z = y.double() * scale.double()[:, :, None, None, None] + shift.double()[:, :, None, None, None]
    ref = torch.where(z > 0, z, 0.01 * z)
    err = (a.double() - ref).abs()
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-30).all()), float(err.max())


@pytest.mark.parametrize("N,D,H,W", [(2, 128, 128, 128), (1, 64, 64, 64), (2, 40, 44, 70), (1, 33, 70, 97), (3, 17, 41, 130)])
def test_loader_prologue_equals_the_conv_over_the_materialised_activation_bit_for_bit(N, D, H, W):
    """conv(lrelu(IN(y0))) two ways: the apply pass materialises a0 and the plain conv reads it; the fused conv reads y0 and
    normalises in its loader.  Same arithmetic (fma, max, one rounding) -> the outputs must be IDENTICAL -- including the
    border tiles, where the zero padding has to apply to a0 (lrelu(shift) != 0 for a padded raw zero)."""
    from multimodal_mvd_seg_amd._lib import call
    y0, w, b, gamma, beta = _rand_block(N, D, H, W, 7 * D + H)
    g = torch.Generator().manual_seed(5)
    scale = (torch.rand(N, 32, generator=g) + 0.3).to(DEV)
    shift = (torch.randn(N, 32, generator=g) * 0.5 + 0.4).to(DEV)     # mostly positive: lrelu(shift) far from zero
    a0 = torch.empty_like(y0)
    call("mvd_instnorm_lrelu_apply_bf16", _p(y0), _p(scale), _p(shift), _p(a0), N, D * H * W, 32, 0.01, _stream())
    y_ref, _s, _g, _n = _conv_fused(a0, w, b, want_stats=False)
    y_fused, stats, got, nt = _conv_fused(y0, w, b, scale, shift, 0.01, want_stats=True)
    assert torch.equal(y_fused, y_ref)
    if nt > 0:   # prologue + statistics together
        assert got == nt
        yd = y_fused.permute(0, 2, 3, 4, 1).reshape(N, -1, 32).double()
        tol_s, tol_q = _sum_tolerances(yd)
        assert bool(((stats.double().sum(1)[..., 1] - (yd * yd).sum(1)).abs() <= tol_q).all())
        assert bool(((stats.double().sum(1)[..., 0] - yd.sum(1)).abs() <= tol_s).all())


def test_loader_prologue_exact_integer_data_against_torch():
    """Small-integer y0, power-of-two scale, integer shift, slope 0.5: a0 and every partial sum are exact -> the fused conv
    must equal torch's conv over torch's activation, rounded once to bf16, bit for bit (ragged 33 x 70 x 97 volume)."""
    N, D, H, W = 1, 33, 70, 97
    g = torch.Generator().manual_seed(11)
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    y0, w, b = ints((N, 32, D, H, W), -4, 4), ints((32, 32, 3, 3, 3), -2, 2), ints((32,), -3, 3)
    scale = torch.tensor([0.5, 1.0, 2.0, 1.0] * 8).repeat(N, 1)
    shift = ints((N, 32), -2, 2)
    z = y0 * scale[:, :, None, None, None] + shift[:, :, None, None, None]
    a0 = torch.where(z > 0, z, 0.5 * z)                      # multiples of 1/4, |a0| <= 10: exact in bf16
    ref = F.conv3d(a0, w, b, 1, 1).to(BF)
    y, _s, _g, _n = _conv_fused(y0.to(BF).to(DEV).contiguous(memory_format=CL), w.to(DEV), b.to(DEV), scale.to(DEV),
                                shift.to(DEV), 0.5, want_stats=False)
    assert torch.equal(y.cpu(), ref)


def test_fused_block_against_fp64():
    """One whole block boundary at the headline shape [2, 32, 128^3]: raw conv output y0 (with its epilogue statistics) ->
    InstanceNorm + LeakyReLU in the next conv's loader -> y1, against fp64: statistics of the bf16 y0 in fp64, the
    activation in fp64 rounded once to bf16, the conv in fp64 over the bf16 operands (three D-slabs).  Bars: a0 within one
    bf16 ulp of the fp64 value (checked through the apply kernel), y1 within 2^-7 relative + 2e-3 of the tensor scale."""
    from multimodal_mvd_seg_amd import ops
    N, D, H, W = 2, 128, 128, 128
    x, w0, b0, gamma, beta = _rand_block(N, D, H, W, 99)
    g = torch.Generator().manual_seed(3)
    w1 = (torch.randn(32, 32, 3, 3, 3, generator=g) / np.sqrt(27 * 32)).to(DEV)
    b1 = (torch.randn(32, generator=g) * 0.1).to(DEV)
    with torch.no_grad():
        y0 = ops.Conv3dFn.apply(x, None, w0, b0, (1, 1, 1))
        y1 = ops.NormActConv3dFn.apply(y0, gamma, beta, 1e-5, 0.01, w1, b1)
    yd = y0.double()
    m = yd.mean((2, 3, 4), keepdim=True)
    v = yd.var((2, 3, 4), unbiased=False, keepdim=True)
    z = (yd - m) / torch.sqrt(v + 1e-5) * gamma.double()[None, :, None, None, None] + beta.double()[None, :, None, None, None]
    a64 = torch.where(z > 0, z, 0.01 * z).to(BF)            # the activation the fp64 path would store
    wq = w1.to(BF).double()
    for z0 in (0, 61, 120):                                  # output planes z0 .. z0+7: both volume faces and the interior
        lo, hi = max(z0 - 1, 0), min(z0 + 9, D)              # the input planes they read (the slab's cut faces are not used)
        ref = F.conv3d(a64[:, :, lo:hi].double(), wq, b1.double(), 1, 1)[:, :, (z0 - lo):(z0 - lo) + 8]
        got = y1[:, :, z0:z0 + 8].double()
        tol = 2.0 ** -7 * ref.abs() + 2e-3 * float(ref.abs().max())
        err = (got - ref).abs()
        assert bool((err <= tol).all()), f"planes {z0}..{z0 + 7}: max err {float(err.max()):.3e}"
    # and the activation itself: what the loader computes == the apply kernel's output (bit-identical by the test above),
    # which must sit within one bf16 ulp of the fp64 activation
    from multimodal_mvd_seg_amd._lib import call, query
    pre = getattr(y0, "_mvd_tile_stats16", None)
    mean, rstd, scale, shift = (torch.empty((N, 32), dtype=torch.float32, device=DEV) for _ in range(4))
    if pre is not None:
        call("mvd_instnorm_finalize_tiles", _p(pre[0]), pre[1], _p(gamma), _p(beta), _p(mean), _p(rstd), _p(scale), _p(shift),
             N, D * H * W, 32, 1e-5, _stream())
    else:
        wsn = torch.empty(query("mvd_instnorm_workspace_bytes", N, D * H * W, 32), dtype=torch.uint8, device=DEV)
        call("mvd_instnorm_stats_bf16", _p(y0), 1, _p(gamma), _p(beta), _p(mean), _p(rstd), _p(scale), _p(shift), N,
             D * H * W, 32, 1e-5, _p(wsn), wsn.numel(), _stream())
    m64, v64 = m.reshape(N, 32), v.reshape(N, 32)
    # (statistics of the fp32 accumulators vs statistics of the stored bf16 tensor: bf16 rounding noise, see above)
    assert float((mean.double() - m64).abs().max()) < 2e-5 * float(yd.abs().max())
    assert float(((rstd.double() - 1.0 / torch.sqrt(v64 + 1e-5)) * torch.sqrt(v64 + 1e-5)).abs().max()) < 2e-5
    a0 = torch.empty_like(y0)
    call("mvd_instnorm_lrelu_apply_bf16", _p(y0), _p(scale), _p(shift), _p(a0), N, D * H * W, 32, 0.01, _stream())
    zz = torch.where(z > 0, z, 0.01 * z)
    err = (a0.double() - zz).abs()
    assert bool((err <= 1.05 * 2.0 ** -7 * zz.abs() + 1e-4).all()), float(err.max())   # one bf16 ulp + the statistics' noise


def test_norm_act_conv_autograd_node_is_bit_identical_to_the_unfused_chain():
    from multimodal_mvd_seg_amd import ops
    N, D, H, W = 1, 64, 64, 64
    x, w0, b0, gamma, beta = _rand_block(N, D, H, W, 21)
    g = torch.Generator().manual_seed(8)
    w1 = (torch.randn(32, 32, 3, 3, 3, generator=g) / np.sqrt(27 * 32)).to(DEV)
    b1 = (torch.randn(32, generator=g) * 0.1).to(DEV)
    gy = torch.randn(N, 32, D, H, W, generator=g).to(BF).to(DEV).contiguous(memory_format=CL)

    def run(fused):
        ps = [t.clone().requires_grad_() for t in (w0, b0, gamma, beta, w1, b1)]
        xx = x.clone().requires_grad_()
        y0 = ops.Conv3dFn.apply(xx, None, ps[0], ps[1], (1, 1, 1))
        if fused:
            y1 = ops.NormActConv3dFn.apply(y0, ps[2], ps[3], 1e-5, 0.01, ps[4], ps[5])
        else:
            a0 = ops.InstanceNormLeakyReLUFn.apply(y0, ps[2], ps[3], 1e-5, 0.01, True)
            y1 = ops.Conv3dFn.apply(a0, None, ps[4], ps[5], (1, 1, 1))
        y1.backward(gy)
        return [y1.detach()] + [xx.grad] + [p.grad for p in ps]
    a, b = run(True), run(False)
    names = ["y1", "dx", "dw0", "db0", "dgamma", "dbeta", "dw1", "db1"]
    for n, u, v in zip(names, a, b):
        assert torch.equal(u, v), n


def test_network_inference_forward_with_and_without_the_loader_fusion_is_bit_identical_and_train_step_uses_the_epilogue():
    from multimodal_mvd_seg_amd import network, ops, trainer
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    plans = trainer.make_plans((64, 64, 64), strides, batch_size=2)
    ds = {"channel_names": {str(i): f"m{i}" for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=DEV)
    tr.precision = "bf16"
    tr.use_hip_graph = False
    torch.manual_seed(0)
    tr.initialize()
    batch = tr.make_dummy_batch(seed=4)
    with torch.no_grad():
        network.FUSE_PROLOGUE[0] = True
        out_f = tr.network(batch["data"])
        network.FUSE_PROLOGUE[0] = False
        out_u = tr.network(batch["data"])
        network.FUSE_PROLOGUE[0] = True
    for u, v in zip(out_f, out_u):
        assert torch.equal(u, v)
    # training: the statistics epilogue against the stand-alone statistics pass (same step, epilogue off): the two
    # reduction orders differ in the last fp32 bits of mean / rstd, so the losses agree to 1e-5, not bit for bit
    sd = {k: v.clone() for k, v in tr.network.state_dict().items()}
    l_on = float(tr.train_step(batch)["loss"])
    tr2 = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=DEV)
    tr2.precision = "bf16"
    tr2.use_hip_graph = False
    tr2.initialize()
    tr2.network.load_state_dict(sd)
    tr2.optimizer.fp.invalidate_packs()
    ops.BF16_CONV_STATS[0] = False
    try:
        l_off = float(tr2.train_step(batch)["loss"])
    finally:
        ops.BF16_CONV_STATS[0] = True
    assert abs(l_on - l_off) <= 2e-4 * abs(l_off), (l_on, l_off)
    worst = max(float((p.detach() - q.detach()).abs().max()) for p, q in
                zip(tr.network.parameters(), tr2.network.parameters()))
    assert worst <= 5e-4, worst


# ------------------------------------------------------------------------------------------------ k_fwd16y (16x16x32 tiles)
def _set_kernel(which):
    from multimodal_mvd_seg_amd._lib import call
    call("mvd_set_bf16_zmarch_kernel", which)


@pytest.mark.parametrize("C1,C2,K,N,D,H,W", [(32, 0, 32, 2, 52, 60, 44), (32, 0, 32, 1, 33, 70, 97), (32, 0, 32, 1, 128, 64, 64),
                                              (32, 32, 32, 1, 40, 44, 70), (32, 32, 32, 2, 64, 64, 64), (64, 0, 32, 1, 36, 72, 66),
                                              (32, 32, 32, 1, 17, 41, 130),
                                              # 64 produce channels (one row group x four channel quarters): the 64^3 stage
                                              (64, 0, 64, 2, 64, 64, 64), (64, 0, 64, 1, 40, 44, 70), (32, 0, 64, 1, 36, 72, 66),
                                              (64, 64, 64, 1, 64, 64, 64), (32, 32, 64, 1, 33, 70, 97)])
def test_fwd16y_exact_integer_data_forward_and_input_gradient(C1, C2, K, N, D, H, W):
    """k_fwd16y on small-integer data: every product and fp32 partial sum is an exact integer, so forward and input gradient
    must equal torch's exact fp32 convolution rounded once to bf16, bit for bit -- 32 reduce channels, 32 + 32 through two
    pointers (the eliminated torch.cat, UNetDecoder.py:107; its gradient = two launches into two tensors), 64 in one tensor;
    ragged volumes (zero-fill through the descriptor range check, dropped out-of-volume stores, the lane-row exchange of
    the store images, the accumulator ring).  The same data through k_fwd16z / the generic kernel must agree too."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(D * 7 + H + C2)
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    C = C1 + C2
    x, w, b = ints((N, C, D, H, W), -2, 2), ints((K, C, 3, 3, 3), -2, 2), ints((K,), -3, 3)
    xr = x.clone().requires_grad_()
    ref = F.conv3d(xr, w, b, 1, 1)
    gy = ints(tuple(ref.shape), -1, 1)
    ref.backward(gy)
    outs = []
    for which in (1, 0):
        _set_kernel(which)
        try:
            g1 = x[:, :C1].to(DEV).to(BF).contiguous(memory_format=CL).requires_grad_()
            g2 = x[:, C1:].to(DEV).to(BF).contiguous(memory_format=CL).requires_grad_() if C2 else None
            gw, gb = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
            y = ops.Conv3dFn.apply(g1, g2, gw, gb, (1, 1, 1))
            y.backward(gy.to(DEV).to(BF).contiguous(memory_format=CL))
        finally:
            _set_kernel(1)
        assert torch.equal(y.detach().cpu(), ref.detach().to(BF)), f"y (kernel {which})"
        assert torch.equal(g1.grad.cpu(), xr.grad[:, :C1].to(BF)), f"dx1 (kernel {which})"
        if C2:
            assert torch.equal(g2.grad.cpu(), xr.grad[:, C1:].to(BF)), f"dx2 (kernel {which})"
        outs.append(y.detach())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("C1,C2", [(32, 0), (32, 32)])
def test_fwd16y_random_data_matches_fp64_and_the_other_kernel(C1, C2):
    N, D, H, W = 1, 64, 64, 64
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(C2 + 3)
    C = C1 + C2
    x = torch.randn(N, C, D, H, W, generator=g).to(BF)
    w = torch.randn(32, C, 3, 3, 3, generator=g) / np.sqrt(27 * C)
    b = torch.randn(32, generator=g) * 0.1
    ref = F.conv3d(x.double(), w.to(BF).double(), b.double(), 1, 1)
    ys = []
    for which in (1, 0):
        _set_kernel(which)
        try:
            with torch.no_grad():
                y = ops.Conv3dFn.apply(x[:, :C1].to(DEV).contiguous(memory_format=CL),
                                       x[:, C1:].to(DEV).contiguous(memory_format=CL) if C2 else None, w.to(DEV), b.to(DEV),
                                       (1, 1, 1))
        finally:
            _set_kernel(1)
        err = (y.cpu().double() - ref).abs()
        tol = 2.0 ** -7 * ref.abs() + 2e-3 * float(ref.abs().max())
        assert bool((err <= tol).all()), f"kernel {which}: max err {float(err.max()):.3e}"
        ys.append(y)
    # the two kernels accumulate the 27 x C products in different orders: equal up to one bf16 rounding
    d = (ys[0].float() - ys[1].float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(ys[0].float().abs().max())


@pytest.mark.parametrize("C,K,N,D,H,W", [(32, 64, 2, 40, 44, 36), (64, 128, 1, 33, 41, 38), (32, 32, 1, 17, 130, 23),
                                          (128, 256, 2, 16, 16, 16)])
def test_fused_stride2_input_gradient_bf16_exact_integer_data_and_accumulate_form(C, K, N, D, H, W):
    """k_dgrad16s: the input gradient of a 3x3x3 stride-2 conv (all eight output-parity classes from one staged dy tile) on
    small-integer data == torch's exact result rounded once to bf16, bit for bit, on even and odd extents; the accumulating
    form (ops._GradShare: dx += ...) == old + gradient, exactly (integers)."""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call, i3, query
    g = torch.Generator().manual_seed(C + D)
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    x, w = ints((N, C, D, H, W), -2, 2), ints((K, C, 3, 3, 3), -2, 2)
    xr = x.clone().requires_grad_()
    ref = F.conv3d(xr, w, None, 2, 1)
    gy = ints(tuple(ref.shape), -1, 1)
    ref.backward(gy)
    gx = x.to(DEV).to(BF).contiguous(memory_format=CL).requires_grad_()
    gw = w.to(DEV).requires_grad_()
    y = ops.Conv3dFn.apply(gx, None, gw, None, (2, 2, 2))
    assert torch.equal(y.detach().cpu(), ref.detach().to(BF))
    dyd = gy.to(DEV).to(BF).contiguous(memory_format=CL)
    y.backward(dyd)
    assert torch.equal(gx.grad.cpu(), xr.grad.to(BF)), "dx"
    assert torch.equal(gw.grad.cpu(), w.grad if False else torch.autograd.grad(F.conv3d(x, w.requires_grad_(), None, 2, 1), w, gy)[0]), "dw"
    # accumulate form through the C ABI
    assert query("mvd_conv3d_dgrad_acc_ok", 1, N, D, H, W, C, K, i3((3, 3, 3)), i3((2, 2, 2))) == 1
    old = ints((N, C, D, H, W), -3, 3)
    buf = old.to(DEV).to(BF).contiguous(memory_format=CL)
    _wf, wb = ops.pack_weight_bf16(w.to(DEV), False)
    ws = torch.empty(max(1024, query("mvd_conv_fwd_workspace_bytes", N, D * H * W, C)), dtype=torch.uint8, device=DEV)
    call("mvd_conv3d_dgrad_bf16_acc", _p(dyd), _p(wb), _p(buf), C, N, D, H, W, K, i3((3, 3, 3)), i3((2, 2, 2)), _p(ws), ws.numel(),
         _stream())
    assert torch.equal(buf.cpu(), (old + xr.grad).to(BF))


def _set_wgrad_kernel(which):
    from multimodal_mvd_seg_amd._lib import call
    call("mvd_set_bf16_wgrad_kernel", which)


@pytest.mark.parametrize("C1,C2,K,N,D,H,W", [(32, 0, 32, 2, 52, 60, 44), (32, 0, 32, 1, 33, 70, 97), (32, 0, 32, 1, 128, 64, 64),
                                              (32, 32, 32, 1, 40, 44, 70), (32, 32, 32, 2, 64, 64, 64), (64, 0, 64, 2, 64, 64, 64),
                                              (64, 64, 64, 1, 40, 41, 66), (128, 0, 128, 2, 32, 32, 32), (128, 128, 128, 1, 32, 32, 32),
                                              (32, 0, 64, 1, 9, 8, 32), (32, 0, 32, 1, 8, 13, 33)])
def test_wgrad16z_exact_integer_data_weight_and_bias_gradient(C1, C2, K, N, D, H, W):
    """k_wgrad16z (z-marching bf16 weight gradient of the plain 3x3x3 stride-1 conv) on small-integer data: every product and
    fp32 partial sum is an exact integer, so dw and db must equal torch's exact fp32 result bit for bit -- one and two
    source tensors (UNetDecoder.py:107), 32 ... 128 channel blocks, ragged volumes (rows / columns / planes outside the
    volume come back as zeros from the buffer descriptors; chunked z ranges with their halo planes), several columns per
    workgroup.  The tiled kernel k_wgrad16 on the same data must agree."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(D * 5 + W + C2)
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    C = C1 + C2
    x, w, b = ints((N, C, D, H, W), -2, 2), ints((K, C, 3, 3, 3), -1, 1), ints((K,), -1, 1)
    gy = ints((N, K, D, H, W), -1, 1)
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    F.conv3d(x, wr, br, 1, 1).backward(gy)
    res = []
    for which in (1, 0):
        _set_wgrad_kernel(which)
        try:
            g1 = x[:, :C1].to(DEV).to(BF).contiguous(memory_format=CL).requires_grad_()
            g2 = x[:, C1:].to(DEV).to(BF).contiguous(memory_format=CL).requires_grad_() if C2 else None
            gw, gb = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
            y = ops.Conv3dFn.apply(g1, g2, gw, gb, (1, 1, 1))
            y.backward(gy.to(DEV).to(BF).contiguous(memory_format=CL))
        finally:
            _set_wgrad_kernel(1)
        assert torch.equal(gw.grad.cpu(), wr.grad), f"dw (kernel {which}): {int((gw.grad.cpu() != wr.grad).sum())} of {wr.grad.numel()} differ"
        assert torch.equal(gb.grad.cpu(), br.grad), f"db (kernel {which})"
        res.append(gw.grad)
    assert torch.equal(res[0], res[1])


def test_wgrad16z_random_data_against_fp64():
    from multimodal_mvd_seg_amd import ops
    N, C, K, D, H, W = 1, 32, 32, 48, 40, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, C, D, H, W, generator=g).to(BF)
    w = torch.randn(K, C, 3, 3, 3, generator=g) / np.sqrt(27 * C)
    gy = torch.randn(N, K, D, H, W, generator=g).to(BF)
    wr = w.double().requires_grad_()
    F.conv3d(x.double(), wr, None, 1, 1).backward(gy.double())
    gx = x.to(DEV).contiguous(memory_format=CL).requires_grad_()
    gw = w.to(DEV).requires_grad_()
    ops.Conv3dFn.apply(gx, None, gw, None, (1, 1, 1)).backward(gy.to(DEV).contiguous(memory_format=CL))
    err = (gw.grad.cpu().double() - wr.grad).abs().max()
    assert float(err) <= 1e-5 * float(wr.grad.abs().max()) + 1e-3, float(err)  # fp32 accumulation of exact bf16 products


@pytest.mark.parametrize("C,K,N,D,H,W", [(32, 32, 2, 52, 60, 44), (32, 32, 1, 33, 70, 97), (64, 64, 2, 40, 41, 66), (32, 64, 1, 9, 8, 32),
                                          (32, 32, 2, 128, 64, 64)])
def test_wgrad_loader_prologue_equals_the_gradient_over_the_materialised_activation_bit_for_bit(C, K, N, D, H, W):
    """mvd_conv3d_wgrad_bf16_fused (k_wgrad16z with the InstanceNorm-apply + LeakyReLU prologue on the staged x planes) ==
    mvd_conv3d_wgrad_bf16 over the tensor mvd_instnorm_lrelu_apply_bf16 writes: same activations bit for bit (one
    expression), same MFMA sequence -> torch.equal on dw and db; per-sample scale / shift, ragged volumes (voxels outside
    the volume must stay ZERO, not lrelu(shift)), chunked z ranges, 64 channels in two blocks."""
    from multimodal_mvd_seg_amd._lib import call, i3, query
    g = torch.Generator().manual_seed(C + D + W)
    y0 = torch.randn(N, C, D, H, W, generator=g).to(BF).to(DEV).contiguous(memory_format=CL)
    dy = torch.randn(N, K, D, H, W, generator=g).to(BF).to(DEV).contiguous(memory_format=CL)
    scale = (torch.rand(N, C, generator=g) + 0.5).to(DEV)
    shift = (torch.randn(N, C, generator=g) * 0.5).to(DEV)   # non-zero shift: lrelu(shift) != 0 would leak through a padding bug
    ks, st = i3((3, 3, 3)), i3((1, 1, 1))
    assert query("mvd_conv3d_wgrad_bf16_prologue_ok", N, D, H, W, C, 0, K, ks, st) == 1
    a0 = torch.empty_like(y0)
    call("mvd_instnorm_lrelu_apply_bf16", _p(y0), _p(scale), _p(shift), _p(a0), N, D * H * W, C, 0.01, _stream())
    ws = torch.empty(query("mvd_conv3d_wgrad_workspace_bytes", C, K, 27, N, D, H, W), dtype=torch.uint8, device=DEV)
    dw1, db1 = torch.empty(K, C, 3, 3, 3, device=DEV), torch.empty(K, device=DEV)
    dw2, db2 = torch.empty_like(dw1), torch.empty_like(db1)
    call("mvd_conv3d_wgrad_bf16", _p(a0), C, None, 0, _p(dy), _p(dw1), _p(db1), N, D, H, W, K, ks, st, _p(ws), ws.numel(), _stream())
    call("mvd_conv3d_wgrad_bf16_fused", _p(y0), C, _p(dy), _p(dw2), _p(db2), N, D, H, W, K, ks, st, _p(scale), _p(shift), 0.01,
         _p(ws), ws.numel(), _stream())
    torch.cuda.synchronize()
    assert torch.equal(dw1, dw2), f"{int((dw1 != dw2).sum())} of {dw1.numel()} differ, max {float((dw1 - dw2).abs().max())}"
    assert torch.equal(db1, db2)
    # and against fp64 over the same activations
    ref = torch.autograd.grad(F.conv3d(a0.double().cpu(), (w := torch.zeros(K, C, 3, 3, 3, dtype=torch.float64, requires_grad=True)),
                                       None, 1, 1), w, dy.double().cpu())[0] if D * H * W * N <= 200000 else None
    if ref is not None:
        assert float((dw2.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-3
    # shapes the kernel does not serve are refused, not mis-computed
    assert query("mvd_conv3d_wgrad_bf16_prologue_ok", N, D, H, 16, C, 0, K, ks, st) == 0
    assert query("mvd_conv3d_wgrad_bf16_prologue_ok", N, D, H, W, C, 32, K, ks, st) == 0


def test_train_step_with_the_fused_block_in_training_is_bit_identical_to_the_unfused_step():
    """bf16 train steps with the block boundary fused in TRAINING (forward: IN-apply + LeakyReLU in the next conv's loader;
    backward: the same prologue in the weight-gradient kernel; the activated tensor is never written) against the same steps
    with MVD_FUSE_PROLOGUE_TRAIN=0: every value that flows is the same, so loss and parameters agree bit for bit."""
    from multimodal_mvd_seg_amd import network, trainer
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    plans = trainer.make_plans((64, 64, 64), strides, batch_size=2)
    ds = {"channel_names": {str(i): f"m{i}" for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    res, counts = [], {}
    saved = network.FUSE_PROLOGUE_TRAIN[0]
    try:
        for mode in ("auto", "0"):
            network.FUSE_PROLOGUE_TRAIN[0] = mode
            tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=DEV)
            tr.precision = "bf16"
            tr.use_hip_graph = False
            torch.manual_seed(0)
            tr.initialize()
            n_fused = [0]
            orig_call = network.ops.call

            def counting_call(name, *a, **k):
                n_fused[0] += name == "mvd_conv3d_wgrad_bf16_fused"
                return orig_call(name, *a, **k)
            network.ops.call = counting_call
            try:
                losses = [float(tr.train_step(tr.make_dummy_batch(seed=4 + i))["loss"]) for i in range(2)]
            finally:
                network.ops.call = orig_call
            counts[mode] = n_fused[0]
            res.append((losses, [p.detach().clone() for p in tr.network.parameters()]))
    finally:
        network.FUSE_PROLOGUE_TRAIN[0] = saved
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for p, q in zip(res[0][1], res[1][1]):
        assert torch.equal(p, q)
    assert counts["auto"] >= 2 and counts["0"] == 0, counts   # the fused weight gradient really ran (once per step) in "auto"


@pytest.mark.parametrize("K,N,D,H,W", [(64, 2, 64, 64, 64), (64, 2, 52, 60, 44), (32, 1, 65, 70, 97)])
def test_loader_prologue_of_a_64_channel_producer_bit_for_bit(K, N, D, H, W):
    """The 64^3 stage (enc1 / dec4 of configs[1]): the producing conv has 64 channels, i.e. TWO 32-channel chunks in the
    consumer's loader, each with its own scale / shift octets.  conv(lrelu(IN(y0))) through the apply pass + plain conv ==
    the fused conv reading y0, bit for bit (ragged volumes: padding is zero in the ACTIVATION)."""
    from multimodal_mvd_seg_amd._lib import call, i3, query
    C = 64
    g = torch.Generator().manual_seed(3 * D + K)
    y0 = (torch.randn(N, C, D, H, W, generator=g) * 1.5 + 0.3).to(BF).to(DEV).contiguous(memory_format=CL)
    w = (torch.randn(K, C, 3, 3, 3, generator=g) / np.sqrt(27 * C)).to(DEV)
    b = (torch.randn(K, generator=g) * 0.1).to(DEV)
    scale = (torch.rand(N, C, generator=g) + 0.3).to(DEV)
    shift = (torch.randn(N, C, generator=g) * 0.5 + 0.4).to(DEV)
    assert query("mvd_conv3d_fwd_bf16_prologue_ok", N, D, H, W, C, 0, K, i3((3, 3, 3)), i3((1, 1, 1))) == 1
    a0 = torch.empty_like(y0)
    call("mvd_instnorm_lrelu_apply_bf16", _p(y0), _p(scale), _p(shift), _p(a0), N, D * H * W, C, 0.01, _stream())
    y_ref, _s, _g, _n = _conv_fused(a0, w, b, want_stats=False)
    y_fused, stats, got, nt = _conv_fused(y0, w, b, scale, shift, 0.01, want_stats=True)
    assert torch.equal(y_fused, y_ref), f"{int((y_fused != y_ref).sum())} differ"
    if nt > 0:
        assert got == nt
        yd = y_fused.permute(0, 2, 3, 4, 1).reshape(N, -1, K).double()
        tol_s, tol_q = _sum_tolerances(yd)
        assert bool(((stats.double().sum(1)[..., 0] - yd.sum(1)).abs() <= tol_s).all())
        assert bool(((stats.double().sum(1)[..., 1] - (yd * yd).sum(1)).abs() <= tol_q).all())
    # two producer TENSORS (the decoder's concatenation) have no single InstanceNorm in front: refused
    assert query("mvd_conv3d_fwd_bf16_prologue_ok", N, D, H, W, 32, 32, K, i3((3, 3, 3)), i3((1, 1, 1))) == 0


@pytest.mark.parametrize("N,D,H,W", [(2, 128, 128, 128), (1, 131, 127, 129), (1, 96, 200, 64)])
def test_fwd16ys_stride2_forward_exact_integer_data(N, D, H, W):
    """k_fwd16ys (z-marching stride-2 conv 32 -> 64, parity-split LDS image) on small-integer data: every product and fp32
    partial sum is exact, so the output must equal torch's exact fp32 convolution rounded once to bf16, bit for bit -- even
    and odd input extents (the last input plane / row / column may or may not exist), several z chunks per column; the generic
    kernel on the same data must agree (mvd_set_bf16_zmarch_kernel(0) switches the z-marching kernels off)."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(D + 3 * W)
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    x, w, b = ints((N, 32, D, H, W), -2, 2), ints((64, 32, 3, 3, 3), -2, 2), ints((64,), -3, 3)
    ref = F.conv3d(x, w, b, 2, 1).to(BF)
    outs = []
    for which in (1, 0):
        _set_kernel(which)
        try:
            with torch.no_grad():
                y = ops.Conv3dFn.apply(x.to(DEV).to(BF).contiguous(memory_format=CL), None, w.to(DEV), b.to(DEV), (2, 2, 2))
        finally:
            _set_kernel(1)
        assert torch.equal(y.cpu(), ref), f"kernel {which}: {int((y.cpu() != ref).sum())} of {ref.numel()} differ"
        outs.append(y)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("C,K,N,D,H,W", [(32, 64, 2, 128, 128, 128), (32, 64, 1, 131, 127, 129), (64, 128, 2, 64, 64, 64),
                                          (32, 64, 1, 16, 24, 70)])
def test_wgrad16zs_stride2_weight_gradient_exact_integer_data(C, K, N, D, H, W):
    """k_wgrad16zs (z-marching weight gradient of the 3x3x3 stride-2 convs: ring of three parity-split x planes replaced in
    place, two dy channel blocks per workgroup) on small-integer data: dw (and db, from the column-sum pass) must equal
    torch's exact fp32 result bit for bit -- even and odd input extents, one and two reduce-channel blocks, chunked z ranges;
    the tiled kernel on the same data must agree."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(D + 3 * W + C)
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    x, w, b = ints((N, C, D, H, W), -2, 2), ints((K, C, 3, 3, 3), -1, 1), ints((K,), -1, 1)
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    y_ref = F.conv3d(x, wr, br, 2, 1)
    gy = ints(tuple(y_ref.shape), -1, 1)
    y_ref.backward(gy)
    res = []
    for which in (1, 0):
        _set_wgrad_kernel(which)
        try:
            gx = x.to(DEV).to(BF).contiguous(memory_format=CL)
            gw, gb = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
            y = ops.Conv3dFn.apply(gx, None, gw, gb, (2, 2, 2))
            y.backward(gy.to(DEV).to(BF).contiguous(memory_format=CL))
        finally:
            _set_wgrad_kernel(1)
        assert torch.equal(gw.grad.cpu(), wr.grad), f"dw (kernel {which}): {int((gw.grad.cpu() != wr.grad).sum())} of {wr.grad.numel()} differ"
        assert torch.equal(gb.grad.cpu(), br.grad), f"db (kernel {which})"
        res.append(gw.grad)
    assert torch.equal(res[0], res[1])


def test_round3_kernels_are_run_to_run_deterministic():
    """The z-marching kernels carry hand-placed schedules (inline-asm MFMAs, counted waits, pinned accumulator resets): a missing
    wait state shows as SPORADICALLY different results, not as a wrong test case (that is how the stale-accumulator bug of
    k_fwd16y was found).  Twelve repetitions of each kernel on the same inputs must be bit-identical."""
    from multimodal_mvd_seg_amd import ops
    from multimodal_mvd_seg_amd._lib import call, i3, query
    g = torch.Generator().manual_seed(77)
    rnd = lambda *s: torch.randn(*s, generator=g)
    reps = 12

    def same(fn, what):
        first = fn()
        for k in range(reps - 1):
            out = fn()
            for a, b in zip(first, out):
                assert torch.equal(a, b), f"{what}: repetition {k + 1} differs ({int((a != b).sum())} values)"

    N, D, H, W = 2, 64, 64, 64
    for C, K, stride in ((32, 32, 1), (64, 64, 1), (32, 64, 2)):   # k_fwd16y (one / two chunks), k_fwd16ys; their weight gradients
        S = 2 * D if stride == 2 else D
        x = rnd(N, C, S, S, S).to(BF).to(DEV).contiguous(memory_format=CL)
        w = (rnd(K, C, 3, 3, 3) / np.sqrt(27 * C)).to(DEV)
        b = (rnd(K) * 0.1).to(DEV)
        gy = rnd(N, K, D, D, D).to(BF).to(DEV).contiguous(memory_format=CL)

        def step():
            xx = x.clone().requires_grad_(stride == 1)
            ww, bb = w.clone().requires_grad_(), b.clone().requires_grad_()
            y = ops.Conv3dFn.apply(xx, None, ww, bb, (stride,) * 3)
            y.backward(gy)
            return [y.detach(), ww.grad, bb.grad] + ([xx.grad] if stride == 1 else [])
        same(step, f"conv {C}->{K} stride {stride}")
    # the fused node (forward loader prologue + statistics epilogue, weight-gradient loader prologue), 32 and 64 channels
    for C in (32, 64):
        y0 = (rnd(N, C, D, D, D) * 1.5 + 0.3).to(BF).to(DEV).contiguous(memory_format=CL)
        w = (rnd(C, C, 3, 3, 3) / np.sqrt(27 * C)).to(DEV)
        b = (rnd(C) * 0.1).to(DEV)
        gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (rnd(C) * 0.2).to(DEV)
        gy = rnd(N, C, D, D, D).to(BF).to(DEV).contiguous(memory_format=CL)

        def fused():
            yy = y0.clone().requires_grad_()
            ps = [t.clone().requires_grad_() for t in (gamma, beta, w, b)]
            y1 = ops.NormActConv3dFn.apply(yy, ps[0], ps[1], 1e-5, 0.01, ps[2], ps[3])
            y1.backward(gy)
            return [y1.detach(), yy.grad] + [p.grad for p in ps]
        same(fused, f"fused block {C} channels")


@pytest.mark.parametrize("dt", ["bf16", "fp32"])
@pytest.mark.parametrize("N,D,H,W", [(2, 64, 64, 64), (1, 33, 36, 44)])
def test_norm_act_seghead_node_is_bit_identical_to_the_two_separate_nodes(N, D, H, W, dt):
    """ops.NormActSegHeadFn (the last decoder block's InstanceNorm + LeakyReLU inside the seg head's loaders, activated tensor
    never written) against InstanceNormLeakyReLUFn -> SegHeadFn on the same raw conv output: logits, d y0, d gamma, d beta,
    dW, db bit for bit."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(D + W)
    y0 = (torch.randn(N, 32, D, H, W, generator=g) * 1.5 + 0.3).to(BF if dt == "bf16" else torch.float32).to(DEV) \
        .contiguous(memory_format=CL)
    gamma, beta = (torch.rand(32, generator=g) + 0.5).to(DEV), (torch.randn(32, generator=g) * 0.2).to(DEV)
    w = (torch.randn(5, 32, 1, 1, 1, generator=g) * 0.2).to(DEV)
    b = (torch.randn(5, generator=g) * 0.1).to(DEV)
    gl = torch.randn(N, 5, D, H, W, generator=g).to(DEV)

    def run(fused):
        ps = [t.clone().requires_grad_() for t in (gamma, beta, w, b)]
        yy = y0.clone().requires_grad_()
        if fused:
            lg = ops.NormActSegHeadFn.apply(yy, ps[0], ps[1], 1e-5, 0.01, ps[2], ps[3])
        else:
            lg = ops.SegHeadFn.apply(ops.InstanceNormLeakyReLUFn.apply(yy, ps[0], ps[1], 1e-5, 0.01, dt == "bf16"), ps[2], ps[3])
        lg.backward(gl)
        return [lg.detach(), yy.grad] + [p.grad for p in ps]
    assert ops.fused_norm_seghead_ok(y0, w)
    for n, u, v in zip(["logits", "dy0", "dgamma", "dbeta", "dW", "db"], run(True), run(False)):
        assert torch.equal(u, v), n


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_train_step_with_the_seg_head_fusion_is_bit_identical_and_it_runs(precision):
    from multimodal_mvd_seg_amd import network, trainer
    strides = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
    plans = trainer.make_plans((64, 64, 64), strides, batch_size=2)
    ds = {"channel_names": {str(i): f"m{i}" for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    res, counts = [], {}
    saved = network.FUSE_SEGHEAD[0]
    try:
        for mode in (True, False):
            network.FUSE_SEGHEAD[0] = mode
            tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=DEV)
            tr.precision = precision
            tr.use_hip_graph = False
            torch.manual_seed(0)
            tr.initialize()
            n_fused = [0]
            orig_call = network.ops.call

            def counting_call(name, *a, **k):
                n_fused[0] += name in ("mvd_seghead_fwd_bf16_fused", "mvd_seghead_fwd_fused")
                return orig_call(name, *a, **k)
            network.ops.call = counting_call
            try:
                losses = [float(tr.train_step(tr.make_dummy_batch(seed=9 + i))["loss"]) for i in range(2)]
            finally:
                network.ops.call = orig_call
            counts[mode] = n_fused[0]
            res.append((losses, [p.detach().clone() for p in tr.network.parameters()]))
    finally:
        network.FUSE_SEGHEAD[0] = saved
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for p, q in zip(res[0][1], res[1][1]):
        assert torch.equal(p, q)
    assert counts[True] == 2 and counts[False] == 0, counts
