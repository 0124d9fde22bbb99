"""GPU tests of the bf16 mixed-precision engine (BASELINE cfg 4/5).  Reference: fp64 evaluation on the SAME
bf16-rounded inputs and weights (products of bf16 numbers are exact in fp32, accumulation is fp32), rounded to bf16 --
so the only legitimate difference is the accumulation order and the final rounding: 1 bf16 ulp (2^-8 relative)."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def bf(t):
    return t.to(torch.bfloat16)


def close_bf16(a, ref, what):
    a, ref = a.float().cpu().double(), ref.double()
    tol = 2.0 ** -7 * ref.abs() + 2e-3 * float(ref.abs().max())
    err = (a - ref).abs()
    assert bool((err <= tol).all()), f"{what}: max err {float(err.max()):.3e} at |ref| {float(ref.abs().max()):.3e}"


@pytest.mark.parametrize("C1,C2,K,sp,stride,N", [
    (32, 0, 32, (12, 10, 14), 1, 2),
    (32, 32, 32, (9, 10, 11), 1, 1),
    (64, 0, 64, (8, 8, 8), 1, 2),
    (32, 0, 64, (16, 12, 20), 2, 1),
    (320, 320, 320, (4, 4, 4), 1, 2),
    (128, 0, 128, (6, 7, 8), (1, 2, 2), 1),
    (32, 0, 32, (38, 70, 60), 1, 2),     # >= 4 tiles per CU: the persistent weights-resident kernel, ragged tiles
])
def test_conv3d_bf16_fwd_dgrad(C1, C2, K, sp, stride, N):
    from multimodal_mvd_seg_amd._lib import call, i3, query
    st = (stride,) * 3 if isinstance(stride, int) else stride
    g = torch.Generator().manual_seed(C1 + K)
    x1 = bf(torch.randn(N, C1, *sp, generator=g))
    x2 = bf(torch.randn(N, C2, *sp, generator=g)) if C2 else None
    w = torch.randn(K, C1 + C2, 3, 3, 3, generator=g) / np.sqrt(27 * (C1 + C2))
    b = torch.randn(K, generator=g) * 0.1
    wq = bf(w).double()
    xs = [t.double().requires_grad_() for t in ([x1, x2] if C2 else [x1])]
    ref = F.conv3d(torch.cat(xs, 1), wq, b.double(), st, 1)
    gy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(gy.double())
    cl = torch.channels_last_3d
    d1 = x1.to(DEV).contiguous(memory_format=cl)
    d2 = x2.to(DEV).contiguous(memory_format=cl) if C2 else None
    T = 27
    wd = w.to(DEV).contiguous()
    wf = torch.empty(T * (C1 + C2) * K, dtype=torch.bfloat16, device=DEV)
    wb = torch.empty(T * (C1 + C2) * K, dtype=torch.bfloat16, device=DEV)
    call("mvd_pack_weight_bf16", _p(wd), _p(wf), _p(wb), K, C1 + C2, T, 0, _stream())
    od = tuple(ref.shape[2:])
    y = torch.empty((N, K, *od), dtype=torch.bfloat16, device=DEV).contiguous(memory_format=cl)
    ws = torch.empty(max(1024, query("mvd_conv_fwd_workspace_bytes", N, sp[0] * sp[1] * sp[2], max(K, C1 + C2))),
                     dtype=torch.uint8, device=DEV)
    D, H, W = sp
    bd = b.to(DEV)
    call("mvd_conv3d_fwd_bf16", _p(d1), C1, _p(d2), C2, _p(wf), _p(bd), _p(y), N, D, H, W, K, i3((3, 3, 3)), i3(st),
         _p(ws), ws.numel(), _stream())
    close_bf16(y, ref.detach(), "y")
    gyd = gy.to(DEV).contiguous(memory_format=cl)
    dx1 = torch.empty_like(d1)
    dx2 = torch.empty_like(d2) if C2 else None
    call("mvd_conv3d_dgrad_bf16", _p(gyd), _p(wb), _p(dx1), C1, _p(dx2), C2, N, D, H, W, K, i3((3, 3, 3)), i3(st), _p(ws),
         ws.numel(), _stream())
    close_bf16(dx1, xs[0].grad, "dx1")
    if C2:
        close_bf16(dx2, xs[1].grad, "dx2")


@pytest.mark.parametrize("N,D,H,W", [(2, 52, 60, 44), (2, 33, 70, 97), (3, 17, 41, 130), (1, 128, 64, 64)])
def test_conv3d_bf16_zmarch_ragged_shapes_exact(N, D, H, W):
    """k_fwd16z (32 -> 32 channels, the z-marching kernel of the headline bf16 block) on volumes whose extents are not
    multiples of its 8 x 32 column, with z chunks that end inside and at the volume faces, forward and input gradient,
    on small-integer data: every product and fp32 partial sum is an exact integer, so the result must equal torch's
    exact fp32 convolution rounded once to bf16, bit for bit (halo zero-fill through the descriptor range check, dropped
    out-of-volume stores, the accumulator ring and the bias re-initialisation all show up as wrong integers)."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(D * 7 + H)
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    x, w, b = ints((N, 32, D, H, W), -2, 2), ints((32, 32, 3, 3, 3), -2, 2), ints((32,), -3, 3)
    xr = x.clone().requires_grad_()
    ref = F.conv3d(xr, w, b, 1, 1)
    gy = ints(tuple(ref.shape), -1, 1)
    ref.backward(gy)
    cl, BF = torch.channels_last_3d, torch.bfloat16
    gx = x.to(DEV).to(BF).contiguous(memory_format=cl).requires_grad_()
    gw, gb = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.Conv3dFn.apply(gx, None, gw, gb, (1, 1, 1))
    y.backward(gy.to(DEV).to(BF).contiguous(memory_format=cl))
    assert torch.equal(y.detach().cpu(), ref.detach().to(BF)), "y"
    assert torch.equal(gx.grad.cpu(), xr.grad.to(BF)), "dx"



@pytest.mark.parametrize("N,C,sp,planar", [(2, 4, (12, 10, 14), True), (1, 4, (33, 70, 97), True), (2, 1, (9, 8, 40), False)])
def test_narrow_input_conv_bf16_exact_integer_data(N, C, sp, planar):
    """The 4-modality input conv under bf16 mixed precision (ops.NarrowInputConv3dBf16Fn: channels zero-padded to 32, bf16
    MFMA engines) on small-integer data: operands exact in bf16, every fp32 partial sum an exact integer -> y must be
    torch's exact fp32 conv rounded once to bf16, the weight / bias gradients (fp32) exact, bit for bit.  Covers the pad
    kernel for planar and NDHWC sources, the padded weight pack, the dropped gradient of the zero channels, and (second
    case) the z-marching kernel fed by the padded tensor."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(11 * C + sp[0])
    ints = lambda shape, lo, hi: torch.randint(lo, hi + 1, shape, generator=g).float()
    x, w, b = ints((N, C, *sp), -2, 2), ints((32, C, 3, 3, 3), -2, 2), ints((32,), -3, 3)
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv3d(x, wr, br, 1, 1)
    gy = ints(tuple(ref.shape), -1, 1)
    ref.backward(gy)
    gx = x.to(DEV) if planar else x.to(DEV).contiguous(memory_format=torch.channels_last_3d)
    gw, gb = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.NarrowInputConv3dBf16Fn.apply(gx, gw, gb)
    assert y.dtype == torch.bfloat16
    y.backward(gy.to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d))
    assert torch.equal(y.detach().cpu(), ref.detach().to(torch.bfloat16)), "y"
    assert gw.grad.dtype == torch.float32 and tuple(gw.grad.shape) == tuple(w.shape)
    assert torch.equal(gw.grad.cpu(), wr.grad), f"dw (max {float((gw.grad.cpu() - wr.grad).abs().max())})"
    assert torch.equal(gb.grad.cpu(), br.grad), "db"


def test_bf16_packs_follow_the_fused_optimizer_in_one_launch():
    """After FusedSGDNesterov.step() (raw-pointer update of the flat buffer) every registered bf16 pack is rebuilt by ONE
    batched launch (ops.repack_all -> mvd_pack_weights_bf16_batch) into fresh tensors: the cached entry must carry the
    new stamp without a per-layer pack, equal the per-layer pack of the updated weight bit for bit (conv and transposed
    conv layouts), and the tensors a graph saved before the step must be untouched."""
    from multimodal_mvd_seg_amd import ops, optim
    g = torch.Generator().manual_seed(5)
    conv = torch.nn.Conv3d(32, 64, 3, padding=1).to(DEV)
    convT = torch.nn.ConvTranspose3d(64, 32, 2, stride=2).to(DEV)
    x = torch.randn(1, 32, 6, 6, 8, generator=g).to(DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last_3d)
    opt = optim.FusedSGDNesterov(list(conv.parameters()) + list(convT.parameters()), 0.1, weight_decay=0.0, momentum=0.9,
                                 max_grad_norm=12)
    y = ops.Conv3dFn.apply(x, None, conv.weight, conv.bias, (1, 1, 1))
    z = ops.ConvTranspose3dFn.apply(y, convT.weight, convT.bias, (2, 2, 2))
    old = conv.weight._mvd_pack16
    old_wf = old[1].clone()
    z.backward(torch.ones_like(z).contiguous(memory_format=torch.channels_last_3d))
    opt.step()
    for mod, tr in ((conv, False), (convT, True)):
        e = mod.weight._mvd_pack16
        assert e[0] == (ops._pack_stamp(mod.weight.detach(), mod.weight), tr, mod.weight.device)
        wf, wb = ops.pack_weight_bf16(mod.weight, tr)
        assert torch.equal(e[1], wf) and torch.equal(e[2], wb)
    assert conv.weight._mvd_pack16[1].data_ptr() != old[1].data_ptr()
    assert torch.equal(old[1], old_wf)  # the pre-step pack is still what the old graph saw
    assert not torch.equal(conv.weight._mvd_pack16[1], old_wf)


@pytest.mark.parametrize("C,K,sp,N", [(64, 32, (4, 5, 6), 2), (320, 256, (2, 2, 2), 2),
                                      (64, 32, (33, 41, 53), 2)])  # the persistent top-level forward, ragged tail
def test_convT3d_bf16_fwd_dgrad(C, K, sp, N):
    from multimodal_mvd_seg_amd._lib import call, i3, query
    g = torch.Generator().manual_seed(C + K)
    x = bf(torch.randn(N, C, *sp, generator=g))
    w = torch.randn(C, K, 2, 2, 2, generator=g) / np.sqrt(C)
    b = torch.randn(K, generator=g) * 0.1
    xr = x.double().requires_grad_()
    ref = F.conv_transpose3d(xr, bf(w).double(), b.double(), 2)
    gy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(gy.double())
    cl = torch.channels_last_3d
    xd = x.to(DEV).contiguous(memory_format=cl)
    wf = torch.empty(8 * C * K, dtype=torch.bfloat16, device=DEV)
    wb = torch.empty(8 * C * K, dtype=torch.bfloat16, device=DEV)
    wd = w.to(DEV).contiguous()
    call("mvd_pack_weight_bf16", _p(wd), _p(wf), _p(wb), K, C, 8, 1, _stream())
    D, H, W = sp
    y = torch.empty((N, K, 2 * D, 2 * H, 2 * W), dtype=torch.bfloat16, device=DEV).contiguous(memory_format=cl)
    ws = torch.empty(max(1024, query("mvd_conv_fwd_workspace_bytes", N, D * H * W, max(C, K))), dtype=torch.uint8,
                     device=DEV)
    bd = b.to(DEV)
    call("mvd_convT3d_fwd_bf16", _p(xd), _p(wf), _p(bd), _p(y), N, D, H, W, C, K, i3((2, 2, 2)), _p(ws), ws.numel(),
         _stream())
    close_bf16(y, ref.detach(), "y")
    dx = torch.empty_like(xd)
    gyd = gy.to(DEV).contiguous(memory_format=cl)  # keep alive across the call
    call("mvd_convT3d_dgrad_bf16", _p(gyd), _p(wb), _p(dx), N, D, H, W, C, K, i3((2, 2, 2)), _p(ws), ws.numel(), _stream())
    close_bf16(dx, xr.grad, "dx")


def close_f32(a, ref, what, rel=2e-5):
    a, ref = a.float().cpu().double(), ref.double()
    err = float((a - ref).abs().max())
    assert err <= rel * float(ref.abs().max()) + 1e-7, f"{what}: max err {err:.3e} at |ref| {float(ref.abs().max()):.3e}"


@pytest.mark.parametrize("C1,C2,K,sp,stride,N", [
    (32, 0, 32, (12, 10, 14), 1, 2),
    (32, 32, 32, (9, 10, 11), 1, 1),
    (32, 0, 64, (16, 12, 20), 2, 1),
    (320, 320, 320, (4, 4, 4), 1, 2),
    (128, 0, 128, (6, 7, 8), (1, 2, 2), 1),
    (32, 0, 32, (34, 44, 52), 1, 2),     # 256-voxel tiles, 756 ragged tiles over 512 workgroups: the x-triple step loop with
                                         # the next tile's buffer loads (out-of-range lanes, zero-record descriptors)
])
def test_conv3d_bf16_wgrad(C1, C2, K, sp, stride, N):
    """dw/dbias from bf16 x and bf16 dy are fp32-accumulated sums of exact products: compare with fp64 on the same
    rounded operands at fp32 tolerance."""
    from multimodal_mvd_seg_amd._lib import call, i3, query
    st = (stride,) * 3 if isinstance(stride, int) else stride
    g = torch.Generator().manual_seed(7 * C1 + K)
    x1 = bf(torch.randn(N, C1, *sp, generator=g))
    x2 = bf(torch.randn(N, C2, *sp, generator=g)) if C2 else None
    C = C1 + C2
    w = torch.zeros(K, C, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(K, dtype=torch.float64, requires_grad=True)
    xin = torch.cat([x1.double(), x2.double()], 1) if C2 else x1.double()
    ref = F.conv3d(xin, w, b, st, 1)
    gy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(gy.double())
    cl = torch.channels_last_3d
    d1 = x1.to(DEV).contiguous(memory_format=cl)
    d2 = x2.to(DEV).contiguous(memory_format=cl) if C2 else None
    gyd = gy.to(DEV).contiguous(memory_format=cl)
    od = tuple(ref.shape[2:])
    ws = torch.empty(max(1024, query("mvd_conv3d_wgrad_workspace_bytes", C, K, 27, N, *od)), dtype=torch.uint8, device=DEV)
    dw = torch.empty(K, C, 3, 3, 3, device=DEV)
    db = torch.empty(K, device=DEV)
    D, H, W = sp
    call("mvd_conv3d_wgrad_bf16", _p(d1), C1, _p(d2), C2, _p(gyd), _p(dw), _p(db), N, D, H, W, K, i3((3, 3, 3)), i3(st),
         _p(ws), ws.numel(), _stream())
    close_f32(dw, w.grad, "dw")
    close_f32(db, b.grad, "dbias")


@pytest.mark.parametrize("C,K,sp,N", [(64, 32, (4, 5, 6), 2), (320, 256, (2, 2, 2), 2)])
def test_convT3d_bf16_wgrad(C, K, sp, N):
    from multimodal_mvd_seg_amd._lib import call, i3, query
    g = torch.Generator().manual_seed(3 * C + K)
    x = bf(torch.randn(N, C, *sp, generator=g))
    w = torch.zeros(C, K, 2, 2, 2, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(K, dtype=torch.float64, requires_grad=True)
    ref = F.conv_transpose3d(x.double(), w, b, 2)
    gy = bf(torch.randn(ref.shape, generator=g))
    ref.backward(gy.double())
    cl = torch.channels_last_3d
    D, H, W = sp
    ws = torch.empty(max(1024, query("mvd_convT3d_wgrad_workspace_bytes", C, K, 8, N, D, H, W)), dtype=torch.uint8,
                     device=DEV)
    dw = torch.empty(C, K, 2, 2, 2, device=DEV)
    db = torch.empty(K, device=DEV)
    xd, gyd = x.to(DEV).contiguous(memory_format=cl), gy.to(DEV).contiguous(memory_format=cl)  # keep alive
    call("mvd_convT3d_wgrad_bf16", _p(xd), _p(gyd), _p(dw), _p(db), N, D, H, W, C, K, i3((2, 2, 2)), _p(ws), ws.numel(),
         _stream())
    close_f32(dw, w.grad, "dw")
    close_f32(db, b.grad, "dbias")


@pytest.mark.parametrize("xb", [False, True])
@pytest.mark.parametrize("N,C,sp", [(2, 32, (8, 8, 8)), (1, 64, (6, 5, 7)), (2, 320, (2, 2, 2))])
def test_instnorm_lrelu_bf16_io(N, C, sp, xb):
    """fp32-or-bf16 conv output -> bf16 activation; statistics/arithmetic as the fp32 kernel, so against fp64 on the
    same (rounded) input the output is 1 bf16 ulp and dgamma/dbeta are fp32-accurate; dx has x's dtype."""
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(N * C + int(xb))
    x = torch.randn(N, C, *sp, generator=g) * 0.7 + 0.2
    if xb:
        x = bf(x)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    gy = bf(torch.randn(N, C, *sp, generator=g))
    xr, gr, br = x.double().requires_grad_(), gamma.double().requires_grad_(), beta.double().requires_grad_()
    ref = F.leaky_relu(F.instance_norm(xr, None, None, gr, br, True, 0.1, 1e-5), 0.01)
    ref.backward(gy.double())
    cl = torch.channels_last_3d
    gx = x.to(DEV).contiguous(memory_format=cl).requires_grad_()
    gg, gb = gamma.to(DEV).requires_grad_(), beta.to(DEV).requires_grad_()
    y = ops.InstanceNormLeakyReLUFn.apply(gx, gg, gb, 1e-5, 0.01, True)
    assert y.dtype == torch.bfloat16 and y.is_contiguous(memory_format=cl)
    close_bf16(y.detach(), ref.detach(), "y")
    y.backward(gy.to(DEV).contiguous(memory_format=cl))
    assert gx.grad.dtype == x.dtype
    if xb:
        close_bf16(gx.grad, xr.grad, "dx")
    else:
        close_f32(gx.grad, xr.grad, "dx")
    close_f32(gg.grad, gr.grad, "dgamma")
    close_f32(gb.grad, br.grad, "dbeta")


def test_seghead_bf16_input():
    from multimodal_mvd_seg_amd import ops
    g = torch.Generator().manual_seed(5)
    N, C, K, sp = 2, 32, 5, (6, 7, 9)
    x = bf(torch.randn(N, C, *sp, generator=g))
    w = torch.randn(K, C, 1, 1, 1, generator=g) * 0.2
    b = torch.randn(K, generator=g) * 0.1
    gl = torch.randn(N, K, *sp, generator=g)
    xr, wr, br = x.double().requires_grad_(), w.double().requires_grad_(), b.double().requires_grad_()
    ref = F.conv3d(xr, wr, br)
    ref.backward(gl.double())
    gx = x.to(DEV).contiguous(memory_format=torch.channels_last_3d).requires_grad_()
    gw, gb = w.to(DEV).requires_grad_(), b.to(DEV).requires_grad_()
    y = ops.SegHeadFn.apply(gx, gw, gb)
    assert y.dtype == torch.float32 and y.is_contiguous()
    close_f32(y.detach(), ref.detach(), "logits")
    y.backward(gl.to(DEV))
    assert gx.grad.dtype == torch.bfloat16
    close_bf16(gx.grad, xr.grad, "dx")
    close_f32(gw.grad, wr.grad, "dw")
    close_f32(gb.grad, br.grad, "db")


@pytest.mark.parametrize("n", [1, 3, 4, 1027, 32 * 33 * 35])
def test_cast_bit_exact(n):
    """round-to-nearest-even fp32 -> bf16 and exact widening, against torch's own conversion"""
    from multimodal_mvd_seg_amd._lib import call
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, generator=g) * 10 ** torch.randint(-6, 6, (n,), generator=g).float()).to(DEV)
    h = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    call("mvd_cast_f32_to_bf16", _p(x), _p(h), n, _stream())
    assert torch.equal(h, x.to(torch.bfloat16))
    back = torch.empty(n, device=DEV)
    call("mvd_cast_bf16_to_f32", _p(h), _p(back), n, _stream())
    assert torch.equal(back, h.float())


def test_unet_bf16_step_no_worse_than_torch_autocast():
    """Whole-network bar for the mixed-precision mode: against the fp64 evaluation of the same 4-stage network, the
    HIP bf16 step must be at least as accurate as the reference's own mixed-precision recipe (torch autocast, here
    CPU bf16: nnUNetTrainer.py:906 with dtype bf16) -- logits, loss and per-parameter gradients."""
    import copy
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_parity import _cfg2_pair
    from multimodal_mvd_seg_amd.network import set_precision
    ora, loss_fn, batch, tr = _cfg2_pair(32, batch_size=2, n_stages=4)
    ora64 = copy.deepcopy(ora).double()
    out64 = ora64(batch["data"].double())
    l64 = loss_fn(out64, [t.double() for t in batch["target"]])
    l64.backward()
    g64 = {n: p.grad for n, p in ora64.named_parameters()}
    oac = copy.deepcopy(ora)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        out_ac = oac(batch["data"])
        l_ac = loss_fn(out_ac, batch["target"])
    l_ac.backward()
    g_ac = {n: p.grad for n, p in oac.named_parameters()}
    set_precision(tr.network, "bf16")
    tr.optimizer.zero_grad()
    out = tr.network(batch["data"].to(DEV))
    l = tr.loss(out, [t.to(DEV) for t in batch["target"]])
    l.backward()
    assert all(o.dtype == torch.float32 for o in out)
    for o, a, r in zip(out, out_ac, out64):
        r = r.detach()
        e_hip = float((o.detach().cpu().double() - r).abs().max())
        e_ac = float((a.detach().double() - r).abs().max())
        assert e_hip <= 1.5 * e_ac + 1e-3 * float(r.abs().max()), f"logits: hip {e_hip:.3e} autocast {e_ac:.3e}"
    assert abs(float(l) - float(l64)) <= max(2 * abs(float(l_ac) - float(l64)), 2e-3 * abs(float(l64)))
    rel_hip, rel_ac = [], []
    for n, p in tr.network.named_parameters():
        r = g64[n]
        nr = float(r.norm())
        if nr < 1e-12:
            continue
        assert p.grad.dtype == torch.float32
        rel_hip.append(float((p.grad.cpu().double() - r).norm()) / nr)
        rel_ac.append(float((g_ac[n].double() - r).norm()) / nr)
    med = lambda v: sorted(v)[len(v) // 2]  # noqa: E731
    assert med(rel_hip) <= 1.25 * med(rel_ac), f"median grad relL2: hip {med(rel_hip):.3e} autocast {med(rel_ac):.3e}"
    assert max(rel_hip) <= 1.5 * max(rel_ac), f"worst grad relL2: hip {max(rel_hip):.3e} autocast {max(rel_ac):.3e}"
    # and one full optimizer step runs (clip + SGD on the fp32 master weights)
    tr.on_train_epoch_start()
    res = tr.train_step({"data": batch["data"].to(DEV), "target": [t.to(DEV) for t in batch["target"]]})
    assert np.isfinite(float(res["loss"]))


@pytest.mark.parametrize("C,T", [(32, 1.0), (8, 4.0)])
def test_feature_kl_bf16_rows(C, T):
    """feature distillation on bf16 NDHWC feature maps: fp32 arithmetic inside, so the loss matches the fp64 oracle
    on the same rounded inputs at fp32 level and the (bf16) gradients to 1 bf16 ulp."""
    from multimodal_mvd_seg_amd import losses
    from oracle import loss_oracle as LO
    g = torch.Generator().manual_seed(C)
    sp = (5, 6, 7)
    a = bf(torch.randn(2, C, *sp, generator=g) * 2)
    b = bf(torch.randn(2, C, *sp, generator=g) * 2)
    ar, br = a.double().requires_grad_(), b.double().requires_grad_()
    ref = LO.l2_loss(ar, br, channel_wise=True, T=T)
    ref.backward()
    cl = torch.channels_last_3d
    ga = a.to(DEV).contiguous(memory_format=cl).requires_grad_()
    gb = b.to(DEV).contiguous(memory_format=cl).requires_grad_()
    out = losses.l2_loss(ga, gb, channel_wise=True, T=T)
    assert abs(float(out) - float(ref)) <= 2e-5 * abs(float(ref)) + 1e-7
    out.backward()
    assert ga.grad.dtype == torch.bfloat16 and gb.grad.dtype == torch.bfloat16
    close_bf16(ga.grad, ar.grad, "d student")
    close_bf16(gb.grad, br.grad, "d teacher")


def test_cfg5_patch_bf16_step_tracks_fp32():
    """BASELINE configs[4]: the 6-stage network on the 160x160x128 patch (batch 1 here) in bf16 mixed precision.  Size
    independent properties: the loss of the bf16 step stays within 1 % of the fp32 step from the same weights, every
    gradient is finite, and the global gradient norms agree within 10 %."""
    from multimodal_mvd_seg_amd import trainer
    from multimodal_mvd_seg_amd.network import set_precision
    strides = [[1, 1, 1]] + [[2, 2, 2]] * 5
    plans = trainer.make_plans((160, 160, 128), strides, batch_size=1)
    ds = {"channel_names": {str(i): str(i) for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    tr = trainer.nnUNetTrainerMI355Benchmark_noDataLoading(plans, "3d_fullres", 0, ds, device=DEV)
    torch.manual_seed(0)
    tr.initialize()
    batch = tr.dummy_batch
    out = {}
    for prec in ("fp32", "bf16"):
        set_precision(tr.network, prec)
        tr.optimizer.zero_grad()
        l = tr.loss(tr.network(batch["data"]), batch["target"])
        l.backward()
        g = tr.optimizer.fp.grad
        assert bool(torch.isfinite(g).all())
        out[prec] = (float(l), float(g.norm()))
    assert abs(out["bf16"][0] - out["fp32"][0]) <= 1e-2 * abs(out["fp32"][0]), out
    assert abs(out["bf16"][1] - out["fp32"][1]) <= 0.1 * out["fp32"][1], out
