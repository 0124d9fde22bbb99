"""GPU parity at the headline configuration (BASELINE configs[1]: 4x128^3, PlainConvUNet 6 stages, fp32).

Four layers of evidence, all through the C ABI:

1. `test_cfg2_full_128cube_step_vs_oracle`: one whole train step (nnUNetTrainer.py:888-925) on the full 4x128^3 patch
   against oracle/step_oracle.py evaluated on this box's host cores: logits 1e-4, loss 1e-5, parameters 1e-5.
2. `test_cfg2_*_exact_integer_data`: every conv / transposed conv / seg head of the network at its REAL shape, batch 2,
   on small-integer data.  All partial sums are integers below 2^24 (the Winograd transforms only add binary
   fractions), so fp32 arithmetic is exact in any summation order and the HIP result must equal torch's CPU result
   BIT FOR BIT: tap tables, halo / border handling, split-K coverage and the two-pointer concat at full size.
3. `test_cfg2_*_block_vs_fp64`: every block at its real shape, batch 2, fed with the activations and upstream gradients
   the oracle network produces at that block (captured from a 2x4x128^3 oracle forward/backward), against an fp64
   evaluation (oracle/fp64_ops.py): y, dX, dW, dgamma, dbeta within 1e-5 relative L2.  Layers above 32^3 compare y /
   dX on three D-slabs (first, middle, last planes) and dW on an 8x8 channel sub-block over ALL voxels -- the same
   sums as the full evaluation, restricted to what a host can compute in fp64 in seconds.
4. `test_cfg2_network_gradients_vs_fp64_truth`: all gradients of the 6-stage network against the fp64 twin.
   LeakyReLU's derivative jumps at 0, so two evaluations that differ by fp32 round-off disagree on the branch of the
   ~1e-6 fraction of voxels with |z| < 1e-6; each such flip moves that layer's gradient by ~1/sqrt(#voxels) relative
   (dbeta but not dgamma, because x_hat ~ 0 there) and everything upstream inherits it.  Round 1 saw this as "HIP 5-13x
   further from fp64 than torch-CPU": both are 1e-3 off, at different layers.  The test evaluates the fp64 truth FOR THE
   BRANCH PATTERN THE HIP FORWARD TOOK (oracle.fp64_ops.MaskedLeakyReLU), checks that the pattern differs from the
   free-running fp64 one only where |z| is at round-off level, and then holds every gradient to a tight bound.
"""
import gc
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
D64 = torch.float64


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from multimodal_mvd_seg_amd import _lib
    _lib.load()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)))
    yield
    gc.collect()
    torch.cuda.empty_cache()


def G(t, requires_grad=False):
    t = t.to(DEV)
    if requires_grad:
        t.requires_grad_()
    return t


def rel_l2(a, ref):
    a, ref = a.detach().cpu().to(D64), ref.detach().to(D64)
    assert a.shape == ref.shape, (tuple(a.shape), tuple(ref.shape))
    return float((a - ref).norm() / (ref.norm() + 1e-300))


# cfg-2 layer table: oracle module path -> (C1, C2, K, input spatial, stride)
def _cfg2_conv_table():
    feats = [32, 64, 128, 256, 320, 320]
    t, sp, cin = {}, 128, 4
    for s, f in enumerate(feats):
        st = 1 if s == 0 else 2
        t[f"encoder.stages.{s}.0.convs.0"] = (cin, 0, f, sp, st)
        sp //= st
        t[f"encoder.stages.{s}.0.convs.1"] = (f, 0, f, sp, 1)
        cin = f
    for d in range(5):  # decoder stage d works at the resolution of encoder stage 4-d
        f = feats[4 - d]
        sp = 128 >> (4 - d)
        t[f"decoder.stages.{d}.convs.0"] = (f, f, f, sp, 1)
        t[f"decoder.stages.{d}.convs.1"] = (f, 0, f, sp, 1)
    return t


CONVS = _cfg2_conv_table()
CONVT = {f"decoder.transpconvs.{d}": ([320, 320, 256, 128, 64][d], [320, 256, 128, 64, 32][d], 4 << d) for d in range(5)}
SEGS = {f"decoder.seg_layers.{d}": ([320, 256, 128, 64, 32][d], 8 << d) for d in range(5)}


# ====================================================================================== 1. the whole step at 128^3
def _cfg2_pair(P, batch_size, n_stages=6):
    from multimodal_mvd_seg_amd import trainer
    from oracle import loss_oracle as LO, step_oracle as SO, unet_oracle as UO
    strides = UO.CONFIGS["cfg2"]["strides"][:n_stages]
    ora = UO.build_plainconv_unet(4, 5, n_stages, strides, seed=0)
    batch = SO.synthetic_batch(batch_size, 4, (P, P, P), strides, num_classes=5, seed=1234)
    loss_fn = LO.build_loss(len(batch["target"]))
    plans = trainer.make_plans((P, P, P), strides, batch_size=batch_size)
    ds = {"channel_names": {str(i): str(i) for i in range(4)},
          "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}
    tr = trainer.nnUNetTrainerMI355(plans, "3d_fullres", 0, ds, device=DEV)
    tr.initialize()
    tr.network.load_state_dict(ora.state_dict())
    return ora, loss_fn, batch, tr


def test_cfg2_full_128cube_step_vs_oracle():
    """BASELINE configs[1] at its full patch size (batch 1): forward logits at all five deep-supervision levels, the
    DC+CE loss, the global gradient norm and every parameter after clip + SGD-Nesterov, HIP vs torch-CPU oracle."""
    from oracle import step_oracle as SO
    ora, loss_fn, batch, tr = _cfg2_pair(128, 1)
    opt = SO.make_optimizer(ora.parameters())
    l_ref, out_ref, gn_ref = SO.train_step(ora, loss_fn, opt, batch)
    gbatch = {"data": batch["data"].to(DEV), "target": [t.to(DEV) for t in batch["target"]]}
    with torch.no_grad():
        out = tr.network(gbatch["data"])
    for i, (o, r) in enumerate(zip(out, out_ref)):
        err = float((o.cpu() - r.detach()).abs().max())
        assert err <= 1e-4, f"logits{i}: max abs err {err:.3e}"
    del out
    tr.on_train_epoch_start()
    res = tr.train_step(gbatch)
    assert abs(float(res["loss"]) - float(l_ref)) <= 1e-5 * max(1.0, abs(float(l_ref)))
    assert abs(float(tr.optimizer.grad_norm()) - gn_ref) <= 1e-3 * gn_ref
    ref_params = dict(ora.named_parameters())
    worst = 0.0
    for n, p in tr.network.named_parameters():
        e = float((p.detach().cpu() - ref_params[n].detach()).abs().max())
        worst = max(worst, e)
        assert e <= 1e-5, f"param after step: {n}: {e:.3e}"
    print(f"[cfg2 128^3 step] loss hip {float(res['loss']):.7f} oracle {float(l_ref):.7f}; worst param err {worst:.2e}")


# ====================================================================================== 2. exact integer data
def _ints(g, shape, lo, hi):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


@pytest.mark.parametrize("name", list(CONVS))
def test_cfg2_conv_exact_integer_data(name):
    from multimodal_mvd_seg_amd import ops
    C1, C2, K, sp, st = CONVS[name]
    g = torch.Generator().manual_seed(sum(name.encode()))
    N = 2
    x1 = _ints(g, (N, C1, sp, sp, sp), -2, 2)
    x2 = _ints(g, (N, C2, sp, sp, sp), -2, 2) if C2 else None
    w = _ints(g, (K, C1 + C2, 3, 3, 3), -2, 2)
    b = _ints(g, (K,), -3, 3)
    xs = [t.clone().requires_grad_() for t in ([x1, x2] if C2 else [x1])]
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv3d(torch.cat(xs, 1) if C2 else xs[0], wr, br, st, 1)
    gy = _ints(g, tuple(ref.shape), -1, 1)
    ref.backward(gy)
    need_dx = C1 != 4  # the 4-modality input needs no gradient on the path
    g1 = G(x1, need_dx)
    g2 = G(x2, True) if C2 else None
    gw, gb = G(w, True), G(b, True)
    y = ops.Conv3dFn.apply(g1, g2, gw, gb, (st,) * 3)
    y.backward(G(gy))
    assert torch.equal(y.detach().cpu(), ref.detach()), f"{name}: y differs (max {float((y.cpu() - ref).abs().max())})"
    if need_dx:
        assert torch.equal(g1.grad.cpu(), xs[0].grad), f"{name}: dx1"
    if C2:
        assert torch.equal(g2.grad.cpu(), xs[1].grad), f"{name}: dx2"
    assert torch.equal(gw.grad.cpu(), wr.grad), f"{name}: dw (max {float((gw.grad.cpu() - wr.grad).abs().max())})"
    assert torch.equal(gb.grad.cpu(), br.grad), f"{name}: db"


@pytest.mark.parametrize("name", [n for n in CONVS if CONVS[n][0] % 32 == 0])
def test_cfg2_conv_bf16_exact_integer_data(name):
    """The bf16 engines (configs[3], [4]) at the real cfg-2 layer shapes, batch 2, on small-integer data: the operands are
    exact in bf16, every product and fp32 partial sum is an exact integer, so y / dX must equal torch's exact fp32 result
    rounded once to bf16 (round-to-nearest-even), and the fp32 weight / bias gradients must be exact -- bit for bit.
    Covers k_fwd16q (32 -> 32 at 128^3), k_fwd16 in all its tile variants incl. the split-reduce path, k_wgrad16."""
    from multimodal_mvd_seg_amd import ops
    C1, C2, K, sp, st = CONVS[name]
    g = torch.Generator().manual_seed(sum(name.encode()) + 1)
    N = 2
    x1 = _ints(g, (N, C1, sp, sp, sp), -2, 2)
    x2 = _ints(g, (N, C2, sp, sp, sp), -2, 2) if C2 else None
    w = _ints(g, (K, C1 + C2, 3, 3, 3), -2, 2)
    b = _ints(g, (K,), -3, 3)
    xs = [t.clone().requires_grad_() for t in ([x1, x2] if C2 else [x1])]
    wr, br = w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv3d(torch.cat(xs, 1) if C2 else xs[0], wr, br, st, 1)
    gy = _ints(g, tuple(ref.shape), -1, 1)
    ref.backward(gy)
    cl = torch.channels_last_3d
    BF = torch.bfloat16
    g1 = x1.to(DEV).to(BF).contiguous(memory_format=cl).requires_grad_()
    g2 = x2.to(DEV).to(BF).contiguous(memory_format=cl).requires_grad_() if C2 else None
    gw, gb = G(w, True), G(b, True)
    y = ops.Conv3dFn.apply(g1, g2, gw, gb, (st,) * 3)
    assert y.dtype == BF
    y.backward(gy.to(DEV).to(BF).contiguous(memory_format=cl))
    assert torch.equal(y.detach().cpu(), ref.detach().to(BF)), f"{name}: y"
    assert torch.equal(g1.grad.cpu(), xs[0].grad.to(BF)), f"{name}: dx1"
    if C2:
        assert torch.equal(g2.grad.cpu(), xs[1].grad.to(BF)), f"{name}: dx2"
    assert gw.grad.dtype == torch.float32
    assert torch.equal(gw.grad.cpu(), wr.grad), f"{name}: dw (max {float((gw.grad.cpu() - wr.grad).abs().max())})"
    assert torch.equal(gb.grad.cpu(), br.grad), f"{name}: db"


@pytest.mark.parametrize("name", list(CONVT))
def test_cfg2_convT_exact_integer_data(name):
    from multimodal_mvd_seg_amd import ops
    C, K, sp = CONVT[name]
    g = torch.Generator().manual_seed(C + K + sp)
    x = _ints(g, (2, C, sp, sp, sp), -2, 2)
    w = _ints(g, (C, K, 2, 2, 2), -2, 2)
    b = _ints(g, (K,), -3, 3)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv_transpose3d(xr, wr, br, 2)
    gy = _ints(g, tuple(ref.shape), -1, 1)
    ref.backward(gy)
    gx, gw, gb = G(x, True), G(w, True), G(b, True)
    y = ops.ConvTranspose3dFn.apply(gx, gw, gb, (2, 2, 2))
    y.backward(G(gy))
    assert torch.equal(y.detach().cpu(), ref.detach()), f"{name}: y"
    assert torch.equal(gx.grad.cpu(), xr.grad), f"{name}: dx"
    assert torch.equal(gw.grad.cpu(), wr.grad), f"{name}: dw"
    assert torch.equal(gb.grad.cpu(), br.grad), f"{name}: db"


@pytest.mark.parametrize("name", list(SEGS))
def test_cfg2_seghead_exact_integer_data(name):
    from multimodal_mvd_seg_amd import ops
    C, sp = SEGS[name]
    g = torch.Generator().manual_seed(C + sp)
    x = _ints(g, (2, C, sp, sp, sp), -2, 2)
    w = _ints(g, (5, C, 1, 1, 1), -2, 2)
    b = _ints(g, (5,), -3, 3)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    ref = F.conv3d(xr, wr, br)
    gy = _ints(g, tuple(ref.shape), -1, 1)
    ref.backward(gy)
    gx, gw, gb = G(x, True), G(w, True), G(b, True)
    y = ops.SegHeadFn.apply(gx, gw, gb)
    y.backward(G(gy))
    assert torch.equal(y.detach().cpu(), ref.detach()), f"{name}: logits"
    assert torch.equal(gx.grad.cpu(), xr.grad), f"{name}: dx"
    assert torch.equal(gw.grad.cpu(), wr.grad), f"{name}: dw"
    assert torch.equal(gb.grad.cpu(), br.grad), f"{name}: db"


# ====================================================================================== 3. blocks on oracle activations
class _Capture:
    """One fp32 oracle forward/backward of the cfg-2 network on a 2x4x128^3 batch with, per block, the tensors each
    HIP op sees in the real network: block input, raw conv output and its gradient, activated output and its gradient."""

    def __init__(self, P=128, B=2):
        from oracle import loss_oracle as LO, step_oracle as SO, unet_oracle as UO
        strides = UO.CONFIGS["cfg2"]["strides"]
        self.net = UO.build_plainconv_unet(4, 5, 6, strides, seed=0)
        batch = SO.synthetic_batch(B, 4, (P, P, P), strides, num_classes=5, seed=1234)
        loss_fn = LO.build_loss(len(batch["target"]))
        self.blocks, self.convT, self.seg = {}, {}, {}
        hooks = []
        mods = dict(self.net.named_modules())
        for name in CONVS:
            m, rec = mods[name], {}
            self.blocks[name] = rec
            hooks.append(m.register_forward_pre_hook(lambda mod, inp, rec=rec: rec.__setitem__("x", inp[0].detach())))

            def conv_hook(mod, inp, out, rec=rec):
                out.retain_grad()
                rec["y_conv"] = out

            def blk_hook(mod, inp, out, rec=rec):
                out.retain_grad()
                rec["y_act"] = out
            hooks.append(m.conv.register_forward_hook(conv_hook))
            hooks.append(m.register_forward_hook(blk_hook))
        for name, store in list((n, self.convT) for n in CONVT) + list((n, self.seg) for n in SEGS):
            rec = {}
            store[name] = rec

            def io_hook(mod, inp, out, rec=rec):
                out.retain_grad()
                rec["x"], rec["y"] = inp[0].detach(), out
            hooks.append(mods[name].register_forward_hook(io_hook))
        out = self.net(batch["data"])
        loss_fn(out, batch["target"]).backward()
        for h in hooks:
            h.remove()
        for rec in self.blocks.values():
            rec["dy_conv"], rec["dy_act"] = rec["y_conv"].grad, rec["y_act"].grad
            rec["y_conv"], rec["y_act"] = rec["y_conv"].detach(), rec["y_act"].detach()
        for rec in list(self.convT.values()) + list(self.seg.values()):
            rec["dy"], rec["y"] = rec["y"].grad, rec["y"].detach()
        self.mods = mods


_CAP = {}


@pytest.fixture(scope="module")
def cap():
    if "c" not in _CAP:
        _CAP["c"] = _Capture()
    yield _CAP["c"]


def _sub(n, k=8):
    """k channel indices spread over [0, n)."""
    return sorted(set(int(i) for i in np.linspace(0, n - 1, min(k, n)).round()))


@pytest.mark.parametrize("name", list(CONVS))
def test_cfg2_conv_block_vs_fp64(cap, name):
    from multimodal_mvd_seg_amd import ops
    from oracle import fp64_ops as O
    C1, C2, K, sp, st = CONVS[name]
    rec, m = cap.blocks[name], cap.mods[name]
    x, dy = rec["x"], rec["dy_conv"]
    assert tuple(x.shape) == (2, C1 + C2, sp, sp, sp), tuple(x.shape)
    w, b = m.conv.weight.detach(), m.conv.bias.detach()
    need_dx = C1 != 4
    g1 = G(x[:, :C1].contiguous(), need_dx)
    g2 = G(x[:, C1:].contiguous(), True) if C2 else None
    gw, gb = G(w, True), G(b, True)
    y = ops.Conv3dFn.apply(g1, g2, gw, gb, (st,) * 3)
    y.backward(G(dy))
    y_h = y.detach().cpu()
    dx_h = torch.cat([g1.grad.cpu()] + ([g2.grad.cpu()] if C2 else []), 1) if need_dx else None
    x64, w64, b64, dy64 = x.to(D64), w.to(D64), b.to(D64), dy.to(D64)
    so = sp // st
    errs = {}
    if sp <= 32:
        y64 = O.conv3d_fwd(x64, w64, b64, st)
        dx64, dw64, db64 = O.conv3d_bwd(x64, w64, dy64, st, need_dx)
        errs["y"] = rel_l2(y_h, y64)
        if need_dx:
            errs["dx"] = rel_l2(dx_h, dx64)
        errs["dw"] = rel_l2(gw.grad, dw64)
    else:
        for tag, d0 in (("first", 0), ("mid", so // 2 - 1), ("last", so - 3)):
            errs[f"y[{tag}]"] = rel_l2(y_h[:, :, d0:d0 + 3], O.conv3d_fwd_planes(x64, w64, b64, st, d0, d0 + 3))
        if need_dx:
            for tag, i0 in (("first", 0), ("mid", sp // 2 - 1), ("last", sp - 3)):
                errs[f"dx[{tag}]"] = rel_l2(dx_h[:, :, i0:i0 + 3], O.conv3d_dx_planes(w64, dy64, st, (sp, sp, sp), i0, i0 + 3))
        ks, cs = _sub(K), _sub(C1 + C2)
        _, dws, _ = O.conv3d_bwd(x64[:, cs].contiguous(), w64[ks][:, cs].contiguous(), dy64[:, ks].contiguous(), st, False)
        errs["dw[8x8 block, all voxels]"] = rel_l2(gw.grad.cpu()[ks][:, cs], dws)
        db64 = dy64.sum((0, 2, 3, 4))
    # the bias gradient is analytically ~0 (dy is an InstanceNorm input gradient: zero mean per instance): absolute bound
    db_tol = 2e-6 * float(dy64.abs().sum((0, 2, 3, 4)).max()) + 1e-12
    assert float((gb.grad.cpu().to(D64) - db64).abs().max()) <= db_tol, f"{name}: db"
    print(f"[{name}] " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v <= 1e-5, f"{name}: {k} relative L2 error {v:.2e} vs fp64"
    # the statistics epilogue of the Winograd kernel against the plain two-pass statistics, at the real shape
    if getattr(y, "_mvd_tile_stats", None) is not None:
        ga, be = G(m.norm.weight.detach()), G(m.norm.bias.detach())
        z1 = ops.InstanceNormLeakyReLUFn.apply(y.detach().requires_grad_(False), ga, be, 1e-5, 0.01)
        # a detached view drops the attribute -> hand the same storage with the statistics attached
        yy = y.detach()
        yy._mvd_tile_stats = y._mvd_tile_stats
        z2 = ops.InstanceNormLeakyReLUFn.apply(yy, ga, be, 1e-5, 0.01)
        d = float((z1 - z2).abs().max())
        assert d <= 5e-6 * max(1.0, float(z1.abs().max())), f"{name}: epilogue statistics vs two-pass: {d:.2e}"


@pytest.mark.parametrize("name", list(CONVS))
def test_cfg2_instnorm_block_vs_fp64(cap, name):
    from multimodal_mvd_seg_amd import ops
    from oracle import fp64_ops as O
    rec, m = cap.blocks[name], cap.mods[name]
    x, dy = rec["y_conv"], rec["dy_act"]
    ga, be = m.norm.weight.detach(), m.norm.bias.detach()
    gx, gg, gb = G(x, True), G(ga, True), G(be, True)
    y = ops.InstanceNormLeakyReLUFn.apply(gx, gg, gb, 1e-5, 0.01)
    y.backward(G(dy))
    y_h = y.detach().cpu()
    y64, z64, xhat, rstd = O.instnorm_lrelu_fwd(x.to(D64), ga.to(D64), be.to(D64))
    mask_h = y_h > 0
    flips = mask_h != (z64 > 0)
    nflip = int(flips.sum())
    if nflip:
        zf = float(z64[flips].abs().max())
        assert zf <= 1e-5, f"{name}: LeakyReLU branch differs from fp64 at |z| = {zf:.2e} (not round-off)"
    assert nflip <= max(4, int(2e-5 * x.numel())), f"{name}: {nflip} branch flips of {x.numel()}"
    dx64, dg64, db64 = O.instnorm_lrelu_bwd(dy.to(D64), xhat, rstd, ga.to(D64), mask_h)
    errs = {"y": rel_l2(y_h, y64), "dx": rel_l2(gx.grad, dx64), "dgamma": rel_l2(gg.grad, dg64),
            "dbeta": rel_l2(gb.grad, db64)}
    print(f"[{name}] flips {nflip}/{x.numel()} " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    assert errs["y"] <= 2e-6, f"{name}: y {errs['y']:.2e}"
    for k in ("dx", "dgamma", "dbeta"):
        assert errs[k] <= 1e-5, f"{name}: {k} relative L2 error {errs[k]:.2e} vs fp64"


@pytest.mark.parametrize("name", list(CONVT))
def test_cfg2_convT_block_vs_fp64(cap, name):
    from multimodal_mvd_seg_amd import ops
    from oracle import fp64_ops as O
    rec, m = cap.convT[name], cap.mods[name]
    x, dy = rec["x"], rec["dy"]
    w, b = m.weight.detach(), m.bias.detach()
    gx, gw, gb = G(x, True), G(w, True), G(b, True)
    y = ops.ConvTranspose3dFn.apply(gx, gw, gb, (2, 2, 2))
    y.backward(G(dy))
    y64 = O.convT3d_fwd(x.to(D64), w.to(D64), b.to(D64), 2)
    dx64, dw64, db64 = O.convT3d_bwd(x.to(D64), w.to(D64), dy.to(D64), 2)
    errs = {"y": rel_l2(y, y64), "dx": rel_l2(gx.grad, dx64), "dw": rel_l2(gw.grad, dw64), "db": rel_l2(gb.grad, db64)}
    print(f"[{name}] " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v <= 1e-5, f"{name}: {k} relative L2 error {v:.2e} vs fp64"


@pytest.mark.parametrize("name", list(SEGS))
def test_cfg2_seghead_block_vs_fp64(cap, name):
    from multimodal_mvd_seg_amd import ops
    rec, m = cap.seg[name], cap.mods[name]
    x, dl = rec["x"], rec["dy"]
    w, b = m.weight.detach(), m.bias.detach()
    gx, gw, gb = G(x, True), G(w, True), G(b, True)
    y = ops.SegHeadFn.apply(gx, gw, gb)
    y.backward(G(dl))
    x64, w64, dl64 = x.to(D64), w.to(D64).view(5, -1), dl.to(D64)
    y64 = torch.einsum('kc,ncdhw->nkdhw', w64, x64) + b.to(D64).view(1, 5, 1, 1, 1)
    errs = {"y": rel_l2(y, y64)}
    if float(dl64.abs().max()) > 0:  # the lowest head has weight 0 in the deep-supervision loss: exact zeros
        errs["dx"] = rel_l2(gx.grad, torch.einsum('kc,nkdhw->ncdhw', w64, dl64))
        errs["dw"] = rel_l2(gw.grad.view(5, -1), torch.einsum('nkdhw,ncdhw->kc', dl64, x64))
        errs["db"] = rel_l2(gb.grad, dl64.sum((0, 2, 3, 4)))
    else:
        assert float(gx.grad.abs().max()) == 0 and float(gw.grad.abs().max()) == 0
    print(f"[{name}] " + " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    for k, v in errs.items():
        assert v <= 1e-5, f"{name}: {k} relative L2 error {v:.2e} vs fp64"


def test_zz_release_capture():
    """frees the ~20 GB of captured activations before the other test modules run"""
    _CAP.clear()
    gc.collect()


# ====================================================================================== 4. all gradients, 6 stages
def test_cfg2_network_gradients_vs_fp64_truth():
    """Six stages, batch 2, 64^3 (InstanceNorms over 8 voxels at the bottleneck).  Bar: every parameter gradient within
    5e-5 relative L2 of the fp64 twin evaluated with the HIP forward's LeakyReLU branch pattern (see the module
    docstring); the pattern may differ from the free-running fp64 pattern only where |z| <= 1e-5.
    Measured on MI355X (round 2): 50 branch flips among 89.3 M activations; worst gradient error 8.4e-6 with the same
    branches against 3.8e-3 versus the free-running fp64 network -- the whole round-1 discrepancy is branch flips."""
    from multimodal_mvd_seg_amd.network import ConvDropoutNormReLU
    from oracle import fp64_ops as O
    ora, loss_fn, batch, tr = _cfg2_pair(64, 2)
    masks, hooks = {}, []
    for n, m in tr.network.named_modules():
        if isinstance(m, ConvDropoutNormReLU):
            hooks.append(m.register_forward_hook(
                lambda mod, inp, out, n=n: masks.__setitem__(n, (out.detach() > 0).cpu().contiguous())))
    tr.optimizer.zero_grad()
    from multimodal_mvd_seg_amd import network as _net
    seg_fuse = _net.FUSE_SEGHEAD[0]
    _net.FUSE_SEGHEAD[0] = False   # the hooks need every block's ACTIVATED output: the last block's exists only un-fused
    try:                           # (round 3: its norm + act otherwise run inside the seg head's loaders; same arithmetic)
        tr.loss(tr.network(batch["data"].to(DEV)), [t.to(DEV) for t in batch["target"]]).backward()
    finally:
        _net.FUSE_SEGHEAD[0] = seg_fuse
    for h in hooks:
        h.remove()
    assert len(masks) == 22
    tgt64 = [t.double() for t in batch["target"]]
    rec = {}
    twin = O.fp64_twin(ora, masks=masks, record=rec)
    loss_fn(twin(batch["data"].double()), tgt64).backward()
    total = nflip = 0
    for n, z in rec.items():
        f = masks[n] != (z > 0)
        total += z.numel()
        k = int(f.sum())
        nflip += k
        if k:
            assert float(z[f].abs().max()) <= 1e-5, f"{n}: branch differs from fp64 at |z| = {float(z[f].abs().max()):.2e}"
    assert nflip <= 1e-5 * total
    # for the record: the same network free-running in fp64 (its own branches) -- what round 1 compared against
    free = O.fp64_twin(ora)
    loss_fn(free(batch["data"].double()), tgt64).backward()
    g64, gfree = dict(twin.named_parameters()), dict(free.named_parameters())
    worst, worst_free = 0.0, 0.0
    for n, p in tr.network.named_parameters():
        r = g64[n].grad
        nr = float(r.norm())
        if nr < 1e-12:  # conv bias in front of InstanceNorm (analytically zero) / the zero-weighted lowest head
            assert float(p.grad.abs().max()) < 1e-5, n
            continue
        e = float((p.grad.cpu().double() - r).norm()) / nr
        ef = float((p.grad.cpu().double() - gfree[n].grad).norm()) / (float(gfree[n].grad.norm()) + 1e-300)
        worst, worst_free = max(worst, e), max(worst_free, ef)
        assert e <= 5e-5, f"{n}: relative L2 error {e:.2e} vs fp64 (HIP branch pattern); vs free-running fp64 {ef:.2e}"
    print(f"[cfg2 gradients, 6 stages] {nflip} LeakyReLU branch flips among {total} activations; worst relative L2 "
          f"error vs fp64 with the same branches {worst:.2e}, vs free-running fp64 {worst_free:.2e}")
