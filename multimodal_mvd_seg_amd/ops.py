"""torch.autograd.Function wrappers over the C ABI (include/mvdseg_hip.h).

PyTorch supplies device memory (caching allocator), the current HIP stream and the autograd tape; every
arithmetic operation of the hot path runs in libmvdseg_hip.so.  Activations are 5-D tensors of logical shape
[N,C,D,H,W] stored NDHWC (torch.channels_last_3d); logits / targets / volumes of the topology losses are planar.
"""
import ctypes
import os
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import call, i3, query

CL3D = torch.channels_last_3d


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("multimodal_mvd_seg_amd ops need tensors on the MI355X (cuda) device; "
                               "there is no CPU fallback")
        if t is not None and t.dtype not in (torch.float32, torch.bfloat16, torch.uint8, torch.int32, torch.int64,
                                             torch.int16):
            raise RuntimeError(f"unsupported dtype {t.dtype} (kernels take fp32, or bf16 activations in mixed precision)")


class _Workspace:
    """Grow-only scratch buffer per device; kernels of one stream run in order so sharing it is safe."""
    _bufs = {}

    @classmethod
    def get(cls, nbytes, device):
        key = (device.index, torch.cuda.current_stream().cuda_stream)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


def _take_grad(param):
    """Direct gradient sink of optim.FlatParams: the tensor to write this parameter's gradient into (its slice of the
    flat gradient buffer) when it has not been written in this step, else None (-> return the gradient to autograd)."""
    take = getattr(param, '_mvd_take_grad', None) if param is not None else None
    t = take() if take is not None else None
    if t is not None and not (t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda):
        raise RuntimeError("flat gradient slice must be a contiguous fp32 cuda tensor")
    return t


def _grad_done(param):
    done = getattr(param, '_mvd_grad_done', None)
    if done is not None:
        done()


def _is_cl3d(t):
    return t.dim() == 5 and t.is_contiguous(memory_format=CL3D)


def empty_cl3d(shape, device, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=device, memory_format=CL3D)


BF16 = torch.bfloat16


def _is_bf16(t):
    return t is not None and t.dtype == BF16


def to_ndhwc(t):
    """[N,C,D,H,W] in any layout -> physical NDHWC (HIP transpose kernel when the tensor is planar)."""
    _require_cuda(t)
    if _is_cl3d(t):
        return t
    if t.dtype == BF16:
        # bf16 activations are produced NDHWC by these ops; a planar one can only come from outside the path
        raise RuntimeError("bf16 activations must already be NDHWC (torch.channels_last_3d)")
    if not t.is_contiguous():
        t = t.contiguous()
    N, C = t.shape[:2]
    V = t[0, 0].numel()
    out = empty_cl3d(t.shape, t.device)
    call("mvd_nchw_to_ndhwc", _p(t), _p(out), N, C, V, _stream())
    return out


def to_planar(t):
    """physical NDHWC -> contiguous NCDHW."""
    _require_cuda(t)
    if t.is_contiguous():
        return t
    if not _is_cl3d(t):
        return t.contiguous()
    N, C = t.shape[:2]
    V = t[0, 0].numel()
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    call("mvd_ndhwc_to_nchw", _p(t), _p(out), N, C, V, _stream())
    return out


def pack_weight(weight, transposed):
    """torch Conv3d [K,C,kd,kh,kw] / ConvTranspose3d [C,K,kd,kh,kw] -> (wf [T,C,K], wb [T,K,C])."""
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    if transposed:
        C, K = w.shape[:2]
    else:
        K, C = w.shape[:2]
    T = w[0, 0].numel()
    wf = torch.empty((T, C, K), dtype=torch.float32, device=w.device)
    wb = torch.empty((T, K, C), dtype=torch.float32, device=w.device)
    call("mvd_pack_weight", _p(w), _p(wf), _p(wb), K, C, T, 1 if transposed else 0, _stream())
    return wf, wb


# ---------------------------------------------------------------------------------------------- packed-weight cache (fp32)
# The packed copies (wf / wb, Winograd uf / ub) of a conv weight live on the weight tensor object (`_mvd_pack`) and are
# valid for one stamp = (global pack epoch, epoch of the flat buffer the parameter is a view of, version of the parameter,
# storage pointer, version of that flat buffer).  torch in-place writes to the parameter or to optim.FlatParams.flat
# (dist.broadcast, `flat -= ...`) bump a version; the fused optimizer -- which updates ITS flat buffer through a raw
# pointer -- bumps that buffer's epoch and re-packs the weights that live in it in ONE launch (repack_all(fp):
# mvd_pack_weights_batch); the weights of another optimizer in the process are not touched (round 3: they used to be
# re-packed and their saved graphs invalidated too).  A stale or missing entry is packed on the spot by the per-layer
# entries.  The one write torch cannot see is `p.data.copy_()` / any other raw-pointer writer: such code must call
# invalidate_packs() (FlatParams.invalidate_packs), which bumps the global epoch.
_PACK_EPOCH = [0]
_PACK_LIVE = []
_PACK16_LIVE = []


def invalidate_packs():
    """Declare every cached packed weight stale (after a write to parameter memory that bypasses torch's version
    counters).  The next forward re-packs per layer."""
    _PACK_EPOCH[0] += 1


def _check_pack_generation(saved, what):
    """`saved` = (entry, generation at forward time).  The fp32 packs are rebuilt IN PLACE; a graph kept across a weight
    update would silently back-propagate through the new weights (torch raises for its own saved tensors: so do we)."""
    if saved is not None and saved[0].gen != saved[1]:
        raise RuntimeError(f"{what}: the weights were updated (optimizer.step() or an in-place write) between the forward "
                           "and the backward pass of this graph; the packed copies saved for backward were rebuilt")


class _PackEntry:
    __slots__ = ("transposed", "K", "C", "T", "wf", "wb", "uf", "ub", "stamp", "gen")


def _flat_of(w, owner=None):
    return getattr(owner if owner is not None else w, "_mvd_flat", None)


def _pack_stamp(w, owner=None):
    # data_ptr: `p.data = other` swaps the storage without a version bump on the parameter object
    flat = _flat_of(w, owner)
    local = getattr(flat, "_mvd_epoch", None) if flat is not None else None
    return (_PACK_EPOCH[0], local[0] if local is not None else 0, w._version, w.data_ptr(),
            flat._version if flat is not None else -1)


def _packed(weight, transposed, want_uf=False, want_ub=False):
    w = weight.detach()
    if not w.is_contiguous():  # no cache for a strided view: pack a contiguous copy
        e = _PackEntry()
        e.gen = 0
        e.wf, e.wb = pack_weight(weight, transposed)
        e.uf = e.ub = None
        if want_uf or want_ub:
            wc = w.contiguous()
            K, C = wc.shape[:2]
            n = query("mvd_wino_weight_elems", C, K)
            e.uf = torch.empty((n,), dtype=torch.float32, device=w.device) if want_uf else None
            e.ub = torch.empty((n,), dtype=torch.float32, device=w.device) if want_ub else None
            call("mvd_pack_weight_wino", _p(wc), _p(e.uf), _p(e.ub), K, C, _stream())
        return e
    e = getattr(weight, "_mvd_pack", None)
    stale = False
    if e is None or e.transposed != transposed or e.wf.device != w.device:
        e = _PackEntry()
        e.transposed = transposed
        if transposed:
            e.C, e.K = w.shape[:2]
        else:
            e.K, e.C = w.shape[:2]
        e.T = w[0, 0].numel()
        e.wf = torch.empty((e.T, e.C, e.K), dtype=torch.float32, device=w.device)
        e.wb = torch.empty((e.T, e.K, e.C), dtype=torch.float32, device=w.device)
        e.uf = e.ub = None
        e.stamp = None
        e.gen = 0
        weight._mvd_pack = e
        if len(_PACK_LIVE) >= 4096:  # inference-only use never calls repack_all: drop dead references here
            _PACK_LIVE[:] = [r for r in _PACK_LIVE if r() is not None]
        _PACK_LIVE.append(weakref.ref(weight))
        stale = True
    new_wino = False
    if (want_uf and e.uf is None) or (want_ub and e.ub is None):
        n = query("mvd_wino_weight_elems", e.C, e.K)
        if want_uf and e.uf is None:
            e.uf = torch.empty((n,), dtype=torch.float32, device=w.device)
        if want_ub and e.ub is None:
            e.ub = torch.empty((n,), dtype=torch.float32, device=w.device)
        new_wino = True
    if stale or e.stamp != _pack_stamp(w, weight):
        e.gen += 1  # wf / wb / uf / ub are overwritten in place
        call("mvd_pack_weight", _p(w), _p(e.wf), _p(e.wb), e.K, e.C, e.T, 1 if transposed else 0, _stream())
        new_wino = e.uf is not None or e.ub is not None
        e.stamp = _pack_stamp(w, weight)
    if new_wino:
        call("mvd_pack_weight_wino", _p(w), _p(e.uf), _p(e.ub), e.K, e.C, _stream())
    return e


def _live(refs, attr, fp):
    """(weight, cache entry) of the live registered weights -- all of them, or those that are views of `fp`'s buffer."""
    out, alive = [], []
    for r in refs:
        w = r()
        e = getattr(w, attr, None) if w is not None else None
        if e is None:
            continue
        alive.append(r)
        if fp is None or _flat_of(w) is fp.flat:
            out.append((w, e))
    refs[:] = alive
    return out


def _repack_all_bf16(fp=None):
    """The bf16 packs of the live weights (of `fp`, or all) that have them, in one launch (mvd_pack_weights_bf16_batch)
    into views of ONE buffer per device.  Default: a FRESH buffer per call -- the tensors an autograd graph saved for
    backward keep their old buffer alive, exactly as with the per-layer packs (fresh tensors per stamp).  With
    `fp.pack16_inplace` (set by a trainer that replays its step as a hipGraph: the captured forward must read the addresses
    the captured repack of the previous replay wrote) the buffer is owned by `fp`, persists and is rewritten in place;
    `fp.pack16_gen` then guards saved autograd graphs the way the fp32 entries' generations do."""
    jobs = []
    for w, e in _live(_PACK16_LIVE, "_mvd_pack16", fp):
        d = w.detach()
        if d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and d.device == e[0][2]:
            jobs.append((d, e[0][1], w))
    inplace = fp is not None and getattr(fp, "pack16_inplace", False)
    for dev in {d.device for d, _, _w in jobs}:
        js = [j for j in jobs if j[0].device == dev]
        sizes = [d.numel() for d, _, _w in js]
        pad = lambda n: (n + 127) // 128 * 128  # 256-byte aligned views
        total = 2 * sum(pad(n) for n in sizes)
        buf = None
        if inplace:
            sig = (dev, tuple((id(w), n) for (_d, _t, w), n in zip(js, sizes)))
            held = getattr(fp, "_pack16_buf", None)
            if held is not None and held[0] == sig:
                buf = held[1]
                fp.pack16_gen[0] += 1
            else:
                buf = torch.empty((total,), dtype=BF16, device=dev)
                fp._pack16_buf = (sig, buf)
        if buf is None:
            buf = torch.empty((total,), dtype=BF16, device=dev)
        views, o = [], 0
        for n in sizes:
            views.append((buf[o:o + n], buf[o + pad(n):o + pad(n) + n]))
            o += 2 * pad(n)
        n = len(js)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
        shp = []
        for (d, tr, _w) in js:
            (C, K) = d.shape[:2] if tr else (d.shape[1], d.shape[0])
            shp.append((K, C, d[0, 0].numel()))
        cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
        with torch.cuda.device(dev):
            call("mvd_pack_weights_bf16_batch", n, cast(PA(*[d.data_ptr() for d, _, _w in js])),
                 cast(PA(*[v[0].data_ptr() for v in views])), cast(PA(*[v[1].data_ptr() for v in views])),
                 cast(IA(*[q[0] for q in shp])), cast(IA(*[q[1] for q in shp])), cast(IA(*[q[2] for q in shp])),
                 cast(IA(*[1 if tr else 0 for _, tr, _w in js])), _stream())
        for (d, tr, w), v in zip(js, views):
            w._mvd_pack16 = ((_pack_stamp(d, w), tr, d.device), v[0], v[1])


def _pack16_guard(weight):
    """(generation holder, generation now) of the in-place bf16 packs `weight` takes part in, or None."""
    flat = _flat_of(weight)
    holder = getattr(flat, "_mvd_pack16_gen", None) if flat is not None else None
    return (holder, holder[0]) if holder is not None else None


def _check_pack16_generation(saved, what):
    if saved is not None and saved[0][0] != saved[1]:
        raise RuntimeError(f"{what}: the weights were updated (optimizer.step()) between the forward and the backward pass "
                           "of this graph; the bf16 packed copies saved for backward were rewritten in place")


def packs_stale(fp):
    """True when a cached pack of one of `fp`'s weights no longer carries the current stamp (a torch-visible write, e.g.
    load_state_dict, or invalidate_packs() since the last repack): a captured step, whose forward reads the persistent
    pack buffers, must not be replayed before repack_all(fp) has refreshed them."""
    for w, e in _live(_PACK_LIVE, "_mvd_pack", fp):
        if e.stamp != _pack_stamp(w.detach(), w):
            return True
    for w, e in _live(_PACK16_LIVE, "_mvd_pack16", fp):
        if e[0][0] != _pack_stamp(w.detach(), w):
            return True
    return False


def repack_all(fp=None):
    """Called by the fused optimizer after its update with its optim.FlatParams: new epoch of that flat buffer, every
    live cached weight that is a view of it re-packed in one launch.  Without `fp`: every live weight of the process
    (new global epoch)."""
    if fp is None:
        _PACK_EPOCH[0] += 1
    else:
        fp.flat._mvd_epoch[0] += 1
    _repack_all_bf16(fp)
    jobs = []
    for w, e in _live(_PACK_LIVE, "_mvd_pack", fp):
        d = w.detach()
        if d.is_cuda and d.dtype == torch.float32 and d.is_contiguous() and d.device == e.wf.device:
            jobs.append((d, e, w))
    if not jobs:
        return
    if any(e.uf is not None or e.ub is not None for _, e, _w in jobs) and query("mvd_wino_mode") != 2:
        return  # the batch entry only writes the F(2x2,3x3) layout; stale entries are packed per layer
    n = len(jobs)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    ptr = lambda t: t.data_ptr() if t is not None else None
    w_ = PA(*[ptr(d) for d, _, _w in jobs])
    wf_ = PA(*[ptr(e.wf) for _, e, _w in jobs])
    wb_ = PA(*[ptr(e.wb) for _, e, _w in jobs])
    uf_ = PA(*[ptr(e.uf) for _, e, _w in jobs])
    ub_ = PA(*[ptr(e.ub) for _, e, _w in jobs])
    K_ = IA(*[e.K for _, e, _w in jobs])
    C_ = IA(*[e.C for _, e, _w in jobs])
    T_ = IA(*[e.T for _, e, _w in jobs])
    tr_ = IA(*[1 if e.transposed else 0 for _, e, _w in jobs])
    cast = lambda a: ctypes.cast(a, ctypes.c_void_p)
    call("mvd_pack_weights_batch", n, cast(w_), cast(wf_), cast(wb_), cast(uf_), cast(ub_), cast(K_), cast(C_), cast(T_),
         cast(tr_), _stream())
    for d, e, w in jobs:
        e.stamp = _pack_stamp(d, w)
        e.gen += 1


def pack_weight_bf16(weight, transposed):
    """fp32 master weight -> bf16 (wf16, wb16) in the MFMA 32x32x16 operand layout (mvd_pack_weight_bf16)."""
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    if transposed:
        C, K = w.shape[:2]
    else:
        K, C = w.shape[:2]
    T = w[0, 0].numel()
    wf = torch.empty((T * C * K,), dtype=BF16, device=w.device)
    wb = torch.empty((T * C * K,), dtype=BF16, device=w.device)
    call("mvd_pack_weight_bf16", _p(w), _p(wf), _p(wb), K, C, T, 1 if transposed else 0, _stream())
    return wf, wb


def _packed_bf16(weight, transposed):
    """bf16 packs cached on the weight tensor under the same (epoch, version) stamp as the fp32 ones: inference packs
    once; in training the fused optimizer re-packs every registered weight in one launch after its update
    (repack_all -> mvd_pack_weights_bf16_batch); a stale or missing entry is packed here, per layer."""
    w = weight.detach()
    if not w.is_contiguous():
        return pack_weight_bf16(weight, transposed)
    e = getattr(weight, "_mvd_pack16", None)
    if e is None or e[0] != (_pack_stamp(w, weight), transposed, w.device):
        if e is None:
            if len(_PACK16_LIVE) >= 4096:
                _PACK16_LIVE[:] = [r for r in _PACK16_LIVE if r() is not None]
            _PACK16_LIVE.append(weakref.ref(weight))
        e = ((_pack_stamp(w, weight), transposed, w.device),) + tuple(pack_weight_bf16(weight, transposed))
        weight._mvd_pack16 = e
    return e[1], e[2]


def _out_dim(i, k, s):
    return (i + 2 * ((k - 1) // 2) - k) // s + 1


# ======================================================================================================== gradient sharing
class _GradShare:
    """A tensor with two consumers gets two gradient contributions, which autograd adds with a torch elementwise kernel
    (a full read-read-write pass over the activation: 0.27 ms of the bf16 step, 0.48 ms of the fp32 step at configs[1]).
    share_grad() tags such a tensor; the consumer whose backward runs FIRST writes its contribution into a fresh buffer and
    publishes it here, the one that runs SECOND adds its contribution into that buffer inside its own kernel (epilogue
    read-modify-write) and returns None to autograd.  Autograd calls the producer's backward only after both consumers
    have run, so the buffer is complete when it is read.  A consumer without an accumulating kernel simply returns its own
    tensor (autograd adds, as before): any order and any mix stays correct."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


def share_grad(t):
    if t.requires_grad:
        t._mvd_gshare = _GradShare()
    return t


def _share_of(t):
    return getattr(t, "_mvd_gshare", None)


def _publish(share, dx):
    """first contribution: remember the buffer (or drop a stale one when this consumer cannot accumulate)"""
    if share is not None:
        share.buf = dx
    return dx


def _joinable(share, shape, dtype):
    b = share.buf if share is not None else None
    return b if (b is not None and tuple(b.shape) == tuple(shape) and b.dtype == dtype and _is_cl3d(b)) else None


# ======================================================================================================== conv
class LaunchTimer:
    """bench.py: HIP events (on the stream the kernels run on) around the forward launches of ONE conv layer shape while
    the timed train steps run -- the in-step duration of the roofline kernel.  key = (bf16, N, C1, C2, K, D, H, W, kernel,
    stride); inactive unless `on`."""

    def __init__(self, key):
        self.key, self.pairs, self.on = tuple(key), [], False

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / len(self.pairs) if self.pairs else None


LAUNCH_TIMER = None


BF16_CONV_STATS = [os.environ.get("MVD_BF16_CONV_STATS", "1") != "0"]


def conv3d_fwd_bf16(x1, C1, x2, C2, wf, bias, y, N, D, H, W, K, ks, stride, ws, in_scale=None, in_shift=None, slope=0.01):
    """mvd_conv3d_fwd_bf16, through the fused entry when the z-marching kernel takes the shape: the InstanceNorm
    statistics of the output then come out of the conv's epilogue (attached to `y` as `_mvd_tile_stats16`, picked up by
    InstanceNormLeakyReLUFn / NormActConv3dFn) and, with in_scale / in_shift, the input is normalised + activated in the
    loader (mvd_conv3d_fwd_bf16_fused)."""
    nt = query("mvd_conv3d_fwd_bf16_stats_tiles", N, D, H, W, C1, C2, K, i3(ks), i3(stride)) \
        if (BF16_CONV_STATS[0] or in_scale is not None) else 0
    if nt <= 0 and in_scale is None:
        call("mvd_conv3d_fwd_bf16", _p(x1), C1, _p(x2), C2, _p(wf), _p(bias), _p(y), N, D, H, W, K, i3(ks), i3(stride),
             _p(ws), ws.numel(), _stream())
        return
    stats = torch.empty((N, nt, K, 2), dtype=torch.float32, device=y.device) if (nt > 0 and BF16_CONV_STATS[0]) else None
    got = ctypes.c_int(0)
    call("mvd_conv3d_fwd_bf16_fused", _p(x1), C1, _p(x2), C2, _p(wf), _p(bias), _p(y), N, D, H, W, K, i3(ks), i3(stride),
         _p(in_scale), _p(in_shift), float(slope), _p(stats), ctypes.byref(got), _p(ws), ws.numel(), _stream())
    if stats is not None and got.value > 0:
        y._mvd_tile_stats16 = (stats, int(got.value))


class Conv3dFn(Function):
    """Conv3d(k in {1,3}, pad=(k-1)/2, stride in {1,2}) over the channel concat of x1 and (optional) x2.
    Replaces nn.Conv3d of ConvDropoutNormReLU and torch.cat((x, skip), 1) (UNetDecoder.py:107)."""

    @staticmethod
    def forward(ctx, x1, x2, weight, bias, stride):
        _require_cuda(x1, x2, weight, bias)
        ctx.share1, ctx.share2 = _share_of(x1), (_share_of(x2) if x2 is not None else None)
        x1 = to_ndhwc(x1)
        x2 = to_ndhwc(x2) if x2 is not None else None
        K, C = weight.shape[:2]
        ks = tuple(weight.shape[2:])
        N, C1, D, H, W = x1.shape
        C2 = x2.shape[1] if x2 is not None else 0
        if C1 + C2 != C:
            raise RuntimeError(f"conv3d: weight expects {C} input channels, got {C1}+{C2}")
        if x2 is not None and tuple(x2.shape[2:]) != (D, H, W):
            raise RuntimeError("conv3d: the two concatenated inputs differ in spatial size")
        bf = _is_bf16(x1)
        if x2 is not None and _is_bf16(x2) != bf:
            raise RuntimeError("conv3d: the two concatenated inputs differ in dtype")
        od = [_out_dim(i, k, s) for i, k, s in zip((D, H, W), ks, stride)]
        y = empty_cl3d((N, K, *od), x1.device, x1.dtype)
        ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, od[0] * od[1] * od[2], K), x1.device)
        ub = None
        tm = LAUNCH_TIMER
        timed = tm is not None and tm.on and tm.key == (bf, N, C1, C2, K, D, H, W, tuple(ks), tuple(stride))
        if timed:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        if bf:
            wf, wb = _packed_bf16(weight, False)
            conv3d_fwd_bf16(x1, C1, x2, C2, wf, bias, y, N, D, H, W, K, ks, stride, ws)
        else:
            # fp32 3x3x3 stride-1 layers with enough tiles run the Winograd kernels (4/9 of the MFMA work); the packed
            # weights come from the per-weight cache (re-packed once per optimizer step, see repack_all)
            wino = query("mvd_conv_wino_applicable", N, D, H, W, C1, C2, K, i3(ks), i3(stride)) if len(ks) == 3 else 0
            pk = _packed(weight, False, bool(wino & 1), bool(wino & 2))
            wf, wb = pk.wf, pk.wb
            uf = pk.uf if wino & 1 else None
            ub = pk.ub if wino & 2 else None
            if uf is not None:
                # the Winograd kernel also emits the per-tile (sum, sum of squares) of its output: the InstanceNorm that
                # follows (InstanceNormLeakyReLUFn picks them up from the tensor) skips its statistics pass
                ntiles = query("mvd_conv_stats_tiles", od[0], od[1], od[2])
                stats = torch.empty((N, ntiles, K, 2), dtype=torch.float32, device=x1.device)
                done = ctypes.c_int(0)
                call("mvd_conv3d_fwd_wino_stats", _p(x1), C1, _p(x2), C2, _p(wf), _p(uf), _p(bias), _p(y), _p(stats),
                     ctypes.byref(done), N, D, H, W, K, i3(ks), i3(stride), _p(ws), ws.numel(), _stream())
                if done.value:
                    y._mvd_tile_stats = (stats, ntiles)
            else:
                call("mvd_conv3d_fwd_wino", _p(x1), C1, _p(x2), C2, _p(wf), _p(uf), _p(bias), _p(y), N, D, H, W, K, i3(ks),
                     i3(stride), _p(ws), ws.numel(), _stream())
        if timed:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            tm.pairs.append((ev0, ev1))
        ctx.bf = bf
        ctx.params = (weight, bias)
        ctx.pack = None if bf else (pk, pk.gen)
        ctx.pack16_gen = _pack16_guard(weight) if bf else None
        ctx.save_for_backward(x1, x2, wb, ub)
        ctx.geom = (N, C1, C2, D, H, W, K, ks, tuple(stride), tuple(od), bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x1, x2, wb, ub = ctx.saved_tensors
        _check_pack_generation(ctx.pack, "conv3d backward")  # (the bf16 packs are fresh tensors per stamp, or guarded:)
        _check_pack16_generation(ctx.pack16_gen, "conv3d backward")
        N, C1, C2, D, H, W, K, ks, stride, od, has_bias = ctx.geom
        dy = to_ndhwc(dy)
        dev = dy.device
        sfx = "_bf16" if ctx.bf else ""
        dx1 = dx2 = dw = db = None
        need1, need2 = ctx.needs_input_grad[0], (x2 is not None and ctx.needs_input_grad[1])
        joined = None
        if need1 and x2 is None and len(ks) == 3:
            # x1 has a second consumer whose backward already ran (a skip connection: the decoder conv read it through its
            # second pointer): add this conv's input gradient into that buffer inside the kernel
            joined = _joinable(ctx.share1, (N, C1, D, H, W), dy.dtype)
            if joined is not None and not query("mvd_conv3d_dgrad_acc_ok", int(ctx.bf), N, D, H, W, C1, K, i3(ks), i3(stride)):
                joined = None
        if joined is not None:
            if ctx.bf:
                ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, D * H * W, C1), dev)
                call("mvd_conv3d_dgrad_bf16_acc", _p(dy), _p(wb), _p(joined), C1, N, D, H, W, K, i3(ks), i3(stride), _p(ws),
                     ws.numel(), _stream())
            else:
                call("mvd_conv3d_dgrad_acc", _p(dy), _p(wb), _p(joined), C1, N, D, H, W, K, i3(ks), i3(stride), _stream())
            ctx.share1.buf = None   # consumed
            need1 = False           # (None to autograd: the contribution already sits in the first consumer's tensor)
        elif need1 or need2:
            dx1 = empty_cl3d((N, C1, D, H, W), dev, dy.dtype)
            dx2 = empty_cl3d((N, C2, D, H, W), dev, dy.dtype) if x2 is not None else None
            ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, D * H * W, C1 + C2), dev)
            if ctx.bf:
                call("mvd_conv3d_dgrad_bf16", _p(dy), _p(wb), _p(dx1), C1, _p(dx2), C2, N, D, H, W, K, i3(ks), i3(stride),
                     _p(ws), ws.numel(), _stream())
            else:
                call("mvd_conv3d_dgrad_wino", _p(dy), _p(wb), _p(ub), _p(dx1), C1, _p(dx2), C2, N, D, H, W, K, i3(ks),
                     i3(stride), _p(ws), ws.numel(), _stream())
            if need1:
                _publish(ctx.share1, dx1)
            if need2:
                _publish(ctx.share2, dx2)
        if ctx.needs_input_grad[2]:
            T = ks[0] * ks[1] * ks[2]
            weight, bias = ctx.params
            sink_w, sink_b = _take_grad(weight), (_take_grad(bias) if has_bias else None)
            dw = sink_w if sink_w is not None else torch.empty((K, C1 + C2, *ks), dtype=torch.float32, device=dev)
            db = (sink_b if sink_b is not None else torch.empty((K,), dtype=torch.float32, device=dev)) if has_bias else None
            nb = query("mvd_conv3d_wgrad_workspace_bytes", C1 + C2, K, T, N, *od)
            ws = _Workspace.get(nb, dev)
            call("mvd_conv3d_wgrad" + sfx, _p(x1), C1, _p(x2), C2, _p(dy), _p(dw), _p(db), N, D, H, W, K, i3(ks), i3(stride),
                 _p(ws), ws.numel(), _stream())
            if sink_w is not None:
                dw = None
                _grad_done(weight)
            if sink_b is not None:
                db = None
                _grad_done(bias)
        return (dx1 if need1 else None), (dx2 if need2 else None), dw, db, None


def _padded_pack_bf16(weight, cpad):
    """bf16 packs of a [K][C][3][3][3] weight zero-padded to cpad reduce channels, cached under the usual stamp."""
    w = weight.detach()
    e = getattr(weight, "_mvd_pack16pad", None)
    key = (_pack_stamp(w, weight), cpad, w.device)
    if e is None or e[0] != key:
        K, C = w.shape[:2]
        wp = torch.zeros((K, cpad, *w.shape[2:]), dtype=torch.float32, device=w.device)
        wp[:, :C] = w
        e = (key,) + tuple(pack_weight_bf16(wp, False))
        weight._mvd_pack16pad = e
    return e[1], e[2]


class NarrowInputConv3dBf16Fn(Function):
    """The network's input conv (4 modalities -> 32 channels, 3x3x3, stride 1) under bf16 mixed precision: the fp32 input
    is converted to bf16 NDHWC with the channels zero-padded to 32 (mvd_pad_channels_bf16) and the conv runs on the
    32-channel bf16 MFMA engines -- bf16 operands, fp32 accumulation, bf16 output, i.e. what the reference's autocast does
    to this layer (nnUNetTrainer.py:906) -- instead of the fp32 direct kernels (0.42 + 0.43 ms per step for a layer whose
    traffic is 0.34 GB).  The weight gradient of the zero channels is computed and dropped; the input needs no gradient."""
    CPAD = 32

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_cuda(x, weight, bias)
        if x.dtype != torch.float32:
            raise RuntimeError("narrow-input bf16 conv: the network input must be fp32")
        if ctx.needs_input_grad[0]:
            # (the backward of this layer has no input-gradient kernel; say so instead of returning None)
            raise RuntimeError("narrow-input bf16 conv: the gradient with respect to the network input is not on the path "
                               "(detach the input, or run the layer in fp32)")
        K, C = weight.shape[:2]
        if tuple(weight.shape[2:]) != (3, 3, 3) or K % 32 != 0 or C > 8 or x.shape[1] != C:
            raise RuntimeError("narrow-input bf16 conv: needs a 3x3x3 conv with <= 8 input and a multiple of 32 output channels")
        N, _, D, H, W = x.shape
        cl = _is_cl3d(x)
        if not cl and not x.is_contiguous():
            x = x.contiguous()
        cp = NarrowInputConv3dBf16Fn.CPAD
        xp = empty_cl3d((N, cp, D, H, W), x.device, BF16)
        call("mvd_pad_channels_bf16", _p(x), _p(xp), N, C, cp, D * H * W, 1 if cl else 0, _stream())
        wf, _wb = _padded_pack_bf16(weight, cp)
        y = empty_cl3d((N, K, D, H, W), x.device, BF16)
        ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, D * H * W, K), x.device)
        conv3d_fwd_bf16(xp, cp, None, 0, wf, bias, y, N, D, H, W, K, (3, 3, 3), (1, 1, 1), ws)
        ctx.save_for_backward(xp)
        ctx.params = (weight, bias)
        ctx.geom = (N, C, D, H, W, K)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (xp,) = ctx.saved_tensors
        N, C, D, H, W, K = ctx.geom
        weight, bias = ctx.params
        dy = to_ndhwc(dy)
        dev = dy.device
        cp = NarrowInputConv3dBf16Fn.CPAD
        dw = db = None
        if ctx.needs_input_grad[1]:
            sink_w = _take_grad(weight)
            sink_b = _take_grad(bias) if bias is not None else None
            dwp = torch.empty((K, cp, 3, 3, 3), dtype=torch.float32, device=dev)
            db = (sink_b if sink_b is not None else torch.empty((K,), dtype=torch.float32, device=dev)) if bias is not None else None
            ws = _Workspace.get(query("mvd_conv3d_wgrad_workspace_bytes", cp, K, 27, N, D, H, W), dev)
            call("mvd_conv3d_wgrad_bf16", _p(xp), cp, None, 0, _p(dy), _p(dwp), _p(db), N, D, H, W, K, i3((3, 3, 3)),
                 i3((1, 1, 1)), _p(ws), ws.numel(), _stream())
            if sink_w is not None:
                sink_w.copy_(dwp[:, :C])
                _grad_done(weight)
            else:
                dw = dwp[:, :C].contiguous()
            if sink_b is not None:
                db = None
                _grad_done(bias)
        return None, dw, db


class ConvTranspose3dFn(Function):
    """ConvTranspose3d with kernel == stride (UNetDecoder.py:56-59)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride):
        _require_cuda(x, weight, bias)
        ctx.share = _share_of(x)
        x = to_ndhwc(x)
        C, K = weight.shape[:2]
        if tuple(weight.shape[2:]) != tuple(stride):
            raise RuntimeError("convT3d: only kernel_size == stride is on the hot path")
        N, Cx, D, H, W = x.shape
        if Cx != C:
            raise RuntimeError("convT3d: channel mismatch")
        bf = _is_bf16(x)
        if bf:
            wf, wb = _packed_bf16(weight, True)
        else:
            pk = _packed(weight, True)
            wf, wb = pk.wf, pk.wb
        y = empty_cl3d((N, K, D * stride[0], H * stride[1], W * stride[2]), x.device, x.dtype)
        ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, D * H * W, K), x.device)
        ctx.bf = bf
        ctx.params = (weight, bias)
        ctx.pack = None if bf else (pk, pk.gen)
        ctx.pack16_gen = _pack16_guard(weight) if bf else None
        call("mvd_convT3d_fwd_bf16" if bf else "mvd_convT3d_fwd", _p(x), _p(wf), _p(bias), _p(y), N, D, H, W, C, K, i3(stride), _p(ws), ws.numel(),
             _stream())
        ctx.save_for_backward(x, wb)
        ctx.geom = (N, C, K, D, H, W, tuple(stride), bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        _check_pack_generation(ctx.pack, "convT3d backward")
        _check_pack16_generation(ctx.pack16_gen, "convT3d backward")
        N, C, K, D, H, W, stride, has_bias = ctx.geom
        dy = to_ndhwc(dy)
        dev = dy.device
        dx = dw = db = None
        sfx = "_bf16" if ctx.bf else ""
        if ctx.needs_input_grad[0]:
            dx = empty_cl3d((N, C, D, H, W), dev, dy.dtype)
            ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, D * H * W, C), dev)
            call("mvd_convT3d_dgrad" + sfx, _p(dy), _p(wb), _p(dx), N, D, H, W, C, K, i3(stride), _p(ws), ws.numel(),
                 _stream())
            _publish(ctx.share, dx)   # (no accumulating form of this kernel: always the first or a separate contribution)
        if ctx.needs_input_grad[1]:
            T = stride[0] * stride[1] * stride[2]
            weight, bias = ctx.params
            sink_w, sink_b = _take_grad(weight), (_take_grad(bias) if has_bias else None)
            dw = sink_w if sink_w is not None else torch.empty((C, K, *stride), dtype=torch.float32, device=dev)
            db = (sink_b if sink_b is not None else torch.empty((K,), dtype=torch.float32, device=dev)) if has_bias else None
            nb = query("mvd_convT3d_wgrad_workspace_bytes", C, K, T, N, D, H, W)
            ws = _Workspace.get(nb, dev)
            call("mvd_convT3d_wgrad" + sfx, _p(x), _p(dy), _p(dw), _p(db), N, D, H, W, C, K, i3(stride), _p(ws), ws.numel(),
                 _stream())
            if sink_w is not None:
                dw = None
                _grad_done(weight)
            if sink_b is not None:
                db = None
                _grad_done(bias)
        return dx, dw, db, None


# ======================================================================================================== norm
class InstanceNormLeakyReLUFn(Function):
    """InstanceNorm3d(eps, affine) + LeakyReLU(slope) fused (get_network_from_plans.py:41-44)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, slope, out_bf16=False):
        _require_cuda(x, gamma, beta)
        x = to_ndhwc(x)
        N, C = x.shape[:2]
        V = x[0, 0].numel()
        xb = _is_bf16(x)
        yb = bool(out_bf16) or xb  # a bf16 input never widens again inside the network
        y = empty_cl3d(x.shape, x.device, BF16 if yb else torch.float32)
        mean = torch.empty((N, C), dtype=torch.float32, device=x.device)
        rstd = torch.empty((N, C), dtype=torch.float32, device=x.device)
        nb = query("mvd_instnorm_workspace_bytes", N, V, C)
        ws = _Workspace.get(nb, x.device)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        pre = getattr(x, '_mvd_tile_stats', None)
        pre16 = getattr(x, '_mvd_tile_stats16', None) if xb else None
        if pre16 is not None and pre16[0].shape[0] == N and pre16[0].shape[2] == C and C % 4 == 0:
            # the bf16 conv's epilogue delivered the statistics: one finalize launch, then the apply pass in the
            # scale / shift form (the arithmetic of the fused loader prologue, mvd_conv3d_fwd_bf16_fused)
            scale, shift = torch.empty_like(mean), torch.empty_like(mean)
            call("mvd_instnorm_finalize_tiles", _p(pre16[0]), pre16[1], _p(g), _p(b), _p(mean), _p(rstd), _p(scale),
                 _p(shift), N, V, C, float(eps), _stream())
            call("mvd_instnorm_lrelu_apply_bf16", _p(x), _p(scale), _p(shift), _p(y), N, V, C, float(slope), _stream())
        elif yb:
            call("mvd_instnorm_lrelu_fwd_bf16", _p(x), int(xb), _p(g), _p(b), _p(y), _p(mean), _p(rstd), N, V, C,
                 float(eps), float(slope), _p(ws), ws.numel(), _stream())
        elif pre is not None and pre[0].shape[0] == N and pre[0].shape[2] == C:
            call("mvd_instnorm_lrelu_fwd_prestats", _p(x), _p(pre[0]), pre[1], _p(g), _p(b), _p(y), _p(mean), _p(rstd), N, V,
                 C, float(eps), float(slope), _p(ws), ws.numel(), _stream())
        else:
            call("mvd_instnorm_lrelu_fwd", _p(x), _p(g), _p(b), _p(y), _p(mean), _p(rstd), N, V, C, float(eps),
                 float(slope), _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(x, g, b, mean, rstd)
        ctx.slope = float(slope)
        ctx.yb = yb
        ctx.params = (gamma, beta)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        x, g, b, mean, rstd = ctx.saved_tensors
        dy = to_ndhwc(dy)
        N, C = x.shape[:2]
        V = x[0, 0].numel()
        dx = empty_cl3d(x.shape, x.device, x.dtype)
        gamma, beta = ctx.params
        sink_g = _take_grad(gamma) if ctx.needs_input_grad[1] else None
        sink_b = _take_grad(beta) if ctx.needs_input_grad[2] else None
        dg = sink_g if sink_g is not None else torch.empty((C,), dtype=torch.float32, device=x.device)
        db = sink_b if sink_b is not None else torch.empty((C,), dtype=torch.float32, device=x.device)
        nb = query("mvd_instnorm_workspace_bytes", N, V, C)
        ws = _Workspace.get(nb, x.device)
        if ctx.yb:
            call("mvd_instnorm_lrelu_bwd_bf16", _p(x), int(_is_bf16(x)), _p(dy), _p(g), _p(b), _p(mean), _p(rstd), _p(dx),
                 _p(dg), _p(db), N, V, C, ctx.slope, _p(ws), ws.numel(), _stream())
        else:
            call("mvd_instnorm_lrelu_bwd", _p(x), _p(dy), _p(g), _p(b), _p(mean), _p(rstd), _p(dx), _p(dg), _p(db), N, V,
                 C, ctx.slope, _p(ws), ws.numel(), _stream())
        if sink_g is not None:
            dg = None
            _grad_done(gamma)
        if sink_b is not None:
            db = None
            _grad_done(beta)
        return dx, dg, db, None, None, None


# NormActConv3dFn.backward: take the weight gradient through the loader prologue of k_wgrad16z (0: re-materialise the
# activated tensor with one apply pass first -- the A/B and cross-check switch)
WGRAD_PROLOGUE = os.environ.get("MVD_WGRAD_PROLOGUE", "1") != "0"


class NormActConv3dFn(Function):
    """The fused block boundary of the north_star (get_network_from_plans.py:41-44, bf16 mixed precision): InstanceNorm3d(affine)
    + LeakyReLU of block k folded into the LOADER of block k+1's Conv3d 3x3x3 (mvd_conv3d_fwd_bf16_fused: z-marching
    kernel, 32 -> 32 channels).  Input `y0` is the RAW bf16 output of block k's conv carrying the statistics its epilogue
    emitted (`_mvd_tile_stats16`); the activated tensor a0 = lrelu(IN(y0)) is never written in the forward pass.
    Backward: dgrad -> d a0; the InstanceNorm backward kernels on (y0, d a0) -> d y0, d gamma, d beta; the weight gradient
    needs a0 as an operand: mvd_conv3d_wgrad_bf16_fused re-computes it from y0 in ITS loader (k_wgrad16z, same arithmetic:
    bit-identical to the gradient over the materialised tensor), so a0 exists in neither pass.  Shapes that kernel does not
    serve fall back to one apply pass in front of the plain weight-gradient call."""

    @staticmethod
    def forward(ctx, y0, gamma, beta, eps, slope, weight, bias):
        _require_cuda(y0, gamma, beta, weight, bias)
        pre = getattr(y0, '_mvd_tile_stats16', None)
        if not (_is_bf16(y0) and _is_cl3d(y0)):
            raise RuntimeError("NormActConv3dFn: needs the raw bf16 NDHWC conv output")
        N, C, D, H, W = y0.shape
        K = weight.shape[0]
        V = D * H * W
        if tuple(weight.shape[1:]) != (C, 3, 3, 3):
            raise RuntimeError("NormActConv3dFn: 3x3x3 conv over the normalised tensor's channels")
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        mean = torch.empty((N, C), dtype=torch.float32, device=y0.device)
        rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
        tm = LAUNCH_TIMER   # bench.py: the fused block (finalize + conv) inside the step
        timed = tm is not None and tm.on and tm.key == (True, N, C, 0, K, D, H, W, (3, 3, 3), (1, 1, 1))
        if timed:
            ev0 = torch.cuda.Event(enable_timing=True)
            ev0.record()
        if pre is not None and pre[0].shape[0] == N and pre[0].shape[2] == C:
            call("mvd_instnorm_finalize_tiles", _p(pre[0]), pre[1], _p(g), _p(b), _p(mean), _p(rstd), _p(scale), _p(shift), N,
                 V, C, float(eps), _stream())
        else:   # the producing conv ran on a kernel without the statistics epilogue: one pass over y0
            ws = _Workspace.get(query("mvd_instnorm_workspace_bytes", N, V, C), y0.device)
            call("mvd_instnorm_stats_bf16", _p(y0), 1, _p(g), _p(b), _p(mean), _p(rstd), _p(scale), _p(shift), N, V, C,
                 float(eps), _p(ws), ws.numel(), _stream())
        wf, wb = _packed_bf16(weight, False)
        y1 = empty_cl3d((N, K, D, H, W), y0.device, BF16)
        ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, V, K), y0.device)
        conv3d_fwd_bf16(y0, C, None, 0, wf, bias, y1, N, D, H, W, K, (3, 3, 3), (1, 1, 1), ws, scale, shift, slope)
        if timed:
            ev1 = torch.cuda.Event(enable_timing=True)
            ev1.record()
            tm.pairs.append((ev0, ev1))
        ctx.save_for_backward(y0, g, b, mean, rstd, scale, shift, wb)
        ctx.params = (gamma, beta, weight, bias)
        ctx.slope = float(slope)
        ctx.pack16_gen = _pack16_guard(weight)
        return y1

    @staticmethod
    @once_differentiable
    def backward(ctx, dy1):
        y0, g, b, mean, rstd, scale, shift, wb = ctx.saved_tensors
        _check_pack16_generation(ctx.pack16_gen, "norm+act+conv3d backward")
        gamma, beta, weight, bias = ctx.params
        N, C, D, H, W = y0.shape
        K = weight.shape[0]
        V = D * H * W
        dev = y0.device
        dy1 = to_ndhwc(dy1)
        ks, st = i3((3, 3, 3)), i3((1, 1, 1))
        dy0 = dg = db_ = dw = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            da0 = empty_cl3d((N, C, D, H, W), dev, BF16)
            ws = _Workspace.get(query("mvd_conv_fwd_workspace_bytes", N, V, C), dev)
            call("mvd_conv3d_dgrad_bf16", _p(dy1), _p(wb), _p(da0), C, None, 0, N, D, H, W, K, ks, st, _p(ws), ws.numel(),
                 _stream())
            dy0 = empty_cl3d(y0.shape, dev, BF16)
            sink_g, sink_b = _take_grad(gamma), _take_grad(beta)
            dg = sink_g if sink_g is not None else torch.empty((C,), dtype=torch.float32, device=dev)
            db_ = sink_b if sink_b is not None else torch.empty((C,), dtype=torch.float32, device=dev)
            ws = _Workspace.get(query("mvd_instnorm_workspace_bytes", N, V, C), dev)
            call("mvd_instnorm_lrelu_bwd_bf16", _p(y0), 1, _p(da0), _p(g), _p(b), _p(mean), _p(rstd), _p(dy0), _p(dg), _p(db_),
                 N, V, C, ctx.slope, _p(ws), ws.numel(), _stream())
            if sink_g is not None:
                dg = None
                _grad_done(gamma)
            if sink_b is not None:
                db_ = None
                _grad_done(beta)
        if ctx.needs_input_grad[5]:
            has_bias = bias is not None
            sink_w, sink_b2 = _take_grad(weight), (_take_grad(bias) if has_bias else None)
            dw = sink_w if sink_w is not None else torch.empty((K, C, 3, 3, 3), dtype=torch.float32, device=dev)
            db = (sink_b2 if sink_b2 is not None else torch.empty((K,), dtype=torch.float32, device=dev)) if has_bias else None
            ws = _Workspace.get(query("mvd_conv3d_wgrad_workspace_bytes", C, K, 27, N, D, H, W), dev)
            if WGRAD_PROLOGUE and query("mvd_conv3d_wgrad_bf16_prologue_ok", N, D, H, W, C, 0, K, ks, st) > 0:
                # the weight-gradient kernel applies the same InstanceNorm + LeakyReLU to y0 in its loader: the activated
                # tensor is never materialised, in neither pass
                call("mvd_conv3d_wgrad_bf16_fused", _p(y0), C, _p(dy1), _p(dw), _p(db), N, D, H, W, K, ks, st, _p(scale),
                     _p(shift), ctx.slope, _p(ws), ws.numel(), _stream())
            else:
                a0 = empty_cl3d(y0.shape, dev, BF16)  # the activated tensor, only now (and only for the weight gradient)
                call("mvd_instnorm_lrelu_apply_bf16", _p(y0), _p(scale), _p(shift), _p(a0), N, V, C, ctx.slope, _stream())
                call("mvd_conv3d_wgrad_bf16", _p(a0), C, None, 0, _p(dy1), _p(dw), _p(db), N, D, H, W, K, ks, st, _p(ws),
                     ws.numel(), _stream())
            if sink_w is not None:
                dw = None
                _grad_done(weight)
            if sink_b2 is not None:
                db = None
                _grad_done(bias)
        return dy0, dg, db_, None, None, dw, db


def fused_norm_conv_ok(y0, weight, stride):
    """True when `y0` (raw bf16 conv output) can feed the next 3x3x3 conv through the fused loader
    prologue (mvd_conv3d_fwd_bf16_fused: the z-marching kernel takes the shape)."""
    if not (_is_bf16(y0) and y0.dim() == 5 and _is_cl3d(y0)):
        return False
    N, C, D, H, W = y0.shape
    K = weight.shape[0]
    if tuple(weight.shape[1:]) != (C, 3, 3, 3) or tuple(stride) != (1, 1, 1):
        return False
    return C % 4 == 0 and query("mvd_conv3d_fwd_bf16_prologue_ok", N, D, H, W, C, 0, K, i3((3, 3, 3)), i3((1, 1, 1))) > 0


# ======================================================================================================== seg head
class SegHeadFn(Function):
    """1x1x1 Conv3d C -> K producing planar logits [N,K,D,H,W] (UNetDecoder.py:70,110)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        _require_cuda(x, weight, bias)
        ctx.share = _share_of(x)
        x = to_ndhwc(x)
        N, C = x.shape[:2]
        K = weight.shape[0]
        V = x[0, 0].numel()
        w = weight.detach().reshape(K, C).contiguous()
        logits = torch.empty((N, K, *x.shape[2:]), dtype=torch.float32, device=x.device)
        call("mvd_seghead_fwd_bf16" if _is_bf16(x) else "mvd_seghead_fwd", _p(x), _p(w), _p(bias), _p(logits), N, V, C, K,
             _stream())
        ctx.save_for_backward(x, w)
        ctx.wshape = tuple(weight.shape)
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        x, w = ctx.saved_tensors
        N, C = x.shape[:2]
        K = w.shape[0]
        V = x[0, 0].numel()
        dl = dl.contiguous()
        dev = x.device
        # the other consumer of x (the next level's transposed conv) has already written its gradient: add ours into it
        joined = _joinable(ctx.share, x.shape, x.dtype) if ctx.needs_input_grad[0] else None
        dx = joined if joined is not None else (empty_cl3d(x.shape, dev, x.dtype) if ctx.needs_input_grad[0] else None)
        dw = torch.empty((K, C), dtype=torch.float32, device=dev)
        db = torch.empty((K,), dtype=torch.float32, device=dev)
        nb = query("mvd_seghead_bwd_workspace_bytes", N, V, C, K)
        ws = _Workspace.get(nb, dev)
        call("mvd_seghead_bwd_bf16" if _is_bf16(x) else "mvd_seghead_bwd", _p(x), _p(w), _p(dl), _p(dx), _p(dw), _p(db), N, V, C, K,
             1 if joined is not None else 0, _p(ws), ws.numel(), _stream())
        if ctx.share is not None:
            ctx.share.buf = None if joined is not None else dx   # consumed / first contribution
        return (None if joined is not None else dx), dw.view(ctx.wshape), db


class NormActSegHeadFn(Function):
    """The LAST decoder block's InstanceNorm3d + LeakyReLU folded into the seg head that is its only consumer (UNetDecoder.py:110
    after get_network_from_plans.py:41-44): `y0` is the RAW output of the block's conv (with the statistics its epilogue
    emitted attached), the head kernels apply the normalisation + activation in their loaders (mvd_seghead_*_fused: bf16 in the
    scale / shift form of the bf16 apply pass, fp32 in the mean / rstd form of the fp32 one) -- the activated tensor of the
    top decoder stage is never written.  Backward: d a from the head's input-gradient kernel, the InstanceNorm backward on
    (y0, d a), dW / db over the re-computed a.  Bit-identical to the two separate nodes."""

    @staticmethod
    def forward(ctx, y0, gamma, beta, eps, slope, weight, bias):
        _require_cuda(y0, gamma, beta, weight, bias)
        bf = _is_bf16(y0)
        if not _is_cl3d(y0) or y0.dtype not in (BF16, torch.float32):
            raise RuntimeError("NormActSegHeadFn: needs the raw NDHWC conv output (bf16 or fp32)")
        N, C, D, H, W = y0.shape
        K = weight.shape[0]
        V = D * H * W
        dev = y0.device
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        mean = torch.empty((N, C), dtype=torch.float32, device=dev)
        rstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
        ws = _Workspace.get(query("mvd_instnorm_workspace_bytes", N, V, C), dev)
        pre16 = getattr(y0, '_mvd_tile_stats16', None) if bf else None
        pre = getattr(y0, '_mvd_tile_stats', None) if not bf else None
        if pre16 is not None and pre16[0].shape[0] == N and pre16[0].shape[2] == C:
            call("mvd_instnorm_finalize_tiles", _p(pre16[0]), pre16[1], _p(g), _p(b), _p(mean), _p(rstd), _p(scale), _p(shift), N,
                 V, C, float(eps), _stream())
        elif pre is not None and pre[0].shape[0] == N and pre[0].shape[2] == C:
            call("mvd_instnorm_stats_from_tiles", _p(pre[0]), pre[1], _p(mean), _p(rstd), N, V, C, float(eps), _p(ws), ws.numel(),
                 _stream())
        else:
            call("mvd_instnorm_stats_bf16", _p(y0), int(bf), _p(g), _p(b), _p(mean), _p(rstd), _p(scale), _p(shift), N, V, C,
                 float(eps), _p(ws), ws.numel(), _stream())
        w = weight.detach().reshape(K, C).contiguous()
        logits = torch.empty((N, K, D, H, W), dtype=torch.float32, device=dev)
        if bf:
            call("mvd_seghead_fwd_bf16_fused", _p(y0), _p(scale), _p(shift), float(slope), _p(w), _p(bias), _p(logits), N, V, C,
                 K, _stream())
        else:
            call("mvd_seghead_fwd_fused", _p(y0), _p(mean), _p(rstd), _p(g), _p(b), float(slope), _p(w), _p(bias), _p(logits), N,
                 V, C, K, _stream())
        ctx.save_for_backward(y0, g, b, mean, rstd, scale, shift, w)
        ctx.params = (gamma, beta)
        ctx.slope = float(slope)
        ctx.wshape = tuple(weight.shape)
        return logits

    @staticmethod
    @once_differentiable
    def backward(ctx, dl):
        y0, g, b, mean, rstd, scale, shift, w = ctx.saved_tensors
        gamma, beta = ctx.params
        bf = _is_bf16(y0)
        N, C, D, H, W = y0.shape
        K = w.shape[0]
        V = D * H * W
        dev = y0.device
        dl = dl.contiguous()
        need_x = ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        da = empty_cl3d(y0.shape, dev, y0.dtype) if need_x else None
        dw = torch.empty((K, C), dtype=torch.float32, device=dev)
        db = torch.empty((K,), dtype=torch.float32, device=dev)
        ws = _Workspace.get(query("mvd_seghead_bwd_workspace_bytes", N, V, C, K), dev)
        if bf:
            call("mvd_seghead_bwd_bf16_fused", _p(y0), _p(scale), _p(shift), ctx.slope, _p(w), _p(dl), _p(da), _p(dw), _p(db), N,
                 V, C, K, 0, _p(ws), ws.numel(), _stream())
        else:
            call("mvd_seghead_bwd_fused", _p(y0), _p(mean), _p(rstd), _p(g), _p(b), ctx.slope, _p(w), _p(dl), _p(da), _p(dw),
                 _p(db), N, V, C, K, 0, _p(ws), ws.numel(), _stream())
        dy0 = dg = db_ = None
        if need_x:
            dy0 = empty_cl3d(y0.shape, dev, y0.dtype)
            sink_g, sink_b = _take_grad(gamma), _take_grad(beta)
            dg = sink_g if sink_g is not None else torch.empty((C,), dtype=torch.float32, device=dev)
            db_ = sink_b if sink_b is not None else torch.empty((C,), dtype=torch.float32, device=dev)
            ws = _Workspace.get(query("mvd_instnorm_workspace_bytes", N, V, C), dev)
            if bf:
                call("mvd_instnorm_lrelu_bwd_bf16", _p(y0), 1, _p(da), _p(g), _p(b), _p(mean), _p(rstd), _p(dy0), _p(dg),
                     _p(db_), N, V, C, ctx.slope, _p(ws), ws.numel(), _stream())
            else:
                call("mvd_instnorm_lrelu_bwd", _p(y0), _p(da), _p(g), _p(b), _p(mean), _p(rstd), _p(dy0), _p(dg), _p(db_), N, V,
                     C, ctx.slope, _p(ws), ws.numel(), _stream())
            if sink_g is not None:
                dg = None
                _grad_done(gamma)
            if sink_b is not None:
                db_ = None
                _grad_done(beta)
        return dy0, dg, db_, None, None, dw.view(ctx.wshape), db


def fused_norm_seghead_ok(y0, weight):
    """True when the seg head can read the raw (bf16 or fp32) conv output `y0` through its fused loaders."""
    if not (y0.dtype in (BF16, torch.float32) and y0.dim() == 5 and _is_cl3d(y0) and y0.is_cuda):
        return False
    N, C = y0.shape[:2]
    return weight.shape[1] == C and query("mvd_seghead_bf16_fused_ok", N, y0[0, 0].numel(), C, weight.shape[0]) > 0


class CastFn(Function):
    """Precision boundary: bf16 activation -> fp32 (and the fp32 gradient back to bf16), layout preserved.  Used where
    a bf16 feature map feeds an fp32 loss kernel (the distillation features)."""

    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        if not _is_bf16(x):
            raise RuntimeError("CastFn widens bf16 tensors")
        x = to_ndhwc(x)
        y = torch.empty_like(x, dtype=torch.float32)  # preserves the NDHWC strides
        call("mvd_cast_bf16_to_f32", _p(x), _p(y), x.numel(), _stream())
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy = to_ndhwc(dy)
        dx = torch.empty_like(dy, dtype=BF16)
        call("mvd_cast_f32_to_bf16", _p(dy), _p(dx), dy.numel(), _stream())
        return dx


def widen(x):
    """fp32 view of an activation for the fp32 loss kernels (identity in the fp32 mode)."""
    return CastFn.apply(x) if _is_bf16(x) else x


# ======================================================================================================== losses
def _flat_target(t, N, V):
    """float label map [N,1,...] or [N,...] -> contiguous [N,V]."""
    t = t.reshape(N, -1)
    if t.shape[1] != V:
        raise RuntimeError(f"target has {t.shape[1]} voxels per sample, logits have {V}")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class DeepSupervisedDCCEFn(Function):
    """sum_i w_i * (w_ce*CE(x_i,t_i) + w_dice*(-softDice(softmax(x_i),t_i))) in one autograd node
    (DeepSupervisionWrapper o DC_and_CE_loss, nnUNetTrainer.py:359-374).  Levels with w_i == 0 contribute exact-zero
    gradients (upstream 2.1.1 evaluates them; see oracle/loss_oracle.py).  `gather` (optional) maps the per-sample
    stats [N,3K+1] to the all-gathered stats for DDP batch-dice (collective C2); it returns (stats_all, my_offset,
    grad_multiplier)."""

    @staticmethod
    def forward(ctx, weights, cfg, gather, targets, *logits):
        batch_dice, do_bg, smooth, w_ce, w_dice = cfg
        dev = logits[0].device
        total = torch.zeros((1,), dtype=torch.float32, device=dev)
        saved = []
        ctx.levels = []
        for i, (x, t) in enumerate(zip(logits, targets)):
            if weights[i] == 0:
                ctx.levels.append(None)
                continue
            _require_cuda(x, t)
            x = x.contiguous()
            N, K = x.shape[:2]
            V = x[0, 0].numel()
            t = _flat_target(t, N, V)
            stats = torch.empty((N, 3 * K + 1), dtype=torch.float32, device=dev)
            nb = query("mvd_dcce_workspace_bytes", N, V, K)
            ws = _Workspace.get(nb, dev)
            call("mvd_dcce_fwd", _p(x), _p(t), _p(stats), N, V, K, _p(ws), ws.numel(), _stream())
            dstats, off, mult = (stats, 0, 1.0)
            if gather is not None and batch_dice:
                dstats, off, mult = gather(stats)
            Nd = dstats.shape[0]
            loss = torch.empty((3,), dtype=torch.float32, device=dev)
            coef = torch.empty((Nd, K, 2), dtype=torch.float32, device=dev)
            call("mvd_dcce_finalize", _p(stats), N, _p(dstats), Nd, _p(loss), _p(coef), V, K, int(batch_dice), int(do_bg),
                 float(smooth), float(w_ce), float(w_dice), _stream())
            call("mvd_axpy", _p(total), _p(loss), float(weights[i]), 1, _stream())
            saved += [x, t, coef[off:off + N].contiguous() if (off or Nd != N) else coef]
            ctx.levels.append((N, K, V, float(weights[i]) , float(mult)))
        ctx.save_for_backward(*saved)
        ctx.w_ce = float(w_ce)
        ctx.shapes = [tuple(x.shape) for x in logits]
        return total.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        saved = ctx.saved_tensors
        g = g.contiguous()
        grads = []
        j = 0
        for lvl, shp in zip(ctx.levels, ctx.shapes):
            if lvl is None:
                grads.append(torch.zeros(shp, dtype=torch.float32, device=g.device))
                continue
            N, K, V, w, mult = lvl
            x, t, coef = saved[j:j + 3]
            j += 3
            if mult != 1.0:
                coef = coef * mult  # ddp batch-dice: AllGatherGrad.backward sums the (identical) grads of all ranks
            dl = torch.empty(shp, dtype=torch.float32, device=g.device)
            call("mvd_dcce_bwd", _p(x), _p(t), _p(coef), _p(g), w, _p(dl), N, V, K, ctx.w_ce, _stream())
            grads.append(dl)
        return (None, None, None, None, *grads)


def argmax_counts(logits, target):
    """validation_step online evaluation (nnUNetTrainer.py:973-990): returns int64 [K,3] = (tp, fp, fn) per class."""
    _require_cuda(logits, target)
    x = logits.contiguous()
    N, K = x.shape[:2]
    V = x[0, 0].numel()
    t = _flat_target(target, N, V)
    counts = torch.empty((K, 3), dtype=torch.int64, device=x.device)
    call("mvd_argmax_counts", _p(x), _p(t), _p(counts), N, V, K, _stream())
    return counts


def _kl_strides(t):
    """(N, C, V, sn, sc, sv) of a [N,C,*spatial] tensor whose spatial block is dense in either planar or NDHWC order."""
    N, C = t.shape[:2]
    V = t[0, 0].numel()
    if t.is_contiguous():
        return N, C, V, C * V, V, 1
    if _is_cl3d(t):
        return N, C, V, C * V, 1, C
    # channel slice of a planar tensor ([N,1,...] view with batch stride K*V)
    st = t.stride()
    sp = t[0, 0]
    if sp.is_contiguous():
        return N, C, V, st[0], st[1], 1
    raise RuntimeError("kl: unsupported memory layout")


class DistillKLFn(Function):
    """distill_kl / l2_loss(channel_wise=True) (other_loss.py:51-64, :67-76)."""

    @staticmethod
    def forward(ctx, ys, yt, T, eps_s, pad_zero_channel):
        _require_cuda(ys, yt)
        if ys.shape != yt.shape:
            raise RuntimeError("kl: shape mismatch")
        gs = _kl_strides(ys)
        if _kl_strides(yt) != gs:
            yt = yt.contiguous(memory_format=CL3D) if _is_cl3d(ys) else yt.contiguous()
            ys = ys if _kl_strides(ys) == _kl_strides(yt) else ys.contiguous()
            gs = _kl_strides(ys)
            if _kl_strides(yt) != gs:
                raise RuntimeError("kl: the two inputs must share a memory layout")
        N, C, V, sn, sc, sv = gs
        out = torch.empty((1,), dtype=torch.float32, device=ys.device)
        nb = query("mvd_kl_workspace_bytes", N, V)
        ws = _Workspace.get(nb, ys.device)
        if _is_bf16(ys) or _is_bf16(yt):
            # mixed precision: the bf16 NDHWC feature maps go into the kernel as they are (fp32 arithmetic inside)
            if not (_is_bf16(ys) and _is_bf16(yt) and _is_cl3d(ys) and (sc, sv, sn) == (1, C, V * C)) or pad_zero_channel:
                raise RuntimeError("kl: bf16 inputs must both be dense NDHWC feature maps")
            call("mvd_kl_fwd_bf16", _p(ys), _p(yt), _p(out), N, C, V, float(T), float(eps_s), _p(ws), ws.numel(), _stream())
            ctx.save_for_backward(ys, yt)
            ctx.cfg = (gs, float(T), float(eps_s), 0)
            return out.reshape(())
        call("mvd_kl_fwd", _p(ys), _p(yt), _p(out), N, C, V, sn, sc, sv, float(T), float(eps_s), int(pad_zero_channel),
             _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(ys, yt)
        ctx.cfg = (gs, float(T), float(eps_s), int(pad_zero_channel))
        return out.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        ys, yt = ctx.saved_tensors
        (N, C, V, sn, sc, sv), T, eps_s, pad = ctx.cfg
        g = g.contiguous()
        if _is_bf16(ys):
            gs = torch.empty_like(ys) if ctx.needs_input_grad[0] else None
            gt = torch.empty_like(yt) if ctx.needs_input_grad[1] else None
            call("mvd_kl_bwd_bf16", _p(ys), _p(yt), _p(g), 1.0, _p(gs), _p(gt), N, C, V, T, eps_s, _stream())
            return gs, gt, None, None, None
        # gradients are written with the inputs' strides into buffers of identical geometry
        def like(t):
            return torch.empty_strided(t.shape, t.stride(), dtype=torch.float32, device=t.device) \
                if (t.is_contiguous() or _is_cl3d(t)) else None
        gs = like(ys) if ctx.needs_input_grad[0] else None
        gt = like(yt) if ctx.needs_input_grad[1] else None
        dense = ys.is_contiguous() or _is_cl3d(ys)
        if not dense:
            # strided channel-slice view: write dense [N,C,V] grads (sn = C*V)
            if ctx.needs_input_grad[0]:
                gs = torch.empty(ys.shape, dtype=torch.float32, device=ys.device)
            if ctx.needs_input_grad[1]:
                gt = torch.empty(yt.shape, dtype=torch.float32, device=yt.device)
            ysd, ytd = ys.contiguous(), yt.contiguous()
            call("mvd_kl_bwd", _p(ysd), _p(ytd), _p(g), 1.0, _p(gs), _p(gt), N, C, V, C * V, V, 1, T, eps_s, pad, _stream())
        else:
            call("mvd_kl_bwd", _p(ys), _p(yt), _p(g), 1.0, _p(gs), _p(gt), N, C, V, sn, sc, sv, T, eps_s, pad, _stream())
        return gs, gt, None, None, None


class MseFn(Function):
    """l2_loss(channel_wise=False) = mean(|a - b|^2) (other_loss.py:77-78)."""

    @staticmethod
    def forward(ctx, a, b):
        _require_cuda(a, b)
        if a.shape != b.shape:
            raise RuntimeError("l2_loss: shape mismatch")
        if a.dtype != torch.float32 or b.dtype != torch.float32:
            raise RuntimeError("l2_loss: fp32 inputs (widen bf16 feature maps first)")
        # elementwise: both operands only have to share ONE dense layout (planar or NDHWC)
        if not (a.is_contiguous() or _is_cl3d(a)):
            a = a.contiguous()
        if a.stride() != b.stride():
            b = b.contiguous(memory_format=CL3D) if (_is_cl3d(a) and not a.is_contiguous()) else b.contiguous()
            if a.stride() != b.stride():
                a = a.contiguous()
                b = b.contiguous()
        n = a.numel()
        out = torch.empty((1,), dtype=torch.float32, device=a.device)
        ws = _Workspace.get(query("mvd_mse_workspace_bytes", n), a.device)
        call("mvd_mse_fwd", _p(a), _p(b), _p(out), n, _p(ws), ws.numel(), _stream())
        ctx.save_for_backward(a, b)
        return out.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        ga = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        gb = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        call("mvd_mse_bwd", _p(a), _p(b), _p(g), _p(ga), _p(gb), a.numel(), _stream())
        return ga, gb


class SoftmaxSelectFn(Function):
    """softmax(logits, 1)[:, sel:sel+1] on planar logits."""

    @staticmethod
    def forward(ctx, logits, sel):
        _require_cuda(logits)
        x = logits.contiguous()
        N, K = x.shape[:2]
        V = x[0, 0].numel()
        p = torch.empty((N, 1, *x.shape[2:]), dtype=torch.float32, device=x.device)
        call("mvd_softmax_select_fwd", _p(x), _p(p), N, V, K, int(sel), _stream())
        ctx.save_for_backward(x)
        ctx.sel = int(sel)
        return p

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        N, K = x.shape[:2]
        V = x[0, 0].numel()
        g = g.contiguous()
        dl = torch.empty_like(x)
        call("mvd_softmax_select_bwd", _p(x), _p(g), _p(dl), N, V, K, ctx.sel, _stream())
        return dl, None


def label_mask(labels, value):
    _require_cuda(labels)
    x = labels.contiguous().float()
    out = torch.empty_like(x)
    call("mvd_label_mask", _p(x), _p(out), x.numel(), float(value), _stream())
    return out


# ======================================================================================================== soft skeleton
def _vol(t):
    if t.dim() != 5:
        raise RuntimeError("soft-skeleton ops take [N,C,D,H,W] volumes")
    t = t.contiguous()
    N, C, D, H, W = t.shape
    return t, N * C, D, H, W


class SoftErodeFn(Function):
    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x, NC, D, H, W = _vol(x)
        y = torch.empty_like(x)
        code = torch.empty(x.shape, dtype=torch.int16, device=x.device) if x.requires_grad or True else None
        call("mvd_soft_erode_fwd", _p(x), _p(y), _p(code), NC, D, H, W, _stream())
        ctx.save_for_backward(code)
        ctx.dims = (NC, D, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (code,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call("mvd_soft_erode_bwd", _p(code), _p(dy), _p(dx), *ctx.dims, _stream())
        return dx


class SoftDilateFn(Function):
    @staticmethod
    def forward(ctx, x):
        _require_cuda(x)
        x, NC, D, H, W = _vol(x)
        y = torch.empty_like(x)
        code = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        call("mvd_soft_dilate_fwd", _p(x), _p(y), _p(code), NC, D, H, W, _stream())
        ctx.save_for_backward(code)
        ctx.dims = (NC, D, H, W)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (code,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        call("mvd_soft_dilate_bwd", _p(code), _p(dy), _p(dx), *ctx.dims, _stream())
        return dx


class SkelUpdateFn(Function):
    """init: skel = relu(img - opened); else skel = skel + relu(delta - skel*delta) (soft_skeleton.py:31,35-36)."""

    @staticmethod
    def forward(ctx, img, opened, skel):
        _require_cuda(img, opened, skel)
        img, opened = img.contiguous(), opened.contiguous()
        init = skel is None
        if not init:
            skel = skel.contiguous()
        out = torch.empty_like(img)
        call("mvd_skel_update_fwd", _p(img), _p(opened), _p(skel), _p(out), img.numel(), int(init), _stream())
        ctx.save_for_backward(img, opened, skel)
        ctx.init = init
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        img, opened, skel = ctx.saved_tensors
        g = g.contiguous()
        d_img, d_open = torch.empty_like(img), torch.empty_like(img)
        d_skel = None if ctx.init else torch.empty_like(img)
        call("mvd_skel_update_bwd", _p(img), _p(opened), _p(skel), _p(g), _p(d_img), _p(d_open), _p(d_skel), img.numel(),
             int(ctx.init), _stream())
        return d_img, d_open, d_skel


def _skel_iter(img, skel, want_codes):
    """mvd_skel_iter_fwd on contiguous volumes; returns (e1 | None, opened, skel_out, c_e1 | None, c_e2, c_o)."""
    NC, D, H, W = img.shape[0] * img.shape[1], *img.shape[2:]
    init = skel is None
    e1 = None if init else torch.empty_like(img)
    opened, skel_out = torch.empty_like(img), torch.empty_like(img)
    c_e1 = torch.empty(img.shape, dtype=torch.int16, device=img.device) if (want_codes and not init) else None
    c_e2 = torch.empty(img.shape, dtype=torch.int16, device=img.device) if want_codes else None
    c_o = torch.empty(img.shape, dtype=torch.uint8, device=img.device) if want_codes else None
    call("mvd_skel_iter_fwd", _p(img), _p(skel), _p(e1), _p(opened), _p(skel_out), _p(c_e1), _p(c_e2), _p(c_o), NC, D, H, W,
         int(init), _stream())
    return e1, opened, skel_out, c_e1, c_e2, c_o


def _skel_iter_bwd_tail(x, opened, skel, c_e2, c_o, g, dims, init):
    """Shared backward of one fused step: (d e1 through the subtraction, d e1 through open(e1), d skel_in)."""
    d_a, d_o = torch.empty_like(x), torch.empty_like(x)
    d_skel = None if init else torch.empty_like(x)
    call("mvd_skel_update_bwd", _p(x), _p(opened), _p(skel), _p(g), _p(d_a), _p(d_o), _p(d_skel), x.numel(), int(init),
         _stream())
    d_e2 = torch.empty_like(x)
    call("mvd_soft_dilate_bwd", _p(c_o), _p(d_o), _p(d_e2), *dims, _stream())
    d_b = torch.empty_like(x)
    call("mvd_soft_erode_bwd", _p(c_e2), _p(d_e2), _p(d_b), *dims, _stream())
    return d_a, d_b, d_skel


class SkelInitFn(Function):
    """skel0 = relu(img - open(img)) (soft_skeleton.py:30-31) in one LDS-tiled launch."""

    @staticmethod
    def forward(ctx, img):
        _require_cuda(img)
        img, NC, D, H, W = _vol(img)
        need = ctx.needs_input_grad[0]
        _, opened, skel0, _, c_e2, c_o = _skel_iter(img, None, need)
        if need:
            ctx.save_for_backward(img, opened, c_e2, c_o)
        ctx.dims = (NC, D, H, W)
        return skel0

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        img, opened, c_e2, c_o = ctx.saved_tensors
        d_a, d_b, _ = _skel_iter_bwd_tail(img, opened, None, c_e2, c_o, g.contiguous(), ctx.dims, True)
        return d_a.add_(d_b)  # autograd's order: the subtraction's share first, then the share through open()


class SkelIterFn(Function):
    """(img, skel) -> (erode(img), skel + relu(d - skel*d)), d = relu(erode(img) - open(erode(img))): one iteration of
    the soft_skel loop (soft_skeleton.py:33-36) in one LDS-tiled launch."""

    @staticmethod
    def forward(ctx, img, skel):
        _require_cuda(img, skel)
        img, NC, D, H, W = _vol(img)
        skel = skel.contiguous()
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        e1, opened, skel_out, c_e1, c_e2, c_o = _skel_iter(img, skel, need)
        if need:
            ctx.save_for_backward(e1, opened, skel, c_e1, c_e2, c_o)
        ctx.dims = (NC, D, H, W)
        return e1, skel_out

    @staticmethod
    @once_differentiable
    def backward(ctx, d_e1, d_skel_out):
        e1, opened, skel, c_e1, c_e2, c_o = ctx.saved_tensors
        g = d_skel_out.contiguous() if d_skel_out is not None else torch.zeros_like(e1)
        d_a, d_b, d_skel = _skel_iter_bwd_tail(e1, opened, skel, c_e2, c_o, g, ctx.dims, False)
        # e1's consumers in creation order: open(e1), the subtraction, the next iteration's erode -> autograd adds their
        # gradients in reverse: (next + subtraction) + open
        total = d_a if d_e1 is None else d_e1.contiguous() + d_a
        total = total.add_(d_b) if total is not d_a else d_a.add_(d_b)
        d_img = torch.empty_like(e1)
        call("mvd_soft_erode_bwd", _p(c_e1), _p(total), _p(d_img), *ctx.dims, _stream())
        return d_img, d_skel


class ClDiceFn(Function):
    """1 - 2*tprec*tsens/(tprec+tsens) from (skel_pred, target, skel_true, pred) (clDice_metric.py:7-36 formula)."""

    @staticmethod
    def forward(ctx, skel_pred, target, skel_true, pred, smooth):
        _require_cuda(skel_pred, target, skel_true, pred)
        sp, tg, st, pr = (t.contiguous() for t in (skel_pred, target, skel_true, pred))
        dev = sp.device
        n = sp.numel()
        sums = torch.empty((4,), dtype=torch.float32, device=dev)
        nb = query("mvd_dot_sum_workspace_bytes", n)
        ws = _Workspace.get(nb, dev)
        call("mvd_dot_sum", _p(sp), _p(tg), _p(sums), n, _p(ws), ws.numel(), _stream())
        call("mvd_dot_sum", _p(st), _p(pr), _p(sums[2:]), n, _p(ws), ws.numel(), _stream())
        out = torch.empty((5,), dtype=torch.float32, device=dev)
        call("mvd_cldice_combine", _p(sums), _p(out), float(smooth), _stream())
        ctx.save_for_backward(tg, st, out)
        return out[0].clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        tg, st, out = ctx.saved_tensors
        # d/d skel_pred = g*(out1*target + out2); d/d pred = g*out3*skel_true  (tiny scalar glue on device)
        c = out * g
        d_sp = torch.addcmul(c[2].expand_as(tg), tg, c[1].expand_as(tg)) if ctx.needs_input_grad[0] else None
        d_pr = st * c[3] if ctx.needs_input_grad[3] else None
        return d_sp, None, None, d_pr, None


# ======================================================================================================== topology (integer)
def cc_label(mask, conn=6):
    """Connected components of a [D,H,W] uint8/bool mask: (int32 labels, int32 count tensor).  Canonical labels
    (1 + smallest linear index of the component) -> bit-exact against oracle/cc_oracle.c."""
    _require_cuda(mask)
    if mask.dim() != 3:
        raise RuntimeError("cc_label takes a [D,H,W] mask")
    m = mask.to(torch.uint8).contiguous()
    D, H, W = m.shape
    labels = torch.empty((D, H, W), dtype=torch.int32, device=m.device)
    count = torch.empty((1,), dtype=torch.int32, device=m.device)
    call("mvd_cc_label", _p(m), _p(labels), _p(count), D, H, W, int(conn), _stream())
    return labels, count


def h0_persistence(f, conn=6, sublevel=True):
    """H0 persistence of a [D,H,W] device field: (birth [N], death [N], death_vertex [N]) as CPU tensors, one bar per
    vertex in linear-index order (the reference's C++ also returns its diagrams on the CPU, functional/sublevel.py:26-49).
    The filtration order of the edges comes from the device (edge keys + radix sort), the elder-rule pairing runs on
    the host (hom.cpp:51-69 restricted to vertices and edges)."""
    _require_cuda(f)
    if f.dim() != 3 or f.dtype != torch.float32:
        raise RuntimeError("h0_persistence takes a [D,H,W] float32 field")
    x = f.detach().contiguous()
    D, H, W = x.shape
    ne = query("mvd_h0_num_edges", D, H, W, int(conn))
    if ne < 0:
        raise RuntimeError("h0_persistence: conn must be 6, 14 or 26")
    nb = query("mvd_h0_workspace_bytes", D, H, W, int(conn))
    ws = _Workspace.get(nb, x.device)
    noff = {6: 3, 14: 7, 26: 13}[int(conn)]
    keys = torch.empty((D * H * W * noff,), dtype=torch.int64, device=x.device)
    call("mvd_h0_sorted_edges", _p(x), _p(keys), D, H, W, int(conn), int(bool(sublevel)), _p(ws), ws.numel(), _stream())
    keys_h = keys[:ne].cpu()          # synchronises the stream
    f_h = x.cpu()
    death = torch.empty((D * H * W,), dtype=torch.float32)
    dv = torch.empty((D * H * W,), dtype=torch.int64)
    ness = _lib.load().mvd_h0_pair_host(_p(f_h), _p(keys_h), ne, D, H, W, int(conn), int(bool(sublevel)), _p(death), _p(dv))
    if ness < 0:
        raise RuntimeError("mvd_h0_pair_host failed: " + _lib.load().mvd_last_error().decode())
    return f_h.reshape(-1), death, dv


class H0DiagramFn(Function):
    """Differentiable H0 diagram of a [D,H,W] field: returns a [N,2] CPU tensor of (birth, death) rows in vertex order,
    essential bars with death = +-inf, like dgms[0] of persistenceForwardHom (hom.cpp:155-185).  backward is the
    reference's persistence_backward (cohom.cpp:148-196): each finite diagram entry's gradient is added onto its critical
    vertex (birth -> the vertex itself, death -> the arg-max vertex of the killing edge), in diagram order."""

    @staticmethod
    def forward(ctx, f, conn, sublevel):
        birth, death, dv = h0_persistence(f, conn, sublevel)
        ctx.save_for_backward(dv)
        ctx.shape, ctx.device = tuple(f.shape), f.device
        return torch.stack([birth, death], 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dv,) = ctx.saved_tensors
        g = g.detach().cpu().float()
        finite = dv >= 0
        grad = g[:, 0].clone()                                          # births: one bar per vertex (every bar has one)
        grad.index_add_(0, dv[finite], g[:, 1][finite])                 # deaths of the finite bars, in diagram order
        return grad.view(ctx.shape).to(ctx.device), None, None


def threshold_mask(f, thr, ge=False):
    _require_cuda(f)
    x = f.contiguous()
    m = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    call("mvd_threshold_mask", _p(x), _p(m), x.numel(), float(thr), int(ge), _stream())
    return m


def seg_label_mask(seg, labels):
    """uint8 mask of the voxels of an int32 segmentation whose label is in `labels` (1..16 ints)."""
    _require_cuda(seg)
    if seg.dtype != torch.int32:
        raise RuntimeError("seg_label_mask takes an int32 segmentation")
    x = seg.contiguous()
    ls = (ctypes.c_int32 * len(labels))(*[int(v) for v in labels])
    m = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    call("mvd_seg_label_mask", _p(x), _p(m), x.numel(), ctypes.cast(ls, ctypes.c_void_p), len(labels), _stream())
    return m


def cc_keep_largest(cc_labels, keep=2):
    """int32[4] device tensor {label0, label1, size0, size1} of the `keep` largest components of a canonical
    label volume (ties -> smaller label)."""
    _require_cuda(cc_labels)
    x = cc_labels.contiguous()
    n = x.numel()
    ws = torch.empty((query("mvd_cc_keep_workspace_bytes", n),), dtype=torch.uint8, device=x.device)
    kept = torch.empty((4,), dtype=torch.int32, device=x.device)
    call("mvd_cc_keep_largest", _p(x), n, int(keep), _p(kept), _p(ws), _stream())
    return kept


def seg_remove_components(seg, cc_labels, kept, background=0):
    _require_cuda(seg)
    x = seg.contiguous()
    out = torch.empty_like(x)
    call("mvd_seg_remove_components", _p(x), _p(cc_labels.contiguous()), _p(kept), _p(out), x.numel(),
         int(background), _stream())
    return out


def set_conv_engine(mode):
    """'auto' (MFMA implicit GEMM when the shape allows) or 'scalar' (gather kernels only; cross-check)."""
    call("mvd_set_conv_engine", {"auto": 0, "scalar": 1}[mode])


def lib_available():
    try:
        _lib.load()
        return True
    except (RuntimeError, OSError, AttributeError):
        return False
