"""Data-parallel plumbing: one process per GPU, torch.distributed ("nccl" == RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  Replaces the reference's DDP wrap (nnUNet/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:220-222),
its per-rank batch split (:304-349) and the batch-dice all-gather (utilities/ddp_allgather.py:25-48).

Gradient reduction (collective C1 of SURVEY 2.4): gradients live in ONE flat buffer (optim.FlatParams) that is cut
into buckets in *backward completion order* (reverse parameter order: decoder tail first, enc0 last).  A
post-accumulate-grad hook counts a bucket's parameters; when the last one lands, the bucket is all-reduced (SUM,
then scaled by 1/world) asynchronously on the communication stream while backward keeps running -- xGMI is a full
mesh of 7 point-to-point links, so a 125 MB gradient costs ~1.4 ms on a ring and is hidden entirely as long as the
big 320-channel buckets are issued mid-backward.  `wait()` fences the compute stream before clip + SGD.
"""
import numpy as np
import torch
import torch.distributed as dist


def ddp_batch_split(global_batch_size, world_size, oversample_foreground_percent=0.33):
    """Per-rank (batch sizes, foreground-oversampling fractions) of a plans batch split over `world_size` ranks --
    the rule of nnUNetTrainer._set_batch_size_and_oversample (:304-349), pinned to that method by
    tests/golden/ddp_split.json.  Rank r takes ceil(G/W) samples, the last ranks what is left; the samples are thought
    of as laid out 0..G-1 and the LAST `oversample` fraction of them is forced-foreground, so a rank's fraction is the
    part of its interval that lies beyond (1 - oversample) * G."""
    G, W = int(global_batch_size), int(world_size)
    if G < W:
        raise AssertionError('Cannot run DDP if the batch size is smaller than the number of GPUs... Duh.')
    per_rank = int(np.ceil(G / W))
    sizes = [per_rank if (r + 1) * per_rank <= G else G - r * per_rank for r in range(W)]
    ends = np.cumsum(sizes)
    cut = 1 - oversample_foreground_percent
    fractions = []
    for size, end in zip(sizes, ends.tolist()):
        lo, hi = (end - size) / G, end / G
        if hi < cut:
            fractions.append(0.0)
        elif lo > cut:
            fractions.append(1.0)
        else:
            fractions.append(float(1 - ((cut - lo) / (hi - lo))))
    return [int(v) for v in sizes], fractions


_HOOK_ON_UNDEFINED = []


def _hook_fires_for_undefined_grad():
    """Does autograd run a leaf's post-accumulate-grad hook when the producing Function returned None for it?  (It does
    in torch 2.x: the AccumulateGrad node executes once ALL uses of the parameter in the graph have run, defined
    gradient or not.)  Probed once on a 1-element CPU graph rather than assumed."""
    if not _HOOK_ON_UNDEFINED:
        class _Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                return x * 1.0

            @staticmethod
            def backward(ctx, dy):
                return dy, None
        w = torch.nn.Parameter(torch.zeros(1))
        fired = []
        h = w.register_post_accumulate_grad_hook(lambda p: fired.append(1))
        _Probe.apply(torch.ones(1, requires_grad=True), w).sum().backward()
        h.remove()
        _HOOK_ON_UNDEFINED.append(bool(fired))
    return _HOOK_ON_UNDEFINED[0]


class BucketedGradReducer:
    """Bucketed, overlapped gradient all-reduce over a FlatParams gradient buffer.

    A bucket is launched when every one of its parameters has reported its gradient complete.  The report is the
    parameter's post-accumulate-grad hook: autograd runs it once per backward pass, after EVERY use of the parameter in
    the graph has executed -- also when the HIP backward kernel wrote the gradient straight into the flat buffer
    (optim.FlatParams' direct sink) and handed autograd None.  (Round 1 counted the sink's own listener as a second
    report: every parameter reported twice and buckets were launched half-filled.)  Only if the running torch does not
    fire the hook for undefined gradients (probed at construction) is the sink listener used as well, idempotently.
    A report for a bucket whose all-reduce is already in flight -- a second backward() before wait() -- raises
    instead of dropping the late gradient.  `optimizer` (FusedSGDNesterov) receives grad_scale = 1/world so that the
    mean is taken inside the optimizer kernel."""

    def __init__(self, flat_params, bucket_bytes=25 * 1024 * 1024, process_group=None, optimizer=None, always_hook=False):
        self.fp = flat_params
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.optimizer = optimizer
        # buckets in reverse parameter order (backward completion order), each a contiguous slice of fp.grad
        n = len(self.fp.params)
        ends = [o + ((p.numel() + self.fp.ALIGN - 1) // self.fp.ALIGN) * self.fp.ALIGN
                for p, o in zip(self.fp.params, self.fp.offsets)]
        self.buckets = []  # (start, end, [param indices])
        # Cut from the END of the buffer (the parameters backward completes first) into >= bucket_bytes pieces; the piece
        # reduced LAST (the network's first layers, complete only when backward ends: its all-reduce cannot hide behind
        # anything) is kept to <= tail_bytes, so that what is exposed after backward is one small collective.
        tail_bytes = min(bucket_bytes, 4 * 1024 * 1024)
        sizes = [self.fp.params[i].numel() * 4 for i in range(n)]
        prefix = [0] * (n + 1)   # bytes of parameters 0 .. i-1
        for i in range(n):
            prefix[i + 1] = prefix[i] + sizes[i]
        cur_end, cur_idx, cur_bytes = ends[-1] if n else 0, [], 0
        for i in range(n - 1, -1, -1):
            cur_idx.append(i)
            cur_bytes += sizes[i]
            left = prefix[i]          # bytes still in front of parameter i
            close = cur_bytes >= bucket_bytes or i == 0
            if not close and 0 < left <= tail_bytes and cur_bytes + left > tail_bytes:
                close = True          # what is left fits the tail bucket, together with this piece it would not
            if close:
                self.buckets.append((self.fp.offsets[i], cur_end, list(cur_idx)))
                cur_end, cur_idx, cur_bytes = self.fp.offsets[i], [], 0
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self._hooks = []
        self.always_hook = always_hook   # measurement aid (tools/ddp_overlap_probe.py): hooks and bucket logic at world 1
        if self.world > 1 or always_hook:
            for i, p in enumerate(self.fp.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
            self.uses_listener = not _hook_fires_for_undefined_grad()
            listeners = getattr(self.fp, 'listeners', None)
            if self.uses_listener and listeners is not None:
                listeners.append(self._on_ready)
            if optimizer is not None and self.world > 1:
                optimizer.grad_scale = 1.0 / self.world
        self.reset()

    def reset(self):
        self._ready = set()
        self._missing = [len(idx) for (_, _, idx) in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []

    def _on_ready(self, i):
        b = self.bucket_of[i]
        if self._launched[b]:
            raise RuntimeError(
                f"gradient of parameter #{i} arrived after its bucket's all-reduce was launched (a second backward() "
                "before reducer.wait()?): it would not be reduced.  Accumulate locally and reduce once per step.")
        if i in self._ready:
            return
        self._ready.add(i)
        self._missing[b] -= 1
        if self._missing[b] == 0:
            self._launch(b)

    def _make_hook(self, i):
        def hook(_param):
            self._on_ready(i)
        return hook

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        buf = self.fp.grad[s:e]
        if self.world <= 1:   # (always_hook at world 1: the bucket bookkeeping without a collective)
            return
        # async_op=True: the collective runs on the process group's own stream, ordered after the kernels already
        # enqueued on the current stream (the wgrad that produced this bucket); backward continues meanwhile.
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, buf))

    def wait(self):
        """Call after backward: launches whatever bucket did not fire (parameters without gradient this step), fences
        the collectives and takes the mean over ranks (DDP semantics) -- inside the optimizer kernel when an optimizer
        was given, else with one scaling pass."""
        if self.world <= 1:
            if self.always_hook:
                self.reset()
            return
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for work, _ in self._works:
            work.wait()
        if self.optimizer is None:
            self.fp.grad.mul_(1.0 / self.world)
        self.reset()

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        listeners = getattr(self.fp, 'listeners', None)
        if listeners is not None and self._on_ready in listeners:
            listeners.remove(self._on_ready)
        if self.optimizer is not None:
            self.optimizer.grad_scale = 1.0


def gather_dice_stats(stats):
    """Batch-dice under DDP (collective C2; ddp_allgather.py:25-48): all-gather the per-sample Dice statistics.
    Returns (stats_all [world*N, 3K+1], offset of this rank's rows, gradient multiplier = world size: the backward
    of AllGatherGrad all-reduces (SUM) the identical per-rank gradients)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    gathered = [torch.zeros_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats.contiguous())
    return torch.cat(gathered, 0), rank * stats.shape[0], float(world)


def broadcast_parameters(flat_params, src=0):
    """DDP broadcasts rank 0's parameters at wrap time (nnUNetTrainer.py:222)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat_params.flat, src=src)
        if hasattr(flat_params, "invalidate_packs"):
            flat_params.invalidate_packs()  # packed weight copies made before the broadcast are stale now
