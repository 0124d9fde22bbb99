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


class BucketedGradReducer:
    """Bucketed, overlapped gradient all-reduce over a FlatParams gradient buffer."""

    def __init__(self, flat_params, bucket_bytes=25 * 1024 * 1024, process_group=None):
        self.fp = flat_params
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # buckets in reverse parameter order (backward completion order), each a contiguous slice of fp.grad
        n = len(self.fp.params)
        ends = [o + ((p.numel() + self.fp.ALIGN - 1) // self.fp.ALIGN) * self.fp.ALIGN
                for p, o in zip(self.fp.params, self.fp.offsets)]
        self.buckets = []  # (start, end, [param indices])
        cur_end, cur_idx, cur_bytes = ends[-1] if n else 0, [], 0
        for i in range(n - 1, -1, -1):
            cur_idx.append(i)
            cur_bytes += self.fp.params[i].numel() * 4
            if cur_bytes >= bucket_bytes or i == 0:
                self.buckets.append((self.fp.offsets[i], cur_end, list(cur_idx)))
                cur_end, cur_idx, cur_bytes = self.fp.offsets[i], [], 0
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._hooks = []
        if self.world > 1:
            for i, p in enumerate(self.fp.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
            # gradients the HIP kernels write straight into the flat buffer never pass through autograd's accumulation
            listeners = getattr(self.fp, 'listeners', None)
            if listeners is not None:
                listeners.append(self._on_ready)
        self.reset()

    def reset(self):
        self._pending = [len(idx) for (_, _, idx) in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = []

    def _on_ready(self, i):
        b = self.bucket_of[i]
        self._pending[b] -= 1
        if self._pending[b] == 0 and not self._launched[b]:
            self._launch(b)

    def _make_hook(self, i):
        def hook(_param):
            self._on_ready(i)
        return hook

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        self._launched[b] = True
        buf = self.fp.grad[s:e]
        # async_op=True: the collective runs on the process group's own stream, ordered after the kernels already
        # enqueued on the current stream (the wgrad that produced this bucket); backward continues meanwhile.
        work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._works.append((work, buf))

    def wait(self):
        """Call after backward: launches whatever bucket did not fire (parameters without gradient), fences the
        collectives and averages (DDP semantics: gradient = mean over ranks)."""
        if self.world <= 1:
            return
        for b in range(len(self.buckets)):
            if not self._launched[b]:  # parameters without a gradient this step (or counted twice): reduce now
                self._launch(b)
        for work, _ in self._works:
            work.wait()
        self.fp.grad.mul_(1.0 / self.world)
        self.reset()

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []


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
