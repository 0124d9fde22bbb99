"""World-size-2 gloo tests (CPU) of the data-parallel host logic: bucketed overlapped gradient all-reduce over the
flat gradient buffer, parameter broadcast and the batch-dice statistics gather.  The reducer is transport-agnostic:
on the GPU box the same code runs over the "nccl" backend (= RCCL over xGMI)."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multimodal_mvd_seg_amd import optim, parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(7, 33), torch.nn.Tanh(), torch.nn.Linear(33, 19), torch.nn.Tanh(),
                               torch.nn.Linear(19, 3))


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        net = _model(100 + rank)  # different init per rank: broadcast must make them equal
        fp = optim.FlatParams(list(net.parameters()))
        parallel.broadcast_parameters(fp)
        red = parallel.BucketedGradReducer(fp, bucket_bytes=256)  # tiny buckets -> several collectives
        assert len(red.buckets) >= 3
        # bucket slices tile the flat buffer, in reverse parameter order
        spans = sorted((s, e) for s, e, _ in red.buckets)
        assert spans[0][0] == 0 and spans[-1][1] == fp.numel
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert red.buckets[0][2][0] == len(fp.params) - 1
        g = torch.Generator().manual_seed(7)
        X = torch.randn(8, 7, generator=g)
        Y = torch.randn(8, 3, generator=g)
        xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
        for step in range(2):
            fp.zero_grad()
            loss = ((net(xs) - ys) ** 2).mean()
            loss.backward()  # hooks fire the bucket all-reduces while backward is still running
            red.wait()
            if step == 0:
                torch.save({"grad": fp.grad.clone(), "flat": fp.flat.clone()}, os.path.join(out_dir, f"r{rank}.pt"))
            with torch.no_grad():
                fp.flat -= 0.1 * fp.grad
        torch.save(fp.flat.clone(), os.path.join(out_dir, f"final{rank}.pt"))
        # batch-dice statistics gather (collective C2)
        stats = torch.full((2, 4), float(rank))
        allst, off, mult = parallel.gather_dice_stats(stats)
        assert allst.shape == (4, 4) and off == rank * 2 and mult == 2.0
        assert torch.equal(allst[:2], torch.zeros(2, 4)) and torch.equal(allst[2:], torch.ones(2, 4))
    finally:
        dist.destroy_process_group()


class _SinkLinear(torch.autograd.Function):
    """y = x W^T whose backward hands dW to optim.FlatParams' direct gradient sink exactly as the HIP wgrad wrappers of
    multimodal_mvd_seg_amd.ops do (_take_grad -> write -> _grad_done), so the listener path runs without a GPU."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w.detach())
        ctx.param = w
        return x @ w.detach().t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dw = dy.t() @ x
        take = getattr(ctx.param, "_mvd_take_grad", None)
        sink = take() if take is not None else None
        if sink is not None:
            sink.copy_(dw)
            ctx.param._mvd_grad_done()
            dw = None
        return dy @ w, dw


class _SinkNet(torch.nn.Module):
    def __init__(self, seed, reuse_first=False):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = torch.nn.Parameter(torch.randn(16, 16, generator=g) * 0.3)
        self.w2 = torch.nn.Parameter(torch.randn(16, 16, generator=g) * 0.3)
        self.head = torch.nn.Linear(16, 3)
        with torch.no_grad():
            self.head.weight.copy_(torch.randn(3, 16, generator=g) * 0.3)
            self.head.bias.zero_()
        self.reuse_first = reuse_first

    def forward(self, x):
        h = torch.tanh(_SinkLinear.apply(x, self.w1))
        h = torch.tanh(_SinkLinear.apply(h, self.w2))
        if self.reuse_first:
            h = torch.tanh(_SinkLinear.apply(h, self.w1))  # second contribution to w1 in the same graph
        return self.head(h)


def _sink_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        net = _SinkNet(200 + rank)
        fp = optim.FlatParams(list(net.parameters()))
        parallel.broadcast_parameters(fp)
        red = parallel.BucketedGradReducer(fp, bucket_bytes=64)  # head (bias + weight), w2, w1
        assert len(red.buckets) == 3
        assert red.uses_listener is False  # torch 2.x: the accumulate hook alone reports (fires for sink-written grads)
        reports = []
        fp.listeners.append(lambda i: reports.append(i))
        g = torch.Generator().manual_seed(9)
        X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 3, generator=g)
        xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
        for step in range(2):
            fp.zero_grad()
            ((net(xs) - ys) ** 2).mean().backward()
            assert sorted(reports[-2:]) == [0, 1]     # w1, w2 were written through the direct sink, the head by autograd
            assert all(red._launched)                 # every bucket fired during backward, none left for wait()
            red.wait()
            if step == 0:
                torch.save(fp.grad.clone(), os.path.join(out_dir, f"sink_grad{rank}.pt"))
            with torch.no_grad():
                fp.flat -= 0.1 * fp.grad
        torch.save(fp.flat.clone(), os.path.join(out_dir, f"sink_final{rank}.pt"))
        # a second backward() before wait(): late contributions must fail loudly, not be dropped
        fp.zero_grad()
        ((net(xs) - ys) ** 2).mean().backward()
        try:
            ((net(xs) - ys) ** 2).mean().backward()
            late = "no error"
        except RuntimeError as e:
            late = str(e)
        red.wait()
        # a parameter used twice in one graph: the first use writes the sink, the second goes through autograd's
        # accumulation; the hook reports once, after both -> the bucket must hold the full gradient
        net2 = _SinkNet(300, reuse_first=True)
        fp2 = optim.FlatParams(list(net2.parameters()))
        parallel.broadcast_parameters(fp2)
        red2 = parallel.BucketedGradReducer(fp2, bucket_bytes=64)
        fp2.zero_grad()
        ((net2(xs) - ys) ** 2).mean().backward()
        red2.wait()
        torch.save({"late": late, "twice_grad": fp2.grad.clone()}, os.path.join(out_dir, f"sink_msgs{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_direct_sink_listener_path_and_late_contributions():
    """The gradient path the HIP trainer uses under DDP (direct sink -> listener -> bucket launch), on gloo/CPU:
    equality with a single process on the full batch, and loud failure for contributions that arrive after a bucket's
    all-reduce was launched (they used to be dropped silently)."""
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_sink_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        g0, g1 = torch.load(os.path.join(d, "sink_grad0.pt")), torch.load(os.path.join(d, "sink_grad1.pt"))
        assert torch.equal(g0, g1)
        net = _SinkNet(200)
        fp = optim.FlatParams(list(net.parameters()))
        g = torch.Generator().manual_seed(9)
        X, Y = torch.randn(8, 16, generator=g), torch.randn(8, 3, generator=g)
        fp.zero_grad()
        ((net(X) - Y) ** 2).mean().backward()
        assert torch.allclose(fp.grad, g0, atol=1e-7)
        assert torch.equal(torch.load(os.path.join(d, "sink_final0.pt")), torch.load(os.path.join(d, "sink_final1.pt")))
        net2 = _SinkNet(300, reuse_first=True)
        fp2 = optim.FlatParams(list(net2.parameters()))
        fp2.zero_grad()
        ((net2(X) - Y) ** 2).mean().backward()
        for r in range(world):
            m = torch.load(os.path.join(d, f"sink_msgs{r}.pt"))
            assert "after its bucket's all-reduce was launched" in m["late"], m["late"]
            assert torch.allclose(m["twice_grad"], fp2.grad, atol=1e-7)  # shared parameter: both contributions reduced


@pytest.mark.timeout(300)
def test_bucketed_reducer_equals_single_process_big_batch():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), d), nprocs=world, join=True)
        r0, r1 = torch.load(os.path.join(d, "r0.pt")), torch.load(os.path.join(d, "r1.pt"))
        assert torch.equal(r0["flat"], r1["flat"])           # broadcast
        assert torch.equal(r0["grad"], r1["grad"])           # all ranks hold the same averaged gradient
        # single-process reference: rank-0 init, full batch of 8 (mean loss == mean of the two rank means)
        net = _model(100)
        fp = optim.FlatParams(list(net.parameters()))
        g = torch.Generator().manual_seed(7)
        X = torch.randn(8, 7, generator=g)
        Y = torch.randn(8, 3, generator=g)
        fp.zero_grad()
        ((net(X) - Y) ** 2).mean().backward()
        assert torch.allclose(fp.grad, r0["grad"], atol=1e-7)
        f0, f1 = torch.load(os.path.join(d, "final0.pt")), torch.load(os.path.join(d, "final1.pt"))
        assert torch.equal(f0, f1)


def test_reducer_is_a_noop_without_process_group():
    net = _model(0)
    fp = optim.FlatParams(list(net.parameters()))
    red = parallel.BucketedGradReducer(fp)
    assert red.world == 1 and not red._hooks
    red.wait()
