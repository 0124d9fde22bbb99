"""The train step replayed as ONE hipGraph launch (trainer._graphed_step) against the eager step.

The graph holds the same kernels with the same arguments in the same order, so everything must be BIT-identical to the
eager path: loss of every step, every parameter after N steps, the momentum buffer -- in fp32 and in bf16 mixed
precision, for the single-branch step (nnUNetTrainer.py:888-925) and for the dual-branch step with the soft-clDice term
and the integer component count (MVDTrainer.py:879-925), with a new batch every step (static input buffers) and a
learning rate that changes between replays (PolyLR, polylr.py:16-20: the optimizer's scalars live in device memory).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
STRIDES = [[1, 1, 1], [2, 2, 2], [2, 2, 2]]
DS = {"channel_names": {str(i): f"m{i}" for i in range(4)}, "labels": {"background": 0, "a": 1, "b": 2, "c": 3, "d": 4}}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu_and_lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from multimodal_mvd_seg_amd import _lib
    _lib.load()


def _make(cls_name, precision, graph, patch=(32, 32, 32), **attrs):
    from multimodal_mvd_seg_amd import trainer
    plans = trainer.make_plans(patch, STRIDES, batch_size=2)
    tr = getattr(trainer, cls_name)(plans, "3d_fullres", 0, DS, device=DEV)
    tr.precision = precision
    tr.use_hip_graph = graph
    for k, v in attrs.items():
        setattr(tr, k, v)
    torch.manual_seed(0)
    tr.initialize()
    return tr


def _batches(tr, n):
    return [tr.make_dummy_batch(seed=100 + i) for i in range(n)]


def _run(tr, batches, lr_epochs):
    losses = []
    for i, b in enumerate(batches):
        if i in lr_epochs:
            tr.lr_scheduler.step(lr_epochs[i])
        losses.append(np.asarray(tr.train_step(b)["loss"]).copy())
    torch.cuda.synchronize()
    return losses


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graphed_single_branch_step_is_bit_identical_to_eager(precision):
    a = _make("nnUNetTrainerMI355", precision, False)
    b = _make("nnUNetTrainerMI355", precision, True)
    b.network.load_state_dict(a.network.state_dict())
    b.optimizer.fp.invalidate_packs()
    batches = _batches(a, 8)
    lr_epochs = {0: 0, 5: 40, 7: 120}          # the schedule moves while the graph is being replayed
    la, lb = _run(a, batches, lr_epochs), _run(b, batches, lr_epochs)
    assert b._step_graph is not None and b._step_graph["graph"] is not None, "the step was never captured"
    assert a._step_graph is None
    for i, (x, y) in enumerate(zip(la, lb)):
        assert np.array_equal(x, y), f"loss of step {i}: eager {x} graph {y}"
    for (n, p), (_, q) in zip(a.network.named_parameters(), b.network.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n
    assert torch.equal(a.optimizer.momentum_buffer, b.optimizer.momentum_buffer)
    assert a.optimizer._steps == b.optimizer._steps == 8
    # an eager forward after the replays sees the weights the last replay wrote (packed copies rewritten in place)
    va, vb = a.validation_step(batches[0]), b.validation_step(batches[0])
    assert np.array_equal(va["loss"], vb["loss"]) and np.array_equal(va["tp_hard"], vb["tp_hard"])


def test_graphed_dual_branch_step_with_topology_terms_is_bit_identical_to_eager_and_counts_match_the_oracle():
    from oracle import cc_oracle
    kw = dict(use_topo=True)
    a = _make("ContrastiveTrainerMI355", "bf16", False, **kw)
    b = _make("ContrastiveTrainerMI355", "bf16", True, **kw)
    b.network.load_state_dict(a.network.state_dict())
    b.optimizer.fp.invalidate_packs()
    batches = _batches(a, 6)
    la, lb = _run(a, batches, {0: 0}), _run(b, batches, {0: 0})
    assert b._step_graph["graph"] is not None
    for i, (x, y) in enumerate(zip(la, lb)):
        assert np.array_equal(x, y), f"loss of step {i}: eager {x} graph {y}"
    for (n, p), (_, q) in zip(a.network.named_parameters(), b.network.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n
    # the integer step: component counts of the last step, eager == graph == the C oracle on the same masks
    ta, tb = a.last_topology, b.last_topology
    for k in ("cc_pred", "cc_true", "betti0_error"):
        assert ta[k].dtype == torch.int32 and torch.equal(ta[k], tb[k]), k
    tgt = batches[-1]["target"][0]
    for n in range(tgt.shape[0]):
        mask = (tgt[n, 0] == a.vessel_channel).cpu().numpy()
        _labels, cnt = cc_oracle.cc_label(mask, 26)
        assert int(ta["cc_true"][n]) == cnt
    with torch.no_grad():
        o1 = a.network(batches[-1]["data"])[0][0]
    # (the weights moved by one step since the counted forward: only check the count of the CURRENT prediction's mask
    # through the same device path against the oracle)
    from multimodal_mvd_seg_amd import ops
    prob = torch.softmax(o1.float(), 1)[:, a.vessel_channel]
    for n in range(prob.shape[0]):
        m = ops.threshold_mask(prob[n].contiguous(), 0.5, ge=True)
        _l, c = ops.cc_label(m, 26)
        assert int(c) == cc_oracle.cc_label(m.cpu().numpy().astype(bool), 26)[1]


def test_graph_is_recaptured_when_the_input_geometry_changes_and_can_be_switched_off():
    tr = _make("nnUNetTrainerMI355", "fp32", True)
    ref = _make("nnUNetTrainerMI355", "fp32", False)
    ref.network.load_state_dict(tr.network.state_dict())
    ref.optimizer.fp.invalidate_packs()
    bs = _batches(tr, 5)
    one = [{"data": b["data"][:1].contiguous(), "target": [t[:1].contiguous() for t in b["target"]]} for b in bs]
    for b in bs:
        tr.train_step(b)
        ref.train_step(b)
    g0 = tr._step_graph["graph"]
    assert g0 is not None
    tr.train_step(one[0])                        # batch 1: another geometry -> the old capture must not be replayed
    ref.train_step(one[0])
    assert tr._step_graph["graph"] is None and tr._step_graph["warm"] == 1
    for b in one[1:] + one[:2]:
        tr.train_step(b)
        ref.train_step(b)
    assert tr._step_graph["graph"] is not None and tr._step_graph["graph"] is not g0
    for (n, p), (_, q) in zip(ref.network.named_parameters(), tr.network.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n
    os.environ["MVD_HIPGRAPH"] = "0"
    try:
        t2 = _make("nnUNetTrainerMI355", "fp32", os.environ.get("MVD_HIPGRAPH", "1") != "0")
        for b in bs:
            t2.train_step(b)
        assert t2._step_graph is None
    finally:
        del os.environ["MVD_HIPGRAPH"]


def test_rccl_all_reduce_inside_a_captured_graph_world_size_one():
    """What MVD_HIPGRAPH_DDP=1 relies on: torch.distributed's nccl (= RCCL) all_reduce with async_op=True issued inside a
    stream capture, fenced by work.wait(), replays correctly.  One rank only (one GPU per box): the multi-GPU behaviour
    stays unverified, which is why graphs are off by default when world_size > 1."""
    import subprocess
    import sys
    code = r'''
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
y = torch.zeros_like(x)
dist.all_reduce(x.clone())                       # communicator up before the capture
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    t = x * 2
    w = dist.all_reduce(t, async_op=True)
    w.wait()
    y.copy_(t + 1)
for k in range(3):
    x.fill_(float(k))
    g.replay()
    torch.cuda.synchronize()
    assert float(y[0]) == 2.0 * k + 1.0, (k, float(y[0]))
dist.destroy_process_group()
print("RCCL_IN_GRAPH_OK")
'''
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "RCCL_IN_GRAPH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_gradient_sharing_of_two_consumer_tensors_matches_autograds_own_sum(precision):
    """ops._GradShare: the skip connections (next encoder stage + decoder conv, UNetDecoder.py:106-108) and the decoder stage
    outputs (seg head + next up-sampling) get ONE gradient buffer; the second consumer adds inside its kernel (k_dgrad32s /
    k_fwd16 epilogue read-modify-write, k_seghead_dx4 accumulate) instead of autograd's elementwise add.  fp32: a + b either
    way -> bit-identical parameters after two steps; bf16: the skip path is bit-identical (fp32 add of two bf16 values, one
    rounding, as torch's add), the seg-head path saves one rounding -> equal to bf16 resolution."""
    from multimodal_mvd_seg_amd import network
    res = []
    for share in (True, False):
        network.SHARE_GRADS[0] = share
        try:
            tr = _make("nnUNetTrainerMI355", precision, False, patch=(64, 64, 64))
            if res:
                tr.network.load_state_dict(res[0][2])
                tr.optimizer.fp.invalidate_packs()
            sd0 = {k: v.clone() for k, v in tr.network.state_dict().items()}
            bs = _batches(tr, 2)
            ls = [float(tr.train_step(b)["loss"]) for b in bs]
            res.append((ls, [p.detach().clone() for p in tr.network.parameters()], sd0))
        finally:
            network.SHARE_GRADS[0] = True
    (la, pa, _), (lb, pb, _) = res
    if precision == "fp32":
        assert la == lb
        for u, v in zip(pa, pb):
            assert torch.equal(u, v)
    else:
        assert abs(la[1] - lb[1]) <= 2e-3 * abs(lb[1])
        worst = max(float((u - v).abs().max()) for u, v in zip(pa, pb))
        assert worst <= 2e-3, worst
