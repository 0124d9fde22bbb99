"""TEST INFRASTRUCTURE -- CPU (torch fp32) restatement of the reference's train step and its host logic.

Follows (reference files under nnUNet/nnunetv2/):
* synthetic batch:     training/nnUNetTrainer/variants/benchmarking/nnUNetTrainerBenchmark_5epochs_noDataLoading.py:16-22
* DS scales:           training/nnUNetTrainer/nnUNetTrainer.py:296-302
* train_step:          training/nnUNetTrainer/nnUNetTrainer.py:888-925 (CPU path: no autocast, no GradScaler)
* optimizer:           training/nnUNetTrainer/nnUNetTrainer.py:473-477 (SGD lr 1e-2, wd 3e-5, mom .99, nesterov)
* PolyLR:              training/lr_scheduler/polylr.py:4-20 (reference-pinned, importable)
* DDP batch split:     training/nnUNetTrainer/nnUNetTrainer.py:304-349
* AllGatherGrad:       utilities/ddp_allgather.py:25-48
* MVD dual-branch step: training/nnUNetTrainer/MVDTrainer.py:879-925, lambdas :132-134 (see loss_oracle for the
  unpinned pieces)
"""
import numpy as np
import torch

from . import loss_oracle as LO


def ds_scales(strides):
    """nnUNetTrainer._get_deep_supervision_scales (:296-302)."""
    return list(list(i) for i in 1 / np.cumprod(np.vstack(strides), axis=0))[:-1]


def synthetic_batch(batch_size, in_ch, patch, strides, num_classes=5, seed=1234, device="cpu"):
    """..._noDataLoading.py:16-22: data = rand(B,C,*patch); target[k] = round(rand(B,1,*patch*scale_k) * max_label)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    data = torch.rand((batch_size, in_ch, *patch), generator=g)
    target = [torch.round(torch.rand((batch_size, 1, *[int(i * j) for i, j in zip(patch, k)]), generator=g) *
                          (num_classes - 1)) for k in ds_scales(strides)]
    return {'data': data.to(device), 'target': [t.to(device) for t in target]}


def poly_lr(initial_lr, step, max_steps, exponent=0.9):
    """polylr.py:16-20."""
    return initial_lr * (1 - step / max_steps) ** exponent


def make_optimizer(params, lr=1e-2, weight_decay=3e-5):
    return torch.optim.SGD(params, lr, weight_decay=weight_decay, momentum=0.99, nesterov=True)


def train_step(network, loss_fn, optimizer, batch, clip=12):
    """nnUNetTrainer.train_step (:888-925), CPU branch.  Returns (loss value, list of logits)."""
    data, target = batch['data'], batch['target']
    optimizer.zero_grad(set_to_none=True)
    output = network(data)
    l = loss_fn(output, target)
    l.backward()
    gn = torch.nn.utils.clip_grad_norm_(network.parameters(), clip)
    optimizer.step()
    return l.detach().cpu().numpy(), output, float(gn)


LAMBDA1, LAMBDA2, LAMBDA3, VESSEL = 0.5, 0.1, 1.0, 2  # MVDTrainer.py:132-134, :897-908


def mvd_loss(network, loss_fn, batch, use_topo=True, skel_iter=3, feat_kl=True, T=1):
    """Build's restatement of ContrastiveTrainer.train_step's loss (MVDTrainer.py:895-925):
    l = L(out1,t) + L(out2,t) + lambda3*L_topo(softmax(out1[0])[:,v], onehot(t)[:,v]) + lambda1*L_KL
    where L_KL = kl_loss_compute1(out1[0][:,v], out2[0][:,v]) (+ l2_loss(feat1, feat2, channel_wise) for cfg 3)
    and L_topo = soft-clDice (the torch_topological loss of the reference is absent)."""
    data, target = batch['data'], batch['target']
    o1, o2, f1, f2 = network(data)
    l = loss_fn(o1, target) + loss_fn(o2, target)
    v = VESSEL
    mutual = LO.kl_loss_compute1(o1[0][:, v], o2[0][:, v], T)
    if feat_kl:
        mutual = mutual + LO.l2_loss(f1, f2, channel_wise=True, T=T)
    l = l + LAMBDA1 * mutual
    if use_topo:
        prob = torch.softmax(o1[0], 1)[:, v:v + 1]
        tgt = (target[0] == v).float()
        l = l + LAMBDA3 * LO.soft_cldice(prob, tgt, skel_iter)
    return l, (o1, o2, f1, f2)


def mvd_train_step(network, loss_fn, optimizer, batch, clip=12, **kw):
    optimizer.zero_grad()
    l, outs = mvd_loss(network, loss_fn, batch, **kw)
    l.backward()
    gn = torch.nn.utils.clip_grad_norm_(network.parameters(), clip)
    optimizer.step()
    return l.detach().cpu().numpy(), outs, float(gn)


def ddp_batch_split(global_batch_size, world_size, oversample_foreground_percent=0.33):
    """nnUNetTrainer._set_batch_size_and_oversample (:304-349): per-rank (batch_size, oversample_percent)."""
    assert global_batch_size >= world_size
    batch_sizes, oversample_percents = [], []
    batch_size_per_GPU = np.ceil(global_batch_size / world_size).astype(int)
    for rank in range(world_size):
        if (rank + 1) * batch_size_per_GPU > global_batch_size:
            batch_size = batch_size_per_GPU - ((rank + 1) * batch_size_per_GPU - global_batch_size)
        else:
            batch_size = batch_size_per_GPU
        batch_sizes.append(int(batch_size))
        sample_id_low = 0 if len(batch_sizes) == 0 else np.sum(batch_sizes[:-1])
        sample_id_high = np.sum(batch_sizes)
        if sample_id_high / global_batch_size < (1 - oversample_foreground_percent):
            oversample_percents.append(0.0)
        elif sample_id_low / global_batch_size > (1 - oversample_foreground_percent):
            oversample_percents.append(1.0)
        else:
            covered = sample_id_high / global_batch_size - sample_id_low / global_batch_size
            oversample_percents.append(float(1 - (((1 - oversample_foreground_percent) -
                                                   sample_id_low / global_batch_size) / covered)))
    return batch_sizes, oversample_percents


class AllGatherGrad(torch.autograd.Function):
    """ddp_allgather.py:25-48."""

    @staticmethod
    def forward(ctx, tensor, group=None):
        ctx.group = group
        gathered = [torch.zeros_like(tensor) for _ in range(torch.distributed.get_world_size())]
        torch.distributed.all_gather(gathered, tensor, group=group)
        return torch.stack(gathered, dim=0)

    @staticmethod
    def backward(ctx, *grad_output):
        grad_output = torch.cat(grad_output)
        torch.distributed.all_reduce(grad_output, op=torch.distributed.ReduceOp.SUM, async_op=False, group=ctx.group)
        return grad_output[torch.distributed.get_rank()], None
