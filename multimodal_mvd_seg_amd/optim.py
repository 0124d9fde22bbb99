"""Fused clip + SGD-Nesterov on flat buffers and the reference's PolyLR schedule.

Replaces `torch.nn.utils.clip_grad_norm_(params, 12)` + `torch.optim.SGD(lr 1e-2, wd 3e-5, momentum .99,
nesterov).step()` (nnUNet/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:473-477, :918-924) and
PolyLRScheduler (nnUNet/nnunetv2/training/lr_scheduler/polylr.py:4-20).
"""
import ctypes

import torch

from . import ops
from ._lib import call, query


class FlatParams:
    """Re-homes the parameters of a module into ONE flat fp32 buffer (and their .grad into another), 16-byte
    aligned per tensor, so that the optimizer, the gradient-norm and the DDP buckets are single contiguous ranges.
    Parameter objects, names and state_dict keys are untouched (p.data becomes a view)."""

    ALIGN = 4  # floats

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        # de-duplicate shared parameters (decoder.encoder.* aliases) preserving order
        seen, uniq = set(), []
        for p in self.params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        self.params = uniq
        dev = self.params[0].device
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        # packed-weight cache (ops.py): epoch of THIS buffer (bumped by the optimizer that updates it through a raw pointer),
        # and -- when a trainer replays its step as a hipGraph -- bf16 packs kept in one persistent buffer owned by this
        # object and rewritten in place, with a generation counter guarding autograd graphs saved across the rewrite
        self.flat._mvd_epoch = [0]
        self.pack16_inplace = False
        self.pack16_gen = [0]
        self.flat._mvd_pack16_gen = self.pack16_gen
        self._pack16_buf = None
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        # direct gradient sink (ops.py): the HIP backward kernels write dW / dbias / dgamma / dbeta straight into this
        # buffer instead of returning fresh tensors that autograd would add into p.grad with one small torch kernel per
        # parameter (~1 ms per step).  A parameter can be taken once per step (zero_grad() starts a new step); a second
        # contribution in the same step goes through autograd's accumulation as before.
        self.step_id = 0
        self._written = [-1] * len(self.params)
        self.listeners = []   # callables(index): "the gradient of parameter index is complete" (BucketedGradReducer)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p._mvd_flat = self.flat   # ops' packed-weight cache also watches the flat buffer's version counter
        self.attach_grads()

    def invalidate_packs(self):
        """Call after writing parameter memory behind torch's back (`p.data.copy_()`, a raw-pointer kernel): the
        packed weight copies the conv kernels read are rebuilt at the next forward."""
        ops.invalidate_packs()

    def attach_grads(self):
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
            p._mvd_take_grad = self._make_take(i)
            p._mvd_grad_done = self._make_done(i)

    def _make_take(self, i):
        def take():
            if self._written[i] == self.step_id:
                return None
            self._written[i] = self.step_id
            return self.params[i].grad
        return take

    def _make_done(self, i):
        def done():
            for fn in self.listeners:
                fn(i)
        return done

    def zero_grad(self):
        self.grad.zero_()
        self.step_id += 1
        self.attach_grads()


class FusedSGDNesterov:
    """Optimizer-like object (zero_grad / step / param_groups / state_dict) driving mvd_grad_sumsq +
    mvd_sgd_nesterov_step.  The clip coefficient is computed on the device: no host sync in the step."""

    def __init__(self, params, lr=1e-2, weight_decay=3e-5, momentum=0.99, nesterov=True, max_grad_norm=12.0):
        if not nesterov:
            raise NotImplementedError("the reference trains with nesterov=True")
        self.fp = params if isinstance(params, FlatParams) else FlatParams(list(params))
        self.param_groups = [{'lr': lr, 'weight_decay': weight_decay, 'momentum': momentum, 'nesterov': True,
                              'params': self.fp.params}]
        self.max_grad_norm = max_grad_norm
        dev = self.fp.flat.device
        self.momentum_buffer = torch.zeros_like(self.fp.flat)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._steps = 0
        self.hyper = self._hyper_host = None   # device copy of the step's scalars (use_device_hyper: hipGraph mode)
        # data parallel: the all-reduce leaves the SUM over ranks in fp.grad; the mean (DDP semantics) is taken inside
        # the optimizer kernel (g * grad_scale) instead of one more pass over the gradient buffer
        self.grad_scale = 1.0

    def zero_grad(self, set_to_none=True):
        # the reference passes set_to_none=True (nnUNetTrainer.py:901); flat gradient views must persist, so zero
        self.fp.zero_grad()

    def grad_norm(self):
        """Device scalar: the global gradient 2-norm of the last step (before clipping)."""
        return self.sumsq.sqrt() * self.grad_scale

    def _hyper_values(self):
        g = self.param_groups[0]
        return (float(g['lr']), float(g['momentum']), float(g['weight_decay']), float(self.max_grad_norm or 0.0),
                float(self.grad_scale), 1.0 if self._steps == 0 else 0.0)

    def use_device_hyper(self):
        """hipGraph mode (trainer.train_step replayed as one graph launch): the step reads lr / momentum / weight decay /
        clip norm / grad_scale / first-step flag from a 6-float device array (mvd_sgd_nesterov_step_dev), refreshed by
        sync_hyper() between replays, so the captured step follows the PolyLR schedule without re-capture."""
        if self.hyper is None:
            self.hyper = torch.zeros(6, dtype=torch.float32, device=self.fp.flat.device)
            self._hyper_host = None
        self.sync_hyper()

    def sync_hyper(self):
        """Host -> device copy of the step's scalars when one of them changed (outside a capture: once per epoch)."""
        if self.hyper is None:
            return
        v = self._hyper_values()
        if v != self._hyper_host:
            self.hyper.copy_(torch.tensor(v, dtype=torch.float32))
            self._hyper_host = v

    def note_replayed_step(self):
        """A graph replay ran the captured step(): keep the host-side step counter (state_dict, first-step flag) true."""
        self._steps += 1

    def step(self):
        fp, g = self.fp, self.param_groups[0]
        if not g['params'][0].is_cuda:
            raise RuntimeError("FusedSGDNesterov runs on the MI355X only (no CPU path)")
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        n = fp.numel
        dev = fp.flat.device
        ws = ops._Workspace.get(query("mvd_sumsq_workspace_bytes", n), dev)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        if (self.max_grad_norm and self.max_grad_norm > 0) or self.hyper is not None:
            call("mvd_grad_sumsq", P(fp.grad), P(self.sumsq), n, P(ws), ws.numel(), s)
        if self.hyper is not None:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()
            call("mvd_sgd_nesterov_step_dev", P(fp.flat), P(fp.grad), P(self.momentum_buffer), P(self.sumsq), n,
                 P(self.hyper), s)
        else:
            call("mvd_sgd_nesterov_step", P(fp.flat), P(fp.grad), P(self.momentum_buffer), P(self.sumsq), n,
                 float(g['lr']), float(g['momentum']), float(g['weight_decay']), float(self.max_grad_norm or 0.0),
                 float(self.grad_scale), 1 if self._steps == 0 else 0, s)
        self._steps += 1
        # the update went through a raw pointer: new weight epoch + every cached packed weight rebuilt in one launch
        ops.repack_all(self.fp)

    def state_dict(self):
        return {'momentum_buffer': self.momentum_buffer.clone(), 'steps': self._steps,
                'param_groups': [{k: v for k, v in self.param_groups[0].items() if k != 'params'}]}

    def load_state_dict(self, sd):
        self.momentum_buffer.copy_(sd['momentum_buffer'])
        self._steps = sd['steps']
        self.param_groups[0].update(sd['param_groups'][0])


class PolyLRScheduler:
    """polylr.py:4-20: lr = initial_lr * (1 - step/max_steps) ** exponent, stepped once per epoch
    (nnUNetTrainer.py:880)."""

    def __init__(self, optimizer, initial_lr: float, max_steps: int, exponent: float = 0.9, current_step: int = None):
        self.optimizer, self.initial_lr, self.max_steps, self.exponent = optimizer, initial_lr, max_steps, exponent
        self.ctr = 0
        self.step(current_step if current_step is not None else -1)

    def step(self, current_step=None):
        if current_step is None or current_step == -1:
            current_step = self.ctr
            self.ctr += 1
        new_lr = self.initial_lr * (1 - current_step / self.max_steps) ** self.exponent
        for param_group in self.optimizer.param_groups:
            param_group['lr'] = new_lr
