"""Host-side mirror of the reference's trainer plugin surface for the train-step hot path.

`nnUNetTrainerMI355` keeps the signatures the reference discovers and calls by name
(nnUNet/nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py): `__init__(plans, configuration, fold, dataset_json,
unpack_dataset, device, specified_cfg)` :69-70, `initialize` :201-228, static `build_network_architecture`
:268-294, `_get_deep_supervision_scales` :296-302, `_set_batch_size_and_oversample` :304-349, `_build_loss`
:351-375, `configure_optimizers` :473-477, `set_deep_supervision_enabled` :802-810, `train_step` :888-925,
`validation_step` :942-1004, `on_validation_epoch_end` :1006-1037.  In the reference tree this class would derive
from nnUNetTrainer and only override `build_network_architecture` / `_build_loss` / `configure_optimizers` /
`initialize`; upstream nnunetv2 cannot be imported here (batchgenerators & co. absent), so the
step-relevant parts of the base class are restated.  Data loading, logging, checkpoint files and final validation
are out of scope.

`ContrastiveTrainerMI355` mirrors the dual-branch mutual-distillation step
(nnUNet/nnunetv2/training/nnUNetTrainer/MVDTrainer.py:879-925; lambdas :132-134).
"""
import os

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

from . import losses, ops
from .network import InitWeights_He, MI355PlainConvUNet, MVDDualBranchNet, set_precision
from .optim import FlatParams, FusedSGDNesterov, PolyLRScheduler
from .parallel import BucketedGradReducer, broadcast_parameters, ddp_batch_split


# ------------------------------------------------------------------------------------------------ plans (input contract)
class ConfigurationManager:
    """The plans.json keys the builder reads (get_network_from_plans.py:26-83; plans_handler.py:32-178)."""

    def __init__(self, configuration_dict: dict):
        self.configuration = configuration_dict

    def __getattr__(self, k):
        cfg = self.__dict__.get('configuration', {})
        if k in cfg:
            return cfg[k]
        raise AttributeError(k)


class LabelManager:
    """label_handling.py:21,230-234 for the plain-labels case (no regions, no ignore label)."""

    def __init__(self, label_dict: dict):
        self.label_dict = label_dict
        self.all_labels = sorted(int(v) for v in label_dict.values() if not isinstance(v, (list, tuple)))
        self.has_regions = any(isinstance(v, (list, tuple)) and len(v) > 1 for v in label_dict.values())
        self.ignore_label = label_dict.get('ignore')
        self.has_ignore_label = self.ignore_label is not None
        if self.has_regions or self.has_ignore_label:
            raise NotImplementedError("regions / ignore label are outside the benchmarked path (SURVEY 8 a-6)")

    @property
    def num_segmentation_heads(self):
        return len(self.all_labels)


class PlansManager:
    def __init__(self, plans: dict):
        self.plans = plans

    def get_configuration(self, name):
        cfgs = self.plans['configurations']
        cfg = dict(cfgs[name])
        if 'inherits_from' in cfg:  # plans_handler.py:197-219
            base = dict(self.get_configuration(cfg['inherits_from']).configuration)
            base.update(cfg)
            cfg = base
        return ConfigurationManager(cfg)

    def get_label_manager(self, dataset_json):
        return LabelManager(dataset_json['labels'])


def determine_num_input_channels(plans_manager, configuration_manager, dataset_json):
    """label_handling.py:283-301 without the cascade branch."""
    key = 'channel_names' if 'channel_names' in dataset_json else 'modality'
    return len(dataset_json[key])


def make_plans(patch_size, strides, batch_size=2, base_features=32, max_features=320, n_conv=2, batch_dice=False,
               conv_kernel_sizes=None):
    """A minimal nnUNetPlans.json-shaped dict for the synthetic configurations of BASELINE.json."""
    n = len(strides)
    return {'plans_name': 'nnUNetPlans', 'configurations': {'3d_fullres': {
        'patch_size': list(patch_size), 'batch_size': batch_size, 'UNet_class_name': 'PlainConvUNet',
        'UNet_base_num_features': base_features, 'unet_max_num_features': max_features,
        'n_conv_per_stage_encoder': [n_conv] * n, 'n_conv_per_stage_decoder': [n_conv] * (n - 1),
        'conv_kernel_sizes': conv_kernel_sizes or [[3, 3, 3]] * n, 'pool_op_kernel_sizes': [list(s) for s in strides],
        'batch_dice': batch_dice}}}


def get_network_from_plans(plans_manager, dataset_json, configuration_manager, num_input_channels,
                           deep_supervision=True):
    """get_network_from_plans.py:15-92 for UNet_class_name == 'PlainConvUNet'."""
    num_stages = len(configuration_manager.conv_kernel_sizes)
    dim = len(configuration_manager.conv_kernel_sizes[0])
    if dim != 3:
        raise NotImplementedError("3d_fullres only")
    if configuration_manager.UNet_class_name != 'PlainConvUNet':
        raise NotImplementedError("north_star names PlainConvUNet; ResidualEncoderUNet is out of scope")
    label_manager = plans_manager.get_label_manager(dataset_json)
    model = MI355PlainConvUNet(
        input_channels=num_input_channels, n_stages=num_stages,
        features_per_stage=[min(configuration_manager.UNet_base_num_features * 2 ** i,
                                configuration_manager.unet_max_num_features) for i in range(num_stages)],
        conv_op=nn.Conv3d, kernel_sizes=configuration_manager.conv_kernel_sizes,
        strides=configuration_manager.pool_op_kernel_sizes, num_classes=label_manager.num_segmentation_heads,
        deep_supervision=deep_supervision, n_conv_per_stage=configuration_manager.n_conv_per_stage_encoder,
        n_conv_per_stage_decoder=configuration_manager.n_conv_per_stage_decoder, conv_bias=True,
        norm_op=nn.InstanceNorm3d, norm_op_kwargs={'eps': 1e-5, 'affine': True}, dropout_op=None,
        dropout_op_kwargs=None, nonlin=nn.LeakyReLU, nonlin_kwargs={'inplace': True})
    model.apply(InitWeights_He(1e-2))
    return model


# ------------------------------------------------------------------------------------------------ trainer
class nnUNetTrainerMI355(object):
    def __init__(self, plans: dict, configuration: str, fold: int, dataset_json: dict, unpack_dataset: bool = True,
                 device: torch.device = torch.device('cuda'), specified_cfg: str = ''):
        self.is_ddp = dist.is_available() and dist.is_initialized()
        self.local_rank = 0 if not self.is_ddp else dist.get_rank()
        self.device = device
        if self.device.type != 'cuda':
            raise RuntimeError("nnUNetTrainerMI355 needs an MI355X (device type 'cuda'); there is no CPU path")
        self.plans_manager = PlansManager(plans)
        self.configuration_manager = self.plans_manager.get_configuration(configuration)
        self.configuration_name = configuration
        self.dataset_json = dataset_json
        self.fold = fold
        self.unpack_dataset = unpack_dataset
        self.specified_cfg = specified_cfg
        # nnUNetTrainer.py:143-149
        self.initial_lr = 1e-2
        self.weight_decay = 3e-5
        self.oversample_foreground_percent = 0.33
        self.num_iterations_per_epoch = 250
        self.num_val_iterations_per_epoch = 50
        self.num_epochs = 200
        self.current_epoch = 0
        self.enable_deep_supervision = True
        self.label_manager = self.plans_manager.get_label_manager(dataset_json)
        self.num_input_channels = None
        self.network = None
        self.optimizer = self.lr_scheduler = None
        self.grad_scaler = None  # fp32 path (the reference's CPU branch: no autocast, no GradScaler, :906,:921-924)
        # "bf16": the reference's autocast path (:906) restated for MI355X -- bf16 activations on the bf16 MFMA engine,
        # fp32 master weights / statistics / losses / optimizer, no GradScaler (network.set_precision)
        self.precision = 'fp32'
        self.loss = None
        self.reducer = None
        self.ddp_bucket_bytes = 25 * 1024 * 1024  # DDP's default bucket_cap_mb (nnUNetTrainer.py:222 passes none)
        self.was_initialized = False
        self.batch_size = None
        # train_step replayed as one hipGraph launch after `hip_graph_warmup` eager steps (MVD_HIPGRAPH=0: always eager)
        self.use_hip_graph = os.environ.get("MVD_HIPGRAPH", "1") != "0"
        self.hip_graph_warmup = 3
        self._step_graph = None

    # -- plugin surface -------------------------------------------------------------------------------------
    @staticmethod
    def build_network_architecture(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                   enable_deep_supervision: bool = True) -> nn.Module:
        return get_network_from_plans(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                      deep_supervision=enable_deep_supervision)

    def initialize(self):
        if self.was_initialized:
            raise RuntimeError("You have called self.initialize even though the trainer was already initialized. "
                               "That should not happen.")
        self.num_input_channels = determine_num_input_channels(self.plans_manager, self.configuration_manager,
                                                               self.dataset_json)
        self.network = self.build_network_architecture(self.plans_manager, self.dataset_json,
                                                       self.configuration_manager, self.num_input_channels,
                                                       self.enable_deep_supervision).to(self.device)
        set_precision(self.network, self.precision)
        self.optimizer, self.lr_scheduler = self.configure_optimizers()
        if self.use_hip_graph:
            self.optimizer.fp.pack16_inplace = True   # the captured repack rewrites the buffers the captured forward reads
        if self.is_ddp:
            # DDP(network): broadcast rank 0's weights, then reduce gradients bucket-wise during backward (:220-222)
            broadcast_parameters(self.optimizer.fp)
            self.reducer = BucketedGradReducer(self.optimizer.fp, self.ddp_bucket_bytes, optimizer=self.optimizer)
        self.loss = self._build_loss()
        self._set_batch_size_and_oversample()
        self.was_initialized = True

    def configure_optimizers(self):
        # flat buffers in forward-execution order (network.parameters_in_execution_order): the gradient all-reduce buckets --
        # contiguous slices cut from the end -- then complete in backward order (parallel.BucketedGradReducer)
        order = getattr(self.network, "parameters_in_execution_order", None)
        params = order() if order is not None else list(self.network.parameters())
        optimizer = FusedSGDNesterov(FlatParams(params), self.initial_lr,
                                     weight_decay=self.weight_decay, momentum=0.99, nesterov=True, max_grad_norm=12)
        lr_scheduler = PolyLRScheduler(optimizer, self.initial_lr, self.num_epochs)
        return optimizer, lr_scheduler

    def _get_deep_supervision_scales(self):
        if self.enable_deep_supervision:
            return list(list(i) for i in 1 / np.cumprod(np.vstack(
                self.configuration_manager.pool_op_kernel_sizes), axis=0))[:-1]
        return None

    def _set_batch_size_and_oversample(self):
        if not self.is_ddp:
            self.batch_size = self.configuration_manager.batch_size
        else:
            bs, ov = ddp_batch_split(self.configuration_manager.batch_size, dist.get_world_size(),
                                     self.oversample_foreground_percent)
            self.batch_size = bs[dist.get_rank()]
            self.oversample_foreground_percent = ov[dist.get_rank()]

    def _build_loss(self):
        loss = losses.DC_and_CE_loss({'batch_dice': self.configuration_manager.batch_dice, 'smooth': 1e-5,
                                      'do_bg': False, 'ddp': self.is_ddp}, {}, weight_ce=1, weight_dice=1,
                                     ignore_label=self.label_manager.ignore_label,
                                     dice_class=losses.MemoryEfficientSoftDiceLoss)
        if self.enable_deep_supervision:
            deep_supervision_scales = self._get_deep_supervision_scales()
            weights = np.array([1 / (2 ** i) for i in range(len(deep_supervision_scales))])
            weights[-1] = 0
            weights = weights / weights.sum()
            loss = losses.DeepSupervisionWrapper(loss, weights)
        return loss

    def set_deep_supervision_enabled(self, enabled: bool):
        self.network.decoder.deep_supervision = enabled

    def on_train_epoch_start(self):
        self.network.train()
        self.lr_scheduler.step(self.current_epoch)

    # -- synthetic batch (the reference's own benchmark harness) ---------------------------------------------
    def make_dummy_batch(self, seed=None):
        """nnUNetTrainerBenchmark_5epochs_noDataLoading.py:16-22."""
        g = torch.Generator(device='cpu')
        g.manual_seed(1234 + self.local_rank if seed is None else seed)
        patch_size = self.configuration_manager.patch_size
        data = torch.rand((self.batch_size, self.num_input_channels, *patch_size), generator=g)
        target = [torch.round(torch.rand((self.batch_size, 1, *[int(i * j) for i, j in zip(patch_size, k)]),
                                         generator=g) * max(self.label_manager.all_labels))
                  for k in self._get_deep_supervision_scales()]
        return {'data': data.to(self.device), 'target': [t.to(self.device) for t in target]}

    # -- the hot path ---------------------------------------------------------------------------------------
    def _forward_loss(self, data, target):
        output = self.network(data)
        return self.loss(output, target), output

    def _step_body(self, data, target):
        """zero_grad -> forward -> loss -> backward -> (gradient all-reduce fence) -> clip + SGD (:901-924)."""
        if self.reducer is not None:
            self.reducer.reset()   # a backward() that raised last step must not leave stale bucket state behind
        self.optimizer.zero_grad(set_to_none=True)
        l, _ = self._forward_loss(data, target)
        l.backward()
        if self.reducer is not None:
            self.reducer.wait()
        self.optimizer.step()  # clip_grad_norm_(12) + SGD fused, clip coefficient stays on the device
        return l.detach()

    # -- the step as ONE hipGraph launch --------------------------------------------------------------------------
    # The ~400 kernels of a step are enqueued by Python in 11-12 ms; the bf16 step needs 13 ms of device time and the
    # reference's per-step `loss.cpu()` (:925) keeps the host from running ahead, so the device idles ~1.5 ms per step
    # between launches (profiles/r02_bf16_step_per_launch.txt: span - sum of durations).  After `hip_graph_warmup` eager
    # steps the step body is captured once per input geometry (torch.cuda.graph = hipGraph on ROCm) and replayed:
    # identical kernels, arguments and order -> results bit-identical to the eager step (tests/test_gpu_graph.py).
    # What keeps that valid: inputs are copied into static buffers; the optimizer's scalars live in device memory
    # (mvd_sgd_nesterov_step_dev: PolyLR needs no re-capture); the packed weight copies are rewritten in place by the
    # captured repack (FlatParams.pack16_inplace); nothing in the body synchronises or depends on host-side data.
    def _graph_allowed(self):
        if not self.use_hip_graph:
            return False
        if self.reducer is not None and self.reducer.world > 1:
            # collectives inside a capture: RCCL supports it, but it is unverified here on more than one GPU
            return os.environ.get("MVD_HIPGRAPH_DDP", "0") == "1" and dist.get_backend() == "nccl"
        return True

    def _graph_key(self, data, target):
        tl = target if isinstance(target, (list, tuple)) else [target]
        return (tuple(data.shape), data.dtype, tuple((tuple(t.shape), t.dtype) for t in tl), isinstance(target, list),
                self.precision, self.network.training, self._graph_flags())

    def _graph_flags(self):
        return (bool(self.network.decoder.deep_supervision),)

    def _graphed_step(self, data, target):
        key = self._graph_key(data, target)
        sg = self._step_graph
        if sg is None or sg['key'] != key:
            sg = self._step_graph = {'key': key, 'graph': None, 'warm': 0}
        if sg['graph'] is None:
            if sg['warm'] < self.hip_graph_warmup:
                sg['warm'] += 1   # eager first: allocator steady state, the optimizer's first-step flag, pack caches
                return self._step_body(data, target)
            sg['data'] = data.clone()
            sg['target'] = [t.clone() for t in target] if isinstance(target, list) else target.clone()
            self.optimizer.use_device_hyper()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    sg['loss'] = self._step_body(sg['data'], sg['target'])
            except Exception as e:  # leave the eager path intact and say so (no silent slow path)
                self.use_hip_graph = False
                self._step_graph = None
                torch.cuda.synchronize()
                import warnings
                warnings.warn(f"hipGraph capture of the train step failed ({type(e).__name__}: {e}); running eagerly")
                return self._step_body(data, target)
            sg['graph'] = g
            captured_now = True    # (the capture ran the body's Python once: the optimizer's step counter already moved)
        else:
            captured_now = False
            if sg['data'].data_ptr() != data.data_ptr():
                sg['data'].copy_(data, non_blocking=True)
            if isinstance(target, list):
                for st, t in zip(sg['target'], target):
                    if st.data_ptr() != t.data_ptr():
                        st.copy_(t, non_blocking=True)
            elif sg['target'].data_ptr() != target.data_ptr():
                sg['target'].copy_(target, non_blocking=True)
        self.optimizer.sync_hyper()
        if ops.packs_stale(self.optimizer.fp):   # load_state_dict / invalidate_packs since the last step: the captured
            ops.repack_all(self.optimizer.fp)    # forward reads the persistent pack buffers -> refresh them first
        sg['graph'].replay()
        if not captured_now:
            self.optimizer.note_replayed_step()
        return sg['loss']

    def train_step(self, batch: dict, return_device_loss: bool = False) -> dict:
        data, target = batch['data'], batch['target']
        data = data.to(self.device, non_blocking=True)
        if isinstance(target, list):
            target = [i.to(self.device, non_blocking=True) for i in target]
        else:
            target = target.to(self.device, non_blocking=True)
        l = self._graphed_step(data, target) if self._graph_allowed() else self._step_body(data, target)
        if return_device_loss:
            return {'loss': l}
        return {'loss': l.cpu().numpy()}  # the reference syncs here every step (:925)

    def validation_step(self, batch: dict) -> dict:
        data, target = batch['data'], batch['target']
        data = data.to(self.device, non_blocking=True)
        target = [i.to(self.device, non_blocking=True) for i in target] if isinstance(target, list) \
            else target.to(self.device, non_blocking=True)
        with torch.no_grad():
            l, output = self._forward_loss(data, target)
            if self.enable_deep_supervision:
                output, target = output[0], target[0]
            counts = ops.argmax_counts(output, target).cpu().numpy()
        tp_hard, fp_hard, fn_hard = counts[1:, 0], counts[1:, 1], counts[1:, 2]  # [1:] removes background (:996-1002)
        return {'loss': l.detach().cpu().numpy(), 'tp_hard': tp_hard, 'fp_hard': fp_hard, 'fn_hard': fn_hard}

    @staticmethod
    def dice_from_counts(tp, fp, fn):
        """on_validation_epoch_end :1033-1034."""
        with np.errstate(divide='ignore', invalid='ignore'):
            per_class = [2 * i / (2 * i + j + k) for i, j, k in zip(tp, fp, fn)]
        return per_class, float(np.nanmean(per_class))


class nnUNetTrainerMI355Benchmark_noDataLoading(nnUNetTrainerMI355):
    """variants/benchmarking/nnUNetTrainerBenchmark_5epochs_noDataLoading.py:8-51: constant synthetic batch."""

    def __init__(self, plans, configuration, fold, dataset_json, unpack_dataset=True, device=torch.device('cuda')):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device)
        self.num_epochs = 5
        self.dummy_batch = None

    def initialize(self):
        super().initialize()
        self.dummy_batch = self.make_dummy_batch()


# ------------------------------------------------------------------------------------------------ MVD dual branch
class ContrastiveTrainerMI355(nnUNetTrainerMI355):
    """Dual-branch mutual-distillation step (MVDTrainer.py:879-925):
    l = L(out1,t) + L(out2,t) + lambda3 * L_topo(out1[0][:,v], onehot(t)[:,v]) + lambda1 * L_KL."""

    def __init__(self, plans, configuration, fold, dataset_json, unpack_dataset=True, device=torch.device('cuda'),
                 specified_cfg=''):
        super().__init__(plans, configuration, fold, dataset_json, unpack_dataset, device, specified_cfg)
        self.lambda1, self.lambda2, self.lambda3 = 0.5, 0.1, 1  # MVDTrainer.py:132-134
        self.vessel_channel = 2                                 # :897-898, :907-908
        self.use_topo, self.skel_iter, self.feat_kl, self.kl_T = True, 3, True, 1
        self.topo_cc = True          # with use_topo: the integer connected-component count beside the soft-clDice term
        self.last_topology = None

    @staticmethod
    def build_network_architecture(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                   enable_deep_supervision: bool = True) -> nn.Module:
        b1 = get_network_from_plans(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                    enable_deep_supervision)
        b2 = get_network_from_plans(plans_manager, dataset_json, configuration_manager, num_input_channels,
                                    enable_deep_supervision)
        return MVDDualBranchNet(b1, b2)

    def set_deep_supervision_enabled(self, enabled: bool):
        self.network.do_ds = enabled  # MVDTrainer.py:802-806

    def _forward_loss(self, data, target):
        o1, o2, f1, f2 = self.network(data)
        v = self.vessel_channel
        l = self.loss(o1, target) + self.loss(o2, target)
        top1 = o1[0] if isinstance(o1, (list, tuple)) else o1
        top2 = o2[0] if isinstance(o2, (list, tuple)) else o2
        tgt0 = target[0] if isinstance(target, (list, tuple)) else target
        mutual = losses.kl_loss_compute1(top1[:, v], top2[:, v], self.kl_T)
        if self.feat_kl:
            mutual = mutual + losses.l2_loss(f1, f2, channel_wise=True, T=self.kl_T)
        l = l + self.lambda1 * mutual
        if self.use_topo:
            prob = ops.SoftmaxSelectFn.apply(top1, v)
            tmask = ops.label_mask(tgt0, v).reshape(prob.shape)
            l = l + self.lambda3 * losses.soft_cldice(prob, tmask, self.skel_iter)
            if self.topo_cc:
                self.last_topology = self._component_counts(prob.detach(), tmask)
        return l, o1

    def _component_counts(self, prob, tmask):
        """The integer topology step of configs[3] ("cfg 3 + soft-clDice + CC count"; the reference's
        per-step cubical-complex pass MVDTrainer.py:907-923 runs on the CPU): number of connected components (= Betti-0,
        26-connectivity: voxels as closed top-dimensional cells) of the predicted vessel mask {softmax >= 0.5} and of the
        label's vessel mask, per sample, on the device (mvd_cc_label: union-find, bit-exact against oracle/cc_oracle.c).
        Stays on the device (no sync): {'cc_pred', 'cc_true', 'betti0_error'} int32 [N]."""
        N = prob.shape[0]
        cp, ct = [], []
        for n in range(N):
            cp.append(ops.cc_label(ops.threshold_mask(prob[n, 0], 0.5, ge=True), 26)[1])
            ct.append(ops.cc_label(ops.threshold_mask(tmask[n, 0], 0.5, ge=True), 26)[1])
        cp, ct = torch.cat(cp), torch.cat(ct)
        return {'cc_pred': cp, 'cc_true': ct, 'betti0_error': (cp - ct).abs()}

    def _graph_flags(self):
        return (bool(self.network.do_ds), self.use_topo, self.topo_cc, self.skel_iter, self.feat_kl, self.kl_T,
                self.lambda1, self.lambda3, self.vessel_channel)
