"""MI355X-native PlainConvUNet behind the reference's network_architecture plugin surface.

Mirrors (names, constructor arguments, attributes, state_dict keys -- SURVEY.md App. C) the classes the reference
instantiates through nnUNet/nnunetv2/utilities/get_network_from_plans.py:70-83 (`PlainConvUNet` of the un-vendored
`dynamic_network_architectures`) and its fork-local decoder nnUNet/nnunetv2/training/my_network/UNetDecoder.py:13-121.
Every tensor operation is a hand-written HIP kernel reached through the C ABI (ops.py); the modules subclass the
torch.nn layer types only so that parameter registration, `state_dict()` keys, `InitWeights_He`
(network_initialization.py:4-12, which dispatches on isinstance(nn.Conv3d / nn.ConvTranspose3d)) and DDP-style
parameter handling behave exactly like the reference's modules.
"""
import os
from typing import List, Sequence, Tuple, Type, Union

import numpy as np
import torch
from torch import nn

from . import ops


# InstanceNorm-apply + LeakyReLU in the consumer conv's loader (bf16, z-marching kernel): on in inference; under autograd
# ("auto", the default) where the weight-gradient kernel has the same prologue (k_wgrad16z: the activated tensor is then
# written in neither pass), "1" = wherever the forward kernel takes it (the weight gradient of other shapes re-materialises
# the tensor), "0" = never
# gradient contributions of a tensor with two consumers summed inside the second consumer's kernel (ops._GradShare)
SHARE_GRADS = [os.environ.get("MVD_SHARE_GRADS", "1") != "0"]
FUSE_PROLOGUE = [os.environ.get("MVD_FUSE_PROLOGUE", "1") != "0"]
FUSE_PROLOGUE_TRAIN = [os.environ.get("MVD_FUSE_PROLOGUE_TRAIN", "auto")]
# the last decoder block's InstanceNorm + LeakyReLU inside the seg head that is its only consumer (ops.NormActSegHeadFn)
FUSE_SEGHEAD = [os.environ.get("MVD_FUSE_SEGHEAD", "1") != "0"]


def _tup3(v):
    if isinstance(v, (int, np.integer)):
        return (int(v),) * 3
    return tuple(int(i) for i in v)


class HipConv3d(nn.Conv3d):
    """nn.Conv3d whose forward is mvd_conv3d_fwd (NDHWC, fp32 MFMA implicit GEMM).  `x2` (optional) is a second
    input concatenated along channels without materialising the cat."""

    def forward(self, x, x2=None):
        return ops.Conv3dFn.apply(x, x2, self.weight, self.bias, self.stride)


class HipConvTranspose3d(nn.ConvTranspose3d):
    def forward(self, x):
        return ops.ConvTranspose3dFn.apply(x, self.weight, self.bias, self.stride)


class HipInstanceNorm3d(nn.InstanceNorm3d):
    """Stand-alone InstanceNorm3d is never used on the path; inside ConvDropoutNormReLU it is fused with the
    LeakyReLU (mvd_instnorm_lrelu_fwd).  Calling it alone applies slope 1 (identity activation)."""

    def forward(self, x):
        return ops.InstanceNormLeakyReLUFn.apply(x, self.weight, self.bias, self.eps, 1.0)


class HipSegLayer(nn.Conv3d):
    """1x1x1 Conv3d producing planar logits (decoder.seg_layers[s])."""

    def forward(self, x):
        return ops.SegHeadFn.apply(x, self.weight, self.bias)


class ConvDropoutNormReLU(nn.Module):
    """conv -> (no dropout) -> InstanceNorm3d -> LeakyReLU; kwargs as in get_network_from_plans.py:39-45."""

    def __init__(self, conv_op, input_channels, output_channels, kernel_size, stride, conv_bias=False, norm_op=None,
                 norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None, nonlin_kwargs=None,
                 nonlin_first=False):
        super().__init__()
        if dropout_op is not None:
            raise NotImplementedError("dropout is not on the reference's path (get_network_from_plans.py:43)")
        if nonlin_first:
            raise NotImplementedError("nonlin_first=False on the reference's path")
        # the fused kernel IS InstanceNorm3d(affine) + LeakyReLU (get_network_from_plans.py:41-44): anything else in the
        # plans would silently compute something different, so refuse it
        if norm_op is not None and not (isinstance(norm_op, type) and issubclass(norm_op, nn.InstanceNorm3d)):
            raise NotImplementedError(f"norm_op {norm_op!r}: only nn.InstanceNorm3d is on the reference's path")
        if norm_op is None:
            raise NotImplementedError("norm_op=None: the fused conv block always normalises (InstanceNorm3d)")
        if nonlin is not None and not (isinstance(nonlin, type) and issubclass(nonlin, nn.LeakyReLU)):
            raise NotImplementedError(f"nonlin {nonlin!r}: only nn.LeakyReLU is on the reference's path")
        if nonlin is None:
            raise NotImplementedError("nonlin=None: the fused conv block always applies LeakyReLU")
        if not dict(norm_op_kwargs or {'affine': True}).get('affine', False):
            raise NotImplementedError("InstanceNorm3d(affine=False) is not on the reference's path")
        self.input_channels, self.output_channels = input_channels, output_channels
        self.stride = _tup3(stride)
        k = _tup3(kernel_size)
        self.conv = HipConv3d(input_channels, output_channels, k, self.stride, padding=[(i - 1) // 2 for i in k],
                              dilation=1, bias=conv_bias)
        norm_op_kwargs = norm_op_kwargs or {'eps': 1e-5, 'affine': True}
        self.norm = HipInstanceNorm3d(output_channels, **norm_op_kwargs)
        nonlin_kwargs = dict(nonlin_kwargs or {'inplace': True})
        self.nonlin = nonlin(**nonlin_kwargs)
        self.all_modules = nn.Sequential(self.conv, self.norm, self.nonlin)
        self.precision = "fp32"  # "bf16": see set_precision()

    def conv_only(self, x, x2=None):
        """The block's convolution alone (raw output; in bf16 with the InstanceNorm statistics from the conv's epilogue
        attached when the kernel emits them)."""
        cw = self.conv.weight
        if (self.precision == "bf16" and x2 is None and x.dtype == torch.float32 and x.shape[1] <= 8 and
                cw.shape[0] % 32 == 0 and tuple(cw.shape[2:]) == (3, 3, 3) and self.stride == (1, 1, 1) and x.is_cuda):
            # the 4-modality input layer under mixed precision: bf16 operands like every other conv of the net
            return ops.NarrowInputConv3dBf16Fn.apply(x, cw, self.conv.bias)
        return self.conv(x, x2)

    def norm_act(self, y):
        return ops.InstanceNormLeakyReLUFn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps,
                                                 self.nonlin.negative_slope, self.precision == "bf16")

    def forward(self, x, x2=None):
        return self.norm_act(self.conv_only(x, x2))

    def forward_from_raw(self, prev, y_raw):
        """This block's conv fed with the RAW conv output of the previous block `prev`: prev's InstanceNorm + LeakyReLU run
        inside this conv's loader (ops.NormActConv3dFn).  Returns this block's raw conv output."""
        return ops.NormActConv3dFn.apply(y_raw, prev.norm.weight, prev.norm.bias, prev.norm.eps,
                                         prev.nonlin.negative_slope, self.conv.weight, self.conv.bias)

    def compute_conv_feature_map_size(self, input_size):
        output_size = [i // j for i, j in zip(input_size, self.stride)]
        return np.prod([self.output_channels, *output_size], dtype=np.int64)


class StackedConvBlocks(nn.Module):
    def __init__(self, num_convs, conv_op, input_channels, output_channels, kernel_size, initial_stride,
                 conv_bias=False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, nonlin_first=False):
        super().__init__()
        if not isinstance(output_channels, (tuple, list)):
            output_channels = [output_channels] * num_convs
        self.convs = nn.Sequential(
            ConvDropoutNormReLU(conv_op, input_channels, output_channels[0], kernel_size, initial_stride, conv_bias,
                                norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                                nonlin_first),
            *[ConvDropoutNormReLU(conv_op, output_channels[i - 1], output_channels[i], kernel_size, 1, conv_bias,
                                  norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs,
                                  nonlin_first) for i in range(1, num_convs)])
        self.output_channels = output_channels[-1]
        self.initial_stride = _tup3(initial_stride)

    def forward(self, x, x2=None, raw_tail=False):
        """conv -> norm -> act per block.  raw_tail (the last decoder stage in front of its seg head, bf16): the LAST block's
        norm + act are left to the consumer -- returns (raw conv output, that block); None when the chain did not end fused.
        bf16 mixed precision, consecutive blocks whose second conv takes the z-marching
        kernel (3x3x3, stride 1, 32 -> 32 channels at the patch resolution): the first block's InstanceNorm + LeakyReLU is
        folded into the second conv's loader (no apply pass, the activated tensor is never written) --
        in inference and, where the weight-gradient kernel has the same prologue, in training (FUSE_PROLOGUE_TRAIN)."""
        blocks = list(self.convs)
        raw = None   # the raw conv output of the previous block when its norm + act are still pending
        for i, blk in enumerate(blocks):
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            fuse_next = nxt is not None and self._can_fuse(blk, nxt, x if raw is None else raw)
            tail = raw_tail and nxt is None and FUSE_PROLOGUE[0]
            if raw is None and not fuse_next and not tail:
                x = blk(x, x2) if i == 0 else blk(x)      # the plain module call (forward hooks fire)
                continue
            y = blk.forward_from_raw(blocks[i - 1], raw) if raw is not None else \
                (blk.conv_only(x, x2) if i == 0 else blk.conv_only(x))
            if fuse_next and ops.fused_norm_conv_ok(y, nxt.conv.weight, nxt.stride):
                raw = y
                continue
            if tail:
                return y, blk
            raw = None
            x = blk.norm_act(y)
        return (x, None) if raw_tail else x

    @staticmethod
    def _can_fuse(blk, nxt, inp):
        """Shape-only test (before anything runs) whether blk's InstanceNorm + LeakyReLU can ride in nxt's conv loader."""
        train = torch.is_grad_enabled()
        mode = str(FUSE_PROLOGUE_TRAIN[0]).lower()
        if not (blk.precision == "bf16" and FUSE_PROLOGUE[0]) or (train and mode in ("0", "false")):
            return False
        w = nxt.conv.weight
        if tuple(w.shape[2:]) != (3, 3, 3) or nxt.stride != (1, 1, 1) or w.shape[1] != blk.output_channels or not inp.is_cuda:
            return False
        sp = [(d + 2 * ((k - 1) // 2) - k) // st + 1 for d, k, st in zip(inp.shape[2:], blk.conv.kernel_size, blk.stride)]
        shape = (inp.shape[0], sp[0], sp[1], sp[2], blk.output_channels, 0, w.shape[0], ops.i3((3, 3, 3)), ops.i3((1, 1, 1)))
        if ops.query("mvd_conv3d_fwd_bf16_prologue_ok", *shape) <= 0:
            return False
        if blk.output_channels > 32 and os.environ.get("MVD_FUSE_PROLOGUE_C64", "1") == "0":   # (A/B: the 64-channel blocks)
            return False
        if train and mode not in ("1", "true"):   # "auto": only where the weight gradient has the prologue too
            return ops.WGRAD_PROLOGUE and ops.query("mvd_conv3d_wgrad_bf16_prologue_ok", *shape) > 0
        return True

    def compute_conv_feature_map_size(self, input_size):
        output = self.convs[0].compute_conv_feature_map_size(input_size)
        size_after_stride = [i // j for i, j in zip(input_size, self.initial_stride)]
        for b in self.convs[1:]:
            output += b.compute_conv_feature_map_size(size_after_stride)
        return output


class PlainConvEncoder(nn.Module):
    def __init__(self, input_channels: int, n_stages: int, features_per_stage: Union[int, Sequence[int]],
                 conv_op: Type = nn.Conv3d, kernel_sizes=3, strides=1, n_conv_per_stage: Union[int, Sequence[int]] = 2,
                 conv_bias: bool = False, norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None,
                 nonlin=None, nonlin_kwargs=None, return_skips: bool = False, nonlin_first: bool = False,
                 pool: str = 'conv'):
        super().__init__()
        if conv_op is not nn.Conv3d:
            raise NotImplementedError("3d_fullres path only (conv_op = nn.Conv3d)")
        if pool != 'conv':
            raise NotImplementedError("the reference's plans use strided convolutions (pool='conv')")
        if isinstance(kernel_sizes, int):
            kernel_sizes = [kernel_sizes] * n_stages
        if isinstance(features_per_stage, int):
            features_per_stage = [features_per_stage] * n_stages
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * n_stages
        if isinstance(strides, int):
            strides = [strides] * n_stages
        assert len(kernel_sizes) == n_stages and len(n_conv_per_stage) == n_stages
        assert len(features_per_stage) == n_stages and len(strides) == n_stages
        stages = []
        for s in range(n_stages):
            stages.append(nn.Sequential(StackedConvBlocks(
                n_conv_per_stage[s], conv_op, input_channels, features_per_stage[s], kernel_sizes[s], strides[s],
                conv_bias, norm_op, norm_op_kwargs, dropout_op, dropout_op_kwargs, nonlin, nonlin_kwargs, nonlin_first)))
            input_channels = features_per_stage[s]
        self.stages = nn.Sequential(*stages)
        self.output_channels = list(features_per_stage)
        self.strides = [_tup3(i) for i in strides]
        self.return_skips = return_skips
        # read by the decoder (UNetDecoder.py:39-65)
        self.conv_op, self.norm_op, self.norm_op_kwargs = conv_op, norm_op, norm_op_kwargs
        self.nonlin, self.nonlin_kwargs = nonlin, nonlin_kwargs
        self.dropout_op, self.dropout_op_kwargs = dropout_op, dropout_op_kwargs
        self.conv_bias, self.kernel_sizes = conv_bias, [_tup3(k) for k in kernel_sizes]

    def forward(self, x):
        ret = []
        last = len(self.stages) - 1
        for i, s in enumerate(self.stages):
            x = s(x)
            if self.return_skips and i != last and SHARE_GRADS[0]:
                ops.share_grad(x)   # a skip: read by the next stage AND by the decoder -> one gradient buffer (ops._GradShare)
            ret.append(x)
        return ret if self.return_skips else ret[-1]

    def compute_conv_feature_map_size(self, input_size):
        output = np.int64(0)
        for s in range(len(self.stages)):
            output += self.stages[s][-1].compute_conv_feature_map_size(input_size)
            input_size = [i // j for i, j in zip(input_size, self.strides[s])]
        return output


class UNetDecoder(nn.Module):
    """UNetDecoder.py:13-121 (without the fork's bottleneck attention insert :75-81,:91-102): per stage
    transpconv -> [cat] -> stacked convs -> 1x1x1 seg layer; outputs high-res first."""

    def __init__(self, encoder: PlainConvEncoder, num_classes: int,
                 n_conv_per_stage: Union[int, Tuple[int, ...], List[int]], deep_supervision,
                 nonlin_first: bool = False):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder
        self.num_classes = num_classes
        n_stages_encoder = len(encoder.output_channels)
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * (n_stages_encoder - 1)
        assert len(n_conv_per_stage) == n_stages_encoder - 1
        stages, transpconvs, seg_layers = [], [], []
        for s in range(1, n_stages_encoder):
            below = encoder.output_channels[-s]
            skip = encoder.output_channels[-(s + 1)]
            st = encoder.strides[-s]
            transpconvs.append(HipConvTranspose3d(below, skip, st, st, bias=encoder.conv_bias))
            stages.append(StackedConvBlocks(
                n_conv_per_stage[s - 1], encoder.conv_op, 2 * skip, skip, encoder.kernel_sizes[-(s + 1)], 1,
                encoder.conv_bias, encoder.norm_op, encoder.norm_op_kwargs, encoder.dropout_op,
                encoder.dropout_op_kwargs, encoder.nonlin, encoder.nonlin_kwargs, nonlin_first))
            seg_layers.append(HipSegLayer(skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(transpconvs)
        self.seg_layers = nn.ModuleList(seg_layers)

    def forward(self, skips, return_last_feature=False):
        """Bottom-up: up-sample, fuse with the encoder feature of that resolution, refine, emit logits.  Same contract
        as UNetDecoder.py:104-121: list of logits, highest resolution first, with deep supervision; one tensor without."""
        feat = skips[-1]
        last = len(self.stages) - 1
        logits = []
        for level, (up, refine, head) in enumerate(zip(self.transpconvs, self.stages, self.seg_layers)):
            # the concatenation of (up-sampled, skip) is never materialised: the first conv reads both pointers
            head_l = head if self.deep_supervision else self.seg_layers[-1]
            if level == last and not return_last_feature and FUSE_SEGHEAD[0] and isinstance(refine, StackedConvBlocks):
                # the top stage's output feeds the seg head alone: its last InstanceNorm + LeakyReLU ride in the head's loaders
                y_raw, blk = refine(up(feat), skips[-(level + 2)], raw_tail=True)
                if blk is not None and ops.fused_norm_seghead_ok(y_raw, head_l.weight):
                    logits.append(ops.NormActSegHeadFn.apply(y_raw, blk.norm.weight, blk.norm.bias, blk.norm.eps,
                                                             blk.nonlin.negative_slope, head_l.weight, head_l.bias))
                    feat = None
                    continue
                feat = y_raw if blk is None else blk.norm_act(y_raw)
            else:
                feat = refine(up(feat), skips[-(level + 2)])
            if self.deep_supervision and level != last and SHARE_GRADS[0]:
                ops.share_grad(feat)   # two consumers (this level's seg head, the next level's up-sampling): one gradient buffer
            if self.deep_supervision:
                logits.append(head(feat))
            elif level == last:
                logits.append(self.seg_layers[-1](feat))
        out = logits[::-1] if self.deep_supervision else logits[0]
        return (out, feat) if return_last_feature else out

    def compute_conv_feature_map_size(self, input_size):
        """Number of feature-map elements the decoder produces for one sample (the planner's VRAM proxy,
        UNetDecoder.py:123-150): per resolution the refined maps, the up-sampled map and the logits that are emitted."""
        sizes, cur = [], list(input_size)
        for st in self.encoder.strides[:-1]:
            cur = [i // j for i, j in zip(cur, st)]
            sizes.append(cur)
        total = np.int64(0)
        n = len(self.stages)
        for level in range(n):
            sp = sizes[-(level + 1)]
            voxels = np.prod(sp, dtype=np.int64)
            total += self.stages[level].compute_conv_feature_map_size(sp)
            total += self.encoder.output_channels[-(level + 2)] * voxels           # transposed-conv output
            if self.deep_supervision or level == n - 1:
                total += self.num_classes * voxels
        return total


class MI355PlainConvUNet(nn.Module):
    """Drop-in for dynamic_network_architectures.architectures.unet.PlainConvUNet (same constructor signature)."""

    def __init__(self, input_channels: int, n_stages: int, features_per_stage: Union[int, Sequence[int]],
                 conv_op: Type = nn.Conv3d, kernel_sizes=3, strides=1, n_conv_per_stage: Union[int, Sequence[int]] = 2,
                 num_classes: int = 2, n_conv_per_stage_decoder: Union[int, Sequence[int]] = 2, conv_bias: bool = False,
                 norm_op=None, norm_op_kwargs=None, dropout_op=None, dropout_op_kwargs=None, nonlin=None,
                 nonlin_kwargs=None, deep_supervision: bool = False, nonlin_first: bool = False):
        super().__init__()
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * n_stages
        if isinstance(n_conv_per_stage_decoder, int):
            n_conv_per_stage_decoder = [n_conv_per_stage_decoder] * (n_stages - 1)
        assert len(n_conv_per_stage) == n_stages
        assert len(n_conv_per_stage_decoder) == (n_stages - 1)
        self.encoder = PlainConvEncoder(input_channels, n_stages, features_per_stage, conv_op, kernel_sizes, strides,
                                        n_conv_per_stage, conv_bias, norm_op, norm_op_kwargs, dropout_op,
                                        dropout_op_kwargs, nonlin, nonlin_kwargs, return_skips=True,
                                        nonlin_first=nonlin_first)
        self.decoder = UNetDecoder(self.encoder, num_classes, n_conv_per_stage_decoder, deep_supervision,
                                   nonlin_first=nonlin_first)

    def forward(self, x, return_last_feature=False):
        if not x.is_cuda:
            raise RuntimeError("MI355PlainConvUNet runs on the MI355X only (no CPU path); move the input to cuda")
        skips = self.encoder(ops.to_ndhwc(x))
        return self.decoder(skips, return_last_feature)

    def compute_conv_feature_map_size(self, input_size):
        return self.encoder.compute_conv_feature_map_size(input_size) + \
            self.decoder.compute_conv_feature_map_size(input_size)

    def parameters_in_execution_order(self):
        """Parameters in the order the forward pass first uses them: encoder stages, then per decoder level the transposed
        conv, the refining convs and the seg layer.  `parameters()` lists them by module type (all stages, all transposed
        convs, all seg layers).  optim.FlatParams lays the flat gradient buffer out in THIS order, so that a suffix of the
        buffer is complete exactly when backward has passed the corresponding part of the network: the reducer's buckets
        (contiguous slices, cut from the end) then fire in backward order while the rest of backward still runs."""
        out = list(self.encoder.parameters())
        for up, refine, head in zip(self.decoder.transpconvs, self.decoder.stages, self.decoder.seg_layers):
            out += list(up.parameters()) + list(refine.parameters()) + list(head.parameters())
        seen, uniq = set(), []
        for p in out + list(self.parameters()):   # (anything not covered above keeps its registration order at the end)
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        return uniq


def set_precision(module: nn.Module, precision: str):
    """Mixed precision of the reference's autocast path (nnUNetTrainer.py:906; BASELINE cfg 4/5), MI355X style:
    "bf16" makes every fused InstanceNorm+LeakyReLU emit bf16, so every conv / transposed conv / seg head reads bf16
    activations and runs on the bf16 MFMA engine (fp32 accumulate); the 4-modality input conv converts its fp32 input to
    bf16 itself (ops.NarrowInputConv3dBf16Fn: channels zero-padded to 32), as autocast does.  Parameters, normalisation
    statistics, logits, losses, weight gradients and the optimizer stay fp32 -- bf16 has fp32's exponent range, so there
    is no GradScaler (the reference needs one for fp16, nnUNetTrainer.py:916-920)."""
    if precision not in ("fp32", "bf16"):
        raise ValueError(f"precision must be 'fp32' or 'bf16', got {precision!r}")
    for m in module.modules():
        if isinstance(m, ConvDropoutNormReLU):
            m.precision = precision
    return module


PlainConvUNet = MI355PlainConvUNet  # the name get_network_from_plans.py:35 maps 'PlainConvUNet' to


class InitWeights_He(object):
    """network_initialization.py:4-12 (host-side, runs on whatever device the module lives on)."""

    def __init__(self, neg_slope=1e-2):
        self.neg_slope = neg_slope

    def __call__(self, module):
        if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            module.weight = nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
            if module.bias is not None:
                module.bias = nn.init.constant_(module.bias, 0)


class MVDDualBranchNet(nn.Module):
    """Dual-branch contract of the mutual-distillation trainer (HybridNetwork.py:1544-1571, MVDTrainer.py:895):
    forward -> (logits_list_1, logits_list_2, feat_1, feat_2); `do_ds` toggles deep supervision
    (MVDTrainer.py:802-806); with DS off single tensors are returned.  Each branch is a MI355PlainConvUNet; the
    feature map is the last decoder stage output (cf. UNetDecoder_return_last_fea, UNetDecoder.py:1012-1027)."""

    def __init__(self, branch1: MI355PlainConvUNet, branch2: MI355PlainConvUNet):
        super().__init__()
        self.branch1, self.branch2 = branch1, branch2

    @property
    def do_ds(self):
        return self.branch1.decoder.deep_supervision

    @do_ds.setter
    def do_ds(self, v):
        self.branch1.decoder.deep_supervision = v
        self.branch2.decoder.deep_supervision = v

    def forward(self, x):
        x = ops.to_ndhwc(x)
        o1, f1 = self.branch1(x, True)
        o2, f2 = self.branch2(x, True)
        return o1, o2, f1, f2

    def parameters_in_execution_order(self):
        return self.branch1.parameters_in_execution_order() + self.branch2.parameters_in_execution_order()
