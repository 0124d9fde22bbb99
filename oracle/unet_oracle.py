"""TEST INFRASTRUCTURE -- CPU (torch fp32) restatement of the reference's PlainConvUNet composition.

Parity status: the arithmetic lives in the un-vendored pip dependency `dynamic-network-architectures>=0.2`
(reference nnUNet/setup.py:15, unpinned version) -- "parity unpinned" for the *composition*; every
individual op is torch.nn (Conv3d / ConvTranspose3d / InstanceNorm3d / LeakyReLU), i.e. the same ATen CPU
kernels the reference's `-device cpu` path runs.  What the reference itself pins and this file follows:

* constructor kwargs: nnUNet/nnunetv2/utilities/get_network_from_plans.py:38-83
  (conv_bias=True, InstanceNorm3d{eps 1e-5, affine}, no dropout, LeakyReLU{inplace}, features
  min(base*2**i, max), strides = pool_op_kernel_sizes, first conv of a stage carries the stride)
* decoder algorithm: nnUNet/nnunetv2/training/my_network/UNetDecoder.py:46-74 (ctor), :104-121 (forward:
  transpconv -> cat((x, skip), 1) -> stacked convs -> 1x1x1 seg layer; outputs reversed high-res first;
  without deep supervision only the last seg layer runs and a bare tensor is returned)
* init: nnUNet/nnunetv2/utilities/network_initialization.py:4-12 (kaiming_normal_(a=1e-2) on Conv3d and
  ConvTranspose3d weights, biases 0) applied via model.apply (get_network_from_plans.py:89)
* module / state_dict names: encoder.stages.{s}.0.convs.{k}.{conv,norm,all_modules.{0,1}},
  decoder.encoder.* (back reference, UNetDecoder.py:37), decoder.stages.{s}.convs.{k}.*,
  decoder.transpconvs.{s}, decoder.seg_layers.{s}
"""
from typing import List, Sequence, Union

import numpy as np
import torch
from torch import nn


def _tup3(v):
    if isinstance(v, (int, np.integer)):
        return (int(v),) * 3
    return tuple(int(i) for i in v)


class ConvDropoutNormReLU(nn.Module):
    def __init__(self, cin, cout, kernel_size, stride, conv_bias=True, eps=1e-5, neg_slope=1e-2):
        super().__init__()
        k = _tup3(kernel_size)
        self.stride = _tup3(stride)
        self.conv = nn.Conv3d(cin, cout, k, self.stride, padding=[(i - 1) // 2 for i in k], dilation=1,
                              bias=conv_bias)
        self.norm = nn.InstanceNorm3d(cout, eps=eps, affine=True)
        self.nonlin = nn.LeakyReLU(neg_slope, inplace=True)
        self.all_modules = nn.Sequential(self.conv, self.norm, self.nonlin)

    def forward(self, x):
        return self.all_modules(x)


class StackedConvBlocks(nn.Module):
    def __init__(self, num_convs, cin, cout, kernel_size, initial_stride, conv_bias=True):
        super().__init__()
        self.convs = nn.Sequential(
            ConvDropoutNormReLU(cin, cout, kernel_size, initial_stride, conv_bias),
            *[ConvDropoutNormReLU(cout, cout, kernel_size, 1, conv_bias) for _ in range(1, num_convs)])
        self.output_channels = cout
        self.initial_stride = _tup3(initial_stride)

    def forward(self, x):
        return self.convs(x)


class PlainConvEncoder(nn.Module):
    def __init__(self, input_channels, n_stages, features_per_stage, kernel_sizes, strides, n_conv_per_stage,
                 conv_bias=True):
        super().__init__()
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * n_stages
        stages = []
        cin = input_channels
        for s in range(n_stages):
            stages.append(nn.Sequential(
                StackedConvBlocks(n_conv_per_stage[s], cin, features_per_stage[s], kernel_sizes[s], strides[s],
                                  conv_bias)))
            cin = features_per_stage[s]
        self.stages = nn.Sequential(*stages)
        self.output_channels = list(features_per_stage)
        self.strides = [_tup3(i) for i in strides]
        self.kernel_sizes = [_tup3(k) for k in kernel_sizes]
        self.conv_bias = conv_bias
        self.return_skips = True

    def forward(self, x):
        ret = []
        for s in self.stages:
            x = s(x)
            ret.append(x)
        return ret


class UNetDecoder(nn.Module):
    """UNetDecoder.py:13-121 without the fork's attention insert (:75-81, :91-102)."""

    def __init__(self, encoder: PlainConvEncoder, num_classes, n_conv_per_stage, deep_supervision):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.encoder = encoder
        self.num_classes = num_classes
        n_enc = len(encoder.output_channels)
        if isinstance(n_conv_per_stage, int):
            n_conv_per_stage = [n_conv_per_stage] * (n_enc - 1)
        stages, transpconvs, seg_layers = [], [], []
        for s in range(1, n_enc):
            below = encoder.output_channels[-s]
            skip = encoder.output_channels[-(s + 1)]
            st = encoder.strides[-s]
            transpconvs.append(nn.ConvTranspose3d(below, skip, st, st, bias=encoder.conv_bias))
            stages.append(StackedConvBlocks(n_conv_per_stage[s - 1], 2 * skip, skip,
                                            encoder.kernel_sizes[-(s + 1)], 1, encoder.conv_bias))
            seg_layers.append(nn.Conv3d(skip, num_classes, 1, 1, 0, bias=True))
        self.stages = nn.ModuleList(stages)
        self.transpconvs = nn.ModuleList(transpconvs)
        self.seg_layers = nn.ModuleList(seg_layers)

    def forward(self, skips, return_last_feature=False):
        lres_input = skips[-1]
        seg_outputs = []
        for s in range(len(self.stages)):
            x = self.transpconvs[s](lres_input)
            x = torch.cat((x, skips[-(s + 2)]), 1)
            x = self.stages[s](x)
            if self.deep_supervision:
                seg_outputs.append(self.seg_layers[s](x))
            elif s == (len(self.stages) - 1):
                seg_outputs.append(self.seg_layers[-1](x))
            lres_input = x
        seg_outputs = seg_outputs[::-1]
        r = seg_outputs if self.deep_supervision else seg_outputs[0]
        if return_last_feature:  # cf. UNetDecoder_return_last_fea, UNetDecoder.py:1012-1027
            return r, lres_input
        return r


class InitWeights_He(object):
    """network_initialization.py:4-12."""

    def __init__(self, neg_slope=1e-2):
        self.neg_slope = neg_slope

    def __call__(self, module):
        if isinstance(module, (nn.Conv3d, nn.Conv2d, nn.ConvTranspose2d, nn.ConvTranspose3d)):
            module.weight = nn.init.kaiming_normal_(module.weight, a=self.neg_slope)
            if module.bias is not None:
                module.bias = nn.init.constant_(module.bias, 0)


class PlainConvUNet(nn.Module):
    def __init__(self, input_channels: int, n_stages: int, features_per_stage: Sequence[int],
                 kernel_sizes, strides, n_conv_per_stage: Union[int, List[int]], num_classes: int,
                 n_conv_per_stage_decoder: Union[int, List[int]], conv_bias: bool = True,
                 deep_supervision: bool = True):
        super().__init__()
        self.encoder = PlainConvEncoder(input_channels, n_stages, features_per_stage, kernel_sizes, strides,
                                        n_conv_per_stage, conv_bias)
        self.decoder = UNetDecoder(self.encoder, num_classes, n_conv_per_stage_decoder, deep_supervision)

    def forward(self, x, return_last_feature=False):
        return self.decoder(self.encoder(x), return_last_feature)


def features_for(n_stages, base=32, max_features=320):
    """get_network_from_plans.py:73-74."""
    return [min(base * 2 ** i, max_features) for i in range(n_stages)]


def build_plainconv_unet(input_channels, num_classes, n_stages, strides, kernel_sizes=None, base=32,
                         max_features=320, n_conv_per_stage=2, n_conv_per_stage_decoder=2,
                         deep_supervision=True, seed=0, features_per_stage=None):
    """get_network_from_plans.py:15-92 restated for PlainConvUNet; He init with a fixed torch CPU seed."""
    if kernel_sizes is None:
        kernel_sizes = [[3, 3, 3]] * n_stages
    feats = features_per_stage or features_for(n_stages, base, max_features)
    torch.manual_seed(seed)
    m = PlainConvUNet(input_channels, n_stages, feats, kernel_sizes, strides, n_conv_per_stage, num_classes,
                      n_conv_per_stage_decoder, True, deep_supervision)
    m.apply(InitWeights_He(1e-2))
    return m


class DualBranchNet(nn.Module):
    """Build's restatement of the MVD dual-branch contract (HybridNetwork.py:1544-1571):
    returns (logits_list_1, logits_list_2, feat_1, feat_2); `do_ds` toggles deep supervision
    (MVDTrainer.py:802-806); with DS off single tensors are returned (HybridNetwork.py:1569-1571)."""

    def __init__(self, net1: PlainConvUNet, net2: PlainConvUNet):
        super().__init__()
        self.branch1 = net1
        self.branch2 = net2

    @property
    def do_ds(self):
        return self.branch1.decoder.deep_supervision

    @do_ds.setter
    def do_ds(self, v):
        self.branch1.decoder.deep_supervision = v
        self.branch2.decoder.deep_supervision = v

    def forward(self, x):
        o1, f1 = self.branch1(x, True)
        o2, f2 = self.branch2(x, True)
        return o1, o2, f1, f2


# topology used by the BASELINE configs (derived with the reference's network_topology.get_pool_and_conv_props,
# pinned by tests/golden/topology_props.json)
CONFIGS = {
    "cfg1": dict(input_channels=1, patch=(64, 64, 64), n_stages=5,
                 strides=[[1, 1, 1]] + [[2, 2, 2]] * 4),
    "cfg2": dict(input_channels=4, patch=(128, 128, 128), n_stages=6,
                 strides=[[1, 1, 1]] + [[2, 2, 2]] * 5),
    "cfg5": dict(input_channels=4, patch=(160, 160, 128), n_stages=6,
                 strides=[[1, 1, 1]] + [[2, 2, 2]] * 5),
}
