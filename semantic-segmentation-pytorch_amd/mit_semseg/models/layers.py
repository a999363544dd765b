"""nn.Module leaves of the MI355X build.  They own parameters with the reference's names/logical shapes
(so reference checkpoints load) and call the HIP operators in mit_semseg.ops."""
import math

import torch
import torch.nn as nn

from .. import ops
from ..lib.nn import SynchronizedBatchNorm2d

BatchNorm2d = SynchronizedBatchNorm2d


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class Conv2d(nn.Module):
    """Drop-in for the nn.Conv2d uses of the reference (square kernels, groups=1).  The weight keeps the
    logical [K,C,R,S] shape but is stored KRSC (channels_last) -- the layout the MFMA kernels read."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        for t in (self.kernel_size, self.stride, self.padding, self.dilation):
            if t[0] != t[1]:
                raise NotImplementedError('only square conv geometry is used by mit_semseg models')
        k = self.kernel_size[0]
        w = torch.empty(out_channels, k, k, in_channels).permute(0, 3, 1, 2)      # KRSC memory, KCRS shape
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        # torch nn.Conv2d default init (kaiming_uniform(a=sqrt(5)) + uniform bias)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return ops.conv2d(x, self.weight, self.bias, self.stride[0], self.padding[0], self.dilation[0])

    def extra_repr(self):
        return '{in_channels}, {out_channels}, kernel_size={kernel_size}, stride={stride}, padding={padding}, ' \
               'dilation={dilation}'.format(**self.__dict__) + (', bias=False' if self.bias is None else '')


class GroupedConv2d(Conv2d):
    """nn.Conv2d(..., groups=g) of the reference's MobileNetV2 (depthwise 3x3, mobilenet.py:48,60) and ResNeXt (g = 32,
    resnext.py:30-31).  FUNCTIONAL coverage on the validated dense kernels: the grouped weight ([K, C/g, R, S], the shape the
    reference checkpoints carry) is expanded to its block-diagonal dense form (zeros outside the group blocks contribute exact
    zeros) and runs through the same h2 convolution path; autograd carries the dense gradient back to the group blocks.
    g-fold redundant MFMA work -- these backbones are outside the BASELINE configs; a dedicated grouped kernel is the
    follow-up."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=False):
        if bias:
            raise NotImplementedError('grouped convolutions of the reference are bias-free')
        if in_channels % groups or out_channels % groups:
            raise ValueError('in_channels and out_channels must be divisible by groups')
        nn.Module.__init__(self)
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        k = self.kernel_size[0]
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, k, k))
        self.bias = None
        self._dense = None          # (parameter version, data_ptr, dense weight): reused under no_grad (evaluation)
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def dense_weight(self):
        """block-diagonal [K, C, R, S] weight in KRSC memory"""
        w = self.weight
        if not torch.is_grad_enabled() and self._dense is not None and self._dense[0] == w._version and \
                self._dense[1] == w.data_ptr():
            return self._dense[2]
        g, r = self.groups, self.kernel_size[0]
        kg, cg = self.out_channels // g, self.in_channels // g
        dense = w.new_zeros(g, kg, r, r, g, cg)
        idx = torch.arange(g, device=w.device)
        dense[idx, :, :, :, idx] = w.view(g, kg, cg, r, r).permute(0, 1, 3, 4, 2)
        dense = dense.view(self.out_channels, r, r, self.in_channels).permute(0, 3, 1, 2)
        if not torch.is_grad_enabled():
            self._dense = (w._version, w.data_ptr(), dense)
        return dense

    def train(self, mode=True):
        self._dense = None          # the fused SGD kernel updates parameters without bumping torch's version counter
        return super().train(mode)

    def forward(self, x):
        if ops.DEPTHWISE_DIRECT and self.groups == self.in_channels == self.out_channels and self.kernel_size == (3, 3) \
                and self.in_channels % 4 == 0:
            return ops.depthwise_conv3x3(x, self.weight, self.stride[0], self.padding[0], self.dilation[0])
        cg, kg = self.in_channels // self.groups, self.out_channels // self.groups
        if ops.GROUPED_DIRECT and self.kernel_size == (3, 3) and cg % 4 == 0 and kg % 4 == 0:
            return ops.grouped_conv3x3(x, self.weight, self.groups, self.stride[0], self.padding[0], self.dilation[0])
        return ops.conv2d(x, self.dense_weight(), None, self.stride[0], self.padding[0], self.dilation[0])

    def extra_repr(self):
        return Conv2d.extra_repr(self) + ', groups=%d' % self.groups


class ReLU6(nn.Module):
    """nn.ReLU6 (mobilenet.py:26,34): min(max(x, 0), 6).  Inside conv -> BN -> ReLU6 units the lower clamp runs fused in the
    BN kernel and only the upper clamp is applied here (`conv_bn_relu6`)."""

    def __init__(self, inplace=False):
        super().__init__()

    def forward(self, x):
        return ops.clamp_max(ops.add_act(x, torch.zeros_like(x), relu=True), 6.0)


def conv_bn_relu6(conv, bn, x):
    """conv -> BN -> ReLU6: the fused conv/BN/ReLU unit followed by the upper clamp (elementwise glue)"""
    return ops.clamp_max(conv_bn(conv, bn, x, relu=True), 6.0)


def conv_bn(conv, bn, x, residual=None, relu=False, passthrough=False, only_feeds=None):
    """bn(conv(x), residual=..., relu=...) -- the conv -> BN [-> +residual] [-> ReLU] unit every block of the reference
    is made of (resnet.py:72-92, models.py:160-167, hrnet.py:45-61) -- dispatched as ONE fused autograd node when the
    h2 path can run it (ops.conv_bn_act), else as the two modules.  passthrough=True returns (y, x'): x has a second consumer
    (the block's shortcut) and x' is the alias to hand to it.  only_feeds=<Conv2d>: the result's ONE consumer is that convolution
    (bn1 -> conv2, bn2 -> conv3 of a block): when it reads its input as planes, the fp32 copy of the result is not written."""
    if passthrough:
        if torch.is_grad_enabled() and x.requires_grad:
            # fork x so that the two gradients are added by the native add kernel instead of autograd's torch-side
            # accumulation; the planes this conv splits serve the shortcut's conv too
            xa, xb = ops.fork(x)
            y = conv_bn(conv, bn, xa, residual=residual, relu=relu, only_feeds=only_feeds)
            ops.share_planes(xa, xb)
            return y, xb
        return conv_bn(conv, bn, x, residual=residual, relu=relu, only_feeds=only_feeds), x
    if conv.bias is None and x.dim() == 4 and type(conv) is Conv2d and isinstance(bn, SynchronizedBatchNorm2d) and bn.affine:
        return ops.conv_bn_act(x, conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                               bn.num_batches_tracked, residual=residual, stride=conv.stride[0],
                               padding=conv.padding[0], dilation=conv.dilation[0], training=bn.training,
                               momentum=bn.momentum, eps=bn.eps, relu=relu,
                               planes_only=(only_feeds is not None and type(only_feeds) is Conv2d and only_feeds.bias is None and
                                            only_feeds.weight.shape[0] % 8 == 0 and not ops.reads_fp32_input(only_feeds)))
    return bn(conv(x), residual=residual, relu=relu)


class ReLU(nn.Module):
    """Standalone ReLU (only used where the reference applies it outside a conv-BN pair)."""

    def __init__(self, inplace=False):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        return ops.add_act(x, torch.zeros_like(x), relu=True)


class MaxPool3x3s2(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (resnet.py:109)."""
    kernel_size, stride, padding = 3, 2, 1

    def forward(self, x):
        return ops.max_pool_3x3_s2(x)


class AdaptiveAvgPool2d(nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        return ops.adaptive_avg_pool(x, self.output_size)


class Dropout2d(nn.Module):
    """nn.Dropout2d (models.py:460,464): per-(n,c) Bernoulli keep mask scaled by 1/(1-p), training only.
    `mask_override` ([N,C] multipliers) replays a fixed mask -- used by the parity tests."""

    def __init__(self, p=0.5):
        super().__init__()
        self.p = p
        self.mask_override = None

    def forward(self, x):
        if not self.training or (self.p == 0 and self.mask_override is None):
            return x
        if self.mask_override is not None:
            mask = self.mask_override
        else:
            mask = ops.dropout_mask(x.shape[0], x.shape[1], self.p, x.device)
        return ops.scale_nc(x, mask)


class ConvBNReLU(nn.Sequential):
    """Conv -> BN -> ReLU with the reference's Sequential child indices ('0' conv, '1' bn[, '2' relu]);
    executes as conv kernel + fused BN/ReLU kernels."""

    def __init__(self, conv, bn, relu=True, first_index=0):
        super().__init__()
        self.add_module(str(first_index), conv)
        self.add_module(str(first_index + 1), bn)
        if relu:
            self.add_module(str(first_index + 2), ReLU(inplace=True))   # executes fused in the BN kernel
        self._relu = relu
        self._conv, self._bn = str(first_index), str(first_index + 1)

    def forward(self, x, residual=None):
        return conv_bn(self._modules[self._conv], self._modules[self._bn], x, residual=residual, relu=self._relu)


def conv3x3_bn_relu(in_planes, out_planes, stride=1):
    """3x3 convolution + BN + relu (models.py:160-167)."""
    return ConvBNReLU(Conv2d(in_planes, out_planes, 3, stride=stride, padding=1, bias=False),
                      BatchNorm2d(out_planes))


# ------------------------------------------------------------------------------------------------
# the deep stem shared by the ResNet and ResNeXt backbones (resnet.py:100-109, resnext.py:69-78)
# ------------------------------------------------------------------------------------------------
_DEEP_STEM = ((3, 64, 2), (64, 64, 1), (64, 128, 1))      # (in, out, stride) of conv1..3, all 3x3 / pad 1 / no bias


def add_deep_stem(module):
    """registers conv{i} / bn{i} / relu{i} (i = 1..3) and `maxpool` on `module`, in the reference's order and names"""
    for i, (cin, cout, stride) in enumerate(_DEEP_STEM, 1):
        module.add_module('conv%d' % i, Conv2d(cin, cout, 3, stride=stride, padding=1, bias=False))
        module.add_module('bn%d' % i, BatchNorm2d(cout))
        module.add_module('relu%d' % i, ReLU(inplace=True))
    module.add_module('maxpool', MaxPool3x3s2())
    return _DEEP_STEM[-1][1]


def run_deep_stem(module, x):
    """three fused conv -> BN -> ReLU units, then the 3x3 / stride-2 max-pool"""
    for i in (1, 2, 3):
        x = conv_bn(getattr(module, 'conv%d' % i), getattr(module, 'bn%d' % i), x, relu=True)
    return module.maxpool(x)
