"""Deep-stem ResNet-18/50/101 backbones of the reference (models/resnet.py:24-205) on the HIP operators.

Module/parameter names match the reference exactly (conv1..3, bn1..3, layer{1-4}.{i}.conv*/bn*/
downsample.{0,1}); each block executes as conv kernels + fused BN(+residual)+ReLU kernels."""
import math

import torch.nn as nn

from .layers import Conv2d, BatchNorm2d, ReLU, ConvBNReLU, conv_bn, add_deep_stem, run_deep_stem
from .utils import load_url

__all__ = ['ResNet', 'resnet18', 'resnet50', 'resnet101']

model_urls = {
    'resnet18': 'http://sceneparsing.csail.mit.edu/model/pretrained_resnet/resnet18-imagenet.pth',
    'resnet50': 'http://sceneparsing.csail.mit.edu/model/pretrained_resnet/resnet50-imagenet.pth',
    'resnet101': 'http://sceneparsing.csail.mit.edu/model/pretrained_resnet/resnet101-imagenet.pth',
}


class _Downsample(ConvBNReLU):
    """1x1 (strided) conv + BN on the shortcut (resnet.py:128-133), no ReLU."""

    def __init__(self, inplanes, outplanes, stride):
        super().__init__(Conv2d(inplanes, outplanes, 1, stride=stride, bias=False), BatchNorm2d(outplanes), relu=False)


class BasicBlock(nn.Module):
    """resnet.py:24-53"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out, x = conv_bn(self.conv1, self.bn1, x, relu=True, passthrough=True,    # the shortcut hangs off conv1's node
                         only_feeds=self.conv2)
        residual = x if self.downsample is None else self.downsample(x)
        return conv_bn(self.conv2, self.bn2, out, residual=residual, relu=True)


class Bottleneck(nn.Module):
    """resnet.py:56-92 (stride on the 3x3)"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        # bn1's output feeds conv2 only, bn2's feeds conv3 only: where the consumer reads planes no fp32 copy is written
        out, x = conv_bn(self.conv1, self.bn1, x, relu=True, passthrough=True,    # the shortcut hangs off conv1's node
                         only_feeds=self.conv2)
        residual = x if self.downsample is None else self.downsample(x)
        out = conv_bn(self.conv2, self.bn2, out, relu=True, only_feeds=self.conv3)
        return conv_bn(self.conv3, self.bn3, out, residual=residual, relu=True)


class ResNet(nn.Module):
    """resnet.py:95-158.  The ImageNet classifier tail (avgpool/fc) is never used by the segmentation
    path (models.py:174-188 drops it) and is omitted; loading a reference checkpoint with fc.* keys
    works through strict=False exactly as in the reference (models.py:106-109)."""

    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = add_deep_stem(self)            # conv1..3 / bn1..3 / relu1..3 / maxpool (resnet.py:100-109); 128 wide
        self.layer1 = self._stage(block, 64, layers[0], 1)
        self.layer2 = self._stage(block, 128, layers[1], 2)
        self.layer3 = self._stage(block, 256, layers[2], 2)
        self.layer4 = self._stage(block, 512, layers[3], 2)
        for m in self.modules():                       # resnet.py:118-124
            if isinstance(m, Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _stage(self, block, planes, count, stride):
        out = planes * block.expansion
        ds = _Downsample(self.inplanes, out, stride) if (stride != 1 or self.inplanes != out) else None
        blocks = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = out
        blocks += [block(out, planes) for _ in range(1, count)]
        return nn.Sequential(*blocks)

    def stem(self, x):
        return run_deep_stem(self, x)

    def forward(self, x):
        x = self.stem(x)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def _make(name, block, layers, pretrained, strict, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        model.load_state_dict(load_url(model_urls[name]), strict=strict)
    return model


def resnet18(pretrained=False, **kwargs):
    return _make('resnet18', BasicBlock, [2, 2, 2, 2], pretrained, False, **kwargs)


def resnet50(pretrained=False, **kwargs):
    return _make('resnet50', Bottleneck, [3, 4, 6, 3], pretrained, False, **kwargs)


def resnet101(pretrained=False, **kwargs):
    return _make('resnet101', Bottleneck, [3, 4, 23, 3], pretrained, False, **kwargs)
