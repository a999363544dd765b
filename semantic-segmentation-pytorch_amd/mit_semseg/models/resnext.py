"""ResNeXt-101 (32 groups) backbone of the reference (models/resnext.py:24-163) on the HIP operators; module / parameter
names match the reference (deep stem conv1..3, layer{1-4}.{i}.conv*/bn*/downsample.{0,1}).  The grouped 3x3 runs through
`GroupedConv2d`, everything else through the fused conv -> BN units of the ResNet path."""
import math

import torch.nn as nn

from .layers import Conv2d, GroupedConv2d, BatchNorm2d, ReLU, MaxPool3x3s2, ConvBNReLU, conv_bn
from .utils import load_url

__all__ = ['ResNeXt', 'resnext101']

model_urls = {
    'resnext101': 'http://sceneparsing.csail.mit.edu/model/pretrained_resnet/resnext101-imagenet.pth',
}


class GroupBottleneck(nn.Module):
    """resnext.py:24-62: 1x1 -> grouped 3x3 (stride) -> 1x1 (x2), shortcut, ReLU"""
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, groups=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = GroupedConv2d(planes, planes, 3, stride=stride, padding=1, groups=groups)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 2, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 2)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out = conv_bn(self.conv1, self.bn1, x, relu=True)
        out = conv_bn(self.conv2, self.bn2, out, relu=True)
        residual = x if self.downsample is None else self.downsample(x)
        return conv_bn(self.conv3, self.bn3, out, residual=residual, relu=True)


class ResNeXt(nn.Module):
    """resnext.py:65-130 without the ImageNet tail (avgpool / fc), which models.Resnet drops anyway (models.py:174-188)"""

    def __init__(self, block, layers, groups=32, num_classes=1000):
        super().__init__()
        self.inplanes = 128
        self.conv1 = Conv2d(3, 64, 3, stride=2, padding=1, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu1 = ReLU(inplace=True)
        self.conv2 = Conv2d(64, 64, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(64)
        self.relu2 = ReLU(inplace=True)
        self.conv3 = Conv2d(64, 128, 3, padding=1, bias=False)
        self.bn3 = BatchNorm2d(128)
        self.relu3 = ReLU(inplace=True)
        self.maxpool = MaxPool3x3s2()
        self.layer1 = self._stage(block, 128, layers[0], 1, groups)
        self.layer2 = self._stage(block, 256, layers[1], 2, groups)
        self.layer3 = self._stage(block, 512, layers[2], 2, groups)
        self.layer4 = self._stage(block, 1024, layers[3], 2, groups)
        for m in self.modules():                            # resnext.py:90-96
            if isinstance(m, Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // getattr(m, 'groups', 1)
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _stage(self, block, planes, count, stride, groups):
        out = planes * block.expansion
        ds = None
        if stride != 1 or self.inplanes != out:
            ds = ConvBNReLU(Conv2d(self.inplanes, out, 1, stride=stride, bias=False), BatchNorm2d(out), relu=False)
        blocks = [block(self.inplanes, planes, stride, groups, ds)]
        self.inplanes = out
        blocks += [block(out, planes, groups=groups) for _ in range(1, count)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = conv_bn(self.conv1, self.bn1, x, relu=True)
        x = conv_bn(self.conv2, self.bn2, x, relu=True)
        x = conv_bn(self.conv3, self.bn3, x, relu=True)
        x = self.maxpool(x)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


def resnext101(pretrained=False, **kwargs):
    model = ResNeXt(GroupBottleneck, [3, 4, 23, 3], **kwargs)
    if pretrained:
        model.load_state_dict(load_url(model_urls['resnext101']), strict=False)
    return model
