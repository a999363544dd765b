"""ResNeXt-101 (32 groups) backbone of the reference (models/resnext.py:24-163) on the HIP operators; module / parameter
names match the reference (deep stem conv1..3, layer{1-4}.{i}.conv*/bn*/downsample.{0,1}), so its checkpoints load.  The
grouped 3x3 runs through `GroupedConv2d`, everything else through the fused conv -> BN units of the ResNet path."""
import math

import torch.nn as nn

from .layers import Conv2d, GroupedConv2d, BatchNorm2d, ReLU, ConvBNReLU, conv_bn, add_deep_stem, run_deep_stem
from .utils import load_url

__all__ = ['ResNeXt', 'resnext101']

model_urls = {
    'resnext101': 'http://sceneparsing.csail.mit.edu/model/pretrained_resnet/resnext101-imagenet.pth',
}
STAGE_WIDTHS = (128, 256, 512, 1024)        # resnext.py:82-85: bottleneck widths, outputs are twice as wide
CARDINALITY = 32


class GroupBottleneck(nn.Module):
    """resnext.py:24-62: 1x1 reduce -> grouped 3x3 (carries the stride) -> 1x1 expand (x2) -> + shortcut -> ReLU"""
    expansion = 2

    def __init__(self, inplanes, planes, stride=1, groups=1, downsample=None):
        super().__init__()
        wide = planes * self.expansion
        self.conv1, self.bn1 = Conv2d(inplanes, planes, 1, bias=False), BatchNorm2d(planes)
        self.conv2 = GroupedConv2d(planes, planes, 3, stride=stride, padding=1, groups=groups)
        self.bn2 = BatchNorm2d(planes)
        self.conv3, self.bn3 = Conv2d(planes, wide, 1, bias=False), BatchNorm2d(wide)
        self.relu = ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride

    def forward(self, x):
        shortcut = self.downsample(x) if self.downsample is not None else x
        y = conv_bn(self.conv1, self.bn1, x, relu=True)
        y = conv_bn(self.conv2, self.bn2, y, relu=True)
        return conv_bn(self.conv3, self.bn3, y, residual=shortcut, relu=True)


class ResNeXt(nn.Module):
    """resnext.py:65-130 without the ImageNet tail (avgpool / fc), which models.Resnet drops anyway (models.py:174-188)"""

    def __init__(self, block, layers, groups=CARDINALITY, num_classes=1000):
        super().__init__()
        width_in = add_deep_stem(self)
        for i, (planes, depth) in enumerate(zip(STAGE_WIDTHS, layers)):
            stage, width_in = self._stage(block, width_in, planes, depth, 1 if i == 0 else 2, groups)
            self.add_module('layer%d' % (i + 1), stage)
        self.inplanes = width_in
        for m in self.modules():                            # resnext.py:90-96: He-normal over the per-group fan-out
            if isinstance(m, Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // getattr(m, 'groups', 1)
                m.weight.data.normal_(0, math.sqrt(2. / fan))
            elif isinstance(m, BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    @staticmethod
    def _stage(block, width_in, planes, depth, stride, groups):
        width_out = planes * block.expansion
        project = None
        if stride != 1 or width_in != width_out:            # resnext.py:98-105
            project = ConvBNReLU(Conv2d(width_in, width_out, 1, stride=stride, bias=False), BatchNorm2d(width_out), relu=False)
        blocks = [block(width_in, planes, stride, groups, project)]
        blocks.extend(block(width_out, planes, groups=groups) for _ in range(depth - 1))
        return nn.Sequential(*blocks), width_out

    def forward(self, x):
        x = run_deep_stem(self, x)
        for name in ('layer1', 'layer2', 'layer3', 'layer4'):
            x = getattr(self, name)(x)
        return x


def resnext101(pretrained=False, **kwargs):
    model = ResNeXt(GroupBottleneck, (3, 4, 23, 3), **kwargs)
    if pretrained:
        model.load_state_dict(load_url(model_urls['resnext101']), strict=False)
    return model
