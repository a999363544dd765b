"""MobileNetV2 backbone of the reference (models/mobilenet.py:22-154) on the HIP operators: parameter names and shapes match
the reference (`features.N.0/1` for the stem / head units, `features.N.conv.M` inside the inverted-residual blocks,
`classifier.1`), so its checkpoints load.  Pointwise convs run the fused conv -> BN(-> ReLU) node, depthwise convs go through
`GroupedConv2d`; ReLU6 = fused ReLU + upper clamp."""
import math

import torch.nn as nn

from .. import ops
from .layers import Conv2d, GroupedConv2d, BatchNorm2d, ReLU6, conv_bn, conv_bn_relu6
from .utils import load_url

__all__ = ['mobilenetv2']

model_urls = {
    'mobilenetv2': 'http://sceneparsing.csail.mit.edu/model/pretrained_resnet/mobilenet_v2.pth.tar',
}

# expansion t, output channels c, repeats n, stride s of the first repeat (mobilenet.py:84-93)
_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


class _ConvBNReLU6(nn.Sequential):
    """children '0' conv, '1' BN, '2' ReLU6 as mobilenet.py:22-35"""

    def __init__(self, conv, bn):
        super().__init__(conv, bn, ReLU6(inplace=True))

    def forward(self, x):
        return conv_bn_relu6(self[0], self[1], x)


class _Units(nn.Sequential):
    """The `conv` Sequential of an inverted-residual block: (conv, BN[, ReLU6]) units executed pairwise"""

    def forward(self, x):
        mods = list(self.children())
        i = 0
        while i < len(mods):
            act = i + 2 < len(mods) and isinstance(mods[i + 2], ReLU6)
            x = conv_bn_relu6(mods[i], mods[i + 1], x) if act else conv_bn(mods[i], mods[i + 1], x)
            i += 3 if act else 2
        return x


class InvertedResidual(nn.Module):
    """mobilenet.py:38-75: [1x1 expand ->] 3x3 depthwise -> 1x1 linear projection, identity shortcut when stride 1 and
    inp == oup"""

    def __init__(self, inp, oup, stride, expand_ratio):
        super().__init__()
        assert stride in (1, 2)
        self.stride = stride
        hidden = round(inp * expand_ratio)
        self.use_res_connect = stride == 1 and inp == oup
        units = []
        if expand_ratio != 1:
            units += [Conv2d(inp, hidden, 1, bias=False), BatchNorm2d(hidden), ReLU6(inplace=True)]
        units += [GroupedConv2d(hidden, hidden, 3, stride=stride, padding=1, groups=hidden), BatchNorm2d(hidden),
                  ReLU6(inplace=True),
                  Conv2d(hidden, oup, 1, bias=False), BatchNorm2d(oup)]
        self.conv = _Units(*units)

    def forward(self, x):
        if not self.use_res_connect:
            return self.conv(x)
        xa, xb = ops.fork(x)
        return ops.add_act(xb, self.conv(xa))


class MobileNetV2(nn.Module):
    """mobilenet.py:78-142.  The ImageNet classifier keeps its parameter names (`classifier.1.*`) so that checkpoints load
    strictly; the segmentation path never runs it (models.py:277 takes `features[:-1]`)."""

    def __init__(self, n_class=1000, input_size=224, width_mult=1.):
        super().__init__()
        assert input_size % 32 == 0
        inp = int(32 * width_mult)
        self.last_channel = int(1280 * width_mult) if width_mult > 1.0 else 1280
        feats = [_ConvBNReLU6(Conv2d(3, inp, 3, stride=2, padding=1, bias=False), BatchNorm2d(inp))]
        for t, c, n, s in _SETTING:
            oup = int(c * width_mult)
            for i in range(n):
                feats.append(InvertedResidual(inp, oup, s if i == 0 else 1, expand_ratio=t))
                inp = oup
        feats.append(_ConvBNReLU6(Conv2d(inp, self.last_channel, 1, bias=False), BatchNorm2d(self.last_channel)))
        self.features = nn.Sequential(*feats)
        self.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(self.last_channel, n_class))
        for m in self.modules():                            # mobilenet.py:127-141
            if isinstance(m, Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                m.weight.data.normal_(0, 0.01)
                m.bias.data.zero_()

    def forward(self, x):
        raise NotImplementedError('the ImageNet classifier head is not part of the segmentation path; use '
                                  'models.MobileNetV2Dilated(mobilenetv2())')


def mobilenetv2(pretrained=False, **kwargs):
    model = MobileNetV2(n_class=1000, **kwargs)
    if pretrained:
        model.load_state_dict(load_url(model_urls['mobilenetv2']), strict=False)
    return model
