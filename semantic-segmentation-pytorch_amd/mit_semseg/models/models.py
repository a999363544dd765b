"""Model API of the MI355X build: ModelBuilder.build_encoder/build_decoder and SegmentationModule with the
reference's signatures, arch strings, module tree names and error behaviour (reference
mit_semseg/models/models.py:9-586), executing on the HIP operators of mit_semseg.ops.

Layout: tensors crossing this API keep the reference's logical [N,C,H,W] shape; memory is NHWC.
"""
import os

import torch
import torch.nn as nn

from .. import ops
from . import resnet, hrnet, mobilenet, resnext
from .layers import (Conv2d, BatchNorm2d, AdaptiveAvgPool2d, Dropout2d, ConvBNReLU, ReLU, conv3x3_bn_relu, conv_bn,
                     run_deep_stem)


class SegmentationModuleBase(nn.Module):
    def __init__(self):
        super(SegmentationModuleBase, self).__init__()

    def pixel_acc(self, pred, label):
        """models.py:12-18 (first-max argmax, valid = label >= 0) -- computed by the fused NLL/acc kernel."""
        _, acc = ops.nll_loss_acc(pred, label, ignore_index=-1)
        return acc


class SegmentationModule(SegmentationModuleBase):
    """models.py:21-47.  forward(feed_dict) -> (loss, acc); forward(feed_dict, segSize=(H,W)) -> softmax probs."""

    def __init__(self, net_enc, net_dec, crit, deep_sup_scale=None):
        super(SegmentationModule, self).__init__()
        self.encoder = net_enc
        self.decoder = net_dec
        self.crit = crit
        self.deep_sup_scale = deep_sup_scale

    def _loss_acc(self, pred, label, want_acc):
        crit = self.crit
        native = type(crit) is nn.NLLLoss and crit.weight is None and crit.reduction == 'mean'
        if native:
            loss, acc = ops.nll_loss_acc(pred, label, ignore_index=crit.ignore_index)
            if want_acc and crit.ignore_index != -1:
                acc = self.pixel_acc(pred, label)
            return loss, acc
        # user-supplied criterion object: evaluated as given (the caller's code, not part of the hot path)
        loss = crit(pred, label)
        return loss, (self.pixel_acc(pred, label) if want_acc else None)

    def forward(self, feed_dict, *, segSize=None):
        if isinstance(feed_dict, (list, tuple)):      # UserScatteredDataParallel passes a 1-element list per GPU
            feed_dict = feed_dict[0]
        feats = self.encoder(feed_dict['img_data'], return_feature_maps=True)
        if segSize is not None:                       # inference
            return self.decoder(feats, segSize=segSize)
        label = feed_dict['seg_label']
        if self.deep_sup_scale is not None:           # deep supervision
            pred, pred_deepsup = self.decoder(feats)
        else:
            pred = self.decoder(feats)
        loss, acc = self._loss_acc(pred, label, True)
        if self.deep_sup_scale is not None:
            loss_deepsup, _ = self._loss_acc(pred_deepsup, label, False)
            loss = loss + loss_deepsup * self.deep_sup_scale
        return loss, acc


class ModelBuilder:
    # custom weights initialization (models.py:52-61)
    @staticmethod
    def weights_init(m):
        classname = m.__class__.__name__
        if classname.find('ConvBN') != -1:
            return
        if classname.find('Conv') != -1:
            nn.init.kaiming_normal_(m.weight.data)
        elif classname.find('BatchNorm') != -1:
            m.weight.data.fill_(1.)
            m.bias.data.fill_(1e-4)

    @staticmethod
    def build_encoder(arch='resnet50dilated', fc_dim=512, weights=''):
        pretrained = True if len(weights) == 0 else False
        arch = arch.lower()
        if arch in ('resnet34', 'resnet34dilated'):
            raise NotImplementedError
        if arch == 'mobilenetv2dilated':
            net_encoder = MobileNetV2Dilated(mobilenet.__dict__['mobilenetv2'](pretrained=pretrained), dilate_scale=8)
        elif arch == 'resnext101':
            net_encoder = Resnet(resnext.__dict__['resnext101'](pretrained=pretrained))      # models.py:96-98
        elif arch in ('resnet18', 'resnet50', 'resnet101'):
            net_encoder = Resnet(resnet.__dict__[arch](pretrained=pretrained))
        elif arch in ('resnet18dilated', 'resnet50dilated', 'resnet101dilated'):
            net_encoder = ResnetDilated(resnet.__dict__[arch[:-7]](pretrained=pretrained), dilate_scale=8)
        elif arch == 'hrnetv2':
            net_encoder = hrnet.__dict__['hrnetv2'](pretrained=pretrained)
        else:
            raise Exception('Architecture undefined!')

        # encoders are usually pretrained
        if len(weights) > 0:
            print('Loading weights for net_encoder')
            net_encoder.load_state_dict(
                torch.load(weights, map_location=lambda storage, loc: storage), strict=False)
        return net_encoder

    @staticmethod
    def build_decoder(arch='ppm_deepsup', fc_dim=512, num_class=150, weights='', use_softmax=False):
        arch = arch.lower()
        table = {
            'c1_deepsup': lambda: C1DeepSup(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax),
            'c1': lambda: C1(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax),
            'ppm': lambda: PPM(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax),
            'ppm_deepsup': lambda: PPMDeepsup(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax),
            'upernet_lite': lambda: UPerNet(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax, fpn_dim=256),
            'upernet': lambda: UPerNet(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax, fpn_dim=512),
        }
        if arch not in table:
            raise Exception('Architecture undefined!')
        net_decoder = table[arch]()
        net_decoder.apply(ModelBuilder.weights_init)
        if len(weights) > 0:
            print('Loading weights for net_decoder')
            net_decoder.load_state_dict(
                torch.load(weights, map_location=lambda storage, loc: storage), strict=False)
        return net_decoder


# ------------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------------
class Resnet(nn.Module):
    """models.py:170-205: the backbone without avgpool/fc, returning the four stage outputs."""

    def __init__(self, orig_resnet):
        super(Resnet, self).__init__()
        for name in ('conv1', 'bn1', 'relu1', 'conv2', 'bn2', 'relu2', 'conv3', 'bn3', 'relu3', 'maxpool',
                     'layer1', 'layer2', 'layer3', 'layer4'):
            setattr(self, name, getattr(orig_resnet, name))

    def forward(self, x, return_feature_maps=False):
        x = run_deep_stem(self, x)
        conv_out = []
        stages = (self.layer1, self.layer2, self.layer3, self.layer4)
        for i, stage in enumerate(stages):
            x = stage(x)
            if return_feature_maps and i + 1 < len(stages):
                # a stage output feeds the next stage AND a decoder head (deep supervision, FPN laterals): fork it, so that
                # the two gradients are summed by the native add kernel (a fork output nobody uses costs nothing)
                keep, x = ops.fork(x)
                conv_out.append(keep)
            else:
                conv_out.append(x)
        return conv_out if return_feature_maps else [x]


class ResnetDilated(Resnet):
    """models.py:208-268: layer3/4 (dilate_scale 8) or layer4 (16) lose their stride and gain dilation."""

    def __init__(self, orig_resnet, dilate_scale=8):
        if dilate_scale == 8:
            self._dilate_stage(orig_resnet.layer3, 2)
            self._dilate_stage(orig_resnet.layer4, 4)
        elif dilate_scale == 16:
            self._dilate_stage(orig_resnet.layer4, 2)
        super(ResnetDilated, self).__init__(orig_resnet)

    @staticmethod
    def _dilate_stage(stage, dilate):
        for m in stage.modules():
            ResnetDilated._nostride_dilate(m, dilate)

    @staticmethod
    def _nostride_dilate(m, dilate):
        """models.py:238-251 applied to this build's Conv2d leaves."""
        if not isinstance(m, Conv2d):
            return
        three = m.kernel_size == (3, 3)
        if m.stride == (2, 2):              # the convolution with stride
            m.stride = (1, 1)
            if three:
                m.dilation = (dilate // 2, dilate // 2)
                m.padding = (dilate // 2, dilate // 2)
        elif three:                         # other convolutions
            m.dilation = (dilate, dilate)
            m.padding = (dilate, dilate)


class MobileNetV2Dilated(nn.Module):
    """models.py:271-323: the MobileNetV2 feature stack without its last 1x1 unit; with dilate_scale 8 the blocks from
    index 7 on lose their stride (dilation 2, from index 14: dilation 4).  Returns the maps after blocks 2, 4, 7, 14 and the
    last one."""

    def __init__(self, orig_net, dilate_scale=8):
        super(MobileNetV2Dilated, self).__init__()
        self.features = orig_net.features[:-1]
        self.total_idx = len(self.features)
        self.down_idx = [2, 4, 7, 14]
        if dilate_scale == 8:
            for i in range(self.down_idx[-2], self.down_idx[-1]):
                ResnetDilated._dilate_stage(self.features[i], 2)
            for i in range(self.down_idx[-1], self.total_idx):
                ResnetDilated._dilate_stage(self.features[i], 4)
        elif dilate_scale == 16:
            for i in range(self.down_idx[-1], self.total_idx):
                ResnetDilated._dilate_stage(self.features[i], 2)

    def forward(self, x, return_feature_maps=False):
        if not return_feature_maps:
            return [self.features(x)]
        conv_out = []
        for i in range(self.total_idx):
            x = self.features[i](x)
            if i in self.down_idx:
                # feeds the next block AND a decoder head: its own alias for each (as Resnet.forward; a fork output nobody uses
                # costs nothing), so the native layer sums the two gradients
                keep, x = ops.fork(x)
                conv_out.append(keep)
        conv_out.append(x)
        return conv_out


# ------------------------------------------------------------------------------------------------
# decoders
# ------------------------------------------------------------------------------------------------
def _head_output(x, use_softmax, segSize):
    """models.py:480-484 / :492: inference = bilinear up-sample of the logits + softmax; training = log_softmax."""
    if use_softmax:
        return ops.upsample_softmax(x, segSize)          # bilinear up-sampling + softmax in one kernel
    return ops.log_softmax(x)


class _PoolBranch(nn.Sequential):
    """PPM branch (models.py:401-408): '0' AdaptiveAvgPool2d, '1' 1x1 conv, '2' BN, '3' ReLU."""

    def __init__(self, fc_dim, scale):
        super().__init__()
        self.add_module('0', AdaptiveAvgPool2d(scale))
        self.add_module('1', Conv2d(fc_dim, 512, kernel_size=1, bias=False))
        self.add_module('2', BatchNorm2d(512))
        self.add_module('3', ReLU(inplace=True))      # fused into the BN kernel; kept so indices match the reference

    def forward(self, x):
        return self.after_pool(self._modules['0'](x))

    def after_pool(self, pooled):
        m = self._modules
        return conv_bn(m['1'], m['2'], pooled, relu=True)


class _ClassifierHead(nn.Sequential):
    """conv_last of PPM/PPMDeepsup (models.py:455-462): '0' 3x3 conv, '1' BN, '2' ReLU, '3' Dropout2d, '4' 1x1."""

    def __init__(self, in_dim, num_class):
        super().__init__()
        self.add_module('0', Conv2d(in_dim, 512, kernel_size=3, padding=1, bias=False))
        self.add_module('1', BatchNorm2d(512))
        self.add_module('2', ReLU(inplace=True))      # fused into the BN kernel; kept so conv_last[3] is the Dropout2d
        self.add_module('3', Dropout2d(0.1))
        self.add_module('4', Conv2d(512, num_class, kernel_size=1))

    def forward(self, x):
        m = self._modules
        x = conv_bn(m['0'], m['1'], x, relu=True)
        return m['4'](m['3'](x))


class C1(nn.Module):
    """models.py:363-385"""

    def __init__(self, num_class=150, fc_dim=2048, use_softmax=False):
        super(C1, self).__init__()
        self.use_softmax = use_softmax
        self.cbr = conv3x3_bn_relu(fc_dim, fc_dim // 4, 1)
        self.conv_last = Conv2d(fc_dim // 4, num_class, 1, 1, 0)

    def forward(self, conv_out, segSize=None):
        x = self.conv_last(self.cbr(conv_out[-1]))
        return _head_output(x, self.use_softmax, segSize)


class C1DeepSup(nn.Module):
    """models.py:327-359"""

    def __init__(self, num_class=150, fc_dim=2048, use_softmax=False):
        super(C1DeepSup, self).__init__()
        self.use_softmax = use_softmax
        self.cbr = conv3x3_bn_relu(fc_dim, fc_dim // 4, 1)
        self.cbr_deepsup = conv3x3_bn_relu(fc_dim // 2, fc_dim // 4, 1)
        self.conv_last = Conv2d(fc_dim // 4, num_class, 1, 1, 0)
        self.conv_last_deepsup = Conv2d(fc_dim // 4, num_class, 1, 1, 0)

    def forward(self, conv_out, segSize=None):
        x = self.conv_last(self.cbr(conv_out[-1]))
        if self.use_softmax:
            return _head_output(x, True, segSize)
        d = self.conv_last_deepsup(self.cbr_deepsup(conv_out[-2]))
        return ops.log_softmax(x), ops.log_softmax(d)


class PPM(nn.Module):
    """models.py:389-434: pyramid pooling (1,2,3,6) -> concat -> 3x3 conv head."""

    def __init__(self, num_class=150, fc_dim=4096, use_softmax=False, pool_scales=(1, 2, 3, 6)):
        super(PPM, self).__init__()
        self.use_softmax = use_softmax
        self.ppm = nn.ModuleList([_PoolBranch(fc_dim, s) for s in pool_scales])
        self.conv_last = _ClassifierHead(fc_dim + len(pool_scales) * 512, num_class)

    def _pyramid(self, conv5):
        size = conv5.shape[2:]
        # all pyramid scales pooled in one pass over conv5 (ops.adaptive_avg_pool_multi)
        conv5, c5 = ops.fork(conv5)                  # two consumers: the pyramid pooling and the concat
        pooled = ops.adaptive_avg_pool_multi(c5, [b._modules['0'].output_size for b in self.ppm])
        # the four pyramid branches (1x1 conv -> BN -> ReLU on an s x s map, then up-sampling) are independent: inside a side-by-side
        # scope (ops.batch_branches) their launches pair up; else one after the other, as before
        fns = [ops.Branch(lambda p, b=b: ops.interpolate_bilinear(b.after_pool(p), size), [b]) for b in self.ppm]
        return ops.concat([conv5] + ops.run_branches(fns, list(pooled), side_streams=False))

    def forward(self, conv_out, segSize=None):
        x = self.conv_last(self._pyramid(conv_out[-1]))
        return _head_output(x, self.use_softmax, segSize)


class PPMDeepsup(PPM):
    """models.py:438-495: PPM + auxiliary head on conv4."""

    def __init__(self, num_class=150, fc_dim=4096, use_softmax=False, pool_scales=(1, 2, 3, 6)):
        super(PPMDeepsup, self).__init__(num_class, fc_dim, use_softmax, pool_scales)
        self.cbr_deepsup = conv3x3_bn_relu(fc_dim // 2, fc_dim // 4, 1)
        self.conv_last_deepsup = Conv2d(fc_dim // 4, num_class, 1, 1, 0)
        self.dropout_deepsup = Dropout2d(0.1)

    def forward(self, conv_out, segSize=None):
        x = self.conv_last(self._pyramid(conv_out[-1]))
        if self.use_softmax:
            return _head_output(x, True, segSize)
        d = self.conv_last_deepsup(self.dropout_deepsup(self.cbr_deepsup(conv_out[-2])))
        return ops.log_softmax(x), ops.log_softmax(d)


class UPerNet(nn.Module):
    """models.py:499-586: PPM on conv5 (pool -> upsample -> 1x1) + FPN top-down + fusion head."""

    def __init__(self, num_class=150, fc_dim=4096, use_softmax=False, pool_scales=(1, 2, 3, 6),
                 fpn_inplanes=(256, 512, 1024, 2048), fpn_dim=256):
        super(UPerNet, self).__init__()
        self.use_softmax = use_softmax
        self.ppm_pooling = nn.ModuleList([AdaptiveAvgPool2d(s) for s in pool_scales])
        self.ppm_conv = nn.ModuleList([
            ConvBNReLU(Conv2d(fc_dim, 512, kernel_size=1, bias=False), BatchNorm2d(512)) for _ in pool_scales])
        self.ppm_last_conv = conv3x3_bn_relu(fc_dim + len(pool_scales) * 512, fpn_dim, 1)
        self.fpn_in = nn.ModuleList([
            ConvBNReLU(Conv2d(c, fpn_dim, kernel_size=1, bias=False), BatchNorm2d(fpn_dim))
            for c in fpn_inplanes[:-1]])                                    # skip the top layer
        self.fpn_out = nn.ModuleList([
            nn.Sequential(conv3x3_bn_relu(fpn_dim, fpn_dim, 1)) for _ in range(len(fpn_inplanes) - 1)])
        self.conv_last = nn.Sequential(
            conv3x3_bn_relu(len(fpn_inplanes) * fpn_dim, fpn_dim, 1),
            Conv2d(fpn_dim, num_class, kernel_size=1))

    def forward(self, conv_out, segSize=None):
        conv5, c5 = ops.fork(conv_out[-1])            # two consumers: the pyramid pooling and the concat
        size = conv5.shape[2:]
        pooled = ops.adaptive_avg_pool_multi(c5, [pool.output_size for pool in self.ppm_pooling])    # one pass over conv5
        # the pyramid branches, and below the lateral 1x1 convs and the 3x3 output convs of the FPN levels, are independent of each
        # other: side-by-side launches inside ops.batch_branches(), one after the other otherwise
        ppm_out = [conv5] + ops.run_branches(
            [ops.Branch(lambda p, conv=conv: conv(ops.interpolate_bilinear(p, size)), [conv]) for conv in self.ppm_conv],
            list(pooled), side_streams=False)
        f = self.ppm_last_conv(ops.concat(ppm_out))
        fpn_feature_list = [f]
        levels = list(range(len(conv_out) - 1))
        laterals = ops.run_branches([self.fpn_in[i] for i in levels], [conv_out[i] for i in levels], side_streams=False)
        tops = {}
        for i in reversed(levels):
            lateral = laterals[i]
            bounds = (ops.absmax_of(lateral), ops.absmax_of(f))
            f = ops.interpolate_bilinear(f, lateral.shape[2:], base=lateral)     # top-down: lateral + up(f)
            if bounds[0] is not None and bounds[1] is not None:
                # |lateral + up(f)| <= bound(lateral) + bound(f): the 3x3 conv that follows then needs no absmax pass over the sum
                # (and may take the Winograd forward, which scales its input transform by a bound)
                ops.attach_absmax(f, ops.bound_sum(bounds))
            tops[i] = f
        outs = ops.run_branches([self.fpn_out[i] for i in levels], [tops[i] for i in levels], side_streams=False)
        fpn_feature_list += [outs[i] for i in reversed(levels)]
        fpn_feature_list.reverse()                                               # [P2 - P5]
        out_size = fpn_feature_list[0].shape[2:]
        fusion = ops.concat([fpn_feature_list[0]] +
                            [ops.interpolate_bilinear(t, out_size) for t in fpn_feature_list[1:]])
        x = self.conv_last(fusion)
        return _head_output(x, self.use_softmax, segSize)
