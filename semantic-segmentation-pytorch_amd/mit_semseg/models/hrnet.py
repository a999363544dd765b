"""HRNetV2-W48 backbone of the reference (models/hrnet.py:105-445) on the HIP operators.

Same module tree / state-dict keys (conv1, bn1, conv2, bn2, layer1.*, transition{1,2,3}.*,
stage{2,3,4}.{m}.branches.{b}.{k}.*, stage*.{m}.fuse_layers.{i}.{j}.*).  Every exchange unit executes as
conv kernels + fused BN kernels; the `y = y + upsample(...)` chains of hrnet.py:231-248 use the bilinear
kernel's accumulate form so no separate add pass is made for the up-sampled terms.
"""
import torch.nn as nn

from .. import ops
from .layers import Conv2d, BatchNorm2d, ReLU, ConvBNReLU, conv_bn
from .utils import load_url

BN_MOMENTUM = 0.1

__all__ = ['hrnetv2']

model_urls = {
    'hrnetv2': 'http://sceneparsing.csail.mit.edu/model/pretrained_resnet/hrnetv2_w48-imagenet.pth',
}


def _bn(c):
    return BatchNorm2d(c, momentum=BN_MOMENTUM)


def _cbr(cin, cout, k, stride, relu):
    return ConvBNReLU(Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=False), _bn(cout), relu=relu)


class BasicBlock(nn.Module):
    """hrnet.py:32-61"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn1 = _bn(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = _bn(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out, x = conv_bn(self.conv1, self.bn1, x, relu=True, passthrough=True,    # the shortcut hangs off conv1's node;
                         only_feeds=self.conv2)                                    # bn1's output feeds conv2 only (planes, no fp32 copy)
        residual = x if self.downsample is None else self.downsample(x)
        return conv_bn(self.conv2, self.bn2, out, residual=residual, relu=True)


class Bottleneck(nn.Module):
    """hrnet.py:64-102"""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = _bn(planes * 4)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        out, x = conv_bn(self.conv1, self.bn1, x, relu=True, passthrough=True,    # the shortcut hangs off conv1's node
                         only_feeds=self.conv2)
        residual = x if self.downsample is None else self.downsample(x)
        out = conv_bn(self.conv2, self.bn2, out, relu=True, only_feeds=self.conv3)
        return conv_bn(self.conv3, self.bn3, out, residual=residual, relu=True)


class HighResolutionModule(nn.Module):
    """hrnet.py:105-250: per-branch 4x BasicBlock then the multi-resolution exchange."""

    def __init__(self, num_branches, blocks, num_blocks, num_inchannels, num_channels, fuse_method,
                 multi_scale_output=True):
        super().__init__()
        if not (num_branches == len(num_blocks) == len(num_channels) == len(num_inchannels)):
            raise ValueError('NUM_BRANCHES(%d) inconsistent with blocks/channels' % num_branches)
        self.num_inchannels = num_inchannels
        self.fuse_method = fuse_method
        self.num_branches = num_branches
        self.multi_scale_output = multi_scale_output
        self.branches = nn.ModuleList([
            self._branch(i, blocks, num_blocks[i], num_channels[i]) for i in range(num_branches)])
        self.fuse_layers = self._fuse_layers()
        self.relu = ReLU(inplace=True)

    def _branch(self, i, block, count, channels):
        cin, cout = self.num_inchannels[i], channels * block.expansion
        ds = _cbr(cin, cout, 1, 1, relu=False) if cin != cout else None
        layers = [block(cin, channels, 1, ds)]
        self.num_inchannels[i] = cout
        layers += [block(cout, channels) for _ in range(1, count)]
        return nn.Sequential(*layers)

    def _fuse_layers(self):
        if self.num_branches == 1:
            return None
        c = self.num_inchannels
        rows = []
        for i in range(self.num_branches if self.multi_scale_output else 1):
            row = []
            for j in range(self.num_branches):
                if j > i:                                   # lower resolution -> 1x1 + BN, then bilinear up
                    row.append(_cbr(c[j], c[i], 1, 1, relu=False))
                elif j == i:
                    row.append(None)
                else:                                       # higher resolution -> chain of 3x3 stride-2 convs
                    chain = []
                    for k in range(i - j):
                        last = k == i - j - 1
                        chain.append(_cbr(c[j], c[i] if last else c[j], 3, 2, relu=not last))
                    row.append(nn.Sequential(*chain))
            rows.append(nn.ModuleList(row))
        return nn.ModuleList(rows)

    def get_num_inchannels(self):
        return self.num_inchannels

    def forward(self, x):
        if self.num_branches == 1:
            return [self.branches[0](x[0])]
        x = ops.run_branches(list(self.branches), x)      # independent until the exchange below (ops.run_branches)
        rows = len(self.fuse_layers)
        # every branch output feeds every row of the exchange: `rows` consumers.  ops.fork hands each row its own alias, so the
        # gradients of the rows are summed by the native add kernel in one fixed order instead of autograd's own accumulation
        xs = [ops.fork(t, rows) for t in x]
        # the conv chains of the exchange -- fuse_layers[i][j] on branch j's output for every row i != j, 12 of them in stage 4 --
        # are independent of each other: inside a side-by-side scope (ops.batch_branches) their launches pair up unit by unit like
        # those of the branches above; otherwise they run one after the other on the current stream, as before (forked onto side
        # streams they made hipStreamEndCapture crash on this ROCm, gpurun r3x)
        pairs = [(i, j) for i in range(rows) for j in range(self.num_branches) if j != i]
        term = dict(zip(pairs, ops.run_branches([self.fuse_layers[i][j] for i, j in pairs], [xs[j][i] for i, j in pairs],
                                                side_streams=False)))
        # ... and so are the accumulation chains of the rows: row i sums, left to right as hrnet.py:232-248, branch i's own output and
        # the terms of the other branches (up-sampled where they come from a coarser one); one branch of a scope per row
        sizes = [t.shape[2:] for t in x]
        last = self.num_branches - 1

        def row(i):
            def run(ts):
                y = ts[0]
                bounds = [ops.absmax_of(y)]                 # |sum| <= sum of the terms' bounds (up-sampling is a convex combination)
                for j in range(1, self.num_branches):
                    ops.batch_unit()                        # the rows pair up their launches step by step
                    relu = j == last                        # the final ReLU (hrnet.py:248) rides on the last add
                    bounds.append(ops.absmax_of(ts[j]))
                    if j > i:
                        y = ops.interpolate_bilinear(ts[j], sizes[i], base=y, relu=relu)
                    else:
                        y = ops.add_act(y, ts[j], relu=relu)
                if all(b is not None for b in bounds):
                    # the exchange output feeds the residual branch of the next module's blocks: with a bound of it their BN kernels
                    # can bound (and emit the split planes of) their own outputs -- without, every conv of the next branch pays an
                    # absmax + split pass over its input (115 such pairs per HRNetV2 step before round 3)
                    ops.attach_absmax(y, ops.bound_sum(bounds))
                return y
            return ops.Branch(run)
        fused = ops.run_branches([row(i) for i in range(rows)],
                                 [[xs[j][i] if j == i else term[(i, j)] for j in range(self.num_branches)] for i in range(rows)],
                                 side_streams=False)
        return fused


def _transition(layers, ys):
    """hrnet.py:401-423: branch i of the next stage is `layers[i](ys[-1])` (a new, coarser branch made from the LAST branch) or
    `ys[i]` itself (layers[i] is None).  The last branch then has two consumers -- and layer1's output both convs of transition1
    --: they get their own aliases (ops.fork), so that the two gradients are summed by the native layer in one fixed order and a
    sum the BN backward kernels form themselves (ops.defer_fork_sums) never meets autograd's own accumulation."""
    src = [len(ys) - 1 if t is not None else i for i, t in enumerate(layers)]
    alias = {j: list(ops.fork(ys[j], src.count(j))) for j in set(src)}
    return [t(alias[j].pop()) if t is not None else alias[j].pop() for t, j in zip(layers, src)]


class HRNetV2(nn.Module):
    """hrnet.py:258-437"""

    def __init__(self, n_class, **kwargs):
        super().__init__()
        cfg = {
            'STAGE2': dict(NUM_MODULES=1, NUM_BRANCHES=2, NUM_BLOCKS=(4, 4), NUM_CHANNELS=(48, 96)),
            'STAGE3': dict(NUM_MODULES=4, NUM_BRANCHES=3, NUM_BLOCKS=(4, 4, 4), NUM_CHANNELS=(48, 96, 192)),
            'STAGE4': dict(NUM_MODULES=3, NUM_BRANCHES=4, NUM_BLOCKS=(4, 4, 4, 4), NUM_CHANNELS=(48, 96, 192, 384)),
        }
        self.conv1 = Conv2d(3, 64, 3, stride=2, padding=1, bias=False)
        self.bn1 = _bn(64)
        self.conv2 = Conv2d(64, 64, 3, stride=2, padding=1, bias=False)
        self.bn2 = _bn(64)
        self.relu = ReLU(inplace=True)
        ds = _cbr(64, 256, 1, 1, relu=False)
        self.layer1 = nn.Sequential(Bottleneck(64, 64, 1, ds), *[Bottleneck(256, 64) for _ in range(3)])

        self.stage2_cfg = cfg['STAGE2']
        ch = list(self.stage2_cfg['NUM_CHANNELS'])
        self.transition1 = self._transition([256], ch)
        self.stage2, pre = self._stage(self.stage2_cfg, ch)
        self.stage3_cfg = cfg['STAGE3']
        ch = list(self.stage3_cfg['NUM_CHANNELS'])
        self.transition2 = self._transition(pre, ch)
        self.stage3, pre = self._stage(self.stage3_cfg, ch)
        self.stage4_cfg = cfg['STAGE4']
        ch = list(self.stage4_cfg['NUM_CHANNELS'])
        self.transition3 = self._transition(pre, ch)
        self.stage4, pre = self._stage(self.stage4_cfg, ch)

    @staticmethod
    def _transition(pre, cur):
        """hrnet.py:309-343"""
        layers = []
        for i, c in enumerate(cur):
            if i < len(pre):
                layers.append(_cbr(pre[i], c, 3, 1, relu=True) if c != pre[i] else None)
            else:
                chain = []
                for j in range(i + 1 - len(pre)):
                    cout = c if j == i - len(pre) else pre[-1]
                    chain.append(_cbr(pre[-1], cout, 3, 2, relu=True))
                layers.append(nn.Sequential(*chain))
        return nn.ModuleList(layers)

    @staticmethod
    def _stage(cfg, num_inchannels):
        mods = []
        for _ in range(cfg['NUM_MODULES']):
            m = HighResolutionModule(cfg['NUM_BRANCHES'], BasicBlock, cfg['NUM_BLOCKS'], num_inchannels,
                                     cfg['NUM_CHANNELS'], 'SUM', True)
            mods.append(m)
            num_inchannels = m.get_num_inchannels()
        return nn.Sequential(*mods), num_inchannels

    def forward(self, x, return_feature_maps=False):
        x = conv_bn(self.conv1, self.bn1, x, relu=True)
        x = conv_bn(self.conv2, self.bn2, x, relu=True)
        x = self.layer1(x)
        ys = self.stage2(_transition(self.transition1, [x]))
        ys = self.stage3(_transition(self.transition2, ys))
        ys = self.stage4(_transition(self.transition3, ys))
        size = ys[0].shape[2:]
        return [ops.concat([ys[0]] + [ops.interpolate_bilinear(t, size) for t in ys[1:]])]


def hrnetv2(pretrained=False, **kwargs):
    model = HRNetV2(n_class=1000, **kwargs)
    if pretrained:
        model.load_state_dict(load_url(model_urls['hrnetv2']), strict=False)
    return model
