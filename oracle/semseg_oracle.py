"""CPU ORACLE (test infrastructure only -- never imported by the product path).

A functional restatement of the CSAILVision/semantic-segmentation-pytorch hot
path (SegmentationModule.forward, training and inference branches) over torch
CPU fp32 operators, driven purely by a *state dict* that uses the reference's
parameter/buffer names.  It contains no nn.Module tree, no CUDA/HIP code and it
never touches /root/reference at run time, so it travels to the GPU box.

Parity pin: `tests/golden/*.pt` hold outputs of the UNMODIFIED reference
(imported from /root/reference by `tests/golden/make_golden.py`, committed) on
seeded synthetic weights/inputs; `tests/test_oracle_golden.py` checks this file
against every one of them.  The arithmetic itself lives in a third-party
dependency of the reference (PyTorch: reference pins torch>=0.4.1 in
setup.py:22; this image has torch 2.10.0) -- the oracle calls the same torch CPU
kernels the reference calls, through torch.nn.functional.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.

Reference citations are relative to /root/reference/mit_semseg/.
"""
import math
import zlib

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
RESNET_BN_MOMENTUM = 0.001   # lib/nn/modules/batchnorm.py:39 (fork default)
HRNET_BN_MOMENTUM = 0.1      # models/hrnet.py:14

RESNET_LAYERS = {            # models/resnet.py:160-205
    'resnet18': ('basic', (2, 2, 2, 2)),
    'resnet50': ('bottleneck', (3, 4, 6, 3)),
    'resnet101': ('bottleneck', (3, 4, 23, 3)),
}


# ----------------------------------------------------------------------------
# deterministic synthetic weights (shared by golden generation, tests, bench)
# ----------------------------------------------------------------------------
def synth_tensor(key, shape, seed=0):
    """Seeded synthetic value for state-dict entry `key` (platform independent:
    torch CPU generator).  Conv weights ~ He-normal(fan_in) so activations stay
    O(1) through 100+ layers; BN affine/running stats are non-trivial so eval
    mode and the affine path are exercised."""
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed) & 0x7FFFFFFF)
    shape = tuple(shape)
    leaf = key.rsplit('.', 1)[-1]
    if leaf == 'num_batches_tracked':
        return torch.zeros(shape, dtype=torch.long)
    if leaf == '_running_iter':
        return torch.ones(shape)
    if len(shape) == 4:                                   # conv weight [K,C,R,S]
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
    if leaf in ('running_var', '_tmp_running_var'):
        return torch.rand(shape, generator=g) + 0.5
    if leaf in ('running_mean', '_tmp_running_mean'):
        return torch.randn(shape, generator=g) * 0.1
    if leaf == 'weight':                                  # BN gamma; 0.9: E[g^2]E[1/running_var]~1 (eval-mode scale stays O(1))
        return (torch.rand(shape, generator=g) + 0.5) * 0.9
    if leaf == 'bias':                                    # BN beta / conv bias
        return torch.randn(shape, generator=g) * 0.1
    raise KeyError('no synthetic rule for %s %s' % (key, shape))


def _residual_tail_gammas(manifest):
    """BN gammas that feed a residual/fuse sum.  They are damped so that eval-mode
    activations (synthetic running stats do not normalise) stay O(1) through
    33-block ResNet-101 / 32-block HRNet streams instead of doubling per block."""
    keys = set()
    for k in manifest:
        if k.endswith('.bn3.weight'):
            keys.add(k)
        elif k.endswith('.bn2.weight') and (k[:-len('bn2.weight')] + 'bn3.weight') not in manifest:
            keys.add(k)
        elif 'fuse_layers' in k and k.endswith('.1.weight') and len(manifest[k]) == 1:
            keys.add(k)
    return keys


def synth_heavy_conv(key, shape, seed=0, channel_scales=True):
    """Conv weight with the statistics of a TRAINED net instead of a fresh init: heavy-tailed elements (a log-normal factor per
    element, sigma 1.5: the largest entries of a tensor are > 1e3 x its median magnitude) and output channels whose scales span
    2.5 decades (10^u, u ~ U[-1.5, 1]: the pre-BN activations of different channels then have variances from 1e-3 to 1e2 -- what
    the BN running_var of a trained checkpoint looks like; `channel_scales=False` for convs that no BN follows -- the classifiers
    -- so that the logits stay O(1)).  Overall second moment per channel = He-normal x its channel scale."""
    g = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * seed + 104729) & 0x7FFFFFFF)
    shape = tuple(shape)
    fan_in = shape[1] * shape[2] * shape[3]
    sigma = 1.5
    w = torch.randn(shape, generator=g) * torch.exp(sigma * torch.randn(shape, generator=g)) * math.exp(-sigma * sigma)
    ch = 10.0 ** (torch.rand(shape[0], generator=g) * 2.5 - 1.5)
    if not channel_scales:
        ch = torch.ones(shape[0])
    return w * math.sqrt(2.0 / fan_in) * ch.view(-1, 1, 1, 1)


def synth_state_dict(manifest, seed=0, style='he'):
    """manifest: {key: shape}.  Returns {key: tensor}.  style 'he': fresh-init statistics (synth_tensor); 'heavy': conv weights
    of synth_heavy_conv (BN running statistics then have to be calibrated, tests/golden/make_golden.py::calibrate_bn)."""
    tails = _residual_tail_gammas(manifest)
    sd = {}
    for k, s in sorted(manifest.items()):
        if style == 'heavy' and len(s) == 4:
            # a conv WITH a bias is one no BN follows (the 1x1 classifiers, models.py:366,459,462,540)
            t = synth_heavy_conv(k, s, seed, channel_scales=(k[:-len('weight')] + 'bias') not in manifest)
        else:
            t = synth_tensor(k, s, seed)
        sd[k] = t * 0.3 if k in tails else t
    return sd


def golden_state_dicts(g):
    """(encoder, decoder) state dicts of a golden case (tests/golden/*.pt): the seeded synthetic weights of the case's style,
    with the BN running statistics the generator calibrated on the reference (stored in the fixture) where the case has them"""
    m = g['meta']
    style = m.get('weights_style', 'he')
    enc = synth_state_dict(g['manifest_enc'], m['seed'], style)
    dec = synth_state_dict(g['manifest_dec'], m['seed'] + 1, style)
    for sd, key in ((enc, 'bn_running_enc'), (dec, 'bn_running_dec')):
        for k, v in (g.get(key) or {}).items():
            assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
            sd[k] = v.clone()
    return enc, dec


def synth_batch(n, h, w, seg_rate, num_class=150, seed=304):
    """SURVEY 8d synthetic inputs: img ~ randn fp32 NCHW, labels in [-1,149]
    int64 at 1/seg_rate resolution (config/defaults.py:75 seed 304)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n, 3, h, w, generator=g)
    lab = torch.randint(-1, num_class, (n, h // seg_rate, w // seg_rate), generator=g)
    return img, lab


def synth_dropout_mask(n, c, p=0.1, seed=0):
    """Replayable Dropout2d mask: per-(n,c) Bernoulli(1-p)/(1-p) (models.py:460)."""
    g = torch.Generator().manual_seed(1000003 + seed)
    keep = (torch.rand(n, c, generator=g) >= p).float()
    return keep / (1.0 - p)


# ----------------------------------------------------------------------------
# primitive layers
# ----------------------------------------------------------------------------
class Ctx:
    """Run-time switches: training (BN batch stats + running-stat update),
    dropout masks to replay ({'main': [N,C], 'deepsup': [N,C]} or None = off)."""
    def __init__(self, training=False, dropout=None):
        self.training = training
        self.dropout = dropout or {}


def _bn(sd, name, x, ctx, momentum):
    """lib/nn/modules/batchnorm.py:58-61 (the single-device / eval branch the CPU
    reference always takes) == F.batch_norm."""
    nbt = sd.get(name + '.num_batches_tracked')
    if ctx.training and nbt is not None:
        nbt += 1                                          # torch _BatchNorm.forward
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'],
                        sd[name + '.weight'], sd[name + '.bias'],
                        ctx.training, momentum, BN_EPS)


def _conv(sd, name, x, stride=1, pad=0, dil=1):
    return F.conv2d(x, sd[name + '.weight'], sd.get(name + '.bias'), stride, pad, dil)


def _up(x, size):
    return F.interpolate(x, size=tuple(size), mode='bilinear', align_corners=False)


# ----------------------------------------------------------------------------
# ResNet / ResnetDilated encoders  (models/resnet.py, models/models.py:170-268)
# ----------------------------------------------------------------------------
def _resnet_plan(arch):
    """Per-block conv geometry after ResnetDilated._nostride_dilate
    (models.py:238-251).  Returns list of layers, each a list of block dicts."""
    dilated = arch.endswith('dilated')
    kind, counts = RESNET_LAYERS[arch.replace('dilated', '')]
    exp = 1 if kind == 'basic' else 4
    plan, inplanes = [], 128                              # resnet.py:98
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), counts)):
        stride = 1 if li == 0 else 2                      # resnet.py:111-114
        dilate = {2: 2, 3: 4}.get(li, 1) if dilated else 1
        blocks = []
        for bi in range(n):
            s = stride if bi == 0 else 1
            has_ds = bi == 0 and (s != 1 or inplanes != planes * exp)   # resnet.py:127-133
            # the strided 3x3 (conv1 of BasicBlock, conv2 of Bottleneck)
            if dilate > 1:
                if s == 2:
                    s_eff, d_str = 1, dilate // 2        # models.py:242-246
                else:
                    s_eff, d_str = s, dilate             # models.py:248-251
                d_other = dilate
            else:
                s_eff, d_str, d_other = s, 1, 1
            blocks.append(dict(kind=kind, stride=s_eff, d_strided=d_str, d_other=d_other,
                               downsample=has_ds))
            inplanes = planes * exp
        plan.append(blocks)
    return plan


def _basic_block(sd, p, x, b, ctx, mom):
    """models/resnet.py:37-53"""
    out = F.relu(_bn(sd, p + 'bn1', _conv(sd, p + 'conv1', x, b['stride'], b['d_strided'], b['d_strided']), ctx, mom))
    out = _bn(sd, p + 'bn2', _conv(sd, p + 'conv2', out, 1, b['d_other'], b['d_other']), ctx, mom)
    res = x
    if b['downsample']:
        res = _bn(sd, p + 'downsample.1', _conv(sd, p + 'downsample.0', x, b['stride']), ctx, mom)
    return F.relu(out + res)


def _bottleneck(sd, p, x, b, ctx, mom):
    """models/resnet.py:72-92"""
    out = F.relu(_bn(sd, p + 'bn1', _conv(sd, p + 'conv1', x), ctx, mom))
    out = F.relu(_bn(sd, p + 'bn2', _conv(sd, p + 'conv2', out, b['stride'], b['d_strided'], b['d_strided']), ctx, mom))
    out = _bn(sd, p + 'bn3', _conv(sd, p + 'conv3', out), ctx, mom)
    res = x
    if b['downsample']:
        res = _bn(sd, p + 'downsample.1', _conv(sd, p + 'downsample.0', x, b['stride']), ctx, mom)
    return F.relu(out + res)


def resnet_encoder(sd, arch, x, ctx):
    """models.py:253-268 / :190-205 with return_feature_maps=True."""
    mom = RESNET_BN_MOMENTUM
    x = F.relu(_bn(sd, 'bn1', _conv(sd, 'conv1', x, 2, 1), ctx, mom))     # resnet.py:100-108
    x = F.relu(_bn(sd, 'bn2', _conv(sd, 'conv2', x, 1, 1), ctx, mom))
    x = F.relu(_bn(sd, 'bn3', _conv(sd, 'conv3', x, 1, 1), ctx, mom))
    x = F.max_pool2d(x, 3, 2, 1)                                          # resnet.py:109
    outs = []
    for li, blocks in enumerate(_resnet_plan(arch)):
        for bi, b in enumerate(blocks):
            p = 'layer%d.%d.' % (li + 1, bi)
            x = (_basic_block if b['kind'] == 'basic' else _bottleneck)(sd, p, x, b, ctx, mom)
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------
# ResNeXt-101 (32 groups) through models.Resnet (models/resnext.py, models.py:96-98)
# ----------------------------------------------------------------------------
RESNEXT_LAYERS = {'resnext101': (3, 4, 23, 3)}           # resnext.py:143-152
RESNEXT_GROUPS = 32                                      # resnext.py:67


def _group_bottleneck(sd, p, x, stride, downsample, ctx, mom):
    """models/resnext.py:41-62"""
    out = F.relu(_bn(sd, p + 'bn1', _conv(sd, p + 'conv1', x), ctx, mom))
    out = F.conv2d(out, sd[p + 'conv2.weight'], None, stride, 1, 1, RESNEXT_GROUPS)
    out = F.relu(_bn(sd, p + 'bn2', out, ctx, mom))
    out = _bn(sd, p + 'bn3', _conv(sd, p + 'conv3', out), ctx, mom)
    res = x
    if downsample:
        res = _bn(sd, p + 'downsample.1', _conv(sd, p + 'downsample.0', x, stride), ctx, mom)
    return F.relu(out + res)


def resnext_encoder(sd, arch, x, ctx):
    mom = RESNET_BN_MOMENTUM
    x = F.relu(_bn(sd, 'bn1', _conv(sd, 'conv1', x, 2, 1), ctx, mom))     # resnext.py:69-78, same deep stem
    x = F.relu(_bn(sd, 'bn2', _conv(sd, 'conv2', x, 1, 1), ctx, mom))
    x = F.relu(_bn(sd, 'bn3', _conv(sd, 'conv3', x, 1, 1), ctx, mom))
    x = F.max_pool2d(x, 3, 2, 1)
    outs, inplanes = [], 128
    for li, (planes, n) in enumerate(zip((128, 256, 512, 1024), RESNEXT_LAYERS[arch])):
        for bi in range(n):
            stride = (1 if li == 0 else 2) if bi == 0 else 1
            ds = bi == 0 and (stride != 1 or inplanes != planes * 2)      # resnext.py:98-105
            x = _group_bottleneck(sd, 'layer%d.%d.' % (li + 1, bi), x, stride, ds, ctx, mom)
            inplanes = planes * 2
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------
# MobileNetV2Dilated (models/mobilenet.py, models.py:271-323, dilate_scale 8)
# ----------------------------------------------------------------------------
MOBILENET_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def _relu6(x):
    return F.hardtanh(x, 0.0, 6.0)                       # nn.ReLU6


def mobilenetv2_dilated_encoder(sd, x, ctx):
    """features[0 .. 17] of mobilenet.py:95-108 after MobileNetV2Dilated._nostride_dilate (models.py:297-311): blocks 7..13
    dilate 2, blocks 14.. dilate 4; a stride-2 depthwise conv becomes stride 1 with dilation dilate // 2."""
    mom = RESNET_BN_MOMENTUM
    x = _relu6(_bn(sd, 'features.0.1', _conv(sd, 'features.0.0', x, 2, 1), ctx, mom))
    outs, idx, inp = [], 1, 32
    for t, c, n, s in MOBILENET_SETTING:
        for i in range(n):
            stride = s if i == 0 else 1
            dilate = 2 if 7 <= idx < 14 else (4 if idx >= 14 else 1)
            if dilate > 1:
                stride, d = (1, dilate // 2) if stride == 2 else (stride, dilate)
            else:
                d = 1
            p = 'features.%d.conv.' % idx
            hidden = inp * t
            y, j = x, 0
            if t != 1:
                y = _relu6(_bn(sd, p + '1', _conv(sd, p + '0', y), ctx, mom))
                j = 3
            y = F.conv2d(y, sd[p + '%d.weight' % j], None, stride, d, d, hidden)           # depthwise 3x3
            y = _relu6(_bn(sd, p + '%d' % (j + 1), y, ctx, mom))
            y = _bn(sd, p + '%d' % (j + 4), _conv(sd, p + '%d' % (j + 3), y), ctx, mom)     # linear projection
            x = x + y if (s if i == 0 else 1) == 1 and inp == c else y                      # mobilenet.py:44 (original stride)
            inp = c
            if idx in (2, 4, 7, 14):
                outs.append(x)
            idx += 1
    outs.append(x)
    return outs


# ----------------------------------------------------------------------------
# HRNetV2-W48 encoder (models/hrnet.py)
# ----------------------------------------------------------------------------
HRNET_STAGES = (('stage2', 1, (48, 96)), ('stage3', 4, (48, 96, 192)), ('stage4', 3, (48, 96, 192, 384)))


def _hr_basic(sd, p, x, ctx):
    """hrnet.py:45-61"""
    m = HRNET_BN_MOMENTUM
    out = F.relu(_bn(sd, p + 'bn1', _conv(sd, p + 'conv1', x, 1, 1), ctx, m))
    out = _bn(sd, p + 'bn2', _conv(sd, p + 'conv2', out, 1, 1), ctx, m)
    return F.relu(out + x)


def _hr_bottleneck(sd, p, x, ctx, downsample):
    """hrnet.py:81-102"""
    m = HRNET_BN_MOMENTUM
    out = F.relu(_bn(sd, p + 'bn1', _conv(sd, p + 'conv1', x), ctx, m))
    out = F.relu(_bn(sd, p + 'bn2', _conv(sd, p + 'conv2', out, 1, 1), ctx, m))
    out = _bn(sd, p + 'bn3', _conv(sd, p + 'conv3', out), ctx, m)
    res = x
    if downsample:
        res = _bn(sd, p + 'downsample.1', _conv(sd, p + 'downsample.0', x), ctx, m)
    return F.relu(out + res)


def _hr_module(sd, p, xs, ctx):
    """HighResolutionModule.forward hrnet.py:225-250 (fuse layers :176-220)."""
    m = HRNET_BN_MOMENTUM
    nb = len(xs)
    xs = list(xs)
    for i in range(nb):
        for k in range(4):
            xs[i] = _hr_basic(sd, '%sbranches.%d.%d.' % (p, i, k), xs[i], ctx)
    outs = []
    for i in range(nb):
        y = None
        for j in range(nb):
            if j == i:
                t = xs[j]
            elif j > i:
                q = '%sfuse_layers.%d.%d.' % (p, i, j)
                t = _bn(sd, q + '1', _conv(sd, q + '0', xs[j]), ctx, m)
                t = _up(t, xs[i].shape[-2:])
            else:
                t = xs[j]
                for k in range(i - j):
                    q = '%sfuse_layers.%d.%d.%d.' % (p, i, j, k)
                    t = _bn(sd, q + '1', _conv(sd, q + '0', t, 2, 1), ctx, m)
                    if k != i - j - 1:
                        t = F.relu(t)
            y = t if y is None else y + t
        outs.append(F.relu(y))
    return outs


def hrnet_encoder(sd, x, ctx):
    """HRNetV2.forward hrnet.py:392-437"""
    m = HRNET_BN_MOMENTUM
    x = F.relu(_bn(sd, 'bn1', _conv(sd, 'conv1', x, 2, 1), ctx, m))
    x = F.relu(_bn(sd, 'bn2', _conv(sd, 'conv2', x, 2, 1), ctx, m))
    for k in range(4):
        x = _hr_bottleneck(sd, 'layer1.%d.' % k, x, ctx, downsample=(k == 0))
    ys = [x]
    pre = (256,)
    for ti, (stage, nmod, chans) in enumerate(HRNET_STAGES):
        # transition layers hrnet.py:309-343
        xs = []
        for i, c in enumerate(chans):
            q = 'transition%d.%d.' % (ti + 1, i)
            if i < len(pre):
                if c != pre[i]:
                    xs.append(F.relu(_bn(sd, q + '1', _conv(sd, q + '0', ys[i], 1, 1), ctx, m)))
                else:
                    xs.append(ys[i])
            else:
                t = ys[-1]
                for j in range(i + 1 - len(pre)):
                    qq = q + '%d.' % j
                    t = F.relu(_bn(sd, qq + '1', _conv(sd, qq + '0', t, 2, 1), ctx, m))
                xs.append(t)
        for mi in range(nmod):
            xs = _hr_module(sd, '%s.%d.' % (stage, mi), xs, ctx)
        ys, pre = xs, chans
    size = ys[0].shape[-2:]
    return [torch.cat([ys[0]] + [_up(t, size) for t in ys[1:]], 1)]


# ----------------------------------------------------------------------------
# decoders (models/models.py:363-586)
# ----------------------------------------------------------------------------
def _cbr(sd, p, x, ctx):
    """conv3x3_bn_relu models.py:160-167"""
    return F.relu(_bn(sd, p + '1', _conv(sd, p + '0', x, 1, 1), ctx, RESNET_BN_MOMENTUM))


def _drop(x, ctx, which):
    mk = ctx.dropout.get(which) if ctx.training else None
    return x if mk is None else x * mk[:, :, None, None]


def _finish(x, seg_size, use_softmax):
    if use_softmax:                                       # models.py:480-484
        return F.softmax(_up(x, seg_size), dim=1)
    return F.log_softmax(x, dim=1)


def decoder_ppm(sd, conv_out, ctx, deepsup, seg_size=None, use_softmax=False):
    """PPM.forward models.py:414-434 / PPMDeepsup.forward models.py:466-495"""
    conv5 = conv_out[-1]
    size = conv5.shape[-2:]
    feats = [conv5]
    for i, s in enumerate((1, 2, 3, 6)):
        q = 'ppm.%d.' % i
        t = F.adaptive_avg_pool2d(conv5, s)
        t = F.relu(_bn(sd, q + '2', _conv(sd, q + '1', t), ctx, RESNET_BN_MOMENTUM))
        feats.append(_up(t, size))
    x = torch.cat(feats, 1)
    x = _cbr(sd, 'conv_last.', x, ctx)
    x = _drop(x, ctx, 'main')
    x = _conv(sd, 'conv_last.4', x)
    if use_softmax:
        return _finish(x, seg_size, True)
    if not deepsup:
        return F.log_softmax(x, dim=1)
    d = _cbr(sd, 'cbr_deepsup.', conv_out[-2], ctx)
    d = _drop(d, ctx, 'deepsup')
    d = _conv(sd, 'conv_last_deepsup', d)
    return F.log_softmax(x, dim=1), F.log_softmax(d, dim=1)


def decoder_c1(sd, conv_out, ctx, deepsup=False, seg_size=None, use_softmax=False):
    """C1.forward models.py:373-385 / C1DeepSup.forward models.py:341-359"""
    x = _cbr(sd, 'cbr.', conv_out[-1], ctx)
    x = _conv(sd, 'conv_last', x)
    if use_softmax:
        return _finish(x, seg_size, True)
    if not deepsup:
        return F.log_softmax(x, dim=1)
    d = _cbr(sd, 'cbr_deepsup.', conv_out[-2], ctx)
    d = _conv(sd, 'conv_last_deepsup', d)
    return F.log_softmax(x, dim=1), F.log_softmax(d, dim=1)


def decoder_upernet(sd, conv_out, ctx, seg_size=None, use_softmax=False):
    """UPerNet.forward models.py:543-586"""
    mom = RESNET_BN_MOMENTUM
    conv5 = conv_out[-1]
    size = conv5.shape[-2:]
    feats = [conv5]
    for i, s in enumerate((1, 2, 3, 6)):
        q = 'ppm_conv.%d.' % i
        t = _up(F.adaptive_avg_pool2d(conv5, s), size)     # pool -> up -> 1x1 (:548-552)
        feats.append(F.relu(_bn(sd, q + '1', _conv(sd, q + '0', t), ctx, mom)))
    f = _cbr(sd, 'ppm_last_conv.', torch.cat(feats, 1), ctx)
    fpn = [f]
    for i in reversed(range(len(conv_out) - 1)):
        q = 'fpn_in.%d.' % i
        lat = F.relu(_bn(sd, q + '1', _conv(sd, q + '0', conv_out[i]), ctx, mom))
        f = lat + _up(f, lat.shape[-2:])
        fpn.append(_cbr(sd, 'fpn_out.%d.0.' % i, f, ctx))
    fpn.reverse()
    osz = fpn[0].shape[-2:]
    x = torch.cat([fpn[0]] + [_up(t, osz) for t in fpn[1:]], 1)
    x = _cbr(sd, 'conv_last.0.', x, ctx)
    x = _conv(sd, 'conv_last.1', x)
    return _finish(x, seg_size, use_softmax)


# ----------------------------------------------------------------------------
# SegmentationModule.forward  (models/models.py:21-47)
# ----------------------------------------------------------------------------
def pixel_acc(pred, label):
    """models.py:12-18"""
    preds = pred.max(dim=1)[1]
    valid = (label >= 0).long()
    acc_sum = (valid * (preds == label).long()).sum()
    return acc_sum.float() / (valid.sum().float() + 1e-10)


def encode(enc_sd, arch_encoder, img, ctx):
    arch_encoder = arch_encoder.lower()
    if arch_encoder == 'hrnetv2':
        return hrnet_encoder(enc_sd, img, ctx)
    if arch_encoder == 'mobilenetv2dilated':
        return mobilenetv2_dilated_encoder(enc_sd, img, ctx)
    if arch_encoder in RESNEXT_LAYERS:
        return resnext_encoder(enc_sd, arch_encoder, img, ctx)
    return resnet_encoder(enc_sd, arch_encoder, img, ctx)


def decode(dec_sd, arch_decoder, conv_out, ctx, seg_size=None, use_softmax=False):
    a = arch_decoder.lower()
    if a in ('ppm', 'ppm_deepsup'):
        return decoder_ppm(dec_sd, conv_out, ctx, a == 'ppm_deepsup', seg_size, use_softmax)
    if a in ('c1', 'c1_deepsup'):
        return decoder_c1(dec_sd, conv_out, ctx, a == 'c1_deepsup', seg_size, use_softmax)
    if a in ('upernet', 'upernet_lite'):
        return decoder_upernet(dec_sd, conv_out, ctx, seg_size, use_softmax)
    raise Exception('Architecture undefined!')


def segmentation_forward(enc_sd, dec_sd, arch_encoder, arch_decoder, img, label,
                         training=False, dropout=None, deep_sup_scale=None, seg_size=None):
    """Training branch -> dict(loss, acc, pred[, pred_deepsup]); inference branch
    (seg_size given) -> softmax probabilities.  Loss = nn.NLLLoss(ignore_index=-1)
    (train.py:154) + deep_sup_scale * deepsup loss (models.py:37-40)."""
    ctx = Ctx(training, dropout)
    feats = encode(enc_sd, arch_encoder, img, ctx)
    if seg_size is not None:
        return decode(dec_sd, arch_decoder, feats, ctx, seg_size, True)
    out = decode(dec_sd, arch_decoder, feats, ctx)
    res = {'feats': feats}
    if isinstance(out, tuple):
        pred, pred_ds = out
        res['pred_deepsup'] = pred_ds
    else:
        pred, pred_ds = out, None
    loss = F.nll_loss(pred, label, ignore_index=-1)
    if deep_sup_scale is not None and pred_ds is not None:
        loss = loss + F.nll_loss(pred_ds, label, ignore_index=-1) * deep_sup_scale
    res.update(pred=pred, loss=loss, acc=pixel_acc(pred, label))
    return res


def sgd_step(params, grads, bufs, lr, momentum=0.9, weight_decay=1e-4):
    """train.py:117-126 + torch.optim.SGD: g += wd*w (conv weights only: 4-D
    tensors, train.py:92-112 group_weight); buf = m*buf + g; w -= lr*buf."""
    with torch.no_grad():
        for k, p in params.items():
            g = grads[k]
            if p.dim() == 4 and weight_decay:
                g = g + weight_decay * p
            b = bufs.get(k)
            b = g.clone() if b is None else b.mul_(momentum).add_(g)
            bufs[k] = b
            p.sub_(lr * b)


def clone_sd(sd, requires_grad=False):
    out = {}
    for k, v in sd.items():
        v = v.clone()
        if requires_grad and v.is_floating_point() and not k.rsplit('.', 1)[-1].startswith(('running', '_tmp', '_running', 'num_batches')):
            v.requires_grad_(True)
        out[k] = v
    return out


def to_dtype(sd, dtype):
    """the state dict with its floating-point tensors cast (float64: the anchor runs of the parity tests)"""
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
