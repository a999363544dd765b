"""Generate tests/golden/*.pt by running the UNMODIFIED reference.

Run in the build container only (the GPU box has no /root/reference):

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

For every case: build the reference model through its public API
(ModelBuilder.build_encoder/build_decoder with a weights= file so no download is
attempted), overwrite the state dict with seeded synthetic tensors
(oracle.semseg_oracle.synth_state_dict), replace nn.Dropout2d by a replayable
mask, run SegmentationModule.forward (+ backward + 2x SGD step as train.py:34-48
does) and record outputs.  The fixtures pin oracle/semseg_oracle.py and, through
it, the HIP path.
"""
import json
import os
import sys
import tempfile

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get('SEMSEG_REFERENCE', '/root/reference')
sys.path.insert(0, REF)

from mit_semseg.models import ModelBuilder, SegmentationModule   # noqa: E402  (the reference)
from mit_semseg.models import resnet, hrnet, mobilenet, resnext, models as ref_models  # noqa: E402
from oracle import semseg_oracle as O                              # noqa: E402

assert os.path.realpath(ref_models.__file__).startswith(os.path.realpath(REF)), ref_models.__file__


class ReplayDropout(nn.Module):
    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        if not self.training or self.mask is None:
            return x
        return x * self.mask[:, :, None, None]


def build_reference(arch_enc, arch_dec, fc_dim, use_softmax=False):
    """SURVEY 8c oracle recipe (1)-(2)."""
    base = arch_enc.replace('dilated', '')
    if arch_enc == 'hrnetv2':
        enc0 = hrnet.hrnetv2(pretrained=False)
    elif arch_enc == 'mobilenetv2dilated':
        enc0 = ref_models.MobileNetV2Dilated(mobilenet.mobilenetv2(pretrained=False), dilate_scale=8)
    elif arch_enc == 'resnext101':
        enc0 = ref_models.Resnet(resnext.resnext101(pretrained=False))
    else:
        r = resnet.__dict__[base](pretrained=False)
        enc0 = ref_models.ResnetDilated(r, 8) if arch_enc.endswith('dilated') else ref_models.Resnet(r)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'enc.pth')
        torch.save(enc0.state_dict(), p)
        enc = ModelBuilder.build_encoder(arch=arch_enc, fc_dim=fc_dim, weights=p)
    dec = ModelBuilder.build_decoder(arch=arch_dec, fc_dim=fc_dim, num_class=150, weights='',
                                     use_softmax=use_softmax)
    return enc, dec


def manifest_of(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


def summarize(t, full_limit=4096):
    """Small tensors in full; large ones as (sum, abs-sum, 16 head + 16 tail values)."""
    t = t.detach().float()
    if t.numel() <= full_limit:
        return {'full': t.clone()}
    f = t.flatten()
    return {'sum': f.double().sum().item(), 'abssum': f.double().abs().sum().item(),
            'head': f[:16].clone(), 'tail': f[-16:].clone(), 'numel': t.numel()}


def sample_index(numel, k=1024):
    """the seeded element sample the anchor records of large tensors hold (tests/util.py rebuilds the same index set)"""
    return torch.randint(0, numel, (k,), generator=torch.Generator().manual_seed(numel % (2 ** 31)))


# fp32 runs of the unmodified reference that differ only in HOW the same arithmetic is executed (see anchor()): memory format,
# thread count (= reduction order of the CPU kernels), and an input perturbed below fp32 resolution
VARIANTS = ('channels_last', 'threads1', 'threads3', 'perturbed')
PERTURB_REL = 1e-7


def anchor(t32s, t64, full_limit=4096):
    """Float64 ANCHOR of one tensor of the reference + the reference's own fp32 REPRODUCIBILITY BAND around it.
    A training step through 50-300 BN/ReLU layers is ill-conditioned: backward amplifies rounding (a 1e-7 relative input
    perturbation -- below fp32 resolution -- moves the fp32 gradients of the seeded ResNet-50 / MobileNetV2 by 0.5 % in L2,
    independent of the image size), and a ReLU whose pre-activation is ~1e-7 resolves either way depending on the summation
    order, which shifts every upstream gradient a little.  Two correct fp32 implementations therefore differ by an amount no
    fixed tolerance describes: on the seeded ResNet-18 case the SAME torch CPU build moves the stem's weight gradient by 2e-5
    (default), 9e-4 (channels_last tensors) or 2e-3 (one thread) of its scale.  `t32s` = the same tensor from 1 + len(VARIANTS)
    fp32 runs of the unmodified reference (default; channels_last memory format; 1 and 3 threads; input image multiplied by
    1 + 1e-7 N(0,1)); the band is their largest deviation from the float64 run, and parity of gradients / updated weights is
    stated as `|native - ref64| <= c * band` (tests/util.check_vs_anchor)."""
    t64 = t64.detach().double().flatten()
    ds = [t.detach().double().flatten() - t64 for t in t32s]
    rec = {'numel': t64.numel(), 'err_max': max(d.abs().max().item() for d in ds), 'err_l2': max(d.norm().item() for d in ds),
           'err_max_plain': ds[0].abs().max().item(), 'err_l2_plain': ds[0].norm().item(),
           'norm': t64.norm().item(), 'absmax': t64.abs().max().item(), 'sum': t64.sum().item()}
    if t64.numel() <= full_limit:
        rec['full'] = t64.float().clone()
    else:
        rec['sample'] = t64[sample_index(t64.numel())].float().clone()
    return rec


def group_weight(module):
    """train.py:92-112 -- decay on conv/linear weights only."""
    decay, no_decay = [], []
    for m in module.modules():
        if isinstance(m, nn.modules.conv._ConvNd):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif isinstance(m, nn.modules.batchnorm._BatchNorm):
            if m.weight is not None:
                no_decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
    return [dict(params=decay), dict(params=no_decay, weight_decay=.0)]


def calibrate_bn(enc, dec, h, w, seed, n=8):
    """'heavy' weights: give every BN the running statistics a TRAINED net would carry -- the statistics of its own input -- by
    one training-mode forward of the unmodified reference modules over a calibration batch with momentum 1 (running <- batch
    statistic; running_var then spans ~1e-3 ... 1e2 like the per-channel scales of the weights).  Returns the calibrated
    buffers {key: tensor} of encoder and decoder (they travel in the fixture: they are not a function of the seed alone)."""
    bns = [m for mod in (enc, dec) for m in mod.modules() if isinstance(m, nn.modules.batchnorm._BatchNorm)]
    saved = [m.momentum for m in bns]
    for m in bns:
        m.momentum = 1.0
    enc.train(); dec.train()
    img = torch.randn(n, 3, h, w, generator=torch.Generator().manual_seed(90001 + seed))
    with torch.no_grad():
        dec(enc(img, return_feature_maps=True))
    for m, mom in zip(bns, saved):
        m.momentum = mom
        m.num_batches_tracked.zero_()
    pick = lambda mod: {k: v.clone() for k, v in mod.state_dict().items()                 # noqa: E731
                        if k.rsplit('.', 1)[-1] in ('running_mean', 'running_var')}
    return pick(enc), pick(dec)


def run_case(name, arch_enc, arch_dec, fc_dim, n, h, w, seg_rate, training, deep_sup_scale,
             step=False, seg_size=None, seed=0, grads_full=False, dtype=torch.float32, variant=None, weights_style='he',
             bn_running=None):
    torch.manual_seed(304)
    enc, dec = build_reference(arch_enc, arch_dec, fc_dim, use_softmax=seg_size is not None)
    man_e, man_d = manifest_of(enc), manifest_of(dec)
    enc.load_state_dict(O.synth_state_dict(man_e, seed, weights_style))
    dec.load_state_dict(O.synth_state_dict(man_d, seed + 1, weights_style))
    if weights_style == 'heavy':
        if bn_running is None:
            bn_running = calibrate_bn(enc, dec, h, w, seed)
        enc.load_state_dict(bn_running[0], strict=False)
        dec.load_state_dict(bn_running[1], strict=False)
    if dtype != torch.float32:
        enc, dec = enc.to(dtype), dec.to(dtype)
    masks = {}
    if training:
        # conv_last.3 is the main-head Dropout2d, dropout_deepsup the deepsup one (models.py:455-465)
        if hasattr(dec, 'conv_last') and isinstance(dec.conv_last, nn.Sequential) and len(dec.conv_last) == 5:
            masks['main'] = O.synth_dropout_mask(n, 512, seed=seed)
            dec.conv_last[3] = ReplayDropout(masks['main'].to(dtype))
        if hasattr(dec, 'dropout_deepsup'):
            masks['deepsup'] = O.synth_dropout_mask(n, fc_dim // 4, seed=seed + 1)
            dec.dropout_deepsup = ReplayDropout(masks['deepsup'].to(dtype))
    crit = nn.NLLLoss(ignore_index=-1)
    sm = SegmentationModule(enc, dec, crit, deep_sup_scale)
    sm.train(training)
    img, lab = O.synth_batch(n, h, w, seg_rate, seed=304 + seed)
    if variant == 'perturbed':
        img = img * (1.0 + PERTURB_REL * torch.randn(img.shape, generator=torch.Generator().manual_seed(1001)))
    if variant == 'channels_last':
        sm = sm.to(memory_format=torch.channels_last)
        img = img.contiguous(memory_format=torch.channels_last)
    feed = {'img_data': img.to(dtype), 'seg_label': lab}
    if dtype != torch.float32 or variant:
        # auxiliary runs for anchor(): the float64 run (same modules, weights, batch; double arithmetic) and the perturbed fp32 runs
        assert step
        opts = [torch.optim.SGD(group_weight(enc), lr=0.02, momentum=0.9, weight_decay=1e-4),
                torch.optim.SGD(group_weight(dec), lr=0.02, momentum=0.9, weight_decay=1e-4)]
        threads = torch.get_num_threads()
        if variant in ('threads1', 'threads3'):
            torch.set_num_threads(int(variant[-1]))
        sm.zero_grad()
        loss, acc = sm(feed)
        loss.backward()
        grads = ({k: p.grad.clone() for k, p in enc.named_parameters()}, {k: p.grad.clone() for k, p in dec.named_parameters()})
        for op in opts:
            op.step()
        torch.set_num_threads(threads)
        return dict(loss=loss.detach(), grads=grads, after=(
            {k: v.clone() for k, v in enc.state_dict().items() if v.is_floating_point()},
            {k: v.clone() for k, v in dec.state_dict().items() if v.is_floating_point()}))
    meta = dict(name=name, arch_encoder=arch_enc, arch_decoder=arch_dec, fc_dim=fc_dim, n=n, h=h, w=w,
                seg_rate=seg_rate, training=training, deep_sup_scale=deep_sup_scale, step=step,
                seg_size=seg_size, seed=seed, lr=0.02, torch=torch.__version__, weights_style=weights_style)
    out = {'meta': meta, 'manifest_enc': man_e, 'manifest_dec': man_d,
           'dropout': {k: v.clone() for k, v in masks.items()}}
    if bn_running is not None:
        out['bn_running_enc'], out['bn_running_dec'] = bn_running
    if seg_size is not None:
        with torch.no_grad():
            out['prob'] = sm(feed, segSize=tuple(seg_size)).clone()
        return out
    # capture decoder outputs through a hook (SegmentationModule returns only loss/acc)
    cap = {}
    hk = dec.register_forward_hook(lambda m, i, o: cap.__setitem__('out', o))
    hk2 = enc.register_forward_hook(lambda m, i, o: cap.__setitem__('feats', o))
    if not step:
        with torch.no_grad():
            loss, acc = sm(feed)
    else:
        opts = [torch.optim.SGD(group_weight(enc), lr=0.02, momentum=0.9, weight_decay=1e-4),
                torch.optim.SGD(group_weight(dec), lr=0.02, momentum=0.9, weight_decay=1e-4)]
        sm.zero_grad()
        loss, acc = sm(feed)
        loss.backward()
    hk.remove(); hk2.remove()
    o = cap['out']
    pred, pred_ds = (o if isinstance(o, tuple) else (o, None))
    out.update(loss=loss.detach().clone(), acc=acc.detach().clone(), pred=pred.detach().clone(),
               pred_deepsup=None if pred_ds is None else pred_ds.detach().clone(),
               feats=[summarize(f) for f in cap['feats']])
    if step:
        out['grads_enc'] = {k: summarize(p.grad) for k, p in enc.named_parameters()}
        out['grads_dec'] = {k: summarize(p.grad) for k, p in dec.named_parameters()}
        enc_grads = {k: p.grad.clone() for k, p in enc.named_parameters()}
        dec_grads = {k: p.grad.clone() for k, p in dec.named_parameters()}
        for op in opts:
            op.step()
        out['after_enc'] = {k: summarize(v) for k, v in enc.state_dict().items() if v.is_floating_point()}
        out['after_dec'] = {k: summarize(v) for k, v in dec.state_dict().items() if v.is_floating_point()}
        aux = dict(name=name, arch_enc=arch_enc, arch_dec=arch_dec, fc_dim=fc_dim, n=n, h=h, w=w, seg_rate=seg_rate,
                   training=training, deep_sup_scale=deep_sup_scale, step=True, seed=seed, weights_style=weights_style,
                   bn_running=bn_running)
        r64 = run_case(dtype=torch.float64, **aux)
        out['loss64'] = r64['loss'].item()
        runs = [dict(grads=(dict(enc_grads), dict(dec_grads)),
                     after=({k: v.clone() for k, v in enc.state_dict().items()}, {k: v.clone() for k, v in dec.state_dict().items()}))]
        runs += [run_case(variant=v, **aux) for v in VARIANTS]
        for i, side in enumerate(('enc', 'dec')):
            out['anchor_grads_' + side] = {k: anchor([r['grads'][i][k] for r in runs], v, 1 << 62 if grads_full else 4096)
                                           for k, v in r64['grads'][i].items()}
            out['anchor_after_' + side] = {k: anchor([r['after'][i][k] for r in runs], v) for k, v in r64['after'][i].items()}
    return out


CASES = [
    # BASELINE.json configs[0]: R18dilated+PPM_deepsup, 1 image 384x384, CPU reference forward+loss (eval, SURVEY 0 trap 2)
    dict(name='cfg0_r18d_ppmds_384_eval', arch_enc='resnet18dilated', arch_dec='ppm_deepsup', fc_dim=512,
         n=1, h=384, w=384, seg_rate=8, training=False, deep_sup_scale=0.4),
    dict(name='r18d_ppmds_64_train', arch_enc='resnet18dilated', arch_dec='ppm_deepsup', fc_dim=512,
         n=2, h=64, w=64, seg_rate=8, training=True, deep_sup_scale=0.4, step=True),
    dict(name='r50d_ppmds_64_train', arch_enc='resnet50dilated', arch_dec='ppm_deepsup', fc_dim=2048,
         n=2, h=64, w=64, seg_rate=8, training=True, deep_sup_scale=0.4, step=True),
    dict(name='r50_upernet_128_train', arch_enc='resnet50', arch_dec='upernet', fc_dim=2048,
         n=2, h=128, w=128, seg_rate=4, training=True, deep_sup_scale=None, step=True),
    dict(name='r101d_ppmds_72x104_eval', arch_enc='resnet101dilated', arch_dec='ppm_deepsup', fc_dim=2048,
         n=1, h=72, w=104, seg_rate=8, training=False, deep_sup_scale=0.4),
    dict(name='hrnetv2_c1_64_train', arch_enc='hrnetv2', arch_dec='c1', fc_dim=720,
         n=2, h=64, w=64, seg_rate=4, training=True, deep_sup_scale=None, step=True),
    # the same step at 128 x 128: at 64 x 64 the coarsest HRNet branch is a 2 x 2 map -- its BNs normalise over 8 values -- and
    # the gradients of that case sit on a knife edge (profiles/r4_anchor_control_h2_vs_f32.txt: the exact-fp32 kernels land 4
    # bands from the anchor in the MEDIAN tensor, the h2 kernels 0.00 or 4 bands depending on the launch plans of the run);
    # the gradient / post-step acceptance of HRNetV2 is therefore taken on this case (32 values per channel at the coarsest level)
    dict(name='hrnetv2_c1_128_train', arch_enc='hrnetv2', arch_dec='c1', fc_dim=720,
         n=2, h=128, w=128, seg_rate=4, training=True, deep_sup_scale=None, step=True, seed=6),
    dict(name='r18d_ppm_infer_64x80', arch_enc='resnet18dilated', arch_dec='ppm', fc_dim=512,
         n=1, h=64, w=80, seg_rate=8, training=False, deep_sup_scale=None, seg_size=[35, 45]),
    dict(name='r50_upernet_infer_64', arch_enc='resnet50', arch_dec='upernet', fc_dim=2048,
         n=1, h=64, w=64, seg_rate=4, training=False, deep_sup_scale=None, seg_size=[40, 40]),
    # the remaining arch strings of ModelBuilder (depthwise / grouped convolutions)
    dict(name='mnv2d_c1ds_64_train', arch_enc='mobilenetv2dilated', arch_dec='c1_deepsup', fc_dim=320,
         n=2, h=64, w=64, seg_rate=8, training=True, deep_sup_scale=0.4, step=True),
    dict(name='resnext101_upernet_128_eval', arch_enc='resnext101', arch_dec='upernet', fc_dim=2048,
         n=2, h=128, w=128, seg_rate=4, training=False, deep_sup_scale=None),
    # grouped / depthwise TRAINING steps at sizes where every BN sees >= 1024 values per channel (deepest maps: /8 of 192^2
    # x 2 images = 1152; /32 of 512^2 x 4 images = 1024); the MobileNetV2 case stores every gradient tensor in full
    dict(name='mnv2d_c1ds_192_train', arch_enc='mobilenetv2dilated', arch_dec='c1_deepsup', fc_dim=320,
         n=2, h=192, w=192, seg_rate=8, training=True, deep_sup_scale=0.4, step=True, seed=2, grads_full=True),
    dict(name='resnext101_c1_512_train', arch_enc='resnext101', arch_dec='c1', fc_dim=2048,
         n=4, h=512, w=512, seg_rate=32, training=True, deep_sup_scale=None, step=True, seed=3),
    # weights and BN statistics of a TRAINED net instead of a fresh init (O.synth_heavy_conv + calibrate_bn): heavy-tailed conv
    # weights (max / median > 1e3), per-channel scales over 2.5 decades, running_var from ~1e-3 to ~1e2 consistent with the
    # activations.  Eval mode consumes the running statistics; the training case starts its EMA from them.
    dict(name='r50d_ppmds_64_trainedlike_eval', arch_enc='resnet50dilated', arch_dec='ppm_deepsup', fc_dim=2048,
         n=2, h=64, w=64, seg_rate=8, training=False, deep_sup_scale=0.4, seed=4, weights_style='heavy'),
    dict(name='r18d_ppmds_64_trainedlike_train', arch_enc='resnet18dilated', arch_dec='ppm_deepsup', fc_dim=512,
         n=2, h=64, w=64, seg_rate=8, training=True, deep_sup_scale=0.4, step=True, seed=5, weights_style='heavy'),
    # the arch strings of ModelBuilder no other case builds: the undilated resnet18 / resnet101 encoders (stride 32) and the
    # 256-wide UPerNet (`upernet_lite`); UPerNet's fpn_inplanes are fixed at (256 ... 2048) in build_decoder, so it only pairs
    # with the bottleneck encoders; the deep-supervision heads read layer3 at the resolution of layer4, so they only pair with the
    # dilated encoders
    dict(name='r101_upernetlite_128_train', arch_enc='resnet101', arch_dec='upernet_lite', fc_dim=2048,
         n=2, h=128, w=128, seg_rate=4, training=True, deep_sup_scale=None, step=True, seed=7),
    dict(name='r18_c1_128_train', arch_enc='resnet18', arch_dec='c1', fc_dim=512,
         n=2, h=128, w=128, seg_rate=32, training=True, deep_sup_scale=None, step=True, seed=8),
    # the inference branch (use_softmax: resize to segSize + softmax) of the C1 heads, which the PPM / UPerNet inference cases do not build
    dict(name='hrnetv2_c1_infer_64x96', arch_enc='hrnetv2', arch_dec='c1', fc_dim=720,
         n=1, h=64, w=96, seg_rate=4, training=False, deep_sup_scale=None, seg_size=[50, 75], seed=9),
    dict(name='mnv2d_c1ds_infer_64x80', arch_enc='mobilenetv2dilated', arch_dec='c1_deepsup', fc_dim=320,
         n=1, h=64, w=80, seg_rate=8, training=False, deep_sup_scale=None, seg_size=[37, 45], seed=10),
]


def main():
    torch.set_num_threads(os.cpu_count())
    only = set(sys.argv[1:])                 # `make_golden.py name ...`: regenerate just these cases (manifests are merged)
    mpath = os.path.join(HERE, 'manifests.json')
    manifests = json.load(open(mpath)) if (only and os.path.exists(mpath)) else {}
    for c in CASES:
        if only and c['name'] not in only:
            continue
        r = run_case(**c)
        path = os.path.join(HERE, c['name'] + '.pt')
        manifests[c['arch_enc']] = r['manifest_enc']
        manifests[c['arch_dec'] + '@%d' % c['fc_dim']] = r['manifest_dec']
        torch.save(r, path)
        extra = '' if 'loss' not in r else ' loss=%.6f acc=%.6f' % (r['loss'].item(), r['acc'].item())
        print('%-28s %7.1f KB%s' % (c['name'], os.path.getsize(path) / 1024, extra))
    with open(os.path.join(HERE, 'manifests.json'), 'w') as f:
        json.dump(manifests, f)


if __name__ == '__main__':
    main()
