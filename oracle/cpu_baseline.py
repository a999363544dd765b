"""CPU timing of the reference path for bench.py's `cpu_baseline` (SURVEY 8d "CPU reference timing").  TEST / MEASUREMENT
INFRASTRUCTURE, not product code: only bench.py's cpu_baseline leg runs it (as a subprocess, so the reference's
`mit_semseg` package and the product's never share an interpreter).

    python oracle/cpu_baseline.py --config resnet50dilated+ppm_deepsup --n 2 --h 512 --w 512 --threads 32 --steps 3

kind "reference": the UNMODIFIED reference (`/root/reference`, train.py:34-48: SegmentationModule forward, backward,
2 x torch.optim.SGD with train.py:92-126's parameter groups) when that tree is importable -- it is in the build container,
not on the GPU box.  kind "port": oracle/semseg_oracle.py (the same torch CPU kernels through torch.nn.functional) otherwise.
Same synthetic batch, same seeded weights, 1 warm-up + `--steps` timed training steps; prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# name -> (arch_encoder, arch_decoder, fc_dim, deep_sup_scale, segm_downsampling_rate)   (config/*.yaml of the reference)
CONFIGS = {
    'resnet18dilated+ppm_deepsup': ('resnet18dilated', 'ppm_deepsup', 512, 0.4, 8),
    'resnet50dilated+ppm_deepsup': ('resnet50dilated', 'ppm_deepsup', 2048, 0.4, 8),
    'resnet50+upernet': ('resnet50', 'upernet', 2048, None, 4),
    'resnet101dilated+ppm_deepsup': ('resnet101dilated', 'ppm_deepsup', 2048, 0.4, 8),
    'hrnetv2+c1': ('hrnetv2', 'c1', 720, None, 4),
}


def time_reference(ref_root, cfg, n, h, w, steps):
    sys.path.insert(0, ref_root)
    sys.path.insert(0, ROOT)
    import torch
    import torch.nn as nn
    from mit_semseg.models import ModelBuilder, SegmentationModule
    from mit_semseg.models import models as ref_models
    assert os.path.realpath(ref_models.__file__).startswith(os.path.realpath(ref_root)), ref_models.__file__
    from tests.golden.make_golden import build_reference, group_weight
    from oracle import semseg_oracle as O
    arch_enc, arch_dec, fc_dim, dss, rate = cfg
    torch.manual_seed(304)
    enc, dec = build_reference(arch_enc, arch_dec, fc_dim)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), dss).train()
    opts = [torch.optim.SGD(group_weight(enc), lr=0.02, momentum=0.9, weight_decay=1e-4),
            torch.optim.SGD(group_weight(dec), lr=0.02, momentum=0.9, weight_decay=1e-4)]
    img, lab = O.synth_batch(n, h, w, rate)
    feed = {'img_data': img, 'seg_label': lab}

    def step():
        sm.zero_grad()
        loss, acc = sm(feed)
        loss.backward()
        for o in opts:
            o.step()
        return loss.item()
    return _time(step, steps)


def time_port(cfg, n, h, w, steps):
    sys.path.insert(0, ROOT)
    import torch
    from oracle import semseg_oracle as O
    arch_enc, arch_dec, fc_dim, dss, rate = cfg
    man = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifests.json')))
    enc = O.clone_sd(O.synth_state_dict(man[arch_enc], 0), True)
    dec = O.clone_sd(O.synth_state_dict(man['%s@%d' % (arch_dec, fc_dim)], 1), True)
    img, lab = O.synth_batch(n, h, w, rate)
    masks = {'main': O.synth_dropout_mask(n, 512), 'deepsup': O.synth_dropout_mask(n, fc_dim // 4, seed=1)}
    bufs = ({}, {})

    def step():
        for sd in (enc, dec):
            for v in sd.values():
                v.grad = None
        res = O.segmentation_forward(enc, dec, arch_enc, arch_dec, img, lab, training=True, dropout=masks,
                                     deep_sup_scale=dss)
        res['loss'].backward()
        for sd, b in zip((enc, dec), bufs):
            params = {k: v for k, v in sd.items() if v.requires_grad}
            O.sgd_step(params, {k: v.grad for k, v in params.items()}, b, 0.02)
        return res['loss'].item()
    return _time(step, steps)


def _time(step, steps):
    t0 = time.perf_counter()
    step()                                   # warm-up (allocator, oneDNN primitive caches)
    warm = time.perf_counter() - t0
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        loss = step()
        ts.append(time.perf_counter() - t0)
    return warm, ts, loss


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='resnet50dilated+ppm_deepsup')
    ap.add_argument('--n', type=int, default=2)
    ap.add_argument('--h', type=int, default=512)
    ap.add_argument('--w', type=int, default=512)
    ap.add_argument('--threads', type=int, default=0)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--reference', default=os.environ.get('SEMSEG_REFERENCE', '/root/reference'))
    ap.add_argument('--impl', default='auto', choices=('auto', 'reference', 'port'))
    args = ap.parse_args()
    import torch
    host = os.cpu_count() or 1
    threads = args.threads or host
    torch.set_num_threads(threads)
    cfg = CONFIGS[args.config]
    have_ref = os.path.isdir(os.path.join(args.reference, 'mit_semseg'))
    impl = args.impl if args.impl != 'auto' else ('reference' if have_ref else 'port')
    if impl == 'reference':
        warm, ts, loss = time_reference(args.reference, cfg, args.n, args.h, args.w, args.steps)
    else:
        warm, ts, loss = time_port(cfg, args.n, args.h, args.w, args.steps)
    mean = sum(ts) / len(ts)
    print(json.dumps({'value': round(args.n / mean, 4), 'unit': 'images/sec', 'cores': threads, 'host_cores': host,
                      'kind': impl,
                      'sample': '%s: 1 warm-up (%.1f s) + %d timed training steps (fwd+loss+bwd+2xSGD) of the same %dx%dx%d '
                                'synthetic batch, torch %s CPU, %d threads of %d host cores; step times %s s'
                                % (args.config, warm, len(ts), args.n, args.h, args.w, torch.__version__, threads, host,
                                   [round(t, 2) for t in ts]),
                      'final_loss': round(loss, 5)}), flush=True)


if __name__ == '__main__':
    main()
