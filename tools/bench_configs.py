"""Single-GPU step time of the OTHER BASELINE.json configs (2: R50+UPerNet, 3: R101dilated+PPM_deepsup, 4: HRNetV2+C1) at
their full size (bs 2, 512x512, synthetic data, seeded reference init), same step as bench.py (fwd + loss + bwd + 2xSGD,
hipGraph replay).  Informational -- bench.py stays on configs[1], the configuration BASELINE.json's metric is quoted on.

    python tools/bench_configs.py [--steps 20] [--warmup 6]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

# name -> (encoder builder, decoder arch, fc_dim, deep-sup scale, label downsampling, train GFLOP/img of SURVEY 8d)
CONFIGS = {
    'resnet50dilated+ppm_deepsup': ('resnet50', True, 'ppm_deepsup', 2048, 0.4, 8, 1224.2),
    'resnet50+upernet': ('resnet50', False, 'upernet', 2048, None, 4, 1468.0),
    'resnet101dilated+ppm_deepsup': ('resnet101', True, 'ppm_deepsup', 2048, 0.4, 8, 1689.6),
    'hrnetv2+c1': ('hrnetv2', False, 'c1', 720, None, 4, 625.4),
}


def build(name, dev):
    from mit_semseg.models import ModelBuilder, SegmentationModule, resnet, hrnet
    from mit_semseg.models.models import Resnet, ResnetDilated
    enc_name, dilated, dec_name, fc_dim, dss, rate, gflop = CONFIGS[name]
    torch.manual_seed(304)
    if enc_name == 'hrnetv2':
        enc = hrnet.hrnetv2(pretrained=False)
    else:
        base = resnet.__dict__[enc_name](pretrained=False)
        enc = ResnetDilated(base, 8) if dilated else Resnet(base)
    dec = ModelBuilder.build_decoder(dec_name, fc_dim=fc_dim, num_class=150)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), dss).to(dev).train()
    return sm, rate, gflop


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=6)
    ap.add_argument('--configs', default=','.join(CONFIGS))
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from mit_semseg.engine import TrainStep
    dev = torch.device('cuda:0')
    for name in args.configs.split(','):
        sm, rate, gflop = build(name, dev)
        g = torch.Generator().manual_seed(304)
        feed = {'img_data': torch.randn(2, 3, 512, 512, generator=g).to(dev),
                'seg_label': torch.randint(-1, 150, (2, 512 // rate, 512 // rate), generator=g).to(dev)}
        ts = TrainStep(sm, max_iters=100000, graph=True)
        for _ in range(args.warmup):
            loss, acc = ts.step(feed)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, acc = ts.step(feed)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        print(json.dumps({'config': name, 'ms_per_step': round(dt * 1e3, 3), 'images_per_sec': round(2 / dt, 2),
                          'step_conv_tflops': round(2 / dt * gflop * 1e-3, 1), 'final_loss': round(loss.item(), 5),
                          'params_M': round(sum(p.numel() for p in sm.parameters()) / 1e6, 2)}), flush=True)
        del ts, sm, feed
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
