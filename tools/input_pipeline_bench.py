"""Input pipeline at speed: decode + device-side batch assembly in images/s against the training step rate.

Synthetic ADE20K-like files (JPEG photos + PNG label maps whose sizes are drawn from the ADE20K size histogram,
tests/golden/ade20k_train_sizes.npz) are written to a temp directory once; then `drivers._Prefetcher` (planner thread + a pool
of decode threads, the replacement of the reference's 16 DataLoader workers, train.py:163-177, dataset.py:110-199) feeds
`TrainDataset.assemble` (csrc/input_pipeline.hip) for `--batches` per-GPU batches of 2 images, for several pool sizes.
Prints one JSON line per pool size: images/s of decode alone (host), of decode + assembly (what a training rank sees), and the
ratio to `--step-rate` (images/s/GPU of the training step, default: the configs[1] headline).

    python tools/input_pipeline_bench.py --workers 1,2,4,8,16 --batches 150
"""
import argparse
import json
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def make_files(root, count, seed=0):
    from PIL import Image
    d = np.load(os.path.join(ROOT, 'tests', 'golden', 'ade20k_train_sizes.npz'))
    rng = np.random.default_rng(seed)
    pick = rng.choice(len(d['width']), size=count, p=d['count'] / d['count'].sum())
    recs = []
    for k, i in enumerate(pick):
        w, h = int(d['width'][i]), int(d['height'][i])
        # a smooth random field + noise: compresses like a photograph (a pure-noise JPEG decodes unrealistically slowly)
        base = rng.integers(0, 256, (h // 16 + 2, w // 16 + 2, 3), dtype=np.uint8)
        img = np.asarray(Image.fromarray(base).resize((w, h), Image.BILINEAR)).astype(np.int16) + rng.integers(-12, 13, (h, w, 3))
        Image.fromarray(np.clip(img, 0, 255).astype(np.uint8)).save(os.path.join(root, 'img%04d.jpg' % k), quality=90)
        seg = np.asarray(Image.fromarray(rng.integers(0, 151, (h // 32 + 2, w // 32 + 2), dtype=np.uint8), mode='L').resize((w, h), Image.NEAREST))
        Image.fromarray(seg, mode='L').save(os.path.join(root, 'seg%04d.png' % k))
        recs.append({'fpath_img': 'img%04d.jpg' % k, 'fpath_segm': 'seg%04d.png' % k, 'width': w, 'height': h})
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', default='1,2,4,8,16')
    ap.add_argument('--batches', type=int, default=150)
    ap.add_argument('--files', type=int, default=96)
    ap.add_argument('--step-rate', type=float, default=144.0, help='training images/s/GPU to compare against (configs[1])')
    args = ap.parse_args()
    import __graft_entry__ as ge
    ge.build()
    from mit_semseg.dataset import TrainDataset
    from mit_semseg.drivers import _Prefetcher
    dev = torch.device('cuda:0')
    opt = types.SimpleNamespace(imgSizes=(300, 375, 450, 525, 600), imgMaxSize=1000, padding_constant=8, segm_downsampling_rate=8)
    with tempfile.TemporaryDirectory() as root:
        t0 = time.time()
        recs = make_files(root, args.files)
        mpx = sum(r['width'] * r['height'] for r in recs) / len(recs) / 1e6
        print('# %d synthetic files (mean %.2f Mpixel) written in %.1f s; host has %d cores' % (len(recs), mpx, time.time() - t0,
                                                                                              os.cpu_count()), flush=True)
        for w in [int(x) for x in args.workers.split(',')]:
            ds = TrainDataset(root, [dict(r) for r in recs], opt, batch_per_gpu=2, device=dev)
            # (a) decode alone: the pool's futures resolved, nothing assembled
            it = _Prefetcher(ds, first_index=0, depth=8, workers=w)
            for _ in range(5):
                item = it.q.get()
                [f.result() for f in item[0]]
            t = time.perf_counter()
            for _ in range(args.batches):
                item = it.q.get()
                [f.result() for f in item[0]]
            dec = 2 * args.batches / (time.perf_counter() - t)
            # (b) decode + assembly on the device, as a training rank consumes it
            ds2 = TrainDataset(root, [dict(r) for r in recs], opt, batch_per_gpu=2, device=dev)
            it2 = _Prefetcher(ds2, first_index=0, depth=8, workers=w)
            for _ in range(5):
                next(it2)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(args.batches):
                feed = next(it2)
            torch.cuda.synchronize()
            both = 2 * args.batches / (time.perf_counter() - t)
            print(json.dumps({'decode_threads': w, 'decode_img_s': round(dec, 1), 'decode_plus_assembly_img_s': round(both, 1),
                              'step_rate_img_s': args.step_rate, 'ratio_to_step_rate': round(both / args.step_rate, 2),
                              'batch_shape': list(feed['img_data'].shape)}), flush=True)


if __name__ == '__main__':
    main()
