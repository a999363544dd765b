"""Golden vectors for the input-pipeline contract from the UNMODIFIED reference class
mit_semseg.dataset.TrainDataset (dataset.py:70-203), run in this container:

    PYTHONPATH=/root/reference python tests/golden/make_input_golden.py

torchvision is not installed here; the one thing dataset.py uses from it (transforms.Normalize, dataset.py:34-36,57) is
stubbed with its documented semantics (tensor.sub_(mean).div_(std) per channel, fp32).  Synthetic RGB / label images are
written as PNG (lossless) so that the decoded arrays are known exactly; the random draws of __getitem__ (np.random.choice:
short-side size, flips) and the records it picked are recorded by wrapping numpy / the class method -- the reference file
itself is untouched.  Output: tests/golden/input_golden.npz (inputs, draws, and the reference's batch tensors)."""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image


class _Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32)
        self.std = torch.tensor(std, dtype=torch.float32)

    def __call__(self, t):
        return t.clone().sub_(self.mean[:, None, None]).div_(self.std[:, None, None])


tv = types.ModuleType('torchvision')
tv.transforms = types.ModuleType('torchvision.transforms')
tv.transforms.Normalize = _Normalize
sys.modules['torchvision'] = tv
sys.modules['torchvision.transforms'] = tv.transforms

import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location('ref_dataset', '/root/reference/mit_semseg/dataset.py')
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

CASES = [
    # (image sizes (h, w), imgSizes, imgMaxSize, padding_constant, segm_downsampling_rate, batch_per_gpu, seed)
    ([(37, 53), (61, 44), (50, 50), (33, 71)], (40, 48, 56), 90, 8, 8, 2, 1),
    ([(45, 80), (52, 77), (48, 64), (90, 41), (70, 35)], (32, 64), 100, 32, 4, 2, 2),
    ([(64, 48), (80, 60), (41, 29)], (72,), 128, 16, 8, 3, 3),
]


def main():
    out = {}
    for ci, (sizes, img_sizes, max_size, pad, rate, bpg, seed) in enumerate(CASES):
        rng = np.random.default_rng(100 + ci)
        with tempfile.TemporaryDirectory() as d:
            recs, imgs, segs = [], {}, {}
            for k, (h, w) in enumerate(sizes):
                # smooth-ish content + noise: exercises the antialiasing taps without being pure noise
                yy, xx = np.mgrid[0:h, 0:w]
                base = (np.stack([np.sin(xx / 5.0 + k), np.cos(yy / 7.0 - k), np.sin((xx + yy) / 9.0)], -1) * 90 + 128)
                img = np.clip(base + rng.normal(0, 25, (h, w, 3)), 0, 255).astype(np.uint8)
                seg = ((yy // 6 + xx // 9 + k) % 151).astype(np.uint8)
                fi, fs = 'img%d.png' % k, 'seg%d.png' % k
                Image.fromarray(img).save(os.path.join(d, fi))
                Image.fromarray(seg, mode='L').save(os.path.join(d, fs))
                recs.append({'fpath_img': fi, 'fpath_segm': fs, 'width': w, 'height': h})
                imgs[fi], segs[fi] = img, seg
            opt = types.SimpleNamespace(imgSizes=img_sizes, imgMaxSize=max_size, padding_constant=pad, segm_downsampling_rate=rate)
            ds = ref.TrainDataset(d, recs, opt, batch_per_gpu=bpg)
            picked = []
            orig_sub = ds._get_sub_batch
            ds._get_sub_batch = lambda: (picked.append(orig_sub()), picked[-1])[1]
            draws = []
            orig_choice = np.random.choice
            np.random.choice = lambda a, *x, **kw: (draws.append(orig_choice(a, *x, **kw)), draws[-1])[1]
            try:
                np.random.seed(seed)
                batch = ds[seed]
            finally:
                np.random.choice = orig_choice
            records = picked[0]
            short = int(draws[0]) if len(img_sizes) > 1 or isinstance(img_sizes, (list, tuple)) else int(img_sizes)
            flips = [int(v) for v in draws[1:1 + bpg]]
            assert len(draws) == 1 + bpg
            pre = 'c%d_' % ci
            out[pre + 'params'] = np.array([short, max_size, pad, rate, bpg], dtype=np.int64)
            out[pre + 'flips'] = np.array(flips, dtype=np.int64)
            for j, r in enumerate(records):
                out[pre + 'img%d' % j] = imgs[r['fpath_img']]
                out[pre + 'seg%d' % j] = segs[r['fpath_img']]
            out[pre + 'img_data'] = batch['img_data'].numpy()
            out[pre + 'seg_label'] = batch['seg_label'].numpy()
            print('case %d: short %d flips %s -> img_data %s seg_label %s' % (ci, short, flips, tuple(batch['img_data'].shape),
                                                                          tuple(batch['seg_label'].shape)))
    # ValDataset (dataset.py:206-255): multi-scale inputs of one image + its label map
    rng = np.random.default_rng(77)
    with tempfile.TemporaryDirectory() as d:
        h, w = 46, 67
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.clip(np.stack([np.sin(xx / 4.0), np.cos(yy / 6.0), np.sin((xx - yy) / 8.0)], -1) * 100 + 128 +
                      rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8)
        seg = ((yy // 5 + xx // 7) % 151).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(d, 'v.png'))
        Image.fromarray(seg, mode='L').save(os.path.join(d, 'vs.png'))
        opt = types.SimpleNamespace(imgSizes=(30, 45, 60, 75), imgMaxSize=100, padding_constant=8)
        ds = ref.ValDataset(d, [{'fpath_img': 'v.png', 'fpath_segm': 'vs.png', 'width': w, 'height': h}], opt)
        item = ds[0]
        out['val_params'] = np.array(list(opt.imgSizes) + [opt.imgMaxSize, opt.padding_constant], dtype=np.int64)
        out['val_img'], out['val_seg'] = img, seg
        assert np.array_equal(item['img_ori'], img)
        for k, t in enumerate(item['img_data']):
            out['val_img_data%d' % k] = t.numpy()
        out['val_seg_label'] = item['seg_label'].numpy()
        print('val: scales', [tuple(t.shape) for t in item['img_data']], 'label', tuple(item['seg_label'].shape))
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'input_golden.npz')
    np.savez_compressed(dst, **out)
    print('wrote', dst, os.path.getsize(dst), 'bytes')


if __name__ == '__main__':
    main()
