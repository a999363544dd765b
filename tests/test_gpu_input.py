"""Input pipeline on the device (csrc/input_pipeline.hip through mit_semseg/dataset.py): bit-exact against the
batches of the unmodified reference TrainDataset (tests/golden/input_golden.npz) and, at the sizes of the real pipeline,
against the oracle (itself pinned against Pillow, tests/test_input_pipeline_cpu.py)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _golden_cases():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'input_golden.npz'))
    n = len([k for k in g.files if k.endswith('_params') and k.startswith('c')])
    for ci in range(n):
        pre = 'c%d_' % ci
        short, mx, pad, rate, bpg = [int(v) for v in g[pre + 'params']]
        yield dict(short=short, max_size=mx, pad=pad, rate=rate, flips=[bool(f) for f in g[pre + 'flips']],
                   images=[g[pre + 'img%d' % j] for j in range(bpg)], segms=[g[pre + 'seg%d' % j] for j in range(bpg)],
                   img_data=g[pre + 'img_data'], seg_label=g[pre + 'seg_label'])


def test_assembler_matches_reference_golden_bit_exact():
    from mit_semseg.dataset import TrainBatchAssembler
    for c in _golden_cases():
        asm = TrainBatchAssembler((c['short'],), c['max_size'], c['pad'], c['rate'], device='cuda:0')
        feed = asm.assemble([torch.from_numpy(i) for i in c['images']], [torch.from_numpy(s) for s in c['segms']], c['flips'],
                            c['short'])
        torch.cuda.synchronize()
        assert feed['img_data'].is_cuda and tuple(feed['img_data'].shape) == c['img_data'].shape
        assert feed['img_data'].permute(0, 2, 3, 1).is_contiguous()             # NHWC memory: what the conv path reads
        assert np.array_equal(feed['img_data'].cpu().numpy(), c['img_data'])
        assert np.array_equal(feed['seg_label'].cpu().numpy(), c['seg_label'])


@pytest.mark.parametrize('short,rate,pad', [(300, 8, 8), (450, 8, 8), (600, 4, 32)])
def test_assembler_at_pipeline_sizes_matches_oracle(short, rate, pad):
    """ADE20K-like photos (683 x 512 landscape, 384 x 512 portrait) at the short-side sizes of config/defaults.py:19-23:
    down- and up-sampling, imgMaxSize clamp, both flips"""
    from mit_semseg.dataset import TrainBatchAssembler
    from oracle import input_oracle as O
    rng = np.random.default_rng(short)
    images, segms = [], []
    for (h, w) in ((512, 683), (512, 384)):
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([np.sin(xx / 11.0), np.cos(yy / 13.0), np.sin((xx + yy) / 17.0)], -1) * 100 + 128
        images.append(np.clip(base + rng.normal(0, 20, (h, w, 3)), 0, 255).astype(np.uint8))
        segms.append(((yy // 37 + xx // 41) % 151).astype(np.uint8))
    flips = [True, False]
    want = O.assemble_train_batch(images, segms, flips, short, 1000, pad, rate)
    asm = TrainBatchAssembler((300, 375, 450, 525, 600), 1000, pad, rate, device='cuda:0')
    feed = asm.assemble([torch.from_numpy(i) for i in images], [torch.from_numpy(s) for s in segms], flips, short)
    torch.cuda.synchronize()
    assert np.array_equal(feed['img_data'].cpu().numpy(), want['img_data'])
    assert np.array_equal(feed['seg_label'].cpu().numpy(), want['seg_label'])
    assert int((feed['seg_label'] < -1).sum()) == 0 and int(feed['seg_label'].max()) <= 149


def test_assembled_batch_feeds_the_training_step():
    """the assembled feed is what SegmentationModule.forward consumes (models.py:29-43): one training step runs on it"""
    import torch.nn as nn
    from mit_semseg.dataset import TrainBatchAssembler
    from mit_semseg.models import ModelBuilder, SegmentationModule
    from mit_semseg.models import resnet
    from mit_semseg.models.models import ResnetDilated
    from mit_semseg.engine import TrainStep
    rng = np.random.default_rng(9)
    images = [torch.from_numpy(rng.integers(0, 256, (70, 70, 3), dtype=np.uint8)) for _ in range(2)]
    segms = [torch.from_numpy(rng.integers(0, 151, (70, 70), dtype=np.uint8)) for _ in range(2)]
    asm = TrainBatchAssembler((64,), 128, 8, 8, device='cuda:0')
    feed = asm.assemble(images, segms, [False, True], 64)
    assert tuple(feed['img_data'].shape) == (2, 3, 64, 64) and tuple(feed['seg_label'].shape) == (2, 8, 8)
    torch.manual_seed(0)
    enc = ResnetDilated(resnet.resnet18(pretrained=False), dilate_scale=8)
    dec = ModelBuilder.build_decoder('ppm_deepsup', fc_dim=512, num_class=150)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), 0.4).to('cuda:0').train()
    loss, acc = TrainStep(sm, max_iters=100).step(feed)
    torch.cuda.synchronize()
    assert torch.isfinite(loss).item() and 0.0 <= acc.item() <= 1.0


def test_eval_assembler_matches_reference_golden_bit_exact():
    """ValDataset / TestDataset inputs (dataset.py:206-296): every scale of the multi-scale list and the label map"""
    from mit_semseg.dataset import EvalImageAssembler
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'input_golden.npz'))
    p = [int(v) for v in g['val_params']]
    sizes, mx, pad = tuple(p[:-2]), p[-2], p[-1]
    asm = EvalImageAssembler(sizes, mx, pad, device='cuda:0')
    feed = asm.assemble(torch.from_numpy(g['val_img']), torch.from_numpy(g['val_seg']))
    torch.cuda.synchronize()
    assert len(feed['img_data']) == len(sizes)
    for k, t in enumerate(feed['img_data']):
        assert np.array_equal(t.cpu().numpy(), g['val_img_data%d' % k])
    assert np.array_equal(feed['seg_label'].cpu().numpy(), g['val_seg_label'])
