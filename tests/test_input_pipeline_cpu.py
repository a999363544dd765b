"""Input-pipeline contract without a GPU:
  * the oracle (oracle/input_oracle.py) is pinned against Pillow itself over random sizes and against the golden batches
    produced by the UNMODIFIED reference TrainDataset (tests/golden/make_input_golden.py);
  * the product's host-side table builders (C, inside libsemseg_hip.so) are pinned against the oracle's;
  * the product's Python assembly (mit_semseg/dataset.py) runs end to end with the three kernel entry points replaced by a
    host build of the SAME per-element code (tests/native/input_emulate.cpp includes csrc/input_pipeline_math.h) and must
    reproduce the reference's batches bit for bit.  The GPU tests (tests/test_gpu_input.py) then only add the launch."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import input_oracle as O  # noqa: E402

GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'input_golden.npz')


def golden_cases():
    g = np.load(GOLDEN)
    n = len([k for k in g.files if k.endswith('_params') and k.startswith('c')])
    for ci in range(n):
        pre = 'c%d_' % ci
        short, mx, pad, rate, bpg = [int(v) for v in g[pre + 'params']]
        yield dict(short=short, max_size=mx, pad=pad, rate=rate, flips=[bool(f) for f in g[pre + 'flips']],
                   images=[g[pre + 'img%d' % j] for j in range(bpg)], segms=[g[pre + 'seg%d' % j] for j in range(bpg)],
                   img_data=g[pre + 'img_data'], seg_label=g[pre + 'seg_label'])


def test_oracle_resize_matches_pillow():
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(0)
    for _ in range(40):
        h, w = int(rng.integers(1, 80)), int(rng.integers(1, 80))
        oh, ow = int(rng.integers(1, 110)), int(rng.integers(1, 110))
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(np.array(Image.fromarray(img).resize((ow, oh), Image.BILINEAR)), O.pil_resize_bilinear(img, (ow, oh)))
        seg = rng.integers(0, 151, (h, w), dtype=np.uint8)
        assert np.array_equal(np.array(Image.fromarray(seg).resize((ow, oh), Image.NEAREST)), O.pil_resize_nearest(seg, (ow, oh)))
    # sizes of the real pipeline: a 683 x 512 photo to short side 300 / 600 (down- and up-sampling)
    img = rng.integers(0, 256, (512, 683, 3), dtype=np.uint8)
    for size in ((400, 300), (800, 600)):
        assert np.array_equal(np.array(Image.fromarray(img).resize(size, Image.BILINEAR)), O.pil_resize_bilinear(img, size))


def test_oracle_matches_reference_golden():
    """bit-exact: float image tensor and int64 labels of the reference's own TrainDataset.__getitem__"""
    for c in golden_cases():
        out = O.assemble_train_batch(c['images'], c['segms'], c['flips'], c['short'], c['max_size'], c['pad'], c['rate'])
        assert out['img_data'].dtype == np.float32 and out['seg_label'].dtype == np.int64
        assert np.array_equal(out['img_data'], c['img_data'])
        assert np.array_equal(out['seg_label'], c['seg_label'])


def test_native_table_builders_match_oracle():
    from mit_semseg import _native
    L = _native.lib()
    rng = np.random.default_rng(1)
    pairs = [(int(rng.integers(1, 900)), int(rng.integers(1, 900))) for _ in range(80)] + [(683, 400), (512, 300), (1, 1), (5, 5)]
    for a, b in pairs:
        ks = L.semseg_input_resample_ksize(a, b)
        bounds, kk = np.zeros((b, 2), np.int32), np.zeros((b, ks), np.int32)
        assert L.semseg_input_resample_coeffs(a, b, bounds.ctypes.data, kk.ctypes.data) == 0
        ob, ok = O.resample_coeffs(a, b)
        assert ok.shape == kk.shape and np.array_equal(ob, bounds) and np.array_equal(ok, kk), (a, b)
        tab = np.zeros(b, np.int32)
        assert L.semseg_input_nearest_table(a, b, tab.ctypes.data) == 0
        assert np.array_equal(tab, O.nearest_index_table(a, b)), (a, b)
    assert L.semseg_input_resample_coeffs(0, 4, None, None) != 0


@pytest.fixture(scope='module')
def emulated_kernels(tmp_path_factory):
    """host build of the kernels' per-element code"""
    out = str(tmp_path_factory.mktemp('emu') / 'libinput_emulate.so')
    src = os.path.join(ROOT, 'tests', 'native', 'input_emulate.cpp')
    inc = os.path.join(ROOT, 'semantic-segmentation-pytorch_amd', 'csrc')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I' + inc, src, '-o', out], check=True)
    return ctypes.CDLL(out)


class _HostKernels:
    """the real library for the host-side entry points, the emulation for the three launches"""

    def __init__(self, real, emu, signatures):
        self._real = real
        for name in ('semseg_input_resample_h_u8', 'semseg_input_resample_v_normalize', 'semseg_input_label_gather'):
            fn = getattr(emu, name)
            fn.restype, fn.argtypes = signatures[name]
            setattr(self, name, fn)

    def __getattr__(self, name):
        return getattr(self._real, name)


def test_product_assembly_on_emulated_kernels_matches_reference(emulated_kernels, monkeypatch):
    from mit_semseg import _native, ops
    from mit_semseg import dataset as D
    lib = _HostKernels(_native.lib(), emulated_kernels, _native.SIGNATURES)
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(D, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(D, '_st', lambda: ctypes.c_void_p(0))
    D._resample_tables.cache_clear()
    D._nearest_table.cache_clear()
    for c in golden_cases():
        asm = D.TrainBatchAssembler((c['short'],), c['max_size'], c['pad'], c['rate'], device='cpu')
        feed = asm.assemble([torch.from_numpy(i) for i in c['images']], [torch.from_numpy(s) for s in c['segms']], c['flips'],
                            c['short'])
        assert tuple(feed['img_data'].shape) == c['img_data'].shape and feed['img_data'].dtype == torch.float32
        assert feed['seg_label'].dtype == torch.int64
        assert np.array_equal(feed['img_data'].numpy(), c['img_data'])
        assert np.array_equal(feed['seg_label'].numpy(), c['seg_label'])


def test_product_dataset_mirror_draws_like_the_reference(emulated_kernels, monkeypatch, tmp_path):
    """TrainDataset mirror: same record grouping and numpy draws as dataset.py:85-125,162 -> for the same np.random state it
    assembles the batch the oracle assembles from the records / flips / size the reference logic picks"""
    Image = pytest.importorskip('PIL.Image')
    import types
    from mit_semseg import _native
    from mit_semseg import dataset as D
    lib = _HostKernels(_native.lib(), emulated_kernels, _native.SIGNATURES)
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(D, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(D, '_st', lambda: ctypes.c_void_p(0))
    rng = np.random.default_rng(5)
    recs, store = [], {}
    for k, (h, w) in enumerate([(40, 60), (66, 45), (50, 52), (35, 70), (72, 38)]):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        seg = rng.integers(0, 151, (h, w), dtype=np.uint8)
        Image.fromarray(img).save(str(tmp_path / ('i%d.png' % k)))
        Image.fromarray(seg, mode='L').save(str(tmp_path / ('s%d.png' % k)))
        recs.append({'fpath_img': 'i%d.png' % k, 'fpath_segm': 's%d.png' % k, 'width': w, 'height': h})
        store['i%d.png' % k] = (img, seg)
    opt = types.SimpleNamespace(imgSizes=(40, 56), imgMaxSize=100, padding_constant=8, segm_downsampling_rate=8)
    ds = D.TrainDataset(str(tmp_path), [dict(r) for r in recs], opt, batch_per_gpu=2, device='cpu')
    feed = ds[7]
    # replay the reference logic (dataset.py:110-125,162) on the same seed to learn what it must have picked
    lst = [dict(r) for r in recs]
    np.random.seed(7)
    np.random.shuffle(lst)
    groups, cur, picked = [[], []], 0, None
    while picked is None:
        r = lst[cur]
        groups[0 if r['height'] > r['width'] else 1].append(r)
        cur += 1
        if cur >= len(lst):
            cur = 0
            np.random.shuffle(lst)
        for g in groups:
            if len(g) == 2:
                picked = g
                break
    short = np.random.choice(opt.imgSizes)
    flips = [bool(np.random.choice([0, 1])) for _ in picked]
    want = O.assemble_train_batch([store[r['fpath_img']][0] for r in picked], [store[r['fpath_img']][1] for r in picked], flips,
                                  short, 100, 8, 8)
    assert np.array_equal(feed['img_data'].numpy(), want['img_data'])
    assert np.array_equal(feed['seg_label'].numpy(), want['seg_label'])


def test_decode_pool_yields_the_sequential_batches(emulated_kernels, monkeypatch, tmp_path):
    """drivers._Prefetcher with a pool of decode threads (the replacement of the reference's 16 DataLoader workers,
    train.py:163-177): batch k of the pooled iterator is exactly dataset[first + k] of a second, sequentially read dataset with
    the same numpy seed -- the planner thread makes the random draws in the reference's order whatever order the pool finishes
    the file decodes in."""
    Image = pytest.importorskip('PIL.Image')
    import types
    from mit_semseg import _native
    from mit_semseg import dataset as D
    from mit_semseg.drivers import _Prefetcher
    lib = _HostKernels(_native.lib(), emulated_kernels, _native.SIGNATURES)
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(D, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(D, '_st', lambda: ctypes.c_void_p(0))
    rng = np.random.default_rng(11)
    recs = []
    for k in range(9):
        h, w = int(rng.integers(30, 60)), int(rng.integers(30, 60))
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(str(tmp_path / ('i%d.png' % k)))
        Image.fromarray(rng.integers(0, 151, (h, w), dtype=np.uint8), mode='L').save(str(tmp_path / ('s%d.png' % k)))
        recs.append({'fpath_img': 'i%d.png' % k, 'fpath_segm': 's%d.png' % k, 'width': w, 'height': h})
    opt = types.SimpleNamespace(imgSizes=(32, 40, 48), imgMaxSize=80, padding_constant=8, segm_downsampling_rate=8)
    seq = D.TrainDataset(str(tmp_path), [dict(r) for r in recs], opt, batch_per_gpu=2, device='cpu')
    want = [seq[3 + k] for k in range(7)]                  # sequential reads: seeds numpy with the first index, then draws on
    pooled = D.TrainDataset(str(tmp_path), [dict(r) for r in recs], opt, batch_per_gpu=2, device='cpu')
    it = _Prefetcher(pooled, first_index=3, depth=3, workers=4)
    # the planner thread re-seeds numpy with the first index and runs ahead by `depth` batches; the sequential read is complete
    got = [next(it) for _ in range(7)]
    for a, b in zip(got, want):
        assert np.array_equal(a['img_data'].numpy(), b['img_data'].numpy())
        assert np.array_equal(a['seg_label'].numpy(), b['seg_label'].numpy())
    # a decode error surfaces in the consumer instead of killing a pool thread silently
    os.remove(str(tmp_path / 'i0.png'))
    with pytest.raises(Exception):
        for _ in range(20):
            next(it)


def val_golden():
    g = np.load(GOLDEN)
    p = [int(v) for v in g['val_params']]
    sizes, mx, pad = tuple(p[:-2]), p[-2], p[-1]
    return dict(sizes=sizes, max_size=mx, pad=pad, img=g['val_img'], seg=g['val_seg'],
                img_data=[g['val_img_data%d' % k] for k in range(len(sizes))], seg_label=g['val_seg_label'])


def test_oracle_eval_inputs_match_reference_golden():
    v = val_golden()
    out = O.eval_image_inputs(v['img'], v['sizes'], v['max_size'], v['pad'], v['seg'])
    assert len(out['img_data']) == len(v['img_data'])
    for a, b in zip(out['img_data'], v['img_data']):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert np.array_equal(out['seg_label'], v['seg_label'])


def test_product_eval_assembly_on_emulated_kernels_matches_reference(emulated_kernels, monkeypatch):
    from mit_semseg import _native
    from mit_semseg import dataset as D
    lib = _HostKernels(_native.lib(), emulated_kernels, _native.SIGNATURES)
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(D, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(D, '_st', lambda: ctypes.c_void_p(0))
    v = val_golden()
    asm = D.EvalImageAssembler(v['sizes'], v['max_size'], v['pad'], device='cpu')
    feed = asm.assemble(torch.from_numpy(v['img']), torch.from_numpy(v['seg']))
    assert len(feed['img_data']) == len(v['img_data'])
    for a, b in zip(feed['img_data'], v['img_data']):
        assert tuple(a.shape) == b.shape and np.array_equal(a.numpy(), b)
    assert np.array_equal(feed['seg_label'].numpy(), v['seg_label'])
    assert 'seg_label' not in asm.assemble(torch.from_numpy(v['img']))           # TestDataset: no label map


def test_val_and_test_dataset_mirrors(emulated_kernels, monkeypatch, tmp_path):
    """ValDataset / TestDataset (dataset.py:206-296): keys, img_ori, info, shard selection, and the tensors the oracle
    computes from the decoded files"""
    Image = pytest.importorskip('PIL.Image')
    import types
    from mit_semseg import _native
    from mit_semseg import dataset as D
    lib = _HostKernels(_native.lib(), emulated_kernels, _native.SIGNATURES)
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(D, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(D, '_st', lambda: ctypes.c_void_p(0))
    rng = np.random.default_rng(11)
    recs, store = [], []
    for k, (h, w) in enumerate([(33, 47), (52, 40), (41, 41)]):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        seg = rng.integers(0, 151, (h, w), dtype=np.uint8)
        Image.fromarray(img).save(str(tmp_path / ('i%d.png' % k)))
        Image.fromarray(seg, mode='L').save(str(tmp_path / ('s%d.png' % k)))
        recs.append({'fpath_img': 'i%d.png' % k, 'fpath_segm': 's%d.png' % k, 'width': w, 'height': h})
        store.append((img, seg))
    opt = types.SimpleNamespace(imgSizes=(24, 40), imgMaxSize=64, padding_constant=8)
    ds = D.ValDataset(str(tmp_path), recs, opt, device='cpu', start_idx=1, end_idx=3)       # eval_multipro.py shard
    assert len(ds) == 2
    item = ds[0]
    img, seg = store[1]
    assert set(item) == {'img_ori', 'img_data', 'seg_label', 'info'} and item['info'] == 'i1.png'
    assert np.array_equal(item['img_ori'], img)
    want = O.eval_image_inputs(img, opt.imgSizes, opt.imgMaxSize, opt.padding_constant, seg)
    assert len(item['img_data']) == 2
    for a, b in zip(item['img_data'], want['img_data']):
        assert np.array_equal(a.numpy(), b)
    assert np.array_equal(item['seg_label'].numpy(), want['seg_label'])
    ts = D.TestDataset([{'fpath_img': str(tmp_path / 'i0.png')}], opt, device='cpu')
    t = ts[0]
    assert set(t) == {'img_ori', 'img_data', 'info'}
    want = O.eval_image_inputs(store[0][0], opt.imgSizes, opt.imgMaxSize, opt.padding_constant)
    for a, b in zip(t['img_data'], want['img_data']):
        assert np.array_equal(a.numpy(), b)
