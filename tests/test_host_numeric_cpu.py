"""The product's Python stack, numerically, on the CPU (tests/cpu_twin.py stands in for the HIP launches): for EVERY golden
case of the unmodified reference -- all model families, train / eval / inference -- ModelBuilder + SegmentationModule +
TrainStep reproduce log-probs, loss, accuracy and, for the training cases, every parameter and BN buffer after the SGD step.
Pins module wiring, state-dict naming, parameter grouping (weight decay on conv weights only), the LR plumbing and the
dropout replay independently of any kernel."""
import os
import tempfile

import pytest
import torch
import torch.nn as nn

from tests import cpu_twin
from tests.util import golden_cases, load_golden, check_summary, anchor_ratios, check_anchor_ratios, post_step_bands, HEAVY_GOLDEN
from oracle import semseg_oracle as O


def _build(g, use_softmax):
    from mit_semseg.models import ModelBuilder, SegmentationModule
    m = g['meta']
    enc_sd, dec_sd = O.golden_state_dicts(g)
    with tempfile.TemporaryDirectory() as d:
        pe, pd = os.path.join(d, 'e.pth'), os.path.join(d, 'd.pth')
        torch.save(enc_sd, pe)
        torch.save(dec_sd, pd)
        enc = ModelBuilder.build_encoder(m['arch_encoder'], fc_dim=m['fc_dim'], weights=pe)
        dec = ModelBuilder.build_decoder(m['arch_decoder'], fc_dim=m['fc_dim'], num_class=150, weights=pd, use_softmax=use_softmax)
    if 'main' in g['dropout']:
        dec.conv_last[3].mask_override = g['dropout']['main']
    if 'deepsup' in g['dropout']:
        dec.dropout_deepsup.mask_override = g['dropout']['deepsup']
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), m['deep_sup_scale'])
    sm.train(m['training'])
    return sm


@pytest.mark.parametrize('name', [n for n in golden_cases() if n != 'cfg0_r18d_ppmds_384_eval' and n not in HEAVY_GOLDEN])      # slow on CPU
def test_python_stack_matches_reference_golden(name, monkeypatch):
    cpu_twin.install(monkeypatch)
    prev = torch.get_num_threads()
    torch.set_num_threads(min(8, prev))            # small tensors: torch's CPU kernels thrash on hundreds of threads
    try:
        _run_case(name)
    finally:
        torch.set_num_threads(prev)


def _run_case(name, step_tol=(1e-4, 1e-3)):
    g = load_golden(name)
    m = g['meta']
    sm = _build(g, use_softmax=m['seg_size'] is not None)
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    feed = {'img_data': img, 'seg_label': lab}
    if m['seg_size'] is not None:
        with torch.no_grad():
            prob = sm(feed, segSize=tuple(m['seg_size']))
        torch.testing.assert_close(prob, g['prob'], atol=1e-5, rtol=1e-4)
        return
    cap = {}
    hk = sm.decoder.register_forward_hook(lambda mod, i, o: cap.__setitem__('out', o))
    # HRNetV2: 307 small convolutions -- torch's CPU autograd takes minutes for them when the suite runs as a whole; its wiring
    # is pinned by the forward pass (training-mode BN), its optimiser plumbing by the stubbed dry run
    do_step = m['step'] and m['arch_encoder'] != 'hrnetv2'
    if do_step:
        from mit_semseg.engine import TrainStep
        ts = TrainStep(sm, lr_encoder=m['lr'], lr_decoder=m['lr'], max_iters=10 ** 9)
        loss, acc = ts.step(feed)
    else:
        with torch.no_grad():
            loss, acc = sm(feed)
    hk.remove()
    out = cap['out']
    pred, pred_ds = out if isinstance(out, tuple) else (out, None)
    torch.testing.assert_close(pred.detach(), g['pred'], atol=2e-4, rtol=1e-3)
    if pred_ds is not None:
        torch.testing.assert_close(pred_ds.detach(), g['pred_deepsup'], atol=2e-4, rtol=1e-3)
    assert abs(loss.item() - g['loss'].item()) < 1e-4 * max(1.0, abs(g['loss'].item()))
    assert abs(acc.item() - g['acc'].item()) < 1e-6
    if not do_step:
        return
    # yardstick: the reference's own fp32 reproducibility band per tensor (tests/util.anchor_ratio) -- the twin composes the
    # same torch kernels in a slightly different order (fused residual adds), which an ill-conditioned net (MobileNetV2 on
    # seeded weights) amplifies beyond any fixed tolerance
    items = []
    for mod, want, side in ((sm.encoder, g['anchor_after_enc'], 'enc.'), (sm.decoder, g['anchor_after_dec'], 'dec.')):
        sd = mod.state_dict()
        for k in want:
            if k.rsplit('.', 1)[-1] in ('_tmp_running_mean', '_tmp_running_var', '_running_iter'):
                continue
            items.append((side + k, sd[k].detach().contiguous(), want[k]))
    print(check_anchor_ratios(anchor_ratios(items, post_step_bands(g, g['meta']['lr'])), name))


def test_mobilenet_golden_with_direct_depthwise_kernels_emulated(monkeypatch, tmp_path):
    """the opt-in depthwise path (SEMSEG_DEPTHWISE_DIRECT=1: layers.GroupedConv2d -> ops.DepthwiseConv3x3Fn -> csrc/depthwise.hip)
    inside the full MobileNetV2 training step: everything else on the torch twin, the depthwise launches on the host build of
    the kernels' per-element code -- against the golden of the unmodified reference"""
    import ctypes
    import subprocess
    from mit_semseg import _native, ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'libdepthwise_emulate.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I' + os.path.join(root, 'semantic-segmentation-pytorch_amd', 'csrc'),
                    os.path.join(root, 'tests', 'native', 'depthwise_emulate.cpp'), '-o', out], check=True)
    emu = ctypes.CDLL(out)
    real = _native.lib()

    class Lib:
        def __getattr__(self, name):
            return getattr(real, name)
    lib = Lib()
    for name in ('semseg_depthwise3x3_workspace_bytes', 'semseg_depthwise3x3_fwd', 'semseg_depthwise3x3_dgrad',
                 'semseg_depthwise3x3_wgrad'):
        fn = getattr(emu, name)
        fn.restype, fn.argtypes = _native.SIGNATURES[name]
        setattr(lib, name, fn)
    cpu_twin.install(monkeypatch, keep=('depthwise_conv3x3',))      # the REAL Function, on the emulated kernels
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(ops, '_st', lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(ops, '_WS', {})
    monkeypatch.setattr(ops, 'DEPTHWISE_DIRECT', True)
    calls = []
    orig = ops.DepthwiseConv3x3Fn.apply
    monkeypatch.setattr(ops.DepthwiseConv3x3Fn, 'apply', staticmethod(lambda *a: (calls.append(1), orig(*a))[1]))
    prev = torch.get_num_threads()
    torch.set_num_threads(min(8, prev))
    try:
        # other fp32 summation order in the depthwise kernels; the stem gradient moves by 0.5 % per 1e-6 of input noise
        _run_case('mnv2d_c1ds_64_train', step_tol=(5e-4, 5e-3))
    finally:
        torch.set_num_threads(prev)
    assert len(calls) == 17                              # one depthwise conv per inverted-residual block


def test_resnext_golden_with_direct_grouped_kernels_emulated(monkeypatch, tmp_path):
    """the opt-in grouped path (SEMSEG_GROUPED_DIRECT=1 -> ops.GroupedConv3x3Fn -> csrc/grouped.hip) inside ResNeXt-101 + UPerNet:
    33 grouped convolutions on the host build of the kernels' per-element code, the rest on the torch twin"""
    import ctypes
    import subprocess
    from mit_semseg import _native, ops
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / 'libgrouped_emulate.so')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I' + os.path.join(root, 'semantic-segmentation-pytorch_amd', 'csrc'),
                    os.path.join(root, 'tests', 'native', 'grouped_emulate.cpp'), '-o', out], check=True)
    emu = ctypes.CDLL(out)
    real = _native.lib()

    class Lib:
        def __getattr__(self, name):
            return getattr(real, name)
    lib = Lib()
    for name in ('semseg_grouped3x3_workspace_bytes', 'semseg_grouped3x3_fwd', 'semseg_grouped3x3_dgrad', 'semseg_grouped3x3_wgrad'):
        fn = getattr(emu, name)
        fn.restype, fn.argtypes = _native.SIGNATURES[name]
        setattr(lib, name, fn)
    cpu_twin.install(monkeypatch, keep=('grouped_conv3x3',))      # the REAL Function, on the emulated kernels
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(ops, '_st', lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(ops, '_WS', {})
    monkeypatch.setattr(ops, 'GROUPED_DIRECT', True)
    calls = []
    orig = ops.GroupedConv3x3Fn.apply
    monkeypatch.setattr(ops.GroupedConv3x3Fn, 'apply', staticmethod(lambda *a: (calls.append(1), orig(*a))[1]))
    prev = torch.get_num_threads()
    torch.set_num_threads(min(8, prev))
    try:
        _run_case('resnext101_upernet_128_eval')
    finally:
        torch.set_num_threads(prev)
    assert len(calls) == 33


@pytest.mark.parametrize('kind,shape,affine,training', [
    ('1d', (16, 10), True, True), ('1d', (16, 10), False, True), ('1d', (16, 10), False, False),
    ('2d', (16, 10, 16, 16), True, True), ('2d', (4, 12, 8, 8), False, True)])
def test_sync_batchnorm_module_affine_false_and_odd_channels(kind, shape, affine, training, monkeypatch):
    """host logic of _SynchronizedBatchNorm (reference batchnorm.py:39-48,79-83 and its unit tests' 10-feature cases): no weight /
    bias entries with affine=False, channel padding for counts the 4-channel lanes do not divide, running statistics copied
    back -- on the CPU twin; the kernels' side is tests/test_gpu_ops.py::test_sync_batchnorm_modules_like_the_reference_unit_tests"""
    import torch.nn as nn
    cpu_twin.install(monkeypatch)
    from mit_semseg.lib.nn import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d
    c = shape[1]
    ref = (nn.BatchNorm1d if kind == '1d' else nn.BatchNorm2d)(c, eps=1e-5, momentum=0.001, affine=affine)
    mine = (SynchronizedBatchNorm1d if kind == '1d' else SynchronizedBatchNorm2d)(c, eps=1e-5, momentum=0.001, affine=affine)
    keys = set(mine.state_dict())
    assert ('weight' in keys) == affine and ('bias' in keys) == affine and '_unit_gamma' not in keys
    assert {'running_mean', 'running_var', 'num_batches_tracked', '_tmp_running_mean', '_tmp_running_var', '_running_iter'} <= keys
    ref.train(training)
    mine.train(training)
    x = torch.rand(shape, generator=torch.Generator().manual_seed(7))
    xr, xm = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    yr, ym = ref(xr), mine(xm)
    (yr * yr).sum().backward()
    (ym * ym).sum().backward()
    assert ym.shape == yr.shape
    torch.testing.assert_close(ym, yr, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(xm.grad, xr.grad, atol=1e-4, rtol=1e-4)
    torch.testing.assert_close(mine.running_mean, ref.running_mean, atol=1e-7, rtol=1e-6)
    torch.testing.assert_close(mine.running_var, ref.running_var, atol=1e-7, rtol=1e-6)
    assert int(mine.num_batches_tracked) == int(training)
