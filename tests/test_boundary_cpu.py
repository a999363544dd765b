"""CPU-side checks of the drop-in boundary (no GPU, no compute through the kernels):
 * the C-ABI library builds, loads and exports every symbol include/semseg_hip.h declares;
 * ModelBuilder / SegmentationModule keep the reference's API surface, arch strings, error behaviour and
   state-dict schema (tests/golden/manifests.json was dumped from the unmodified reference);
 * the product path refuses to run without the HIP device (no CPU fallback)."""
import json
import os
import re

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from mit_semseg import _native
    header = open(os.path.join(ROOT, 'include', 'semseg_hip.h')).read()
    declared = set(re.findall(r'\b(semseg_\w+)\s*\(', header))
    assert declared, 'no declarations found'
    lib = _native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert declared == set(_native.SIGNATURES), (declared ^ set(_native.SIGNATURES))
    assert lib.semseg_abi_version() == 1
    # workspace queries are pure host functions: callable without a GPU
    assert lib.semseg_conv2d_workspace_bytes(2, 64, 64, 4096, 512, 3, 3, 1, 1, 1) >= 0
    assert lib.semseg_bn_workspace_bytes(8192, 512) > 0


def _manifest(m):
    return {k: list(v.shape) for k, v in m.state_dict().items()}


@pytest.fixture(scope='module')
def manifests():
    return json.load(open(os.path.join(ROOT, 'tests', 'golden', 'manifests.json')))


@pytest.mark.parametrize('arch', ['resnet18dilated', 'resnet50dilated', 'resnet50', 'resnet101dilated', 'hrnetv2'])
def test_encoder_state_dict_schema(arch, manifests, tmp_path):
    from mit_semseg.models import ModelBuilder
    from oracle import semseg_oracle as O
    sd = O.synth_state_dict(manifests[arch], 0)
    p = str(tmp_path / 'enc.pth')
    torch.save(sd, p)
    enc = ModelBuilder.build_encoder(arch=arch, fc_dim=2048, weights=p)     # weights file => no download
    assert _manifest(enc) == manifests[arch]
    got = enc.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    # conv weights are stored KRSC (channels_last) under the reference's logical [K,C,R,S] shape
    w = enc.state_dict()['conv1.weight']
    assert w.shape == (64, 3, 3, 3) and w.permute(0, 2, 3, 1).is_contiguous()


@pytest.mark.parametrize('arch,fc_dim', [('ppm_deepsup', 512), ('ppm_deepsup', 2048), ('upernet', 2048), ('c1', 720),
                                         ('ppm', 512)])
def test_decoder_state_dict_schema(arch, fc_dim, manifests):
    from mit_semseg.models import ModelBuilder
    dec = ModelBuilder.build_decoder(arch=arch, fc_dim=fc_dim, num_class=150)
    assert _manifest(dec) == manifests['%s@%d' % (arch, fc_dim)]
    # ModelBuilder.weights_init (models.py:52-61): BN gamma=1, beta=1e-4
    for k, v in dec.state_dict().items():
        if k.endswith('running_var') or (k.endswith('.weight') and v.dim() == 1):
            assert torch.all(v == 1), k
        if k.endswith('.bias') and k[:-5] + '.running_mean' in dec.state_dict():
            assert torch.allclose(v, torch.full_like(v, 1e-4)), k


def test_arch_strings_and_errors():
    from mit_semseg.models import ModelBuilder
    with pytest.raises(NotImplementedError):
        ModelBuilder.build_encoder(arch='resnet34', weights='x')
    with pytest.raises(NotImplementedError):
        ModelBuilder.build_encoder(arch='resnet34dilated', weights='x')
    with pytest.raises(Exception, match='Architecture undefined!'):
        ModelBuilder.build_encoder(arch='vgg16', weights='x')
    with pytest.raises(Exception, match='Architecture undefined!'):
        ModelBuilder.build_decoder(arch='fcn')
    for arch in ('c1_deepsup', 'c1', 'ppm', 'ppm_deepsup', 'upernet_lite', 'upernet'):
        ModelBuilder.build_decoder(arch=arch.upper(), fc_dim=512)      # case-insensitive like the reference


def test_public_import_surface():
    from mit_semseg.models import ModelBuilder, SegmentationModule                                     # noqa
    from mit_semseg.lib.nn import (UserScatteredDataParallel, user_scattered_collate, async_copy_to,   # noqa
                                   patch_replication_callback, SynchronizedBatchNorm2d,
                                   SynchronizedBatchNorm1d, SynchronizedBatchNorm3d, DataParallelWithCallback)
    bn = SynchronizedBatchNorm2d(8)
    assert bn.momentum == 0.001 and bn.eps == 1e-5                     # batchnorm.py:39
    assert user_scattered_collate([1, 2]) == [1, 2]
    sm = SegmentationModule(nn.Identity(), nn.Identity(), nn.NLLLoss(ignore_index=-1), 0.4)
    assert sm.deep_sup_scale == 0.4 and hasattr(sm, 'encoder') and hasattr(sm, 'decoder') and hasattr(sm, 'crit')


def test_no_cpu_fallback():
    """The HIP path must fail loudly, not fall back, when tensors are not on the GPU."""
    from mit_semseg.models import ModelBuilder
    dec = ModelBuilder.build_decoder(arch='c1', fc_dim=64, num_class=150)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dec([torch.randn(1, 64, 8, 8)])


def test_product_never_imports_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(base, f)).read()
                if re.search(r'^\s*(from|import)\s+oracle\b', src, re.M) or 'semseg_oracle' in src:
                    bad.append(f)
    assert not bad, bad


def test_side_by_side_scope_state_machine_on_the_host():
    """csrc/batch.hip without a GPU: one scope per process, branches inside their range, ordinals only inside a scope, an empty scope
    closes cleanly, abort always leaves no scope behind; the plan knob returns its previous value"""
    from mit_semseg import _native
    import ctypes
    L = _native.lib()
    EINVAL = -1
    assert L.semseg_batch_active() == 0
    assert L.semseg_batch_next_op() == EINVAL and L.semseg_batch_next_unit() == EINVAL and L.semseg_batch_branch(0) == EINVAL
    assert L.semseg_batch_end() == EINVAL
    assert L.semseg_batch_begin(0, None) == EINVAL and L.semseg_batch_begin(17, None) == EINVAL
    c0 = (ctypes.c_longlong * 4)()
    L.semseg_batch_stats(c0)
    assert L.semseg_batch_begin(4, None) == 0 and L.semseg_batch_active() == 1
    assert L.semseg_batch_begin(2, None) == EINVAL                                 # a scope inside a scope is refused
    assert L.semseg_batch_branch(3) == 0 and L.semseg_batch_branch(4) == EINVAL and L.semseg_batch_branch(-1) == EINVAL
    assert L.semseg_batch_next_op() == 0 and L.semseg_batch_next_unit() == 0
    assert L.semseg_batch_flush() == 0                                             # nothing recorded: nothing launched
    assert L.semseg_batch_end() == 0 and L.semseg_batch_active() == 0
    assert L.semseg_batch_begin(1, None) == 0 and L.semseg_batch_abort() == 0 and L.semseg_batch_active() == 0
    c1 = (ctypes.c_longlong * 4)()
    L.semseg_batch_stats(c1)
    assert c1[0] - c0[0] == 2 and c1[1] == c0[1] and c1[2] == c0[2] and c1[3] == c0[3]
    prev = L.semseg_batch_plan(15, 0)
    assert L.semseg_batch_plan(prev, 0) == 15
    assert L.semseg_batch_stats(None) == EINVAL
