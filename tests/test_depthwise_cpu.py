"""Depthwise and grouped 3x3 kernels (csrc/depthwise.hip, csrc/grouped.hip) without a GPU: the per-element code (csrc/depthwise_math.h) is compiled for the
host behind the same C ABI (tests/native/depthwise_emulate.cpp) and the product's autograd wrapper
(ops.DepthwiseConv3x3Fn) runs on it against torch's grouped convolution in float64 -- forward, data gradient, weight
gradient; strides, dilations, odd sizes, strided (channel-slice) inputs."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _host_build(tmp_path_factory, stem):
    out = str(tmp_path_factory.mktemp(stem) / ('lib%s.so' % stem))
    src = os.path.join(ROOT, 'tests', 'native', stem + '.cpp')
    inc = os.path.join(ROOT, 'semantic-segmentation-pytorch_amd', 'csrc')
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-I' + inc, src, '-o', out], check=True)
    return ctypes.CDLL(out)


@pytest.fixture(scope='module')
def emu(tmp_path_factory):
    return _host_build(tmp_path_factory, 'depthwise_emulate')


@pytest.fixture(scope='module')
def emu_grouped(tmp_path_factory):
    return _host_build(tmp_path_factory, 'grouped_emulate')


class _HostKernels:
    def __init__(self, real, emu, signatures, family='depthwise3x3'):
        self._real = real
        for name in ('semseg_%s_workspace_bytes' % family, 'semseg_%s_fwd' % family, 'semseg_%s_dgrad' % family,
                     'semseg_%s_wgrad' % family):
            fn = getattr(emu, name)
            fn.restype, fn.argtypes = signatures[name]
            setattr(self, name, fn)

    def __getattr__(self, name):
        return getattr(self._real, name)


CASES = [  # n, c, h, w, stride, pad, dil
    (2, 32, 16, 16, 1, 1, 1), (1, 96, 17, 13, 2, 1, 1), (2, 144, 9, 11, 1, 2, 2), (1, 960, 8, 8, 1, 4, 4),
    (2, 8, 7, 5, 2, 1, 1), (1, 16, 5, 5, 1, 1, 1), (3, 24, 12, 10, 2, 2, 2)]


@pytest.mark.parametrize('case', CASES, ids=str)
def test_depthwise_function_on_emulated_kernels(case, emu, monkeypatch):
    from mit_semseg import _native, ops
    lib = _HostKernels(_native.lib(), emu, _native.SIGNATURES)
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(ops, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(ops, '_st', lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(ops, '_WS', {})
    n, c, h, w, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, 1, 3, 3, generator=g) / 3.0
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad, dil, c)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xg = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = wt.clone().requires_grad_(True)
    y = ops.depthwise_conv3x3(xg, wg, stride, pad, dil)
    assert y.shape == yr.shape
    y.backward(gy.contiguous(memory_format=torch.channels_last))

    def rel(a, b):
        return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert rel(y.detach(), yr.detach()) < 2e-6, rel(y.detach(), yr.detach())
    assert rel(xg.grad, xr.grad) < 2e-6, rel(xg.grad, xr.grad)
    assert wg.grad.shape == wt.shape and rel(wg.grad, wr.grad) < 2e-5, rel(wg.grad, wr.grad)


def test_depthwise_channel_slice_input_and_grouped_layer_switch(emu, monkeypatch):
    """x as a channel slice of a wider NHWC tensor (pixel pitch > C), and GroupedConv2d routing to the kernels"""
    from mit_semseg import _native, ops
    from mit_semseg.models.layers import GroupedConv2d
    lib = _HostKernels(_native.lib(), emu, _native.SIGNATURES)
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(ops, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(ops, '_st', lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(ops, '_WS', {})
    monkeypatch.setattr(ops, 'DEPTHWISE_DIRECT', True)
    g = torch.Generator().manual_seed(3)
    wide = torch.randn(2, 10, 9, 40, generator=g)                       # N, H, W, 40 channels
    x = wide[..., 8:24].permute(0, 3, 1, 2)                             # 16-channel slice, ld = 40
    assert ops.as_nhwc(x)[1] == 40
    m = GroupedConv2d(16, 16, 3, stride=1, padding=2, dilation=2, groups=16)
    y = m(x)
    ref = F.conv2d(x.double(), m.weight.detach().double(), None, 1, 2, 2, 16)
    assert ((y.detach().double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    y.sum().backward()
    assert m.weight.grad is not None and m.weight.grad.shape == m.weight.shape


GROUPED_CASES = [  # n, c, k, groups, h, w, stride, pad, dil
    (2, 128, 128, 32, 9, 11, 1, 1, 1), (1, 256, 256, 32, 8, 8, 2, 1, 1), (2, 64, 32, 4, 7, 5, 1, 2, 2), (1, 512, 512, 32, 4, 4, 1, 1, 1),
    (2, 32, 64, 8, 12, 10, 2, 1, 1)]


@pytest.mark.parametrize('case', GROUPED_CASES, ids=str)
def test_grouped_function_on_emulated_kernels(case, emu_grouped, monkeypatch):
    from mit_semseg import _native, ops
    lib = _HostKernels(_native.lib(), emu_grouped, _native.SIGNATURES, 'grouped3x3')
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(ops, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(ops, '_st', lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(ops, '_WS', {})
    n, c, k, groups, h, w, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c // groups, 3, 3, generator=g) / (3.0 * (c // groups) ** 0.5)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad, dil, groups)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    xg = x.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = wt.clone().requires_grad_(True)
    y = ops.grouped_conv3x3(xg, wg, groups, stride, pad, dil)
    assert y.shape == yr.shape
    y.backward(gy.contiguous(memory_format=torch.channels_last))

    def rel(a, b):
        return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert rel(y.detach(), yr.detach()) < 3e-6, rel(y.detach(), yr.detach())
    assert rel(xg.grad, xr.grad) < 3e-6, rel(xg.grad, xr.grad)
    assert wg.grad.shape == wt.shape and rel(wg.grad, wr.grad) < 2e-5, rel(wg.grad, wr.grad)


def test_grouped_layer_switch(emu_grouped, monkeypatch):
    from mit_semseg import _native, ops
    from mit_semseg.models.layers import GroupedConv2d
    lib = _HostKernels(_native.lib(), emu_grouped, _native.SIGNATURES, 'grouped3x3')
    monkeypatch.setattr(_native, 'lib', lambda: lib)
    monkeypatch.setattr(ops, '_require_cuda', lambda *a: None)
    monkeypatch.setattr(ops, '_st', lambda: ctypes.c_void_p(0))
    monkeypatch.setattr(ops, '_WS', {})
    monkeypatch.setattr(ops, 'GROUPED_DIRECT', True)
    m = GroupedConv2d(128, 128, 3, stride=2, padding=1, groups=32)
    x = torch.randn(2, 128, 10, 12).contiguous(memory_format=torch.channels_last)
    y = m(x)
    ref = F.conv2d(x.double(), m.weight.detach().double(), None, 2, 1, 1, 32)
    assert ((y.detach().double() - ref).abs().max() / ref.abs().max()).item() < 3e-6
    y.sum().backward()
    assert m.weight.grad is not None and m.weight.grad.shape == m.weight.shape
