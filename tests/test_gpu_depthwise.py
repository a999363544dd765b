"""Depthwise and grouped 3x3 kernels on the device (csrc/depthwise.hip, csrc/grouped.hip; opt-in SEMSEG_DEPTHWISE_DIRECT=1 /
SEMSEG_GROUPED_DIRECT=1) against torch's grouped
convolution in float64; the per-element code is the one tests/test_depthwise_cpu.py checks on the host."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('case', [(2, 32, 64, 64, 1, 1, 1), (2, 96, 33, 29, 2, 1, 1), (2, 384, 16, 16, 1, 2, 2),
                                  (2, 960, 8, 8, 1, 4, 4), (1, 144, 40, 56, 2, 1, 1)], ids=str)
def test_depthwise3x3_vs_float64(case):
    from mit_semseg import ops
    n, c, h, w, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(c, 1, 3, 3, generator=g) / 3.0
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad, dil, c)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    dev = torch.device('cuda:0')
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = wt.to(dev).requires_grad_(True)
    y = ops.depthwise_conv3x3(xg, wg, stride, pad, dil)
    y.backward(gy.to(dev).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert rel(y.detach(), yr.detach()) < 2e-6
    assert rel(xg.grad, xr.grad) < 2e-6
    assert rel(wg.grad, wr.grad) < 2e-5


@pytest.mark.parametrize('case', [(2, 128, 128, 32, 32, 32, 1, 1, 1), (2, 256, 256, 32, 17, 19, 2, 1, 1), (2, 512, 512, 32, 16, 16, 1, 1, 1),
                                  (1, 1024, 1024, 32, 8, 8, 2, 1, 1), (2, 64, 32, 4, 9, 11, 1, 2, 2)], ids=str)
def test_grouped3x3_vs_float64(case):
    from mit_semseg import ops
    n, c, k, groups, h, w, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, c, h, w, generator=g)
    wt = torch.randn(k, c // groups, 3, 3, generator=g) / (3.0 * (c // groups) ** 0.5)
    xr, wr = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, stride, pad, dil, groups)
    gy = torch.randn(yr.shape, generator=g)
    yr.backward(gy.double())
    dev = torch.device('cuda:0')
    xg = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wg = wt.to(dev).requires_grad_(True)
    y = ops.grouped_conv3x3(xg, wg, groups, stride, pad, dil)
    y.backward(gy.to(dev).contiguous(memory_format=torch.channels_last))
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    assert rel(y.detach(), yr.detach()) < 3e-6
    assert rel(xg.grad, xr.grad) < 3e-6
    assert rel(wg.grad, wr.grad) < 2e-5
