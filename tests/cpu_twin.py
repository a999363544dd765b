"""TEST INFRASTRUCTURE: a torch-CPU twin of the `mit_semseg.ops` surface.  With it installed (`install(monkeypatch)`) the WHOLE
Python side of the product -- ModelBuilder, every model family, SegmentationModule, TrainStep / FusedSGD, parameter
grouping, dropout replay -- runs numerically on the CPU, so its wiring can be checked against the goldens of the unmodified
reference without a GPU (tests/test_host_numeric_cpu.py).  The twin replaces exactly the functions that would launch HIP
kernels with the torch operator the reference itself calls at that site; it says nothing about the kernels (that is the
-m gpu suite) and is never imported by the product."""
import torch
import torch.nn.functional as F


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    return F.conv2d(x, weight, bias, stride, padding, dilation)


def depthwise_conv3x3(x, weight, stride=1, padding=1, dilation=1):
    return F.conv2d(x, weight, None, stride, padding, dilation, x.shape[1])


def batch_norm_act(z, gamma, beta, running_mean, running_var, residual=None, training=False, momentum=0.1, eps=1e-5,
                   relu=False, num_batches_tracked=None):
    if training and num_batches_tracked is not None:
        num_batches_tracked += 1
    y = F.batch_norm(z, running_mean, running_var, gamma, beta, training, momentum, eps)
    if residual is not None:
        y = y + residual
    return F.relu(y) if relu else y


def conv_bn_act(x, weight, gamma, beta, running_mean, running_var, num_batches_tracked, residual=None, stride=1, padding=0,
                dilation=1, training=False, momentum=0.1, eps=1e-5, relu=False, passthrough=False, planes_only=False):
    y = batch_norm_act(F.conv2d(x, weight, None, stride, padding, dilation), gamma, beta, running_mean, running_var, residual,
                       training, momentum, eps, relu, num_batches_tracked)
    return (y, x) if passthrough else y


def grouped_conv3x3(x, weight, groups, stride=1, padding=1, dilation=1):
    return F.conv2d(x, weight, None, stride, padding, dilation, groups)


def clamp_max(x, cap):
    return torch.clamp(x, max=cap)


def fork(x, n=2):
    return (x,) * n


def share_planes(src, dst):
    return None


def dropout_mask(n, c, p, device):
    return (torch.rand(n, c, device=device) >= p).float() / (1.0 - p)


def add_act(a, b, relu=False):
    y = a + b
    return F.relu(y) if relu else y


def concat(xs):
    return torch.cat(list(xs), 1)


def scale_nc(x, mask):
    return x * mask[:, :, None, None]


def max_pool_3x3_s2(x):
    return F.max_pool2d(x, 3, 2, 1)


def adaptive_avg_pool(x, size):
    return F.adaptive_avg_pool2d(x, size)


def adaptive_avg_pool_multi(x, sizes):
    return [F.adaptive_avg_pool2d(x, s) for s in sizes]


def interpolate_bilinear(x, size, base=None, relu=False):
    y = F.interpolate(x, size=(int(size[0]), int(size[1])), mode='bilinear', align_corners=False)
    if base is not None:
        y = y + base
    return F.relu(y) if relu else y


def upsample_softmax(z, size):
    from mit_semseg import ops
    y = F.softmax(F.interpolate(z, size=(int(size[0]), int(size[1])), mode='bilinear', align_corners=False), dim=1)
    out, w, acc = ops._HEAD['out'], ops._HEAD['weight'], ops._HEAD['accumulate']
    ops._HEAD['used'] = True
    if out is None:
        return y * w if w != 1.0 else y
    if acc:
        out += y * w
    else:
        out.copy_(y * w)
    return out


def log_softmax(z):
    return F.log_softmax(z, dim=1)


def softmax(z):
    return F.softmax(z, dim=1)


def nll_loss_acc(logp, label, ignore_index=-1):
    loss = F.nll_loss(logp, label, ignore_index=ignore_index)
    preds = logp.max(dim=1)[1]
    valid = (label >= 0).long()
    acc = (valid * (preds == label).long()).sum().float() / (valid.sum().float() + 1e-10)
    return loss, acc.detach()


def sgd_step(params, grads, bufs, first_step, weight_decays, lr_tensor, momentum=0.9, grad_scale=1.0):
    """train.py:117-126 semantics of the fused kernel: g = grad * scale + wd * p; buf = g (first step) | m * buf + g; p -= lr buf"""
    lr = float(lr_tensor.reshape(-1)[0])
    if isinstance(first_step, (bool, int)):
        first_step = [first_step] * len(params)
    with torch.no_grad():
        for p, g, b, wd, first in zip(params, grads, bufs, weight_decays, first_step):
            d = g * grad_scale + wd * p
            if first:
                b.copy_(d)
            else:
                b.mul_(momentum).add_(d)
            p.sub_(lr * b)


def install(monkeypatch, keep=()):
    """`keep`: names of ops functions to leave in place (they then need a native library, e.g. a host-emulated one)"""
    from mit_semseg import ops
    g = globals()
    for name in ('conv2d', 'depthwise_conv3x3', 'grouped_conv3x3', 'clamp_max', 'fork', 'share_planes', 'dropout_mask', 'batch_norm_act', 'conv_bn_act', 'add_act', 'concat', 'scale_nc',
                 'max_pool_3x3_s2', 'adaptive_avg_pool', 'adaptive_avg_pool_multi', 'interpolate_bilinear', 'upsample_softmax', 'log_softmax',
                 'softmax', 'nll_loss_acc', 'sgd_step'):
        if name not in keep:
            monkeypatch.setattr(ops, name, g[name])
    monkeypatch.setattr(ops, 'prepare_conv_weights', lambda ws: 0)
    monkeypatch.setattr(ops, '_require_cuda', lambda *a: None)
