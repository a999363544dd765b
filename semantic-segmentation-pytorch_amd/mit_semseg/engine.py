"""Training-step engine: train.py:34-48 semantics (zero_grad -> forward -> backward -> 2x SGD with the
poly LR of train.py:130-139) executed as HIP kernels, optionally captured ONCE into a hipGraph and
replayed (the whole step is capture-safe: no host sync, no allocation outside torch's graph pool, the
learning rate lives in device memory)."""
import torch
import torch.nn as nn

from . import ops
from .models.layers import Conv2d
from .parallel import GradientBuckets, world_size


def group_weight(module):
    """train.py:92-112: weight decay on conv weights only; biases and BN affine get none.
    Returns (decay_params, no_decay_params)."""
    decay, no_decay = [], []
    for m in module.modules():
        if isinstance(m, (Conv2d, nn.modules.conv._ConvNd, nn.Linear)):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif hasattr(m, 'running_mean') and hasattr(m, 'weight'):
            if m.weight is not None:
                no_decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
    assert len(list(module.parameters())) == len(decay) + len(no_decay)
    return decay, no_decay


class FusedSGD:
    """torch.optim.SGD(momentum=0.9, weight_decay on group 0) for both encoder and decoder groups in one
    multi-tensor HIP kernel (train.py:115-127).  lr per group is a device scalar."""

    def __init__(self, groups, momentum=0.9):
        """groups: list of dict(params=[...], lr=float, weight_decay=float)"""
        self.groups = groups
        self.momentum = momentum
        self.state = {}
        self.steps = 0
        for g in groups:
            dev = g['params'][0].device
            g['lr_t'] = torch.tensor([g['lr']], device=dev, dtype=torch.float32)

    def set_lr(self, group_index, lr):
        g = self.groups[group_index]
        g['lr'] = lr
        g['lr_t'].fill_(lr)

    def zero_grad(self):
        for g in self.groups:
            for p in g['params']:
                p.grad = None

    def step(self, grad_scale=1.0):
        first = self.steps == 0
        for g in self.groups:
            ps = [p for p in g['params'] if p.grad is not None]
            if not ps:
                continue
            bufs = []
            for p in ps:
                b = self.state.get(p)
                if b is None:
                    b = torch.empty_like(p.grad)
                    self.state[p] = b
                bufs.append(b)
            ops.sgd_step(ps, [p.grad for p in ps], bufs, first, [g['weight_decay']] * len(ps), g['lr_t'],
                         self.momentum, grad_scale)
        self.steps += 1


class TrainStep:
    """One training iteration of train.py:34-48 for a SegmentationModule.

    step(feed) -> (loss, acc) device scalars.  With `graph=True` the first `warmup_eager` calls run eagerly
    (they size the workspace and the allocator pools), then fwd+bwd+all-reduce+SGD is captured into a
    hipGraph; later calls copy the batch into static buffers and replay."""

    def __init__(self, segmentation_module, lr_encoder=0.02, lr_decoder=0.02, momentum=0.9, weight_decay=1e-4,
                 lr_pow=0.9, max_iters=100000, graph=False, group=None, bucket_bytes=64 << 20):
        self.sm = segmentation_module
        enc, dec = segmentation_module.encoder, segmentation_module.decoder
        groups = []
        for net, lr in ((enc, lr_encoder), (dec, lr_decoder)):
            decay, no_decay = group_weight(net)
            groups.append(dict(params=decay, lr=lr, weight_decay=weight_decay, base_lr=lr))
            groups.append(dict(params=no_decay, lr=lr, weight_decay=0.0, base_lr=lr))
        self.opt = FusedSGD(groups, momentum)
        self.lr_pow, self.max_iters = lr_pow, max_iters
        self.iter = 0
        self.group = group
        self.world = world_size(group)
        self.buckets = None
        if self.world > 1:
            dev = next(segmentation_module.parameters()).device
            side = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
            self.buckets = GradientBuckets(list(enc.parameters()) + list(dec.parameters()), bucket_bytes, group, side)
        # h2 path: the split planes of every conv weight are rebuilt by ONE multi-tensor launch after each optimiser
        # step (csrc/weights_prep.hip) instead of 5 small launches per conv inside forward/backward
        self._conv_weights = [m.weight for m in segmentation_module.modules() if isinstance(m, Conv2d)]
        self._weights_ready = False
        self.use_graph = graph
        self._graph = None
        self._static = None
        self._out = None
        self.warmup_eager = 2

    def adjust_learning_rate(self):
        """train.py:130-139 poly schedule"""
        scale = (1.0 - float(self.iter) / self.max_iters) ** self.lr_pow
        for i, g in enumerate(self.opt.groups):
            self.opt.set_lr(i, g['base_lr'] * scale)

    def _prepare_weights(self):
        if ops.CONV_MODE == 'h2' and ops.FUSE:
            ops.prepare_conv_weights(self._conv_weights)
            self._weights_ready = True

    def _eager(self, feed):
        if not self._weights_ready:
            self._prepare_weights()
        self.opt.zero_grad()
        loss, acc = self.sm(feed)
        if self.buckets is not None:
            self.buckets.prepare()            # hooks launch each bucket's all-reduce as backward completes it
        loss.backward()
        scale = 1.0
        if self.buckets is not None:
            self.buckets.finish()
            scale = 1.0 / self.world          # loss.mean() over replicas (train.py:42)
        self.opt.step(grad_scale=scale)
        self._prepare_weights()               # planes of the UPDATED weights, for the next step
        return loss.detach(), acc.detach()

    def step(self, feed):
        self.adjust_learning_rate()
        self.iter += 1
        if not self.use_graph or self.world > 1:
            return self._eager(feed)
        if self._graph is None:
            if self.opt.steps < self.warmup_eager:
                return self._eager(feed)
            self._static = {k: v.clone() for k, v in feed.items() if torch.is_tensor(v)}
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._out = self._eager(self._static)
            # capture records but does not execute: fall through to the replay of this very step
        for k, v in self._static.items():
            v.copy_(feed[k])
        self._graph.replay()
        return self._out
