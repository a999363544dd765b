"""SynchronizedBatchNorm for one-process-per-GPU data parallelism on MI355X.

Mirrors the public surface of reference lib/nn/modules/batchnorm.py:38-139 (class names, ctor
arguments eps=1e-5 / momentum=0.001 / affine, parameter and buffer names incl. the fork's
`_tmp_running_mean/_tmp_running_var/_running_iter`) so reference checkpoints load unchanged.

Numerics follow the path the CPU reference executes, F.batch_norm (batchnorm.py:58-61): biased
variance + eps for normalisation, unbiased variance in the running-average EMA.  Cross-replica
synchronisation is NOT the reference's thread rendezvous (comm.py, batchnorm.py:63-117): each rank
computes [sum, sum^2, n] with a HIP kernel and the vector is all-reduced over RCCL
(ops.set_sync_bn_group), so every rank finalises identical statistics.
"""
import torch
import torch.nn as nn

from .... import ops

__all__ = ['SynchronizedBatchNorm1d', 'SynchronizedBatchNorm2d', 'SynchronizedBatchNorm3d']


class _SynchronizedBatchNorm(nn.Module):
    def __init__(self, num_features, eps=1e-5, momentum=0.001, affine=True):
        super().__init__()
        self.num_features = num_features
        self.eps = eps
        self.momentum = momentum
        self.affine = affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))
        else:
            # batchnorm.py:79-83 `affine=False`: y = (x - mean) * inv_std.  As in torch's _BatchNorm the module then has NO
            # weight / bias entries (state-dict parity); the kernels get a constant gamma = 1 / beta = 0 that needs no gradient
            self.register_parameter('weight', None)
            self.register_parameter('bias', None)
            self.register_buffer('_unit_gamma', torch.ones(num_features), persistent=False)
            self.register_buffer('_zero_beta', torch.zeros(num_features), persistent=False)
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        # fork-only bookkeeping buffers of the reference (batchnorm.py:50-52); kept for state-dict parity
        self.register_buffer('_tmp_running_mean', torch.zeros(num_features))
        self.register_buffer('_tmp_running_var', torch.ones(num_features))
        self.register_buffer('_running_iter', torch.ones(1))
        self._nbt_in_line = 0       # value of num_batches_tracked at which the three buffers above matched running_*

    def _as4d(self, x):
        raise NotImplementedError

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        """The fork's multi-GPU branch recomputes `running_* = _tmp_running_* / _running_iter` on every step
        (batchnorm.py:132-137) with `tmp <- tmp (1-m) + stat`, `iter <- iter (1-m) + 1`.  The kernels here keep the plain EMA
        `running <- (1-m) running + m stat` (the F.batch_norm path, batchnorm.py:58-61) and do not touch the bookkeeping
        buffers, so they are brought in line when a checkpoint is written: `_running_iter = 1/m` (the fixed point of its
        recursion, at which the fork's update IS the plain EMA) and `_tmp_running_* = running_* / m`.  A checkpoint trained here
        and resumed in the reference's multi-GPU training then continues from these running statistics instead of
        overwriting them from the init values.  Only done when training steps have run since the buffers were last in line
        (`num_batches_tracked` moved), so a loaded checkpoint is written back unchanged."""
        nbt = int(self.num_batches_tracked)
        if nbt != self._nbt_in_line:
            with torch.no_grad():
                it = 1.0 / self.momentum if self.momentum else 1.0
                self._running_iter.fill_(it)
                self._tmp_running_mean.copy_(self.running_mean * it)
                self._tmp_running_var.copy_(self.running_var * it)
            self._nbt_in_line = nbt
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._nbt_in_line = int(self.num_batches_tracked)

    def forward(self, input, residual=None, relu=False):
        """y = BN(input); the fused form y = relu(BN(input) + residual) is what the blocks call."""
        x, shape = self._as4d(input)
        gamma, beta = (self.weight, self.bias) if self.affine else (self._unit_gamma, self._zero_beta)
        c = x.shape[1]
        if c % 4:
            y = self._forward_odd_channels(x, gamma, beta, residual, relu)
        else:
            y = ops.batch_norm_act(x, gamma, beta, self.running_mean, self.running_var,
                                   residual=residual, training=self.training, momentum=self.momentum,
                                   eps=self.eps, relu=relu, num_batches_tracked=self.num_batches_tracked)
        return y if shape is None else y.reshape(shape)

    def _forward_odd_channels(self, x, gamma, beta, residual, relu):
        """channel counts that are not a multiple of 4 (no mit_semseg model has one; the reference's own unit tests use 10,
        tests/test_sync_batchnorm.py:66-107): the kernels move 4 channels per lane, so the tensors are padded with constant
        channels (torch glue, never on the hot path), normalised, and cut back; the running statistics of the real channels
        are copied back from the padded buffers."""
        c = x.shape[1]
        pad = 4 - c % 4
        fill = lambda t, v: torch.cat([t, torch.full((pad,), v, device=t.device, dtype=t.dtype)])      # noqa: E731
        widen = lambda t: torch.nn.functional.pad(t, (0, 0, 0, 0, 0, pad))                             # noqa: E731
        rm, rv = fill(self.running_mean, 0.0), fill(self.running_var, 1.0)
        y = ops.batch_norm_act(widen(x), fill(gamma, 1.0), fill(beta, 0.0), rm, rv,
                               residual=widen(residual) if residual is not None else None, training=self.training,
                               momentum=self.momentum, eps=self.eps, relu=relu, num_batches_tracked=self.num_batches_tracked)
        if self.training:
            with torch.no_grad():
                self.running_mean.copy_(rm[:c])
                self.running_var.copy_(rv[:c])
        return y[:, :c]

    def extra_repr(self):
        return '{num_features}, eps={eps}, momentum={momentum}, affine={affine}'.format(**self.__dict__)


class SynchronizedBatchNorm2d(_SynchronizedBatchNorm):
    def _as4d(self, x):
        if x.dim() != 4:
            raise ValueError('expected 4D input (got {}D input)'.format(x.dim()))
        return x, None


class SynchronizedBatchNorm1d(_SynchronizedBatchNorm):
    def _as4d(self, x):
        if x.dim() == 2:
            return x[:, :, None, None], x.shape
        if x.dim() == 3:
            return x[:, :, :, None], x.shape
        raise ValueError('expected 2D or 3D input (got {}D input)'.format(x.dim()))


class SynchronizedBatchNorm3d(_SynchronizedBatchNorm):
    def _as4d(self, x):
        if x.dim() != 5:
            raise ValueError('expected 5D input (got {}D input)'.format(x.dim()))
        n, c, d, h, w = x.shape
        return x.reshape(n, c, d * h, w), x.shape
