"""Evaluation metrics of the reference (mit_semseg/utils.py:128-156, used by eval.py:74-104 / eval_multipro.py:64-75) on
the device: the argmax over classes and the accuracy / intersection / union tallies run in HIP kernels
(csrc/head.hip: semseg_argmax_metrics, semseg_label_metrics), so an evaluation loop never copies a full-resolution score
or label map to the host -- only 2 + 3*numClass integers per image (or per dataset: the tallies accumulate).

Same names and return conventions as the reference for label-map inputs (`accuracy`, `intersectionAndUnion`,
`AverageMeter`); inputs are CUDA tensors (there is no CPU fallback: the reference's numpy code is the CPU path).
`segmentation_metrics` is the fused form for a score map.
"""
import numpy as np
import torch

from . import _native
from . import ops

__all__ = ['AverageMeter', 'accuracy', 'intersectionAndUnion', 'segmentation_metrics', 'MetricTally']


class AverageMeter(object):
    """utils.py:64-97: running weighted average (val/avg/sum/count; `initialized` on first update)."""

    def __init__(self):
        self.initialized = False
        self.val = self.avg = self.sum = self.count = None

    def initialize(self, val, weight):
        self.val, self.avg, self.sum, self.count, self.initialized = val, val, val * weight, weight, True

    def update(self, val, weight=1):
        if not self.initialized:
            self.initialize(val, weight)
        else:
            self.add(val, weight)

    def add(self, val, weight):
        self.val = val
        self.sum += val * weight
        self.count += weight
        self.avg = self.sum / self.count

    def value(self):
        return self.val

    def average(self):
        return self.avg


def _as_i64_flat(t, what):
    if not torch.is_tensor(t):
        raise TypeError('%s: expected a CUDA tensor (the numpy path is the reference; this build has no CPU fallback)' % what)
    ops._require_cuda(t)
    return t.reshape(-1).to(torch.int64).contiguous()


def _label_counts(pred, label, num_class):
    p, l = _as_i64_flat(pred, 'pred'), _as_i64_flat(label, 'label')
    if p.numel() != l.numel():
        raise ValueError('prediction has %d pixels, label has %d' % (p.numel(), l.numel()))
    counts = torch.zeros(2 + 3 * num_class, dtype=torch.int64, device=p.device)
    _native.check(_native.lib().semseg_label_metrics(ops._p(p), ops._p(l), p.numel(), int(num_class), ops._p(counts),
                                                     ops._st()), 'label_metrics')
    return counts


def accuracy(preds, label):
    """utils.py:128-133: (acc, valid_sum) with acc = #(valid & preds == label) / (valid_sum + 1e-10)."""
    c = _label_counts(preds, label, 1).cpu()
    acc_sum, valid_sum = int(c[0]), int(c[1])
    return float(acc_sum) / (valid_sum + 1e-10), valid_sum


def intersectionAndUnion(imPred, imLab, numClass):
    """utils.py:136-156: (area_intersection, area_union) as int64 numpy arrays of length numClass."""
    c = _label_counts(imPred, imLab, numClass).cpu().numpy()
    inter, pred, lab = c[2:2 + numClass], c[2 + numClass:2 + 2 * numClass], c[2 + 2 * numClass:]
    return inter.copy(), (pred + lab - inter)


class MetricTally:
    """Device-resident accumulator over images: acc_sum, valid_sum, area_intersection/pred/lab (int64 [2 + 3C])."""

    def __init__(self, num_class, device):
        self.num_class = int(num_class)
        self.counts = torch.zeros(2 + 3 * self.num_class, dtype=torch.int64, device=device)

    def summary(self):
        """(pixel accuracy, per-class IoU, mean IoU) as eval.py:98-105 computes them from its meters."""
        c = self.counts.cpu().numpy().astype(np.float64)
        n = self.num_class
        inter, union = c[2:2 + n], c[2 + n:2 + 2 * n] + c[2 + 2 * n:] - c[2:2 + n]
        iou = inter / (union + 1e-10)
        return c[0] / (c[1] + 1e-10), iou, float(iou.mean())


def segmentation_metrics(scores, label=None, tally=None):
    """eval.py:74-84 fused: pred = argmax over classes of `scores` ([1|N, C, H, W] CUDA tensor, first maximum like
    torch.max), and -- with `label` ([N, H, W] or [H, W], < 0 = unlabeled) -- the tallies of accuracy() and
    intersectionAndUnion() added to `tally` (a MetricTally; created if None).  Returns (pred int64 [N, H, W], tally)."""
    s, ld = ops.as_nhwc(scores.detach())
    n, c, h, w = s.shape
    P = n * h * w
    pred = torch.empty((n, h, w), dtype=torch.int64, device=s.device)
    lab = None
    if label is not None:
        lab = _as_i64_flat(label, 'label')
        if lab.numel() != P:
            raise ValueError('scores cover %d pixels, label has %d' % (P, lab.numel()))
        if tally is None:
            tally = MetricTally(c, s.device)
        if tally.num_class != c:
            raise ValueError('tally has %d classes, scores have %d' % (tally.num_class, c))
    _native.check(_native.lib().semseg_argmax_metrics(ops._p(s), ld, ops._p(lab), P, c, ops._p(pred),
                                                      ops._p(tally.counts if lab is not None else None), ops._st()),
                  'argmax_metrics')
    return pred, tally
