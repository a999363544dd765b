"""Evaluation metrics of the reference (mit_semseg/utils.py:128-156, used by eval.py:74-104 / eval_multipro.py:64-75) on
the device: the argmax over classes and the accuracy / intersection / union tallies run in HIP kernels
(csrc/head.hip: semseg_argmax_metrics, semseg_label_metrics), so an evaluation loop never copies a full-resolution score
or label map to the host -- only 2 + 3*numClass integers per image (or per dataset: the tallies accumulate).

Same names and return conventions as the reference for label-map inputs (`accuracy`, `intersectionAndUnion`,
`AverageMeter`); inputs are CUDA tensors, or numpy arrays / host tensors as eval.py:74-84 passes them (uploaded to the current
HIP device: the tallies still run in the kernels -- there is no CPU fallback).  `segmentation_metrics` is the fused form for a
score map.  The host-side helpers of the reference's drivers (`parse_devices`, `setup_logger`, `find_recursive`,
`colorEncode`; utils.py:10-31,111-125,159-200) are plain Python / numpy and keep their semantics.
"""
import fnmatch
import logging
import os
import re
import sys

import numpy as np
import torch

from . import _native
from . import ops

__all__ = ['AverageMeter', 'accuracy', 'intersectionAndUnion', 'segmentation_metrics', 'MetricTally', 'parse_devices',
           'setup_logger', 'find_recursive', 'colorEncode', 'unique', 'NotSupportedCliException']


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
    """label / prediction map -> flat int64 tensor on the HIP device (numpy arrays and host tensors are uploaded)"""
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(np.ascontiguousarray(t))
    if not torch.is_tensor(t):
        raise TypeError('%s: expected a tensor or a numpy array, got %s' % (what, type(t).__name__))
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError('mit_semseg (MI355X build): the evaluation tallies run on a HIP device; there is no CPU fallback')
        t = t.to(torch.device('cuda', torch.cuda.current_device()))
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


# ---------------------------------------------------------------------------------------------------------------------------
# host-side helpers of the reference's drivers
# ---------------------------------------------------------------------------------------------------------------------------
def setup_logger(distributed_rank=0, filename='log.txt'):
    """utils.py:10-22: a DEBUG logger named "Logger" printing to stdout on rank 0, silent on the other ranks"""
    logger = logging.getLogger('Logger')
    logger.setLevel(logging.DEBUG)
    if distributed_rank > 0 or logger.handlers:
        return logger
    handler = logging.StreamHandler(stream=sys.stdout)
    handler.setLevel(logging.DEBUG)
    handler.setFormatter(logging.Formatter('[%(asctime)s %(levelname)s %(filename)s line %(lineno)d %(process)d] %(message)s'))
    logger.addHandler(handler)
    return logger


def find_recursive(root_dir, ext='.jpg'):
    """utils.py:25-30: every file below `root_dir` whose name ends in `ext`"""
    return [os.path.join(root, name) for root, _, names in os.walk(root_dir) for name in fnmatch.filter(names, '*' + ext)]


def unique(ar, return_index=False, return_inverse=False, return_counts=False):
    """utils.py:68-108 (a vendored numpy.unique of 2016): numpy's own does the same"""
    return np.unique(np.asanyarray(ar).flatten(), return_index=return_index, return_inverse=return_inverse,
                     return_counts=return_counts)


def colorEncode(labelmap, colors, mode='RGB'):
    """utils.py:111-125: [H, W] class indices -> [H, W, 3] uint8 colours (`colors[label]`); negative labels stay black"""
    labelmap = np.asarray(labelmap).astype('int')
    colors = np.asarray(colors)
    rgb = np.zeros(labelmap.shape + (3,), dtype=np.uint8)
    valid = labelmap >= 0
    rgb[valid] = colors[labelmap[valid]].astype(np.uint8)
    return rgb[:, :, ::-1] if mode == 'BGR' else rgb


class NotSupportedCliException(Exception):
    pass


_DEVICE_PATTERNS = (re.compile(r'^(?:gpu)?(\d+)$'), re.compile(r'^(?:gpu)?(\d+)-(?:gpu)?(\d+)$'))


def parse_devices(input_devices):
    """utils.py:180-200: "0-3" / "0,1,2,3" / "gpu0-gpu2,5" -> ['gpu0', 'gpu1', ...] (ranges are inclusive and may be given
    high-to-low; duplicates are dropped, order of first mention is kept)"""
    out = []
    for item in input_devices.split(','):
        item = item.lower().strip()
        single, span = _DEVICE_PATTERNS[0].match(item), _DEVICE_PATTERNS[1].match(item)
        if single:
            ids = [int(single.group(1))]
        elif span:
            a, b = sorted((int(span.group(1)), int(span.group(2))))
            ids = range(a, b + 1)
        else:
            raise NotSupportedCliException('Can not recognize device: "{}"'.format(item))
        for i in ids:
            if 'gpu%d' % i not in out:
                out.append('gpu%d' % i)
    return out
