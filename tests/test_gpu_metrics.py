"""Evaluation metrics on the device: argmax + accuracy / intersection / union tallies of
eval.py:74-84 / utils.py:128-156, bit-exact against outputs of the unmodified reference functions
(tests/golden/metrics_golden.npz) -- integer work, no tolerance."""
import numpy as np
import pytest
import torch

from tests.test_oracle_golden import _metrics_cases

pytestmark = pytest.mark.gpu


def test_segmentation_metrics_match_reference_golden():
    from mit_semseg import utils as U
    dev = torch.device('cuda:0')
    n = 0
    for name, C, scores, label, want in _metrics_cases():
        s = torch.from_numpy(scores)[None].to(dev).contiguous(memory_format=torch.channels_last)
        lab = torch.from_numpy(label).to(dev)
        pred, tally = U.segmentation_metrics(s, lab)
        torch.cuda.synchronize()
        assert np.array_equal(pred[0].cpu().numpy(), want['pred'].astype(np.int64)), name
        c = tally.counts.cpu().numpy()
        inter, union = c[2:2 + C], c[2 + C:2 + 2 * C] + c[2 + 2 * C:] - c[2:2 + C]
        assert int(c[1]) == int(want['pix']) and float(c[0]) / (int(c[1]) + 1e-10) == float(want['acc']), name
        assert np.array_equal(inter, want['inter']) and np.array_equal(union, want['union']), name
        # the reference-named functions on label maps
        acc, pix = U.accuracy(pred[0], lab)
        assert pix == int(want['pix']) and acc == float(want['acc']), name
        i2, u2 = U.intersectionAndUnion(pred[0], lab, C)
        assert np.array_equal(i2, want['inter']) and np.array_equal(u2, want['union']), name
        n += 1
    assert n == 6


def test_metric_tally_accumulates_and_reads_channel_slices():
    """two images into one tally == the sum of the per-image tallies; scores given as a channel slice of a wider NHWC
    buffer (ld > C) and NCHW-contiguous scores give the same answer"""
    from mit_semseg import utils as U
    dev = torch.device('cuda:0')
    cases = [c for c in _metrics_cases() if c[0] in ('a', 'b')]
    tally = None
    total = None
    for name, C, scores, label, want in cases:
        wide = torch.randn(1, 37 + C, scores.shape[1], scores.shape[2], device=dev).contiguous(memory_format=torch.channels_last)
        wide[:, 37:] = torch.from_numpy(scores)[None].to(dev)
        lab = torch.from_numpy(label).to(dev)
        pred, tally = U.segmentation_metrics(wide[:, 37:], lab, tally)
        pred2, t2 = U.segmentation_metrics(torch.from_numpy(scores)[None].to(dev), lab)          # NCHW-contiguous input
        assert torch.equal(pred, pred2)
        total = t2.counts.clone() if total is None else total + t2.counts
    assert torch.equal(tally.counts, total)
    acc, iou, miou = tally.summary()
    inter = sum(c[4]['inter'] for c in cases).astype(np.float64)
    union = sum(c[4]['union'] for c in cases).astype(np.float64)
    assert np.allclose(iou, inter / (union + 1e-10)) and abs(miou - (inter / (union + 1e-10)).mean()) < 1e-12
