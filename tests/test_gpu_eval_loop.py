"""The multi-scale evaluation loop (engine.evaluate, eval.py:40-105) on the device against the oracle."""
import pytest
import torch

from tests.util import load_golden
from tests.test_gpu_models import build_native, argmax_check
from oracle import semseg_oracle as O

pytestmark = pytest.mark.gpu


def test_evaluate_multiscale_loop_vs_oracle():
    """engine.evaluate (eval.py:40-105): multi-scale average of the softmax scores at the label size, argmax, tallies over
    two items -- scores against the oracle's, tallies exact for the predictions made, predictions equal to the oracle's
    outside its own near-ties"""
    from mit_semseg.engine import evaluate
    from oracle import metrics_oracle as M
    import numpy as np
    g = load_golden('r18d_ppm_infer_64x80')
    m = g['meta']
    dev = torch.device('cuda:0')
    sm, enc_sd, dec_sd = build_native(g, dev, use_softmax=True)
    items = []
    gen = torch.Generator().manual_seed(11)
    for k, (lh, lw) in enumerate([(70, 90), (96, 72)]):
        imgs = [torch.randn(1, 3, 64 + 16 * s, 80 + 16 * s, generator=gen) for s in range(2)]
        lab = torch.randint(-1, 150, (1, lh, lw), generator=gen)
        items.append({'img_data': imgs, 'seg_label': lab, 'info': 'item%d' % k})
    preds = {}
    acc, iou, miou, tally = evaluate(sm, [[it] for it in items], 150, on_item=lambda it, p: preds.__setitem__(it['info'], p.cpu()))
    torch.cuda.synchronize()
    counts = np.zeros(2 + 3 * 150, dtype=np.int64)
    e, d = O.clone_sd(enc_sd), O.clone_sd(dec_sd)
    for it in items:
        lab = it['seg_label'][0]
        ref = torch.zeros(1, 150, lab.shape[0], lab.shape[1])
        for img in it['img_data']:
            with torch.no_grad():
                ref = ref + O.segmentation_forward(e, d, m['arch_encoder'], m['arch_decoder'], img, None,
                                                   seg_size=tuple(lab.shape)) / len(it['img_data'])
        pred = preds[it['info']]
        assert tuple(pred.shape) == (1,) + tuple(lab.shape)
        argmax_check(torch.log(torch.zeros_like(ref).scatter_(1, pred[:, None], 1.0) + 1e-30), ref.log(), it['info'])
        p, l = pred[0].numpy(), lab.numpy()
        valid = l >= 0
        counts[0] += int(((p == l) & valid).sum())
        counts[1] += int(valid.sum())
        i, u = M.intersection_and_union(p, l, 150)
        counts[2:152] += i
        counts[152:302] += np.bincount(p[valid], minlength=150)
        counts[302:452] += np.bincount(l[valid], minlength=150)
    assert np.array_equal(tally.counts.cpu().numpy(), counts)
    assert abs(acc - counts[0] / (counts[1] + 1e-10)) < 1e-12
    inter, union = counts[2:152], counts[152:302] + counts[302:452] - counts[2:152]
    assert abs(miou - (inter / (union + 1e-10)).mean()) < 1e-12
