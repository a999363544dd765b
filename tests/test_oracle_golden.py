"""Pin oracle/semseg_oracle.py against outputs of the UNMODIFIED reference
(tests/golden/*.pt, produced by tests/golden/make_golden.py).  CPU only."""
import os

import pytest
import torch

from tests.util import golden_cases, load_golden, oracle_run, check_summary, HEAVY_GOLDEN

# same torch CPU kernels on both sides -> agreement is at rounding level
ATOL, RTOL = 1e-5, 1e-5


@pytest.mark.parametrize('name', golden_cases())
def test_oracle_matches_reference(name):
    g = load_golden(name)
    heavy = name in HEAVY_GOLDEN
    res, enc, dec, grads = oracle_run(g, with_step=False if heavy else None)
    if 'prob' in g:
        torch.testing.assert_close(res['prob'], g['prob'], atol=ATOL, rtol=RTOL)
        assert torch.equal(res['prob'].argmax(1), g['prob'].argmax(1))
        return
    torch.testing.assert_close(res['pred'], g['pred'], atol=ATOL * 10, rtol=RTOL)
    assert torch.equal(res['pred'].argmax(1), g['pred'].argmax(1))
    if g['pred_deepsup'] is not None:
        torch.testing.assert_close(res['pred_deepsup'], g['pred_deepsup'], atol=ATOL * 10, rtol=RTOL)
    torch.testing.assert_close(res['loss'], g['loss'], atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(res['acc'], g['acc'], atol=0, rtol=0)
    for f, w in zip(res['feats'], g['feats']):
        check_summary(f, w, 1e-4, 1e-4, 'feat')
    if not g['meta']['step'] or heavy:
        return
    for sd_grads, want in ((grads[0], g['grads_enc']), (grads[1], g['grads_dec'])):
        assert set(sd_grads) == set(want)
        for k in want:
            check_summary(sd_grads[k], want[k], 1e-4, 1e-3, 'grad ' + k)
    for sd, want in ((enc, g['after_enc']), (dec, g['after_dec'])):
        for k in want:
            if k.rsplit('.', 1)[-1] in ('_tmp_running_mean', '_tmp_running_var', '_running_iter'):
                continue   # fork-only DP buffers, untouched on the single-device path (batchnorm.py:50-52)
            check_summary(sd[k], want[k], 1e-4, 1e-3, 'after-step ' + k)
    if 'anchor_grads_enc' in g:
        # the well-conditioned acceptance of the GPU tests (tests/util.check_directions: cosine / dot product per gradient tensor
        # against the float64 anchor) applied to the reference's own fp32 arithmetic: pins the yardstick and its calibration on the CPU
        from tests.util import check_directions
        items = [(side + '.' + k, gr[k], g['anchor_grads_' + side][k]) for side, gr in (('enc', grads[0]), ('dec', grads[1])) for k in gr]
        print(check_directions(items, name + ' oracle gradients'))


# forward of every full-size fixture; the backward too (every gradient tensor, the state after the step) where the CPU suite can
# afford it: configs[1], the metric's config (SEMSEG_FULLSIZE_BACKWARD=all runs it for every case: ~10 min on 8 cores)
_ALL_BWD = os.environ.get('SEMSEG_FULLSIZE_BACKWARD', '') == 'all'


@pytest.mark.parametrize('case,backward', [('cfg1_r50d_ppmds_512', True), ('cfg2_r50_upernet_512', _ALL_BWD),
                                           ('cfg3_r101d_ppmds_376x504', _ALL_BWD), ('cfg3_r101d_ppmds_456x680', _ALL_BWD),
                                           ('cfg4_hrnetv2_c1_512', _ALL_BWD)])
def test_oracle_matches_reference_at_full_size(case, backward):
    """BASELINE configs[1..4] at full size (2 x 512 x 512; two variable-size shapes for configs[3]): the oracle's training step
    against the stored step of the unmodified reference (tests/golden/make_fullsize_golden.py) -- forward: log-probabilities of the
    pixel sample, the arg-max of every pixel, loss, accuracy; backward: EVERY gradient tensor and the state after the SGD step
    against the float64 anchors in units of the reference's own fp32 band (the acceptance the GPU test applies to the HIP path,
    here applied to the oracle, so the yardstick itself is pinned on the CPU)"""
    import json
    from oracle import semseg_oracle as O
    from tests.util import (load_fullsize_golden, anchor_ratios, check_anchor_ratios, post_step_bands, post_step_record, is_head_tensor,
                            scale_error, HEAD_SCALE_ERR)
    fx = load_fullsize_golden(case)
    assert fx is not None
    m = fx['meta']
    man = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'manifests.json')))
    enc_sd = O.synth_state_dict(man[m['arch_encoder']], m['seed_weights'][0])
    dec_sd = O.synth_state_dict(man['%s@%d' % (m['arch_decoder'], m['fc_dim'])], m['seed_weights'][1])
    drop = {'main': O.synth_dropout_mask(2, 512, seed=m['seed_dropout'][0])} if 'ppm' in m['arch_decoder'] else {}
    if m['arch_decoder'] == 'ppm_deepsup':
        drop['deepsup'] = O.synth_dropout_mask(2, m['fc_dim'] // 4, seed=m['seed_dropout'][1])
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=m['seed_batch'])
    e, d = O.clone_sd(enc_sd, backward), O.clone_sd(dec_sd, backward)
    with torch.set_grad_enabled(backward):
        res = O.segmentation_forward(e, d, m['arch_encoder'], m['arch_decoder'], img, lab,
                                     training=True, dropout=drop, deep_sup_scale=m['deep_sup_scale'])
    pred = res['pred'].detach()
    n, c, h, w = pred.shape
    assert [n, c, h, w] == list(m['pred_shape'])
    rows = pred.permute(0, 2, 3, 1).reshape(n * h * w, c)
    torch.testing.assert_close(rows[fx['pixels']], fx['logp'], atol=ATOL * 10, rtol=RTOL)
    assert torch.equal(rows.argmax(1).reshape(n, h, w), fx['argmax'].long())
    torch.testing.assert_close(res['loss'].detach(), fx['loss'], atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(res['acc'], fx['acc'], atol=0, rtol=0)
    assert abs(fx['loss64'] - fx['loss'].item()) < 1e-4
    for side in ('enc', 'dec'):
        assert all(k.rsplit('.', 1)[-1] in ('running_mean', 'running_var') for k in fx['anchor_after_' + side])
    if not backward:
        return
    res['loss'].backward()
    lr = m['lr']
    items, heads = [], []
    for sd, side in ((e, 'enc'), (d, 'dec')):
        want = fx['anchor_grads_' + side]
        params = {k: v for k, v in sd.items() if v.requires_grad}
        assert sorted(params) == sorted(want)
        for k, v in params.items():
            items.append((side + '.' + k, v.grad, want[k]))
            if side == 'dec' and is_head_tensor(k, v):
                heads.append((scale_error(v.grad, want[k]), k))
    print(check_anchor_ratios(anchor_ratios(items), case + ' oracle gradients'))
    from tests.util import check_directions
    print(check_directions(items, case + ' oracle gradients'))
    assert heads and max(heads)[0] <= HEAD_SCALE_ERR, heads
    items = []
    for sd, sd0, side in ((e, enc_sd, 'enc'), (d, dec_sd, 'dec')):
        params = {k: v for k, v in sd.items() if v.requires_grad}
        O.sgd_step(params, {k: v.grad for k, v in params.items()}, {}, lr)
        for k, v in params.items():
            wd = 1e-4 if v.dim() == 4 else 0.0          # train.py:92-112: conv weights only
            items.append((side + '.' + k, v, post_step_record(fx['anchor_grads_' + side][k], sd0[k], lr, wd)))
        for k, rec in fx['anchor_after_' + side].items():
            items.append((side + '.' + k, sd[k], rec))
    print(check_anchor_ratios(anchor_ratios(items, post_step_bands(fx, lr)), case + ' oracle after-step state'))


# ---------------------------------------------------------------------------------------------------------
# evaluation metrics: the numpy oracle against outputs of the unmodified reference functions
# ---------------------------------------------------------------------------------------------------------
def _metrics_cases():
    import importlib.util
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    spec = importlib.util.spec_from_file_location('make_metrics_golden', os.path.join(here, 'make_metrics_golden.py'))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = np.load(os.path.join(here, 'metrics_golden.npz'))
    for name in sorted(k[:-4] for k in gold.files if k.endswith('_cfg')):
        C, H, W, seed = (int(v) for v in gold[name + '_cfg'])
        scores, label = gen.synth_case(name, C, H, W, seed)
        if name == 'perfect':
            label = gold[name + '_pred'].astype('int64')
        yield name, C, scores, label, {k: gold[name + '_' + k] for k in ('pred', 'acc', 'pix', 'inter', 'union')}


def test_metrics_oracle_matches_reference_golden():
    import numpy as np
    from oracle import metrics_oracle as M
    n = 0
    for name, C, scores, label, want in _metrics_cases():
        pred = M.argmax_first(scores)
        assert np.array_equal(pred, want['pred'].astype(np.int64)), name
        # torch.max(dim) (eval.py:74) picks the same first maximum on the tie-laden inputs
        assert np.array_equal(torch.max(torch.from_numpy(scores)[None], dim=1)[1][0].numpy(), pred), name
        acc, pix = M.accuracy(pred, label)
        assert pix == int(want['pix']) and acc == float(want['acc']), name
        inter, union = M.intersection_and_union(pred, label, C)
        assert np.array_equal(inter, want['inter']) and np.array_equal(union, want['union']), name
        n += 1
    assert n == 6
