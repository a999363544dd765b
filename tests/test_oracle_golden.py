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


@pytest.mark.parametrize('case', ['cfg1_r50d_ppmds_512', 'cfg2_r50_upernet_512'])
def test_oracle_matches_reference_at_full_size(case):
    """BASELINE configs[1] / configs[2] at 2 x 512 x 512: the oracle's training-mode forward against the stored forward of the
    unmodified reference (tests/golden/make_fullsize_golden.py) -- the oracle the 512 x 512 GPU parity test runs on the box is
    pinned at that size too"""
    import json
    from oracle import semseg_oracle as O
    from tests.util import load_fullsize_golden
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
    with torch.no_grad():
        res = O.segmentation_forward(O.clone_sd(enc_sd, False), O.clone_sd(dec_sd, False), m['arch_encoder'], m['arch_decoder'], img, lab,
                                     training=True, dropout=drop, deep_sup_scale=m['deep_sup_scale'])
    n, c, h, w = res['pred'].shape
    rows = res['pred'].permute(0, 2, 3, 1).reshape(n * h * w, c)
    torch.testing.assert_close(rows[fx['pixels']], fx['logp'], atol=ATOL * 10, rtol=RTOL)
    assert torch.equal(rows.argmax(1).reshape(n, h, w), fx['argmax'].long())
    torch.testing.assert_close(res['loss'], fx['loss'], atol=ATOL, rtol=RTOL)
    torch.testing.assert_close(res['acc'], fx['acc'], atol=0, rtol=0)


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
