"""Pin oracle/semseg_oracle.py against outputs of the UNMODIFIED reference
(tests/golden/*.pt, produced by tests/golden/make_golden.py).  CPU only."""
import pytest
import torch

from tests.util import golden_cases, load_golden, oracle_run, check_summary

# same torch CPU kernels on both sides -> agreement is at rounding level
ATOL, RTOL = 1e-5, 1e-5


@pytest.mark.parametrize('name', golden_cases())
def test_oracle_matches_reference(name):
    g = load_golden(name)
    res, enc, dec, grads = oracle_run(g)
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
    if not g['meta']['step']:
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
