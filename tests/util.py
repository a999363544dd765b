"""Shared helpers for the parity tests (oracle side + fixture comparison)."""
import glob
import os

import torch

from oracle import semseg_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

# golden cases whose ~150 convolution geometries (dense forms of grouped convs, 101-layer backbones) no other test uses: they
# run on the library's heuristic launch plans instead of timing every tile x split candidate
HEURISTIC_PLAN_GOLDEN = ('mnv2d_c1ds_64_train', 'resnext101_upernet_128_eval')


def golden_cases():
    return sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN, '*.pt')))


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name + '.pt'), weights_only=False)


def check_summary(got, want, atol, rtol, what):
    """Compare a tensor with a make_golden.summarize() record."""
    got = got.detach().float().cpu()
    if 'full' in want:
        torch.testing.assert_close(got.reshape(want['full'].shape), want['full'], atol=atol, rtol=rtol, msg=lambda m: what + ': ' + m)
        return
    f = got.flatten()
    assert f.numel() == want['numel'], what
    torch.testing.assert_close(f[:16], want['head'], atol=atol, rtol=rtol, msg=lambda m: what + ' head: ' + m)
    torch.testing.assert_close(f[-16:], want['tail'], atol=atol, rtol=rtol, msg=lambda m: what + ' tail: ' + m)
    s, a = f.double().sum().item(), f.double().abs().sum().item()
    tol = atol * f.numel() ** 0.5 + rtol * want['abssum']
    assert abs(s - want['sum']) <= tol, (what, s, want['sum'], tol)
    assert abs(a - want['abssum']) <= tol, (what, a, want['abssum'], tol)


def oracle_run(g, with_step=None):
    """Run the oracle on the golden case's recipe.  Returns (result dict, enc_sd, dec_sd, grads)."""
    m = g['meta']
    step = m['step'] if with_step is None else with_step
    enc = O.clone_sd(O.synth_state_dict(g['manifest_enc'], m['seed']), requires_grad=step)
    dec = O.clone_sd(O.synth_state_dict(g['manifest_dec'], m['seed'] + 1), requires_grad=step)
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    if m['seg_size'] is not None:
        with torch.no_grad():
            prob = O.segmentation_forward(enc, dec, m['arch_encoder'], m['arch_decoder'], img, lab,
                                          seg_size=m['seg_size'])
        return {'prob': prob}, enc, dec, None
    with torch.set_grad_enabled(step):
        res = O.segmentation_forward(enc, dec, m['arch_encoder'], m['arch_decoder'], img, lab,
                                     training=m['training'], dropout=g['dropout'],
                                     deep_sup_scale=m['deep_sup_scale'])
    grads = None
    if step:
        res['loss'].backward()
        grads = ({k: v.grad for k, v in enc.items() if v.requires_grad},
                 {k: v.grad for k, v in dec.items() if v.requires_grad})
        for sd, gr in ((enc, grads[0]), (dec, grads[1])):
            params = {k: v for k, v in sd.items() if v.requires_grad}
            O.sgd_step(params, gr, {}, m['lr'])
    return res, enc, dec, grads
