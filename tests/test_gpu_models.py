"""End-to-end parity of the HIP hot path (through ModelBuilder / SegmentationModule, i.e. through the C ABI)
 (a) against the committed golden outputs of the UNMODIFIED reference (tests/golden/*.pt), and
 (b) against the CPU oracle at BASELINE.json configs[1] full size (R50dilated+PPM_deepsup, 2x512x512).
Tolerances (north_star): log-probs within 1e-3 fp32, argmax label maps identical (pixels whose top-2
margin is below 1e-4 in the reference are reported separately)."""
import os
import tempfile

import pytest
import torch
import torch.nn as nn

from tests.util import (parity_line, check_directions, load_fullsize_golden, check_fullsize_golden, golden_cases, load_golden, anchor_ratios, check_anchor_ratios, is_head_tensor, scale_error, post_step_bands, post_step_record,
                        HEAD_SCALE_ERR, HEURISTIC_PLAN_GOLDEN, KNIFE_EDGE_GOLDEN, KNIFE_EDGE_SCALE_ERR, KNIFE_EDGE_MEDIAN_SCALE_ERR, KNIFE_EDGE_COSINE_MIN)
from oracle import semseg_oracle as O

pytestmark = pytest.mark.gpu
LOGP_ATOL = 1e-3


def build_native(g, dev, use_softmax=False):
    from mit_semseg.models import ModelBuilder, SegmentationModule
    m = g['meta']
    enc_sd, dec_sd = O.golden_state_dicts(g)
    with tempfile.TemporaryDirectory() as d:
        pe, pd = os.path.join(d, 'e.pth'), os.path.join(d, 'd.pth')
        torch.save(enc_sd, pe)
        torch.save(dec_sd, pd)
        enc = ModelBuilder.build_encoder(m['arch_encoder'], fc_dim=m['fc_dim'], weights=pe)
        dec = ModelBuilder.build_decoder(m['arch_decoder'], fc_dim=m['fc_dim'], num_class=150, weights=pd,
                                         use_softmax=use_softmax)
    if 'main' in g['dropout']:
        dec.conv_last[3].mask_override = g['dropout']['main'].to(dev)
    if 'deepsup' in g['dropout']:
        dec.dropout_deepsup.mask_override = g['dropout']['deepsup'].to(dev)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), m['deep_sup_scale']).to(dev)
    sm.train(m['training'])
    return sm, enc_sd, dec_sd


def argmax_check(got_logp, ref_logp, what):
    """bit-exact argmax except (reported) near-ties of the reference itself"""
    got, ref = got_logp.argmax(1), ref_logp.argmax(1)
    top2 = ref_logp.topk(2, dim=1)[0]
    margin = top2[:, 0] - top2[:, 1]
    diff = got != ref
    hard = diff & (margin >= 1e-4)
    parity_line('%s: %d/%d argmax flips, %d of them outside near-ties (margin>=1e-4); min margin %.2e' % (
        what, diff.sum().item(), diff.numel(), hard.sum().item(), margin.min().item()))
    assert hard.sum().item() == 0


@pytest.mark.parametrize('name', golden_cases())
def test_native_matches_reference_golden(name, monkeypatch):
    if name in HEURISTIC_PLAN_GOLDEN:
        from mit_semseg import tuner
        monkeypatch.setattr(tuner, 'ENABLED', False)
    g = load_golden(name)
    m = g['meta']
    dev = torch.device('cuda:0')
    sm, _, _ = build_native(g, dev, use_softmax=m['seg_size'] is not None)
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    feed = {'img_data': img.to(dev), 'seg_label': lab.to(dev)}
    if m['seg_size'] is not None:
        with torch.no_grad():
            prob = sm(feed, segSize=tuple(m['seg_size']))
        assert tuple(prob.shape) == tuple(g['prob'].shape)
        torch.testing.assert_close(prob.cpu().contiguous(), g['prob'], atol=1e-4, rtol=1e-3)
        argmax_check(prob.cpu().log(), g['prob'].log(), name)
        return
    cap = {}
    hk = sm.decoder.register_forward_hook(lambda mod, i, o: cap.__setitem__('out', o))
    if m['step']:
        from mit_semseg.engine import TrainStep
        ts = TrainStep(sm, lr_encoder=m['lr'], lr_decoder=m['lr'], max_iters=10 ** 9)
        loss, acc = ts.step(feed)
    else:
        with torch.no_grad():
            loss, acc = sm(feed)
    hk.remove()
    torch.cuda.synchronize()
    out = cap['out']
    pred, pred_ds = out if isinstance(out, tuple) else (out, None)
    pred = pred.detach().cpu().contiguous()
    parity_line('%s: max|dlogp| %.3e loss %.6f vs %.6f' % (name, (pred - g['pred']).abs().max().item(), loss.item(), g['loss'].item()))
    torch.testing.assert_close(pred, g['pred'], atol=LOGP_ATOL, rtol=0)
    argmax_check(pred, g['pred'], name)
    if pred_ds is not None:
        torch.testing.assert_close(pred_ds.detach().cpu().contiguous(), g['pred_deepsup'], atol=LOGP_ATOL, rtol=0)
    assert abs(loss.item() - g['loss'].item()) < 1e-3 * max(1.0, abs(g['loss'].item()))
    assert abs(acc.item() - g['acc'].item()) < 1e-6
    if not m['step'] or name in KNIFE_EDGE_GOLDEN:
        return
    # TrainStep has stepped already: compare the post-step state (weights, BN running stats) -- it pins grads, weight decay,
    # momentum and lr -- against the reference's float64 anchor with the reference's own fp32 deviation as the yardstick
    # (tests/util.anchor_ratio / check_anchor_ratios)
    items = []
    for mod, want, side in ((sm.encoder, g['anchor_after_enc'], 'enc.'), (sm.decoder, g['anchor_after_dec'], 'dec.')):
        sd = mod.state_dict()
        for k in want:
            if k.rsplit('.', 1)[-1] in ('_tmp_running_mean', '_tmp_running_var', '_running_iter'):
                continue
            items.append((side + k, sd[k], want[k]))
    parity_line(check_anchor_ratios(anchor_ratios(items, post_step_bands(g, m['lr'])), name + ' after-step state'))


def _native_grads(g, dev):
    m = g['meta']
    sm, enc_sd, dec_sd = build_native(g, dev)
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    loss, acc = sm({'img_data': img.to(dev), 'seg_label': lab.to(dev)})
    loss.backward()
    torch.cuda.synchronize()
    return sm


@pytest.mark.parametrize('name', ['r18d_ppmds_64_train', 'r50d_ppmds_64_train', 'r50_upernet_128_train', 'hrnetv2_c1_128_train',
                                  'mnv2d_c1ds_64_train', 'mnv2d_c1ds_192_train', 'r18d_ppmds_64_trainedlike_train',
                                  'r101_upernetlite_128_train', 'r18_c1_128_train'])
def test_native_gradients_vs_reference_anchor(name, monkeypatch):
    """EVERY parameter gradient of one backward against the float64 anchor of the unmodified reference
    (tests/golden/make_golden.py::anchor): elementwise on small tensors and on a seeded 1024-element sample of large ones --
    for `mnv2d_c1ds_192_train` on EVERY element of every gradient tensor (stored in full).  Yardstick = the reference's own
    fp32 reproducibility band per tensor: the largest deviation from the float64 run over five fp32 runs of the unmodified
    reference that differ only in execution (default / channels_last / 1 thread / 3 threads / input perturbed by 1e-7).
    Backward through these nets is ill-conditioned at ANY image size (ReLU gates whose pre-activation is ~1e-7 resolve either
    way; the same torch build moves ResNet-18's stem gradient by 2e-5 ... 2e-3 of its scale between those runs), so no fixed
    elementwise bound separates right from wrong; this one does: a kernel bug moves a gradient by O(1) of its norm."""
    if name in HEURISTIC_PLAN_GOLDEN or name.startswith('mnv2d'):
        from mit_semseg import tuner
        monkeypatch.setattr(tuner, 'ENABLED', False)
    g = load_golden(name)
    sm = _native_grads(g, torch.device('cuda:0'))
    items, heads = [], []
    for mod, want, side in ((sm.encoder, g['anchor_grads_enc'], 'enc.'), (sm.decoder, g['anchor_grads_dec'], 'dec.')):
        for k, p in mod.named_parameters():
            items.append((side + k, p.grad, want[k]))
            if side == 'dec.' and is_head_tensor(k, p):
                heads.append((scale_error(p.grad, want[k]), side + k))
    parity_line(check_anchor_ratios(anchor_ratios(items), name + ' gradients'))
    parity_line(check_directions(items, name + ' gradients'))                 # the well-conditioned twin: cosine / dot product per tensor
    # the classifier convs: no ReLU gate between them and the loss, so their gradients agree elementwise at roundoff level
    # (measured 2e-6 ... 1.2e-5 of the tensor's scale on every case, h2 and exact-fp32 alike)
    assert heads, 'no classifier tensors found'
    parity_line('%s classifier gradients: max |err| / scale %.2e (%s)' % ((name,) + max(heads)))
    assert max(heads)[0] <= HEAD_SCALE_ERR, heads


FULL_SIZE = {
    # BASELINE.json configs[1..4] at their full sizes: (arch_encoder, arch_decoder, fc_dim, deep_sup_scale, seg_rate, H, W); configs[3]
    # is variable-size: two non-square batch shapes of the multi-scale rule (dataset.py:121-142), the second with odd feature-map
    # sizes (57 x 85).  Every case has a fixture of the UNMODIFIED reference (tests/golden/make_fullsize_golden.py).
    'cfg1_r50d_ppmds_512': ('resnet50dilated', 'ppm_deepsup', 2048, 0.4, 8, 512, 512),
    'cfg2_r50_upernet_512': ('resnet50', 'upernet', 2048, None, 4, 512, 512),
    'cfg3_r101d_ppmds_376x504': ('resnet101dilated', 'ppm_deepsup', 2048, 0.4, 8, 376, 504),
    'cfg3_r101d_ppmds_456x680': ('resnet101dilated', 'ppm_deepsup', 2048, 0.4, 8, 456, 680),
    'cfg4_hrnetv2_c1_512': ('hrnetv2', 'c1', 720, None, 4, 512, 512),
}


@pytest.mark.parametrize('case', sorted(FULL_SIZE))
def test_full_size_vs_oracle(case):
    """BASELINE.json configs[1..4] at FULL size (bs 2; 512x512, and two variable-size shapes for configs[3]): one full training
    step on the device (TrainStep: forward, loss, backward, 2 x SGD) against
      * the fixture of the UNMODIFIED reference for this case: the log-probabilities of the stored pixel sample within 1e-3, the
        arg-max of EVERY pixel identical outside the reference's own near-ties, loss / accuracy; EVERY parameter gradient against
        its float64 anchor in units of the reference's fp32 reproducibility band (median <= 1, p95 <= 3, max <= 12 over all
        tensors, tests/util.check_anchor_ratios -- the acceptance of the small goldens, now on the BASELINE shapes and on the launch
        forms that only exist there: 256 x 256 tiles, the fused Winograd data gradient, batched / deferred weight gradients);
        the classifier gradients elementwise at 1e-4 of their scale; every parameter and BN running statistic after the step;
      * the oracle (torch CPU) running the same forward on the box's host cores: log-probs of EVERY pixel within 1e-3, arg-max."""
    from mit_semseg.engine import TrainStep, group_weight
    import json
    arch_enc, arch_dec, fc_dim, dss, rate, H, W = FULL_SIZE[case]
    fx = load_fullsize_golden(case)
    assert fx is not None and 'anchor_grads_enc' in fx, 'tests/golden/fullsize/%s.pt is missing or predates the gradient anchors' % case
    lr = fx['meta']['lr']
    dev = torch.device('cuda:0')
    man = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'manifests.json')))
    drop = {'main': O.synth_dropout_mask(2, 512, seed=3)} if 'ppm' in arch_dec else {}
    if arch_dec == 'ppm_deepsup':
        drop['deepsup'] = O.synth_dropout_mask(2, fc_dim // 4, seed=4)
    g = dict(meta=dict(arch_encoder=arch_enc, arch_decoder=arch_dec, fc_dim=fc_dim, seed=3, training=True, deep_sup_scale=dss,
                       seg_size=None),
             manifest_enc=man[arch_enc], manifest_dec=man['%s@%d' % (arch_dec, fc_dim)], dropout=drop)
    sm, enc_sd, dec_sd = build_native(g, dev)
    img, lab = O.synth_batch(2, H, W, rate, seed=307)
    cap = {}
    hk = sm.decoder.register_forward_hook(lambda mod, i, o: cap.__setitem__('out', o))
    ts = TrainStep(sm, lr_encoder=lr, lr_decoder=lr, max_iters=10 ** 9)
    loss, acc = ts.step({'img_data': img.to(dev), 'seg_label': lab.to(dev)})
    hk.remove()
    torch.cuda.synchronize()
    out = cap['out']
    pred = (out[0] if isinstance(out, tuple) else out).detach().cpu().contiguous()

    # forward: the fixture of the unmodified reference, then every pixel against the oracle on this box
    check_fullsize_golden(pred, loss.item(), acc.item(), fx, case, LOGP_ATOL)
    torch.set_num_threads(min(os.cpu_count(), 32))    # torch CPU convs thrash beyond ~32 threads (203 s at 256)
    with torch.no_grad():
        ref = O.segmentation_forward(O.clone_sd(enc_sd, False), O.clone_sd(dec_sd, False), arch_enc, arch_dec, img, lab,
                                     training=True, dropout=drop, deep_sup_scale=dss)
    rp = ref['pred'].detach()
    parity_line('%s vs the oracle on this box: max|dlogp| %.3e  loss %.6f vs %.6f' % (case, (pred - rp).abs().max().item(), loss.item(),
                                                                                      ref['loss'].item()))
    torch.testing.assert_close(pred, rp, atol=LOGP_ATOL, rtol=0)
    argmax_check(pred, rp, case)
    assert abs(loss.item() - ref['loss'].item()) < 1e-3
    assert abs(acc.item() - ref['acc'].item()) < 1e-6

    # backward: EVERY gradient tensor against the reference's float64 anchor (the SGD kernel reads the gradients, it does not
    # modify them: p.grad after the step is the gradient of the step)
    items, heads = [], []
    for mod, want, side in ((sm.encoder, fx['anchor_grads_enc'], 'enc.'), (sm.decoder, fx['anchor_grads_dec'], 'dec.')):
        names = [k for k, _ in mod.named_parameters()]
        assert sorted(names) == sorted(want), (side, set(names) ^ set(want))
        for k, p in mod.named_parameters():
            items.append((side + k, p.grad, want[k]))
            if side == 'dec.' and is_head_tensor(k, p):
                heads.append((scale_error(p.grad, want[k]), side + k))
    parity_line(check_anchor_ratios(anchor_ratios(items), case + ' gradients'))
    parity_line(check_directions(items, case + ' gradients'))
    assert heads, 'no classifier tensors found'
    parity_line('%s classifier gradients: max |err| / scale %.2e (%s)' % ((case,) + max(heads)))
    assert max(heads)[0] <= HEAD_SCALE_ERR, heads

    # state after the step: parameters against w0 - lr (g64 + wd w0), BN running statistics against their own anchors
    items = []
    bands = post_step_bands(fx, lr)
    for mod, sd0, gk, ak, side in ((sm.encoder, enc_sd, 'anchor_grads_enc', 'anchor_after_enc', 'enc.'),
                                   (sm.decoder, dec_sd, 'anchor_grads_dec', 'anchor_after_dec', 'dec.')):
        decay = set(id(p) for p in group_weight(mod)[0])
        got = mod.state_dict()
        for k, p in mod.named_parameters():
            wd = 1e-4 if id(p) in decay else 0.0
            items.append((side + k, got[k], post_step_record(fx[gk][k], sd0[k], lr, wd)))
        for k, rec in fx[ak].items():
            items.append((side + k, got[k], rec))
    parity_line(check_anchor_ratios(anchor_ratios(items, bands), case + ' after-step state'))


@pytest.mark.parametrize('name', KNIFE_EDGE_GOLDEN)
def test_knife_edge_case_gradients_stay_within_a_loose_elementwise_limit(name):
    """`hrnetv2_c1_64_train` is exempt from the band acceptance of its gradients (tests/util.KNIFE_EDGE_GOLDEN: its coarsest branch
    is a 2 x 2 map, the result sits 0.00 or 3.7 bands from the anchor depending on the launch plans).  It is NOT exempt from
    being right: the MEDIAN gradient tensor within KNIFE_EDGE_MEDIAN_SCALE_ERR of its scale (a flip of the knife edge moves it by
    2.8e-3), every tensor within KNIFE_EDGE_SCALE_ERR (single tensors of the 2 x 2 branch move by 13 ... 25 % on a flip), the
    classifier gradients at roundoff level as everywhere -- so a regression that moves the 64 x 64 HRNet gradients by a few per
    cent across the board, or breaks the 2 x 2 BN path in backward, fails here."""
    g = load_golden(name)
    sm = _native_grads(g, torch.device('cuda:0'))
    worst, heads, errs, items = (0.0, ''), [], [], []
    for mod, want, side in ((sm.encoder, g['anchor_grads_enc'], 'enc.'), (sm.decoder, g['anchor_grads_dec'], 'dec.')):
        for k, p in mod.named_parameters():
            assert torch.isfinite(p.grad).all(), side + k
            items.append((side + k, p.grad, want[k]))
            e = scale_error(p.grad, want[k])
            errs.append(e)
            worst = max(worst, (e, side + k))
            if side == 'dec.' and is_head_tensor(k, p):
                heads.append((e, side + k))
    median = sorted(errs)[len(errs) // 2]
    parity_line('%s gradients (knife-edge case, loose limits): |err| / scale median %.2e, worst %.2e (%s), classifier %.2e' % (
        name, median, worst[0], worst[1], max(heads)[0]))
    assert median <= KNIFE_EDGE_MEDIAN_SCALE_ERR, median
    assert worst[0] <= KNIFE_EDGE_SCALE_ERR, worst
    # ... and to the direction check of every other case (round-5 review): a knife edge resolving the other way moves single
    # elements of the 2 x 2 branch, not the direction of the tensors
    parity_line(check_directions(items, name + ' gradients (knife-edge case)', min_cos=KNIFE_EDGE_COSINE_MIN))
    assert max(heads)[0] <= HEAD_SCALE_ERR, heads


def test_deferred_wgrad_reduce_is_bit_identical(monkeypatch):
    """TrainStep sums the slabs of all split weight gradients in ONE multi-tensor launch after backward (ops.defer_wgrad_reduces)
    and runs the small weight gradients themselves side by side in one launch per 24 (semseg_conv2d_wgrad_multi_h2): same blocks,
    same slab order per tensor as the per-layer launches, so the state after two steps is bit-identical to the step that computes
    and reduces every weight gradient where autograd reaches it"""
    from mit_semseg import ops, tuner
    from mit_semseg.engine import TrainStep
    monkeypatch.setattr(tuner, 'ENABLED', False)          # heuristic plans: all runs launch the same plans
    g = load_golden('r18d_ppmds_64_train')
    m = g['meta']
    dev = torch.device('cuda:0')
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    feed = {'img_data': img.to(dev), 'seg_label': lab.to(dev)}
    batched = []
    real = ops.flush_wgrad_reduces

    def counting(*a, **kw):
        batched.append(len(ops._PENDING_WGRADS))
        return real(*a, **kw)
    monkeypatch.setattr(ops, 'flush_wgrad_reduces', counting)
    states = []
    for defer, launch in ((True, True), (True, False), (False, False)):
        monkeypatch.setattr(ops, 'DEFER_WGRAD_REDUCE', defer)
        monkeypatch.setattr(ops, 'DEFER_WGRAD_LAUNCH', launch)
        sm, _, _ = build_native(g, dev)
        ts = TrainStep(sm, lr_encoder=m['lr'], lr_decoder=m['lr'], max_iters=10 ** 9)
        del batched[:]
        for _ in range(2):
            loss, _ = ts.step(feed)
        torch.cuda.synchronize()
        assert not ops._PENDING_SLABS and not ops._PENDING_WGRADS and not ops._SLABS_FOR_SGD
        assert (max(batched) >= 8) == (defer and launch), batched      # the path under test ran (r18: more than 8 small layers)
        states.append(({k: v.clone() for k, v in sm.state_dict().items()}, loss.clone()))
    for (b, lb) in states[1:]:
        a, la = states[0]
        assert torch.equal(la, lb)
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_inference_graph_replay_equals_eager():
    """engine.InferenceGraph: the captured inference branch returns exactly what the eager call returns, for new inputs too"""
    from mit_semseg.engine import InferenceGraph
    g = load_golden('r18d_ppm_infer_64x80')
    m = g['meta']
    dev = torch.device('cuda:0')
    sm, _, _ = build_native(g, dev, use_softmax=True)
    run = InferenceGraph(sm)
    for seed in (1, 2, 3):
        img, _ = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=seed)
        img = img.to(dev)
        with torch.no_grad():
            want = sm({'img_data': img}, segSize=tuple(m['seg_size'])).clone()
        got = run(img, tuple(m['seg_size']))
        torch.cuda.synchronize()
        assert torch.equal(got, want), seed



def _batch_stats():
    import ctypes
    from mit_semseg import _native
    out = (ctypes.c_longlong * 4)()
    _native.check(_native.lib().semseg_batch_stats(out), 'batch_stats')
    return dict(scopes=out[0], recorded=out[1], issued=out[2], forced=out[3])


def test_hrnet_branch_streams_equal_one_stream(monkeypatch):
    """ops.run_branches: HRNetV2's parallel branches (hrnet.py:225-227) train to exactly the weights of the one-stream step
      * on side HIP streams (the form of rounds 2-5), with eager launches and as a captured hipGraph whose branches are parallel chains;
      * as SIDE-BY-SIDE launches (round 6, csrc/batch.h: the launches of all branches recorded and issued position by position through
        the many-problem trampoline, forward and -- ops.BranchesFn -- backward), eagerly and inside a captured hipGraph
    -- same kernels, same order inside every branch, so the state after four steps is compared with torch.equal."""
    from mit_semseg import ops, tuner
    from mit_semseg.engine import TrainStep
    monkeypatch.setattr(tuner, 'ENABLED', False)          # heuristic launch plans: the runs must sum in the same order
    from mit_semseg import _native
    prev_tile = _native.lib().semseg_batch_plan(-1, 0)    # ... inside the scopes too (their own tile form changes the summation order)
    monkeypatch.setattr(ops, 'BATCH_CHECK', True)         # and no torch-side kernel may run inside a scope (it would overtake the records)
    try:
        _hrnet_branch_forms(monkeypatch, ops, TrainStep)
    finally:
        _native.lib().semseg_batch_plan(prev_tile, 0)


def _hrnet_branch_forms(monkeypatch, ops, TrainStep):
    g = load_golden('hrnetv2_c1_64_train')
    m = g['meta']
    dev = torch.device('cuda:0')
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=m['seed'] + 2)
    feed = {'img_data': img.to(dev), 'seg_label': lab.to(dev)}
    res = {}
    for streams, graph, batch in ((False, False, False), (True, False, False), (True, True, False), (True, False, True), (True, True, True)):
        monkeypatch.setattr(ops, 'BRANCH_STREAMS', streams)
        monkeypatch.setattr(ops, 'BATCH_BRANCHES', batch)
        sm, _, _ = build_native(g, dev)
        ts = TrainStep(sm, max_iters=1000, graph=graph)
        before = _batch_stats()
        for _ in range(4):
            loss, acc = ts.step(feed)
        torch.cuda.synchronize()
        after = _batch_stats()
        if graph:
            assert ts.stats['replayed'] >= 2
        if batch:
            # the path under test ran: scopes were opened (forward + backward of every HighResolutionModule with > 1 branch) and
            # their launches carried several problems each
            scopes, recorded, issued, forced = (after[k] - before[k] for k in ('scopes', 'recorded', 'issued', 'forced'))
            assert scopes >= 16 and recorded >= 1.5 * issued and forced == 0, (before, after)
            parity_line('hrnetv2_c1_64_train side-by-side launches (%s): %d scopes, %d recorded launches left as %d (%.2f problems per '
                        'launch), %d direct launches inside a scope' % ('graph capture' if graph else 'eager', scopes, recorded, issued,
                                                                       recorded / max(1, issued), forced))
        else:
            assert after['scopes'] == before['scopes']
        res[(streams, graph, batch)] = ({k: v.detach().clone() for k, v in sm.state_dict().items()}, loss.item())
    assert ops._BRANCH_POOL and any(k[2].startswith('branch') for k in ops._WS)       # the branch streams really ran, on their own scratch
    base = res[(False, False, False)]
    for key in res:
        assert res[key][1] == base[1], (key, res[key][1], base[1])
        for k, v in base[0].items():
            assert torch.equal(res[key][0][k], v), (key, k)


@pytest.mark.parametrize('name', ['r50d_ppmds_64_train', 'r50_upernet_128_train', 'hrnetv2_c1_128_train'])
def test_side_by_side_scopes_hold_no_torch_kernels(name, monkeypatch):
    """the scopes of every model family that opens them (PPM pyramid branches; UPerNet pyramid / lateral / output branches; HRNetV2
    branches, exchange chains and rows) under ops.BATCH_CHECK: two training steps in which every torch operator dispatched inside a
    scope must be launch-free -- a torch kernel there would run ahead of the recorded launches it depends on"""
    from mit_semseg import ops
    from mit_semseg.engine import TrainStep
    if not (ops.FUSE and ops.CONV_MODE == 'h2' and ops.BATCH_BRANCHES):
        pytest.skip('no scopes in this process (a switch turns them off)')
    monkeypatch.setattr(ops, 'BATCH_CHECK', True)
    g = load_golden(name)
    m = g['meta']
    dev = torch.device('cuda:0')
    sm, _, _ = build_native(g, dev)
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    ts = TrainStep(sm, lr_encoder=m['lr'], lr_decoder=m['lr'], max_iters=10 ** 9)
    before = _batch_stats()
    for _ in range(2):
        loss, _ = ts.step({'img_data': img.to(dev), 'seg_label': lab.to(dev)})
    torch.cuda.synchronize()
    after = _batch_stats()
    assert torch.isfinite(loss) and after['scopes'] - before['scopes'] >= 2, (before, after)
    parity_line('%s: %d side-by-side scopes in two steps, %d recorded launches left as %d, %d direct launches inside a scope; no torch '
                'kernel inside any of them' % (name, after['scopes'] - before['scopes'], after['recorded'] - before['recorded'],
                                               after['issued'] - before['issued'], after['forced'] - before['forced']))


# every environment switch of the product that survives in the code base, with the golden case(s) that exercise what it changes:
# the parity suite must hold on BOTH sides of each switch (they are read at import time, hence one child process per switch)
@pytest.mark.parametrize('name', ['r50d_ppmds_64_train', 'hrnetv2_c1_128_train'])
def test_deferred_fork_sums_are_bit_identical(name, monkeypatch):
    """the gradient sums at the forks (block output -> next block's first conv + its shortcut; HRNet's branch outputs -> every row of
    the exchange) formed inside the BN backward kernels that consume them (ops.defer_fork_sums, semseg_bn_bwd_*_sum2) give the same
    bits as the add launches they replace: one fp32 add per element either way; state after two steps compared with torch.equal"""
    from mit_semseg import ops, tuner
    from mit_semseg.engine import TrainStep
    if not (ops.FUSE and ops.CONV_MODE == 'h2'):
        pytest.skip('the sums are only deferred on the fused h2 path (this process runs with a switch that turns it off)')
    monkeypatch.setattr(tuner, 'ENABLED', False)          # heuristic plans: both runs launch the same plans
    g = load_golden(name)
    m = g['meta']
    dev = torch.device('cuda:0')
    img, lab = O.synth_batch(m['n'], m['h'], m['w'], m['seg_rate'], seed=304 + m['seed'])
    feed = {'img_data': img.to(dev), 'seg_label': lab.to(dev)}
    taken = []
    real = ops._take_addend

    def counting(t):
        r = real(t)
        if r is not None:
            taken.append(tuple(t.shape))
        return r
    monkeypatch.setattr(ops, '_take_addend', counting)
    states = []
    for defer in (True, False):
        monkeypatch.setattr(ops, 'DEFER_FORK_SUMS', defer)
        sm, _, _ = build_native(g, dev)
        ts = TrainStep(sm, lr_encoder=m['lr'], lr_decoder=m['lr'], max_iters=10 ** 9)
        del taken[:]
        for _ in range(2):
            loss, _ = ts.step(feed)
        torch.cuda.synchronize()
        assert not ops._ADDENDS
        assert (len(taken) >= 8) == defer, (defer, len(taken))          # the path under test ran
        states.append(({k: v.clone() for k, v in sm.state_dict().items()}, loss.clone()))
    (a, la), (b, lb) = states
    assert torch.equal(la, lb)
    for k in a:
        assert torch.equal(a[k], b[k]), k


SWITCH_CASES = [
    ('SEMSEG_CONV=s3', 'r18d_ppmds_64_train'),               # 3-way bf16 split family end to end
    ('SEMSEG_CONV=f32', 'r18d_ppmds_64_train'),              # exact-fp32 MFMA family end to end
    ('SEMSEG_FUSE=0', 'r50d_ppmds_64_train'),                # no plane hand-over, unfused conv / BN nodes
    ('SEMSEG_WINOGRAD=0', 'r50d_ppmds_64_train'),            # conv_last (4096 channels) through the direct kernels
    ('SEMSEG_WINOGRAD_WGRAD=0', 'r50d_ppmds_64_train'),
    ('SEMSEG_WINOGRAD_DGRAD=0', 'r50d_ppmds_64_train'),      # conv_last / deepsup / layer4 data gradients through the direct kernels
    ('SEMSEG_WINOGRAD_MIN_C=256', 'r50d_ppmds_64_train'),    # Winograd for every eligible 3x3 conv of the net
    ('SEMSEG_WINOGRAD_MIN_C=1024', 'r50d_ppmds_64_train'),   # round 2's threshold: layer4's 512-channel convs direct
    ('SEMSEG_TUNE=0', 'r50d_ppmds_64_train'),                # the library's heuristic launch plans
    ('SEMSEG_EPILOGUE_STATS=0', 'r50d_ppmds_64_train'),
      # BN statistics by the separate sweep instead of the conv epilogue
    ('SEMSEG_TUNE_BUCKETS=0', 'r18d_ppmds_64_train'),
    ('SEMSEG_FORCE_SYNC_PATH=1', 'r18d_ppmds_64_train'),     # the unfused SyncBN kernel sequence on one rank
    ('SEMSEG_BRANCH_STREAMS=0', 'hrnetv2_c1_128_train'),
    ('SEMSEG_BATCH_BRANCHES=0', 'hrnetv2_c1_128_train'),    # the branches on side streams (rounds 2-5) instead of side-by-side launches
    ('SEMSEG_PLANES_ONLY=0', 'r50d_ppmds_64_train'),         # every BN output also as fp32 (round 4: bn1 / bn2 of a block are planes only)
    ('SEMSEG_WINOGRAD_FUSED=0', 'r50d_ppmds_64_train'),      # Winograd data gradients as batched GEMM + output transform
    ('SEMSEG_TUNE_DB=0', 'r18d_ppmds_64_train'),
    ('SEMSEG_DEFER_WGRAD_REDUCE=0', 'r50d_ppmds_64_train'),  # one reduce launch per split weight gradient instead of ONE per step             # no shipped launch plans: every geometry timed in the process
    ('SEMSEG_DEFER_WGRAD_LAUNCH=0', 'hrnetv2_c1_128_train'), # every small weight gradient its own launch instead of 24 per launch
    ('SEMSEG_WGRAD_MEMBER_PLAN=0', 'hrnetv2_c1_128_train'),  # every small weight gradient on its own plan instead of joining the batched launch
    ('SEMSEG_DEFER_FORK_SUMS=0', 'r50d_ppmds_64_train'),     # the gradient sums at the forks by add launches instead of inside the BN backward kernels
    ('SEMSEG_DEFER_FORK_SUMS=0', 'hrnetv2_c1_128_train'),
    ('SEMSEG_DMA64_SPREAD=0', 'r50d_ppmds_64_train'),        # the 64-deep GEMM tiles with their DMA pieces in one burst per k-tile
    ('SEMSEG_WINO_WGRAD_FORM=0', 'r50d_ppmds_64_train'),     # the batched Winograd weight-gradient GEMM on the plain 2-slot loop
    ('SEMSEG_SGD_FUSED=0', 'r50d_ppmds_64_train'),           # reduce launch + plain SGD kernel + absmax pass instead of the one fused pass over the weights
    ('SEMSEG_DEPTHWISE_DIRECT=0', 'mnv2d_c1ds_64_train'),
    ('SEMSEG_GROUPED_DIRECT=0', 'resnext101_upernet_128_eval'),
]


_SWITCH_FARM = {}


def _switch_child(switch, case):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    k, v = switch.split('=')
    env = dict(os.environ, SEMSEG_SWITCH_CHILD='1', **{k: v})
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_models.py'), '-q', '-x', '-m', 'gpu',
                        '-k', '%s and (matches_reference_golden or gradients_vs_reference_anchor)' % case],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    return r.returncode, r.stdout[-1500:], r.stderr[-1500:]


def _switch_result(switch, case):
    """the children of all switch cases run from a small pool, SEMSEG_SWITCH_WORKERS at a time -- ONE by default: measured on the
    MI355X box, two at a time take as long as one after the other (375 s for 24 children either way, gpurun r8n: a child's ~14 s are
    device context + library load + its golden tests, and processes sharing one GPU are time-sliced), six at a time made the children
    that time launch plans crawl (820 s for one child, gpurun r8l).  The suite is ~600 s of the driver's 1 200 s with them."""
    if not _SWITCH_FARM:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=int(os.environ.get('SEMSEG_SWITCH_WORKERS', '1')))
        for sw, cs in SWITCH_CASES:
            _SWITCH_FARM[(sw, cs)] = pool.submit(_switch_child, sw, cs)
        pool.shutdown(wait=False)
    return _SWITCH_FARM[(switch, case)].result()


@pytest.mark.parametrize('switch,case', SWITCH_CASES, ids=[s for s, _ in SWITCH_CASES])
def test_env_switch_keeps_model_parity(switch, case):
    """the golden tests of `case` (test_native_matches_reference_golden: forward bounds + post-step state; where stored,
    test_native_gradients_vs_reference_anchor: every gradient against the float64 anchors, band and direction) in a child process
    with `switch` set"""
    if os.environ.get('SEMSEG_SWITCH_CHILD'):
        pytest.skip('already inside a switch child')
    rc, tail, err = _switch_result(switch, case)
    assert rc == 0 and ' passed' in tail, '%s: %s\n%s' % (switch, tail, err)
