"""The analytic multi-GPU model (mit_semseg/scaling_model.py) on hand-made timelines: pure host arithmetic, no GPU."""
import torch

from mit_semseg import scaling_model as smod
from mit_semseg.parallel import plan_bucket_groups, GradientBuckets


def _params(sizes):
    return [torch.nn.Parameter(torch.zeros(n)) for n in sizes]


def test_bucket_groups_are_what_gradient_buckets_builds():
    ps = _params([1000, 2000, 300, 50, 4000, 10, 20])
    order = list(reversed(ps))
    groups = plan_bucket_groups(order, bucket_bytes=4 * 4500, tail_bytes=4 * 400)
    gb = GradientBuckets(ps, bucket_bytes=4 * 4500, tail_bytes=4 * 400, overlap=False)
    assert [[id(p) for p, _, _ in b['items']] for b in gb.buckets] == [[id(p) for p in g] for g in groups]
    assert sum(len(g) for g in groups) == len(ps)
    # backward order 20, 10, 4000, 50, 300 | 2000, 1000: no parameter of the last bucket fits the 400-element tail, so nothing
    # is carved off; with a 1100-element tail the last parameter (the first layer) travels alone
    assert [[p.numel() for p in g] for g in groups] == [[20, 10, 4000, 50, 300], [2000, 1000]]
    g2 = plan_bucket_groups(order, bucket_bytes=4 * 8000, tail_bytes=4 * 1100)
    assert [[p.numel() for p in g] for g in g2] == [[20, 10, 4000, 50, 300, 2000], [1000]]


def test_allreduce_time_is_ring_arithmetic():
    a = dict(smod.ASSUMPTIONS)
    n, nbytes = 8, 64 << 20
    t = smod.allreduce_ms(nbytes, n, a)
    wire = 2.0 * (n - 1) / n * nbytes / (a['xgmi_link_GBps'] * 1e9 * a['ring_efficiency']) * 1e3
    fixed = (a['allreduce_launch_us'] + 2 * (n - 1) * a['ring_hop_us']) * 1e-3
    assert abs(t - (wire + fixed)) < 1e-9
    assert smod.allreduce_ms(nbytes, 1, a) == 0.0
    assert smod.allreduce_ms(nbytes, 2, a) < smod.allreduce_ms(nbytes, 8, a)


def test_prediction_exposes_only_what_outlives_backward():
    # backward from 5 ms to 13 ms; two big buckets early, a tiny tail bucket at the very end
    marks = {'step_begin': 0.0, 'fwd_end': 5.0, 'bucket0': 7.0, 'bucket1': 10.0, 'bucket2': 12.9, 'bwd_end': 13.0, 'step_end': 14.0}
    buckets = [64 << 20, 64 << 20, 1 << 20]
    bn = [64] * 10
    quiet = {'overlap_slowdown': 0.0, 'syncbn_peer_us': 0.0, 'syncbn_peer_skew_us': 0.0}
    p = smod.predict(14.0, marks, buckets, bn, 8, quiet)
    t_big, t_tail = smod.allreduce_ms(64 << 20, 8, smod.ASSUMPTIONS), smod.allreduce_ms(1 << 20, 8, smod.ASSUMPTIONS)
    # bucket0 7.0 -> 7+t, bucket1 starts at max(10, 7+t), tail starts at max(12.9, end of bucket1)
    end1 = max(10.0, 7.0 + t_big) + t_big
    end = max(12.9, end1) + t_tail
    assert abs(p['exposed_allreduce_ms'] - max(0.0, end - 13.0)) < 2e-3
    assert abs(p['ms_per_step'] - (14.0 + p['exposed_allreduce_ms'])) < 2e-3
    assert 0 < p['efficiency_vs_1gpu'] <= 1
    # SyncBN through RCCL costs 20 exchanges x syncbn_rccl_us more
    r = smod.predict(14.0, marks, buckets, bn, 8, quiet, syncbn='rccl')
    assert abs((r['ms_per_step'] - p['ms_per_step']) - 20 * smod.ASSUMPTIONS['syncbn_rccl_us'] * 1e-3) < 2e-3


def test_model_line_scales_ticks_to_the_measured_step():
    ticks = {'step_begin': 1000, 'fwd_end': 1500, 'bucket0': 1800, 'bucket1': 1990, 'bwd_end': 2000, 'step_end': 2100}
    line = smod.model_line(11.0, ticks, [10 << 20, 1 << 20], [64, 128, 256])
    tl = line['timeline_ms']
    assert tl['step_begin'] == 0.0 and abs(tl['step_end'] - 11.0) < 1e-9 and abs(tl['fwd_end'] - 5.0) < 1e-9
    assert set(line['predicted']) == {'2', '4', '8'}
    assert line['syncbn_exchanges_per_step'] == 6
    assert line['syncbn_payload_doubles'] == {'fwd': 2 * 448 + 3, 'bwd': 2 * 448}
    assert 'MODEL' in line['kind']
    e = [line['predicted'][n]['peer_exchange']['efficiency_vs_1gpu'] for n in ('2', '4', '8')]
    assert e[0] >= e[1] >= e[2]


def test_model_line_builds_on_the_rank_form_of_the_step():
    """the single-GPU step batches its weight gradients after backward; a rank under gradient buckets cannot (its hooks read every
    gradient when autograd accumulates it): predictions are built on the step timed in the rank's form, efficiencies are quoted
    against the single-GPU step"""
    ticks = {'step_begin': 1000, 'fwd_end': 1500, 'bucket0': 1800, 'bucket1': 1990, 'bwd_end': 2000, 'step_end': 2100}
    base = smod.model_line(11.0, ticks, [10 << 20, 1 << 20], [64, 128, 256])
    line = smod.model_line(11.0, ticks, [10 << 20, 1 << 20], [64, 128, 256], t_rank_ms=11.5)
    assert line['measured_1gpu_ms_per_step'] == 11.0 and line['measured_rank_form_ms_per_step'] == 11.5
    assert abs(line['timeline_ms']['step_end'] - 11.5) < 1e-9
    for n in ('2', '4', '8'):
        p, q = line['predicted'][n]['peer_exchange'], base['predicted'][n]['peer_exchange']
        assert p['ms_per_step'] > q['ms_per_step'] + 0.45            # the rank's compute is 0.5 ms longer
        assert abs(p['efficiency_vs_1gpu'] - 11.0 / p['ms_per_step']) < 1e-3
        assert p['efficiency_vs_1gpu'] < q['efficiency_vs_1gpu']
    assert base['measured_rank_form_ms_per_step'] == 11.0
