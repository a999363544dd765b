"""TrainStep's per-shape graph policy (engine.TrainStep.step; BASELINE configs[3]: variable-size per-GPU batches, dataset.py:110-142),
host logic only: capture and eager passes are replaced by recorders, no GPU, no native library."""
import torch
import torch.nn as nn

from mit_semseg import engine, tuner


class _Graph:
    def __init__(self, log, key):
        self.log, self.key = log, key

    def replay(self):
        self.log.append(('replay', self.key))


def _step(monkeypatch, first_sight, missing=(), unrecordable=()):
    """a TrainStep whose capture / eager passes only write to a log; capturing a shape in `missing` meets a geometry without a launch plan"""
    monkeypatch.setenv('SEMSEG_CAPTURE_FIRST_SIGHT', '1' if first_sight else '0')
    net = nn.Module()
    net.encoder, net.decoder = nn.Linear(2, 2), nn.Linear(2, 2)
    ts = engine.TrainStep(net, graph=True)
    log = []
    ts.launch_mode = lambda: 'graph'

    def eager(feed):
        ts.opt.steps += 1
        log.append(('eager', tuple(feed['x'].shape)))
        return torch.zeros(()), torch.zeros(())

    def capture(key, feed, mode='graph'):
        shape = tuple(feed['x'].shape)
        log.append(('capture', shape))
        if shape in unrecordable and shape not in timed:
            raise RuntimeError('operation not permitted when stream is capturing')
        if shape in missing and shape not in timed:
            tuner.stats['missed_capturing'] += 1
        rec = ts._graphs[key] = (_Graph(log, shape), {}, (torch.zeros(()), torch.zeros(())))
        ts.stats['captured'] += 1
        return rec

    timed = set()
    orig_eager = eager

    def eager_times(feed):
        timed.add(tuple(feed['x'].shape))           # an eager pass is where the tuner times what it could not inherit
        return orig_eager(feed)

    monkeypatch.setattr(ts, '_eager', eager_times)
    monkeypatch.setattr(ts, '_capture', capture)
    monkeypatch.setattr(ts, 'adjust_learning_rate', lambda: None)
    return ts, log


def _run(ts, log, shapes):
    for hw in shapes:
        ts.step({'x': torch.zeros(hw)})
        if log[-1][0] == 'replay':
            ts.opt.steps += 1
    return log


A, B, C = (1, 2), (1, 3), (1, 4)


def test_new_shapes_are_captured_at_first_sight_after_the_warm_up(monkeypatch):
    ts, log = _step(monkeypatch, True)
    _run(ts, log, [A, B, A, C, C, B])
    assert log == [('eager', A), ('eager', B),                       # the two warm-up steps run eagerly whatever their shape
                   ('capture', A), ('replay', A),                    # seen before: captured
                   ('capture', C), ('replay', C),                    # NEW shape: captured at once, no eager pass
                   ('replay', C),
                   ('capture', B), ('replay', B)]
    assert ts.stats['provisional'] == 0 and ts.stats['eager'] == 2 and ts.stats['captured'] == 3


def test_the_old_order_is_one_switch_away(monkeypatch):
    ts, log = _step(monkeypatch, False)
    _run(ts, log, [A, B, A, C, C, C])
    assert log == [('eager', A), ('eager', B), ('capture', A), ('replay', A), ('eager', C), ('capture', C), ('replay', C), ('replay', C)]


def test_a_capture_that_met_an_unplanned_geometry_is_provisional(monkeypatch):
    """replayed once (the step must run), then the old order: an eager pass in which the tuner can time, and a capture for good"""
    ts, log = _step(monkeypatch, True, missing={C})
    _run(ts, log, [A, B, C, C, C, C])
    assert log == [('eager', A), ('eager', B),
                   ('capture', C), ('replay', C),                    # provisional: ran on the library's default plans
                   ('eager', C),                                     # times the missing plans
                   ('capture', C), ('replay', C),                    # for good
                   ('replay', C)]
    assert ts.stats['provisional'] == 1 and not ts._provisional


def test_a_shape_that_cannot_be_recorded_at_first_sight_falls_back_to_the_old_order(monkeypatch):
    """a failed first-sight recording executes nothing: the step runs eagerly instead, and the shape is recorded at its next sight"""
    ts, log = _step(monkeypatch, True, unrecordable={C})
    _run(ts, log, [A, B, C, C, C])
    assert log == [('eager', A), ('eager', B), ('capture', C), ('eager', C), ('capture', C), ('replay', C), ('replay', C)]
    assert ts.stats['capture_failed'] == 1 and ts.stats['captured'] == 1


def test_only_capture_refusals_fall_back_other_errors_propagate(monkeypatch):
    """an out-of-memory error or a kernel / argument error met while recording is NOT answered by an eager rerun (the
    rerun hid real bugs until the shape was seen a second time): it propagates, and the host state of the refused recording
    (optimiser step count) is put back before a legitimate eager retry"""
    import pytest
    from mit_semseg import _native
    assert _native.is_capture_error(RuntimeError('operation not permitted when stream is capturing'))
    assert _native.is_capture_error(_native.NativeError('conv2d_fwd_h2', 900))
    assert not _native.is_capture_error(_native.NativeError('conv2d_fwd_h2', -1))
    assert not _native.is_capture_error(torch.cuda.OutOfMemoryError('HIP out of memory while capturing'))
    for exc in (torch.cuda.OutOfMemoryError('HIP out of memory'), _native.NativeError('bn_apply_h2', -1), RuntimeError('shape mismatch')):
        ts, log = _step(monkeypatch, True)
        _run(ts, log, [A, B])

        def capture(key, feed, mode='graph', exc=exc):
            raise exc
        monkeypatch.setattr(ts, '_capture', capture)
        with pytest.raises(type(exc)):
            ts.step({'x': torch.zeros(C)})
        assert ts.stats['capture_failed'] == 0 and ts.stats['eager'] == 2
    # a refusal: the recording pass had advanced the optimiser's step count; the eager retry starts from the count before it
    ts, log = _step(monkeypatch, True)
    _run(ts, log, [A, B])
    seen = []

    def capture(key, feed, mode='graph'):
        ts.opt.steps += 1                           # what _eager does inside the recording
        raise RuntimeError('hipErrorStreamCaptureUnsupported: operation not permitted when stream is capturing')
    orig = ts._eager

    def eager(feed):
        seen.append(ts.opt.steps)
        return orig(feed)
    monkeypatch.setattr(ts, '_capture', capture)
    monkeypatch.setattr(ts, '_eager', eager)
    ts.step({'x': torch.zeros(C)})
    assert seen == [2] and ts.opt.steps == 3 and ts.stats['capture_failed'] == 1
