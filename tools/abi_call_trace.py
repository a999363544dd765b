"""Sequence of C-ABI calls of ONE training step with their integer arguments (CPU only: the ABI is replaced by a stub that records and
computes nothing, as tests/test_host_logic_dryrun.py does).  Lines up with the dispatch order of a kernel trace of the eager step
(tools/probes/step_traffic.py), so a launch in a PMC table can be given its geometry.   python tools/abi_call_trace.py [--config 1]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
import torch  # noqa: E402


class Recorder:
    def __init__(self, signatures):
        self.calls = []
        self.real = None                # the built library: its launch-plan queries are host code
        for name, (res, args) in signatures.items():
            setattr(self, name, self._make(name, res, args))

    def _make(self, name, res, argtypes=()):
        def fn(*args):
            # byte counts (size_t arguments) are not geometry: leave them out so that the columns of a call are always the same
            args = [a for a, t in zip(args, list(argtypes) + [None] * len(args)) if t is not ctypes.c_size_t]
            ints = [a if isinstance(a, int) else getattr(a, 'value', None) for a in args]
            self.calls.append((name, [v for v in ints if isinstance(v, int) and 0 <= v < (1 << 24)]))
            if name == 'semseg_winograd_tiles':          # return values the caller's later calls depend on
                n, h, w, d = ints[:4]
                return n * d * d * (-(-(-(-h // d)) // 2)) * (-(-(-(-w // d)) // 2))
            if name == 'semseg_conv2d_wgrad_tile_h2':    # tile 1 = a small weight gradient, batched after backward (ops.flush_wgrad_reduces)
                return self.real.semseg_conv2d_wgrad_tile_h2(*ints[:10]) if self.real is not None else 0
            return 1 << 20 if res is ctypes.c_size_t else 0
        return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', type=int, default=1)
    ap.add_argument('--grep', default='conv2d,winograd')
    args = ap.parse_args()
    from mit_semseg import _native, ops, tuner
    rec = Recorder(_native.SIGNATURES)
    try:
        rec.real = _native.lib()         # which weight gradients are batched is decided by the library's plan of the geometry
    except Exception:
        rec.real = None
    _native.lib = lambda: rec
    ops._require_cuda = lambda *a: None
    ops._st = lambda: ctypes.c_void_p(0)
    ops._WS = {}
    tuner.ENABLED = False
    # nothing is timed here, but WHICH launch form runs (Winograd data gradient: three launches or the fused kernel) is a measured
    # choice: take it from the shipped performance database, as the step on the device does
    import json
    if tuner.USE_PERFDB and os.path.exists(tuner.PERFDB):
        for k, v in json.load(open(tuner.PERFDB)).items():
            parts = k.split(',')
            if not k.startswith('_') and parts[1] == '4':
                tuner._done[(parts[0],) + tuple(int(t) for t in parts[1:])] = tuple(v)
            if not k.startswith('_') and parts[0] == 'h2' and parts[1] == '2' and int(v[0]) >= 0 and rec.real is not None:
                rec.real.semseg_conv2d_h2_set_plan(2, *[int(t) for t in parts[2:12]], int(v[0]), int(v[1]))
    import bench
    from mit_semseg.engine import TrainStep
    cfg = bench.CONFIGS[args.config]
    sm = bench.build_model(torch.device('cpu'), cfg)
    feed = bench.synth_feed(torch.device('cpu'), 0, cfg)
    step = TrainStep(sm, lr_encoder=0.02, lr_decoder=0.02, max_iters=10 ** 5, graph=False)
    step.step(feed)
    rec.calls.clear()
    step.step(feed)
    keys = [k for k in args.grep.split(',') if k]
    for name, ints in rec.calls:
        if not keys or any(k in name for k in keys):
            print('%-36s %s' % (name.replace('semseg_', ''), ' '.join(str(v) for v in ints)))


if __name__ == '__main__':
    main()
