"""Host-side cost of ONE eager training step with the C ABI stubbed out (no GPU needed): what the Python glue of
mit_semseg costs per step when every native call returns immediately -- the floor of the eager (multi-GPU) step.
    python tools/probes/host_overhead_stub.py [--profile]
"""
import cProfile
import ctypes
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

from mit_semseg import _native, ops  # noqa: E402
from mit_semseg.models import ModelBuilder, SegmentationModule  # noqa: E402
from mit_semseg.engine import TrainStep  # noqa: E402


class StubLib:
    def __init__(self, signatures):
        self.n = 0
        for name, (res, _) in signatures.items():
            setattr(self, name, self._make(res))

    def _make(self, res):
        rv = (1 << 16) if res is ctypes.c_size_t else 0

        def fn(*args):
            self.n += 1
            return rv
        return fn


def main():
    lib = StubLib(_native.SIGNATURES)
    lib.semseg_winograd_tiles = lambda n, h, w, d: n * d * d * ((-(-h // d) + 1) // 2) * ((-(-w // d) + 1) // 2)
    _native.lib = lambda: lib
    ops._require_cuda = lambda *a: None
    ops._st = lambda: ctypes.c_void_p(0)
    torch.manual_seed(0)
    from mit_semseg.models import resnet
    from mit_semseg.models.models import ResnetDilated
    enc = ResnetDilated(resnet.resnet50(pretrained=False), dilate_scale=8)
    dec = ModelBuilder.build_decoder('ppm_deepsup', fc_dim=2048, num_class=150)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), 0.4).train()
    feed = {'img_data': torch.randn(2, 3, 64, 64), 'seg_label': torch.randint(-1, 150, (2, 8, 8))}
    ts = TrainStep(sm, max_iters=1000)
    for _ in range(3):
        ts.step(feed)
    lib.n = 0
    t = time.perf_counter()
    K = 10
    for _ in range(K):
        ts.step(feed)
    dt = (time.perf_counter() - t) / K
    print('host time per eager step (stubbed ABI, CPU tensors): %.2f ms, %d native calls per step' % (dt * 1e3, lib.n // K))
    if '--profile' in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            ts.step(feed)
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(28)


if __name__ == '__main__':
    main()
