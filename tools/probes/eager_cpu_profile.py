"""cProfile of the eager (no hipGraph) training step on the GPU box: where the HOST time of the ~800 launches per step goes
(the multi-GPU path launches eagerly, so its step time is bounded below by this)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.build()
from mit_semseg.engine import TrainStep  # noqa: E402

dev = torch.device('cuda:0')
sm = bench.build_model(dev)
feed = bench.synth_feed(dev, 0)
ts = TrainStep(sm, max_iters=100000, graph=False)
for _ in range(4):
    ts.step(feed)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    ts.step(feed)
host = time.perf_counter() - t
torch.cuda.synchronize()
total = time.perf_counter() - t
print('10 eager steps: host-side issue time %.2f ms/step, wall (incl. GPU drain) %.2f ms/step' % (host * 100, total * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    ts.step(feed)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
