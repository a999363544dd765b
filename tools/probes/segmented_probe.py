"""Where does a SegmentedStep replay spend its time?  One rank, SyncBN kernel sequence forced (SEMSEG_FORCE_SYNC_PATH=1), so the
step has its ~120 segment boundaries but no transport.  Prints ms/step of: eager, one whole hipGraph (fused single-rank path),
the segmented executor; then the host time of every graph.replay() call of one segmented step."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
import torch  # noqa: E402


def run(mode, arch, steps=20):
    import bench
    from mit_semseg import ops
    from mit_semseg.engine import TrainStep
    dev = torch.device('cuda:0')
    cfg = dict(bench.CONFIGS[1])
    if arch == 'r18':
        cfg.update(enc='resnet18', fc_dim=512)
    sm = bench.build_model(dev, cfg)
    feed = bench.synth_feed(dev, 0, cfg)
    ops._SYNC_GROUP['force'] = mode in ('segmented', 'eager_sync')
    ts = TrainStep(sm, max_iters=10 ** 6, graph=mode in ('graph', 'segmented'))
    for _ in range(5):
        ts.step(feed)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ts.step(feed)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print('%-5s %-10s %8.2f ms/step   stats %s' % (arch, mode, ms, ts.stats), flush=True)
    if mode == 'segmented':
        seg = ts._graph
        print('   items:', seg.counts())
        host = []
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for kind, obj, owner in seg.items:
            if kind == 'graph':
                t = time.perf_counter()
                obj.replay()
                host.append(time.perf_counter() - t)
        t_issue = time.perf_counter() - t_all
        torch.cuda.synchronize()
        t_done = time.perf_counter() - t_all
        host.sort()
        print('   one replay pass: issue %.2f ms, complete %.2f ms; per graph.replay() host us: min %.0f median %.0f p90 %.0f max %.0f'
              % (t_issue * 1e3, t_done * 1e3, host[0] * 1e6, host[len(host) // 2] * 1e6, host[int(len(host) * .9)] * 1e6, host[-1] * 1e6))
    ops._SYNC_GROUP['force'] = False
    del ts, sm
    torch.cuda.empty_cache()


if __name__ == '__main__':
    import __graft_entry__ as ge
    ge.build()
    for arch in ('r18', 'r50'):
        for mode in ('eager', 'eager_sync', 'graph', 'segmented'):
            run(mode, arch)
