"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel path that replaces the reference's
UserScatteredDataParallel + SyncBN rendezvous: gradient buckets, the SyncBN statistics protocol with unequal
per-rank batch sizes (train.py feeds one variable-size dict per GPU), logging means, and the full TrainStep
host logic under world_size 2 (with the stub ABI of test_host_logic_dryrun -- no kernels run on CPU)."""
import ctypes
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn
import torch.nn.functional as F


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(fn, world=2):
    port = _free_port()
    mp.spawn(_entry, args=(world, port, fn), nprocs=world, join=True)


def _entry(rank, world, port, fn):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, 'semantic-segmentation-pytorch_amd')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        globals()[fn](rank, world)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------
def _w_gradient_buckets(rank, world):
    from mit_semseg.parallel import GradientBuckets
    torch.manual_seed(0)
    conv_w = nn.Parameter(torch.empty(8, 3, 3, 4).permute(0, 3, 1, 2))          # KRSC memory like models.layers.Conv2d
    ps = [nn.Parameter(torch.zeros(7)), conv_w, nn.Parameter(torch.zeros(5, 6)), nn.Parameter(torch.zeros(()))]
    gb = GradientBuckets(ps, bucket_bytes=4 * 100)                               # forces several buckets
    assert len(gb.buckets) >= 2 and sum(b['flat'].numel() for b in gb.buckets) == sum(p.numel() for p in ps)
    for i, p in enumerate(ps):
        g = torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) * (rank + 1) + i
        p.grad = g.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else g
    ps[2].grad = None                                                            # a parameter without gradient this step
    gb.prepare()
    gb.finish()
    tot = sum(r + 1 for r in range(world))
    for i, p in enumerate(ps):
        if i == 2:
            continue
        want = torch.arange(p.numel(), dtype=torch.float32).reshape(p.shape) * tot + i * world
        assert torch.equal(p.grad, want), (i, rank)
        assert p.grad.stride() == p.stride()
        flat = [b for b in gb.buckets if any(q is p for q, _, _ in b['items'])][0]['flat']
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        assert lo <= p.grad.data_ptr() < hi                                      # grad is a view of the reduced bucket


def test_gradient_buckets_allreduce():
    _run('_w_gradient_buckets')


def _w_gradient_buckets_overlap(rank, world):
    """hooks launch a bucket's all-reduce as soon as backward has produced its last gradient"""
    from mit_semseg.parallel import GradientBuckets
    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(6, 8), nn.ReLU(), nn.Linear(8, 8), nn.ReLU(), nn.Linear(8, 3))
    gb = GradientBuckets(net.parameters(), bucket_bytes=4 * 60)
    assert len(gb.buckets) >= 3
    launched = []
    orig = gb._launch
    gb._launch = lambda b: (launched.append(len(launched)), orig(b))[1]
    x = torch.full((4, 6), float(rank + 1))
    for _ in range(2):                                  # two steps: counters must re-arm
        for p in net.parameters():
            p.grad = None
        launched.clear()
        gb.prepare()
        net(x).sum().backward()
        assert len(launched) == len(gb.buckets)         # all launched by the hooks, during backward
        gb.finish()
        ref = [torch.zeros_like(p) for p in net.parameters()]
        for r in range(world):
            for p in net.parameters():
                p_grad = torch.autograd.grad(net(torch.full((4, 6), float(r + 1))).sum(), p)[0]
                ref[[id(q) for q in net.parameters()].index(id(p))] += p_grad
        for p, want in zip(net.parameters(), ref):
            torch.testing.assert_close(p.grad, want, atol=1e-5, rtol=1e-5)


def test_gradient_buckets_overlap_hooks():
    _run('_w_gradient_buckets_overlap')


def _w_syncbn_protocol(rank, world):
    """[sum, sum^2, n] all-reduce -> every rank finalises the statistics of the CONCATENATED batch, also when the
    ranks hold different numbers of pixels (reference sums python ints, batchnorm.py:106)."""
    from mit_semseg import ops
    ops.set_sync_bn_group(None, enabled=True)
    g = torch.Generator().manual_seed(1)
    shards = [torch.randn(2, 6, 5 + 3 * r, 4, generator=g) * (1 + r) + 0.3 for r in range(world)]
    x = shards[rank].double()
    c = x.shape[1]
    stats = torch.cat([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3)), torch.tensor([x.numel() / c], dtype=torch.float64)])
    ops._maybe_allreduce(stats)
    n = stats[2 * c]
    mean = stats[:c] / n
    var = stats[c:2 * c] / n - mean * mean
    full = torch.cat([s.permute(1, 0, 2, 3).reshape(c, -1) for s in shards], 1).double()
    torch.testing.assert_close(mean, full.mean(1), atol=1e-12, rtol=1e-12)
    torch.testing.assert_close(var, full.var(1, unbiased=False), atol=1e-12, rtol=1e-12)
    assert n.item() == full.shape[1]
    # normalising the local shard with the global statistics == F.batch_norm over the concatenation
    y = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)
    rm, rv = torch.zeros(c, dtype=torch.float64), torch.ones(c, dtype=torch.float64)
    off = sum(s.shape[0] * s.shape[2] * s.shape[3] for s in shards[:rank])
    ref = F.batch_norm(full.t().reshape(1, -1, c).permute(0, 2, 1).contiguous(), rm, rv, None, None, True, 0.1, 1e-5)
    mine = y.permute(1, 0, 2, 3).reshape(c, -1)
    torch.testing.assert_close(mine, ref[0][:, off:off + mine.shape[1]], atol=1e-9, rtol=1e-9)
    ops.set_sync_bn_group(None, enabled=False)


def test_syncbn_statistics_protocol_unequal_shards():
    _run('_w_syncbn_protocol')


def _w_scatter_and_means(rank, world):
    from mit_semseg.parallel import NativeDataParallel, mean_over_ranks

    class Probe(nn.Module):
        def __init__(self):
            super().__init__()
            self.p = nn.Parameter(torch.zeros(1))

        def forward(self, d, **kw):
            return d['v'] + self.p

    dp = NativeDataParallel(Probe(), device_ids=list(range(world)))
    batch = [{'v': torch.tensor([10.0 * r])} for r in range(world)]          # train.py:170-177 list of per-GPU dicts
    out = dp(batch)
    assert out.item() == 10.0 * rank
    loss, acc = mean_over_ranks(torch.tensor(float(rank)), torch.tensor(1.0))
    assert abs(loss.item() - (world - 1) / 2) < 1e-12 and acc.item() == 1.0


def test_scatter_and_logging_means():
    _run('_w_scatter_and_means')


class _StubLib:
    """signature-checking stub of the C ABI (no arithmetic), see tests/test_host_logic_dryrun.py"""

    def __init__(self, signatures):
        self.calls = []
        for name, (res, args) in signatures.items():
            setattr(self, name, self._make(name, res, args))

    def _make(self, name, res, argtypes):
        def fn(*args):
            assert len(args) == len(argtypes), name
            for a, t in zip(args, argtypes):
                t.from_param(a)
            self.calls.append((name, args))
            return 1 << 20 if res is ctypes.c_size_t else 0
        return fn


def _w_trainstep_world2(rank, world):
    from mit_semseg import _native, ops
    from mit_semseg.models import ModelBuilder, SegmentationModule
    from mit_semseg.models import resnet
    from mit_semseg.models.models import ResnetDilated
    from mit_semseg.parallel import NativeDataParallel
    from mit_semseg.engine import TrainStep
    lib = _StubLib(_native.SIGNATURES)
    _native.lib = lambda: lib
    ops._require_cuda = lambda *a: None
    ops._st = lambda: ctypes.c_void_p(0)
    torch.manual_seed(0)
    enc = ResnetDilated(resnet.resnet18(pretrained=False), 8)
    dec = ModelBuilder.build_decoder('ppm_deepsup', fc_dim=512, num_class=150)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), 0.4).train()
    NativeDataParallel(sm)                                                    # switches SyncBN all-reduce on
    assert ops._SYNC_GROUP['enabled']
    size = 64 + 32 * rank                                                     # ranks see different image sizes
    feed = {'img_data': torch.randn(2, 3, size, size), 'seg_label': torch.randint(-1, 150, (2, size // 8, size // 8))}
    ts = TrainStep(sm, max_iters=100, bucket_bytes=8 << 20)
    assert ts.buckets is not None and len(ts.buckets.buckets) > 1
    ts.step(feed)
    ts.step(feed)
    # every gradient now lives in a reduced bucket; the fused SGD averaged with 1/world
    flats = [(b['flat'].data_ptr(), b['flat'].data_ptr() + b['flat'].numel() * 4) for b in ts.buckets.buckets]
    for n, p in sm.named_parameters():
        assert any(lo <= p.grad.data_ptr() < hi for lo, hi in flats), n
    sgd = [a for name, a in lib.calls if name in ('semseg_sgd_step', 'semseg_sgd_step_fused')]
    assert sgd and all(abs(a[4] - 1.0 / world) < 1e-12 for a in sgd)
    nstats = sum(1 for name, _ in lib.calls if name in ('semseg_bn_stats', 'semseg_bn_stats_mm'))
    assert nstats > 0
    # with SyncBN active the statistics are all-reduced between the partial sums and their use: the unfused BN entry
    # points (stats -> all-reduce -> finalize, reduce -> all-reduce -> bound) must be the ones called
    names = {name for name, _ in lib.calls}
    assert {'semseg_bn_finalize_mm', 'semseg_bn_bwd_reduce_mm', 'semseg_bn_bwd_bound'} <= names
    assert not ({'semseg_bn_fwd_stats_fused', 'semseg_bn_bwd_reduce_fused'} & names)


def test_trainstep_host_logic_world2():
    _run('_w_trainstep_world2')


# ---------------------------------------------------------------------------------------------------------
def _w_sharded_evaluate(rank, world):
    """engine.evaluate over per-rank shards (eval_multipro.py:148-160 start_idx / end_idx) == one process over the whole list"""
    import numpy as np
    from mit_semseg import engine, utils
    from oracle import metrics_oracle as M
    C = 5

    def cpu_metrics(scores, label=None, tally=None):
        pred = M.argmax_first(scores.numpy()[0])[None]
        if tally is None:
            tally = utils.MetricTally(C, 'cpu')
        lab = label.numpy()
        valid = lab >= 0
        tally.counts[0] += int(((pred[0] == lab) & valid).sum())
        tally.counts[1] += int(valid.sum())
        for c in range(C):
            tally.counts[2 + c] += int(((pred[0] == c) & (lab == c)).sum())
            tally.counts[2 + C + c] += int(((pred[0] == c) & valid).sum())
            tally.counts[2 + 2 * C + c] += int((lab == c).sum())
        return torch.from_numpy(pred), tally
    utils.segmentation_metrics = cpu_metrics

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(1))

        def forward(self, feed, segSize=None):
            g = torch.Generator().manual_seed(int(feed['img_data'].abs().sum().item() * 1000) % 100003)
            return torch.softmax(torch.randn(1, C, segSize[0], segSize[1], generator=g) * 2, dim=1)
    rng = np.random.default_rng(0)
    items = [{'img_data': [torch.from_numpy(rng.standard_normal((1, 3, 8, 8)).astype(np.float32)) for _ in range(2)],
              'seg_label': torch.from_numpy(rng.integers(-1, C, (1, 6 + k, 7)))} for k in range(5)]
    whole = engine.evaluate(Fake(), items, C, device='cpu', use_graph=False, reduce=False)
    n = len(items)
    lo, hi = rank * n // world, (rank + 1) * n // world
    mine = engine.evaluate(Fake(), items[lo:hi], C, device='cpu', use_graph=False)
    assert torch.equal(mine[3].counts, whole[3].counts)
    assert mine[0] == whole[0] and mine[2] == whole[2]


def test_sharded_evaluate_equals_single_process():
    _run('_w_sharded_evaluate')


def test_bench_ddp_graph_selftest_fails_closed(monkeypatch):
    """bench.ddp_graph_selftest: a child that cannot finish (here: its peer never shows up) is killed at the time limit and
    no stage counts as reached -- the data-parallel run then uses torch.distributed collectives and eager launches"""
    import time
    import bench
    for k, v in dict(RANK='0', WORLD_SIZE='2', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29713',
                     TORCHELASTIC_USE_AGENT_STORE='True').items():
        monkeypatch.setenv(k, v)
    t = time.time()
    assert bench.ddp_graph_selftest(timeout_s=6) == {'comm': False, 'peer': False, 'segmented': False, 'graph': False}
    assert time.time() - t < 30


def test_gradient_buckets_small_tail():
    """The last bucket (the first layers of the network: complete only when backward is) is kept small, every parameter is in
    exactly one bucket, and the buckets follow the reverse registration order."""
    import torch
    from mit_semseg.parallel import GradientBuckets
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (10, 20, 30, 400, 50, 60, 700, 80)]       # registration order: first layer first
    gb = GradientBuckets(ps, bucket_bytes=4 * 1000, overlap=False)                                # tail = 1/16 of a bucket = 62 elements
    sizes = [b['flat'].numel() for b in gb.buckets]
    order = [p for b in gb.buckets for p, _, _ in b['items']]
    assert order == list(reversed(ps)) and sum(sizes) == sum(p.numel() for p in ps)
    assert sizes[-1] <= 62 and sizes[-1] == 10 + 20 + 30 and len(gb.buckets) == 3, sizes
    # one parameter alone, or a tail that is small already: nothing to split
    assert len(GradientBuckets(ps[:1], bucket_bytes=4 * 1000, overlap=False).buckets) == 1
    gb2 = GradientBuckets(ps, bucket_bytes=4 * 1000, overlap=False, tail_bytes=4 * 10 ** 6)
    assert [b['flat'].numel() for b in gb2.buckets] == [80 + 700 + 60 + 50, 400 + 30 + 20 + 10]
    for b in gb.buckets:                       # offsets are contiguous inside a bucket
        off = 0
        for p, o, n in b['items']:
            assert o == off and n == p.numel()
            off += n


def _bench_cmd(*args, **env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + list(args), env=e, capture_output=True, text=True,
                          timeout=300)


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (train.py:184-190 drives all GPUs from one
    command) and the line says n_gpus 2; SEMSEG_BENCH_LAUNCH_PROBE=1 stops every rank after the rendezvous (no GPU here)."""
    import json
    r = _bench_cmd('--gpus', '2', '--steps', '1', '--warmup', '0', SEMSEG_BENCH_LAUNCH_PROBE='1')
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['ranks_counted'] == 2 and line['gpus_arg'] == 2 and line['self_launched'] is True


def test_bench_refuses_a_mislabelled_world():
    """a launcher that started a different number of ranks than --gpus says, or a node with fewer GPUs than --gpus: no JSON
    line, exit code 2 (never a line whose n_gpus differs from --gpus)"""
    r = _bench_cmd('--gpus', '8', SEMSEG_BENCH_LAUNCH_PROBE='1', WORLD_SIZE='1', RANK='0')
    assert r.returncode == 2 and 'refusing' in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    r = _bench_cmd('--gpus', '8')                    # this container has no GPU: 8 ranks cannot each have one
    assert r.returncode == 2 and 'refusing' in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_bench_roofline_constants_come_from_the_pmc_index():
    """roofline.traffic / clock are looked up in profiles/pmc_index.json by the plan the tuner picked; an unknown plan gives None"""
    import bench
    geom = (2, 64, 64, 4096, 512, 3, 3, 1, 1, 1)
    e = bench.pmc_lookup('h2', 1, geom, (14, 1))
    assert e is not None and os.path.exists(os.path.join(bench.ROOT, e['source'])) and e['fetch_kib'] > 0
    assert bench.pmc_lookup('h2', 1, geom, (3, 2)) is None and bench.pmc_lookup('h2', 1, geom, None) is None
    # dy planes 16.8 MB + weight planes 75.5 MB + fp32 dx 134.2 MB
    assert abs(bench.algorithmic_bytes(1, geom, 'h2') - 226.5e6) < 0.1e6
