"""Self-test of the data-parallel step, one process per GPU (launched by bench.py on every rank in a child process before the real
run, or by hand under torchrun): a small model (ResNet18dilated + PPM_deepsup, 2 x 64 x 64 per rank, every rank on different data)
goes through exactly the production code paths, in stages, and prints a marker per stage it passed:
  COMM_OK       the C ABI's own RCCL communicators came up and summed correctly (NativeDataParallel -> comm.init),
  PEER_OK       the one-node peer exchange of the SyncBN sums came up on every rank and summed correctly (comm.peer_init),
  SEGMENTED_OK  the segmented hipGraph executor trained 40 steps, no exchange timed out, the replicas are bit-identical (they could
                not be if any exchange or all-reduce were dropped or stale),
  GRAPH_OK      the whole step incl. the RCCL all-reduces replayed as ONE hipGraph with identical replicas -- only tried when the
                peer exchange is NOT up (GRAPH_SKIPPED otherwise: the real run then stays segmented, see bench.py).
bench.py enables for the real run what every rank's child reached; an exception, or a hang that the parent kills at its time limit,
leaves the later stages off."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'semantic-segmentation-pytorch_amd'))
os.environ['SEMSEG_DDP_GRAPH'] = '1'

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    from mit_semseg.models import ModelBuilder, SegmentationModule, resnet
    from mit_semseg.models.models import ResnetDilated
    from mit_semseg.parallel import init_distributed, NativeDataParallel
    from mit_semseg.engine import TrainStep
    rank, world, local = init_distributed()
    assert world > 1, 'run with one process per GPU (WORLD_SIZE > 1)'
    if os.environ.get('SEMSEG_BENCH_DEVICE'):          # several ranks on one GPU (gloo transport): functional check of this script
        local = int(os.environ['SEMSEG_BENCH_DEVICE'])
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)
    torch.manual_seed(304)
    enc = ResnetDilated(resnet.resnet18(pretrained=False), dilate_scale=8)
    dec = ModelBuilder.build_decoder('ppm_deepsup', fc_dim=512, num_class=150)
    sm = SegmentationModule(enc, dec, nn.NLLLoss(ignore_index=-1), 0.4).to(dev).train()
    dp = NativeDataParallel(sm)                         # brings up the C ABI's own RCCL communicator (self-tested sums)
    # every rank: its own parent reads its own child's log
    print('COMM_%s native RCCL communicator through the C ABI: %s' % ('OK' if dp.native_comm else 'OFF', dp.native_comm), flush=True)
    # peer_init is unanimous by construction (every rank mapped every inbox and summed correctly, or nobody uses it)
    print('PEER_%s one-node peer exchange of the SyncBN payloads (csrc/peer.hip): %s' % ('OK' if dp.peer_exchange else 'OFF', dp.peer_exchange),
          flush=True)

    def replicas_identical():
        sums = torch.stack([p.detach().double().abs().sum() for p in sm.parameters()] +
                           [b.detach().double().abs().sum() for n, b in sm.named_buffers() if n.endswith('running_var')])
        hi, lo = sums.clone(), sums.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        assert torch.equal(hi, lo), 'replicas diverged: max |hi - lo| = %g' % (hi - lo).abs().max().item()

    g = torch.Generator().manual_seed(1000 + rank)
    feed = {'img_data': torch.randn(2, 3, 64, 64, generator=g).to(dev),
            'seg_label': torch.randint(-1, 150, (2, 8, 8), generator=g).to(dev)}
    # stage 1: the default data-parallel launch mode (SegmentedStep: hipGraph segments, collectives between the replays)
    os.environ['SEMSEG_DDP_GRAPH'] = '0'
    ts0 = TrainStep(sm, max_iters=1000, graph=True, bucket_bytes=8 << 20)
    for _ in range(40):          # ~3 000 SyncBN exchanges and ~200 bucket all-reduces through the replayed segments
        loss, acc = ts0.step(feed)
    torch.cuda.synchronize()
    assert ts0.launch_mode() == 'segmented' and ts0.stats['replayed'] >= 2 and torch.isfinite(loss).item()
    from mit_semseg import comm
    comm.peer_check()
    replicas_identical()        # every rank trains on different data: equal replicas <=> every exchange delivered every payload
    dist.barrier()
    print('SEGMENTED_OK loss %.5f' % loss.item(), flush=True)
    if dp.peer_exchange and os.environ.get('SEMSEG_SELFTEST_GRAPH', '0') != '1':
        # with the peer exchange up the real run stays segmented (a hipGraph with the buckets' side stream inside it is submitted
        # node by node): the whole-step graph is not needed, so it is not risked either
        print('GRAPH_SKIPPED the segmented executor with the peer exchange is the preferred mode', flush=True)
        comm.peer_destroy()
        dist.destroy_process_group()
        return
    # stage 2: RCCL captured inside ONE hipGraph
    os.environ['SEMSEG_DDP_GRAPH'] = '1'
    ts = TrainStep(sm, max_iters=1000, graph=True, bucket_bytes=8 << 20)
    assert ts.buckets is not None and len(ts.buckets.buckets) > 1
    loss = None
    for _ in range(6):
        loss, acc = ts.step(feed)
    torch.cuda.synchronize()
    assert ts._graph is not None, 'the step did not run as a graph'
    assert torch.isfinite(loss).item(), 'loss is not finite'
    comm.peer_check()
    replicas_identical()
    # bucket-sized message through a graph as well (the real run reduces 64 MiB gradient buckets)
    big = torch.ones(16 << 20, device=dev, dtype=torch.float32)
    dist.all_reduce(big)
    torch.cuda.synchronize()
    gg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gg):
        dist.all_reduce(big)
    for _ in range(2):
        gg.replay()
    torch.cuda.synchronize()
    want = float(world) ** 3                     # 1 -> eager reduce -> two replays (the capture itself executes nothing)
    assert float(big[0]) == want and float(big[-1]) == want, (float(big[0]), want)
    ref = torch.tensor([loss.item()], device=dev, dtype=torch.float64)
    dist.all_reduce(ref)                          # also proves an eager collective still works after the replays
    dist.barrier()
    print('GRAPH_OK ddp graph selftest ok: world %d, loss %.5f' % (world, loss.item()), flush=True)
    comm.peer_destroy()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
