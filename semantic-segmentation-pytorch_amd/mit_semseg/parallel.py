"""One-process-per-GPU data parallelism over RCCL/xGMI -- the MI355X replacement of the reference's
single-process `UserScatteredDataParallel` + thread-rendezvous SyncBN (lib/nn/parallel/data_parallel.py,
lib/nn/modules/{batchnorm,comm,replicate}.py, train.py:184-190).

 * weights are resident on every rank (no per-iteration parameter broadcast, reference replicate());
 * SyncBN statistics: ops.set_sync_bn_group -> all-reduce of [sum, sum^2, n] (fwd) and
   [sum dy, sum dy*xhat] (bwd), 2C(+1) fp64 per BN layer, on the compute stream;
 * gradients: flat fp32 buckets filled in reverse parameter order as backward produces them and
   all-reduced (sum, then /world inside the fused SGD via grad_scale) on a side HIP stream so the
   transfers overlap the rest of backward; xGMI is point-to-point, so buckets are large (default 64 MiB)
   to stay bandwidth- rather than latency-bound on the ring.
 * loss/acc logging mean = mean of per-rank means (train.py:42-43).

Works with any torch.distributed backend: `nccl` (= RCCL) on GPUs, `gloo` in the CPU unit tests.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops


def init_distributed(backend=None):
    """Initialise torch.distributed from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    if not dist.is_initialized():
        if backend is None:
            # SEMSEG_DIST_BACKEND=gloo: several ranks on ONE GPU (RCCL refuses that) -- how the single-GPU test box runs the
            # multi-rank code path end to end
            backend = os.environ.get('SEMSEG_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def plan_bucket_groups(params_in_backward_order, bucket_bytes=64 << 20, tail_bytes=None):
    """parameters (already in the order backward produces their gradients) -> list of buckets (lists of parameters).
    Greedy fill up to `bucket_bytes`; the LAST bucket (the first layers of the network) is complete only when backward is, so
    its all-reduce overlaps with nothing: it is kept small (`tail_bytes`, default 1/16 of a bucket) -- what backward produces in
    its last fraction of a millisecond -- and everything before it travels while backward is still running.  Shared by
    GradientBuckets and the analytic scaling model (mit_semseg/scaling_model.py), which prices exactly these buckets."""
    groups, cur, cur_n = [], [], 0
    cap = max(1, bucket_bytes // 4)
    for p in params_in_backward_order:
        n = p.numel()
        if cur and cur_n + n > cap:
            groups.append(cur)
            cur, cur_n = [], 0
        cur.append(p)
        cur_n += n
    if cur:
        groups.append(cur)
    tail = max(1, (bucket_bytes // 16 if tail_bytes is None else tail_bytes) // 4)
    if groups and len(groups[-1]) > 1 and sum(p.numel() for p in groups[-1]) > tail:
        last, keep, n = groups.pop(), [], 0
        while len(last) > 1 and n + last[-1].numel() <= tail:
            n += last[-1].numel()
            keep.insert(0, last.pop())
        groups.append(last)
        if keep:
            groups.append(keep)
    return groups


class GradientBuckets:
    """Flat gradient buckets + all-reduce overlapped with backward.

    Parameters are taken in REVERSE registration order (the order backward produces gradients) and packed
    into flat fp32 buckets.  A post-accumulate-grad hook on every parameter counts its bucket down; when the
    last gradient of a bucket has been produced the bucket is staged (gradients copied into their slices,
    `p.grad` re-pointed at the slice) and its all-reduce is launched on `comm_stream` behind an event recorded
    on the compute stream -- so the transfer runs while backward continues on the earlier layers.
    `finish()` launches whatever is left (parameters that received no gradient), joins the streams and leaves
    every `p.grad` as a VIEW of a reduced bucket (sum over ranks; the 1/world factor is folded into the SGD
    kernel's grad_scale)."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None, comm_stream=None, overlap=True, tail_bytes=None):
        self.group = group
        self.params = [p for p in reversed(list(params)) if p.requires_grad]
        self.buckets = []          # dict(flat, items=[(param, offset, numel)], pending, launched)
        groups = plan_bucket_groups(self.params, bucket_bytes, tail_bytes)
        for grp in groups:
            items, off = [], 0
            for p in grp:
                items.append((p, off, p.numel()))
                off += p.numel()
            self.buckets.append(self._make_bucket(items, off))
        self.comm_stream = comm_stream
        self._works = []
        self._bucket_of = {}
        self._hooks = []
        self.armed = False
        # weight gradients deferred by ops.defer_wgrad_reduces are completed by the hook that completes their bucket (_on_grad),
        # which needs the hooks; without them (overlap=False / an old torch) every gradient is finished where it is produced
        self.flushes_deferred = False
        for bi, b in enumerate(self.buckets):
            for p, _, _ in b['items']:
                self._bucket_of[p] = bi
        if overlap and hasattr(torch.Tensor, 'register_post_accumulate_grad_hook'):
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            self.flushes_deferred = os.environ.get('SEMSEG_DEFER_WGRAD_BUCKETS', '1') != '0'

    def _make_bucket(self, items, total):
        dev = items[0][0].device
        return dict(flat=torch.zeros(total, device=dev, dtype=torch.float32), items=list(items),
                    pending=len(items), launched=False)

    def bucket_sizes(self):
        return [b['flat'].numel() * 4 for b in self.buckets]

    def prepare(self):
        """Call before backward: re-arm the per-bucket counters."""
        for b in self.buckets:
            b['pending'] = len(b['items'])
            b['launched'] = False
        self.armed = True

    def _on_grad(self, p):
        if not self.armed:
            return
        b = self.buckets[self._bucket_of[p]]
        b['pending'] -= 1
        if b['pending'] == 0 and not b['launched']:
            self._launch(b)

    def _stage(self, b):
        """copy p.grad into the bucket slices (physical order), re-point p.grad at the slice.  One multi-tensor copy per
        bucket (torch._foreach_copy_) instead of one copy launch per parameter -- the data-parallel step launches eagerly and
        is host-bound, and a ResNet-50 has 161 parameters -- and the slice views are built once per gradient layout."""
        views = b.setdefault('views', {})
        dsts, srcs, moved = [], [], []
        for p, off, n in b['items']:
            g = p.grad
            if g is None:
                b['flat'][off:off + n].zero_()
                continue
            key = (tuple(g.shape), tuple(g.stride()))
            rec = views.get(p)
            if rec is None or rec[0] != key:
                flat = b['flat'][off:off + n]
                view = flat.as_strided(g.shape, g.stride()) if g.dim() > 0 else flat.view(())
                rec = views[p] = (key, view)
            view = rec[1]
            if g.data_ptr() != view.data_ptr():
                dsts.append(view)
                srcs.append(g)
                moved.append(p)
        if dsts:
            if hasattr(torch, '_foreach_copy_'):
                torch._foreach_copy_(dsts, srcs)
            else:
                for d, g in zip(dsts, srcs):
                    d.copy_(g)
            for p, d in zip(moved, dsts):
                p.grad = d

    def reduce_async(self, flat):
        """all-reduce of one staged bucket, overlapped with whatever the compute stream does next: on `comm_stream` behind
        an event recorded on the compute stream (GPU), or as an async work object (CPU tests)"""
        if self.comm_stream is not None and flat.is_cuda:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                ops.allreduce_sum(flat, self.group, channel='bucket')
        else:
            self._works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def join(self):
        """the compute stream waits for every bucket all-reduce launched so far"""
        for w in self._works:
            w.wait()
        self._works = []
        if self.comm_stream is not None and self.buckets and self.buckets[0]['flat'].is_cuda:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def _launch(self, b):
        b['launched'] = True
        if ops.deferring() or ops._PENDING_SLABS or ops._PENDING_WGRADS:
            # every gradient of this bucket has been ACCUMULATED (the hooks counted them) but the deferred ones are still slabs /
            # not yet launched: finish them before anything reads the bucket's gradients
            ops.flush_wgrad_reduces(mid_backward=self.armed and ops.deferring())
        self._stage(b)
        if ops._SEGMENTS is not None:
            # segmented hipGraph capture (engine.SegmentedStep): the staging copies above belong to the segment being
            # captured; the all-reduce itself is issued eagerly between segment replays
            ops._SEGMENTS.collective('bucket', b['flat'], self)
            return
        if world_size(self.group) <= 1:
            return
        self.reduce_async(b['flat'])

    def all_reduce(self):
        """Launch every bucket that the hooks have not launched yet (no hooks / parameters without gradient)."""
        for b in self.buckets:
            if not b['launched']:
                self._launch(b)

    def finish(self):
        self.all_reduce()
        if ops._SEGMENTS is not None:
            ops._SEGMENTS.collective('join', None, self)
        else:
            self.join()
        self.armed = False


class NativeDataParallel(nn.Module):
    """Reference-compatible wrapper (`UserScatteredDataParallel(module, device_ids=gpus)` /
    `DataParallelWithCallback`): runs the resident module on this rank's element of the scattered batch.

    forward(batch): `batch` is the reference's list of per-GPU dicts (train.py:170-177,
    data_parallel.py:54-62) -- this rank consumes element [rank % len(batch)] -- or a single dict.
    Returns what the wrapped module returns; losses are per-rank means exactly as each DataParallel
    replica produced them (train.py:42 then averages; `mean_over_ranks` does that across processes)."""

    def __init__(self, module, device_ids=None, output_device=None, dim=0, group=None, sync_bn=True):
        super().__init__()
        self.module = module
        self.device_ids = device_ids
        self.group = group
        self.rank = dist.get_rank(group) if (dist.is_available() and dist.is_initialized()) else 0
        if sync_bn:
            ops.set_sync_bn_group(group, enabled=world_size(group) > 1)
        # collectives through the C ABI's own RCCL communicator (csrc/comm.hip) when the group runs on RCCL: one ctypes call per
        # all-reduce instead of a c10d work object.  Collective + self-tested; any failure leaves torch.distributed in charge.
        from . import comm
        self.native_comm = self.peer_exchange = False
        if world_size(group) > 1:
            import contextlib
            dev = next(module.parameters()).device       # communicators and the inbox live on the module's device
            with (torch.cuda.device(dev) if dev.type == 'cuda' else contextlib.nullcontext()):
                self.native_comm = comm.init(group)
                # SyncBN payloads: the one-node peer exchange (csrc/peer.hip) -- the sums cross xGMI inside the BN kernels (or in
                # one small kernel), inside the step's hipGraph segments; RCCL keeps the gradient buckets.  Collective +
                # self-tested as well.
                self.peer_exchange = comm.peer_init(group) if sync_bn else False

    def scatter(self, batch):
        """This rank's element of the reference's per-GPU list (data_parallel.py:54-62).  A list must hold ONE dict (this
        rank's own batch) or one dict per rank of the group.  Anything else is the single-process recipe of the reference
        (`train.py --gpus 0-7`: one process, a list of 8 per-GPU dicts) run without torchrun -- taking an element of it
        would silently train on 1/8 of every batch, so it raises."""
        if isinstance(batch, (list, tuple)):
            world = world_size(self.group)
            if len(batch) == 1:
                return batch[0]
            if len(batch) == world:
                return batch[self.rank]
            raise ValueError(
                'NativeDataParallel got a list of %d per-GPU batches but runs as rank %d of %d process(es)%s: this build is one '
                'process per GPU -- launch with `python -m torch.distributed.run --nproc-per-node N` and give every rank '
                'its own batch (a 1-element list) or the full list of N' %
                (len(batch), self.rank, world, '' if not self.device_ids else ' (device_ids=%s)' % (list(self.device_ids),)))
        return batch

    def forward(self, batch, **kwargs):
        item = self.scatter(batch)
        dev = next(self.module.parameters()).device
        if isinstance(item, dict):
            item = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in item.items()}
        return self.module(item, **kwargs)


def mean_over_ranks(*scalars, group=None):
    """train.py:42-43 `loss.mean(), acc.mean()` across replicas -> across processes (logging only)."""
    if world_size(group) <= 1:
        return scalars
    buf = torch.stack([s.detach().float().reshape(()) for s in scalars])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    buf /= world_size(group)
    return tuple(buf[i] for i in range(len(scalars)))
