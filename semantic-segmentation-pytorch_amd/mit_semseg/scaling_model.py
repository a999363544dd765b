"""Analytic model of the data-parallel training step at 2 / 4 / 8 GPUs of one node, fed with what ONE GPU can measure
("2/4/8-GPU throughput only where >= 2 GPUs are visible -- otherwise report measured 1-GPU + analytic comm
model and say so").  MODEL, NOT A MEASUREMENT: every number it returns is labelled so in bench.py's line.

What it prices (reference: train.py:42 `loss.mean()` over replicas = gradient average; lib/nn/modules/batchnorm.py:98-117 the
SyncBN master/slave exchange, one per BN layer forward and backward):

 * gradient all-reduce (parallel.GradientBuckets): the buckets of `plan_bucket_groups` in backward order; bucket i becomes
   ready at the time the replayed single-GPU step marks it (TimelineProbe: a timestamp kernel in the post-accumulate hook of the
   bucket's last parameter, captured into the step's hipGraph), travels on the side stream one bucket at a time as a ring
   all-reduce  t(S, N) = launch + 2 (N-1) hop + 2 (N-1)/N * S / (eff * link)  -- xGMI is point to point, a ring moves at ONE
   link's rate -- and whatever is still in flight when backward ends is exposed (the join before the SGD kernel);
 * the kernels of backward slow down while RCCL's copy kernels share the CUs and HBM (`overlap_slowdown` of the overlapped time);
 * SyncBN: one exchange per BN layer and direction, priced per transport (the one-node peer exchange inside the fused BN finish
   kernels, csrc/peer.hip; or one small RCCL all-reduce each);
 * the segmented executor's graph launches between the collectives (engine.SegmentedStep).

The assumptions are arguments with defaults and are echoed in the result, so the line shows what was assumed."""
import ctypes

import torch

from .parallel import plan_bucket_groups

ASSUMPTIONS = {
    'xgmi_link_GBps': 153.0,            # per direction per link (7 links per GPU, one per peer)
    'ring_efficiency': 0.75,            # fraction of the link rate a RCCL ring sustains on large messages
    'allreduce_launch_us': 12.0,        # host enqueue + kernel start of one ncclAllReduce on the side stream
    'ring_hop_us': 2.5,                 # per ring step (2 (N-1) of them): xGMI store + flag visibility
    'overlap_slowdown': 0.04,           # compute kernels run this much slower while an all-reduce shares CUs / HBM
    'syncbn_peer_us': 2.5,              # extra latency of one in-kernel peer exchange over its single-GPU form (xGMI round trip)
    'syncbn_peer_skew_us': 0.5,         # + this much per additional peer (arrival skew of the ranks)
    'syncbn_rccl_us': 22.0,             # one small ncclAllReduce on the compute stream (launch + log-depth protocol)
    'segment_launch_us': 14.0,          # host cost of one hipGraph segment launch (profiles/r3c_segmented_probe.txt, median)
}


class TimelineProbe:
    """Markers of a (captured) training step: mark(name) enqueues semseg_probe_timestamp on the current stream; bucket marks come
    from post-accumulate hooks on the last parameter of every gradient bucket.  read() -> {name: tick} of the last replay."""

    def __init__(self, params, bucket_bytes=64 << 20, tail_bytes=None, device=None):
        from . import _native
        self.L = _native.lib()
        self.params = [p for p in reversed(list(params)) if p.requires_grad]
        self.groups = plan_bucket_groups(self.params, bucket_bytes, tail_bytes)
        self.bucket_bytes = [4 * sum(p.numel() for p in g) for g in self.groups]
        dev = device or self.params[0].device
        self.names = ['step_begin', 'fwd_end'] + ['bucket%d' % i for i in range(len(self.groups))] + ['bwd_end', 'step_end']
        self.slots = torch.zeros(len(self.names), dtype=torch.int64, device=dev)
        self._index = {n: i for i, n in enumerate(self.names)}
        self._bucket_of, self._pending, self._hooks = {}, [], []
        for bi, g in enumerate(self.groups):
            for p in g:
                self._bucket_of[p] = bi
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self.armed = False

    def mark(self, name):
        from . import _native
        i = self._index[name]
        _native.check(self.L.semseg_probe_timestamp(ctypes.c_void_p(self.slots.data_ptr() + 8 * i),
                                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'probe_timestamp')

    def arm(self):
        self._pending = [len(g) for g in self.groups]
        self.armed = True

    def _on_grad(self, p):
        if not self.armed:
            return
        bi = self._bucket_of[p]
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            from . import ops
            if ops.deferring():              # what GradientBuckets._launch does on a rank before a bucket travels
                ops.flush_wgrad_reduces(mid_backward=True)
            self.mark('bucket%d' % bi)

    def detach(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.armed = False

    def read(self):
        t = self.slots.cpu().tolist()
        return {n: t[i] for i, n in enumerate(self.names)}


def syncbn_payloads(module):
    """[(channels)] of every SynchronizedBatchNorm2d of the module, in registration order: the forward exchange carries 2C+1
    doubles, the backward one 2C (batchnorm.py:98-117)"""
    return [m.num_features for m in module.modules() if hasattr(m, 'running_mean') and hasattr(m, 'num_features')]


def allreduce_ms(nbytes, n, a):
    if n <= 1:
        return 0.0
    bw = a['xgmi_link_GBps'] * 1e9 * a['ring_efficiency']
    return (a['allreduce_launch_us'] + 2 * (n - 1) * a['ring_hop_us']) * 1e-3 + 2.0 * (n - 1) / n * nbytes / bw * 1e3


def predict(t1_ms, marks_ms, bucket_bytes, bn_channels, n, assumptions=None, syncbn='peer', segments=None, t_ref_ms=None):
    """marks_ms: {'fwd_end', 'bucket<i>', 'bwd_end', 'step_end'} in ms from the begin of the replayed single-GPU step, scaled so
    that step_end == t1_ms.  Returns the predicted step of an n-rank job and its parts.  t1_ms is the compute of ONE RANK of the
    job; t_ref_ms (default: the same) is the single-GPU step the efficiency is quoted against -- they differ when the single-GPU
    step uses something a rank cannot (the deferred weight gradients of engine.TrainStep: gradient buckets read every gradient as soon
    as autograd has accumulated it)."""
    a = dict(ASSUMPTIONS)
    a.update(assumptions or {})
    free, busy = 0.0, 0.0
    first_ready = None
    for i, nbytes in enumerate(bucket_bytes):
        ready = marks_ms['bucket%d' % i]
        first_ready = ready if first_ready is None else first_ready
        t = allreduce_ms(nbytes, n, a)
        start = max(ready, free)
        free = start + t
        busy += t
    bwd_end = marks_ms['bwd_end']
    exposed = max(0.0, free - bwd_end)
    overlapped = max(0.0, min(busy, bwd_end - (first_ready or bwd_end)))
    slowdown = a['overlap_slowdown'] * overlapped
    nx = 2 * len(bn_channels)
    if syncbn == 'peer':
        sync = nx * (a['syncbn_peer_us'] + a['syncbn_peer_skew_us'] * (n - 2)) * 1e-3
        nseg = segments if segments is not None else len(bucket_bytes) + 2
    else:
        sync = nx * a['syncbn_rccl_us'] * 1e-3
        nseg = segments if segments is not None else nx + len(bucket_bytes) + 2
    # the segment launches are host work beside a GPU-bound step (3.3 ms of issue time against 14 ms of kernels,
    # profiles/r3c_segmented_probe.txt): reported, not on the critical path
    step = t1_ms + exposed + slowdown + sync
    return {'n': n, 'ms_per_step': round(step, 3), 'img_s': round(2.0 * n / step * 1e3, 1),
            'efficiency_vs_1gpu': round((t_ref_ms if t_ref_ms is not None else t1_ms) / step, 4),
            'exposed_allreduce_ms': round(exposed, 3), 'overlap_slowdown_ms': round(slowdown, 3), 'syncbn_ms': round(sync, 3),
            'allreduce_busy_ms': round(busy, 3), 'syncbn_transport': syncbn, 'segments': nseg,
            'host_segment_launch_ms': round(nseg * a['segment_launch_us'] * 1e-3, 3)}


def model_line(t1_ms, ticks, bucket_bytes, bn_channels, assumptions=None, t_rank_ms=None):
    """ticks: TimelineProbe.read() of one replay; scaled with the measured step -> the block bench.py prints.  t_rank_ms: the measured
    step in the form a rank of an N > 1 job runs it (the timeline's form: every weight gradient complete when autograd accumulates it);
    default t1_ms.  Predictions are built on t_rank_ms, efficiencies quoted against t1_ms."""
    t_ref = t1_ms
    if t_rank_ms is not None:
        t1_ms = t_rank_ms
    t0, t_end = ticks['step_begin'], ticks['step_end']
    span = max(1, t_end - t0)
    marks = {k: (v - t0) / span * t1_ms for k, v in ticks.items()}
    a = dict(ASSUMPTIONS)
    a.update(assumptions or {})
    out = {'kind': 'MODEL, not measured (no multi-GPU box was available to this run): single-GPU replay timeline + ring '
                   'all-reduce over one xGMI link per hop + per-exchange SyncBN cost; mit_semseg/scaling_model.py',
           'measured_1gpu_ms_per_step': round(t_ref, 3),
           'measured_rank_form_ms_per_step': round(t1_ms, 3),
           'rank_form': 'the same step with the deferred / batched weight gradients finished once per gradient bucket, by the hook '
                        'that completes the bucket (what a rank under gradient buckets runs; the single-GPU step finishes all of '
                        'them in one go after backward), timed on this GPU',
           'timeline_ms': {k: round(v, 3) for k, v in marks.items()},
           'bucket_bytes': list(bucket_bytes), 'gradient_bytes': int(sum(bucket_bytes)),
           'syncbn_exchanges_per_step': 2 * len(bn_channels),
           'syncbn_payload_doubles': {'fwd': int(sum(2 * c + 1 for c in bn_channels)), 'bwd': int(sum(2 * c for c in bn_channels))},
           'assumptions': a, 'predicted': {}}
    for n in (2, 4, 8):
        out['predicted'][str(n)] = {'peer_exchange': predict(t1_ms, marks, bucket_bytes, bn_channels, n, a, 'peer', t_ref_ms=t_ref),
                                    'rccl_syncbn': predict(t1_ms, marks, bucket_bytes, bn_channels, n, a, 'rccl', t_ref_ms=t_ref)}
    return out
