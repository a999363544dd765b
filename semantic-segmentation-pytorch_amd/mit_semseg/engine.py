"""Training-step engine: train.py:34-48 semantics (zero_grad -> forward -> backward -> 2x SGD with the
poly LR of train.py:130-139) executed as HIP kernels, optionally captured ONCE into a hipGraph and
replayed (the whole step is capture-safe: no host sync, no allocation outside torch's graph pool, the
learning rate lives in device memory)."""
import collections
import os
import time

import torch
import torch.nn as nn

from . import ops
from .models.layers import Conv2d
from .parallel import GradientBuckets, world_size


def group_weight(module):
    """train.py:92-112: weight decay on conv weights only; biases and BN affine get none.
    Returns (decay_params, no_decay_params)."""
    decay, no_decay = [], []
    for m in module.modules():
        if isinstance(m, (Conv2d, nn.modules.conv._ConvNd, nn.Linear)):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif hasattr(m, 'running_mean') and hasattr(m, 'weight'):
            if m.weight is not None:
                no_decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
    assert len(list(module.parameters())) == len(decay) + len(no_decay)
    return decay, no_decay


class FusedSGD:
    """torch.optim.SGD(momentum=0.9, weight_decay on group 0) for both encoder and decoder groups in one
    multi-tensor HIP kernel (train.py:115-127).  lr per group is a device scalar."""

    def __init__(self, groups, momentum=0.9):
        """groups: list of dict(params=[...], lr=float, weight_decay=float)."""
        self.groups = groups
        self.momentum = momentum
        self.state = {}
        self.steps = 0
        for g in groups:
            dev = g['params'][0].device
            g['lr_t'] = torch.full((1,), float(g['lr']), device=dev, dtype=torch.float32)

    def set_lr(self, group_index, lr):
        g = self.groups[group_index]
        g['lr'] = lr
        g['lr_t'].fill_(lr)

    def zero_grad(self):
        for g in self.groups:
            for p in g['params']:
                p.grad = None

    def step(self, grad_scale=1.0):
        for g in self.groups:
            ps = [p for p in g['params'] if p.grad is not None]
            if not ps:
                continue
            bufs, first = [], []
            for p in ps:
                b = self.state.get(p)
                first.append(b is None)           # per parameter, as torch.optim.SGD: a parameter that gets its first gradient
                if b is None:                     # at a later step (unfrozen layer, conditional head) starts its buffer then
                    b = torch.empty_like(p.grad)
                    self.state[p] = b
                bufs.append(b)
            ops.sgd_step(ps, [p.grad for p in ps], bufs, first, [g['weight_decay']] * len(ps), g['lr_t'], self.momentum,
                         grad_scale)
        self.steps += 1


class SegmentedStep:
    """The data-parallel training step as a CHAIN of hipGraphs with the collectives between them.

    A step of R50dilated+PPM at world > 1 is ~800 kernel launches, 122 SyncBN all-reduces and a handful of gradient-bucket
    all-reduces; launched eagerly it is host-bound (25-30 ms of launch work against 14.6 ms of kernels).  Capturing the whole
    step, collectives included, removes the host but puts RCCL inside a hipGraph (opt-in, SEMSEG_DDP_GRAPH=1, behind bench.py's
    self-test).  This executor keeps RCCL OUT of the graphs: one capture pass runs the step with `ops._SEGMENTS = self`; every
    collective site (`ops._maybe_allreduce`, `GradientBuckets._launch` / `finish`) ends the segment being captured, records
    (kind, tensor) and begins the next segment on the same allocator pool.  A replay launches the segments in order and issues
    the recorded collectives eagerly in between -- ~125 graph launches + ~125 all-reduce calls per step, no per-kernel host
    work -- on any transport (RCCL, or gloo in the two-ranks-on-one-GPU test).  Backward runs on the calling thread during
    the capture pass (single-threaded autograd), so begin/end capture never cross threads."""

    def __init__(self):
        self.items = []           # ('graph', CUDAGraph) | ('allreduce', tensor, group) | ('bucket', flat, buckets) | ('join', None, buckets)
        self.pool = None
        self._g = None
        self.out = None

    def _begin(self):
        self._g = torch.cuda.CUDAGraph()
        self._g.capture_begin(pool=self.pool, capture_error_mode='relaxed')

    def _end(self):
        self._g.capture_end()
        self.items.append(('graph', self._g, None))
        self._g = None

    def collective(self, kind, tensor, owner):
        self._end()
        self.items.append((kind, tensor, owner))
        self._begin()

    def capture(self, run, pool):
        """`run()` executes one eager step on static buffers; returns what it returns"""
        self.pool = pool
        # no device synchronize, no gc.collect() (TrainStep._capture): recording executes nothing, the capture stream is ordered after the
        # work already issued by the event wait below
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        prev = ops._SEGMENTS
        try:
            with torch.cuda.stream(stream), torch.autograd.set_multithreading_enabled(False):
                ops._SEGMENTS = self
                self._begin()
                self.out = run()
                self._end()
        finally:
            ops._SEGMENTS = prev
            if self._g is not None:            # an exception inside a segment: close the capture before re-raising
                try:
                    self._g.capture_end()
                except Exception:
                    pass
                self._g = None
        torch.cuda.current_stream().wait_stream(stream)
        return self.out

    def replay(self):
        for kind, obj, owner in self.items:
            if kind == 'graph':
                obj.replay()
            elif kind == 'allreduce':
                ops.allreduce_sum(obj, owner)
            elif kind == 'bucket':
                if world_size(owner.group) > 1:
                    owner.reduce_async(obj)
            else:
                owner.join()
        return self.out

    def counts(self):
        c = collections.Counter(k for k, _, _ in self.items)
        return dict(c)


def feed_key(feed):
    """shape signature of a feed dict: one captured graph per signature (per-GPU batches of train.py:170-177 come in many
    H x W, dataset.py:121-142)"""
    return tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(feed.items()) if torch.is_tensor(v))


class _TimedGraph(torch.cuda.CUDAGraph):
    """a CUDAGraph that remembers how long ending the capture took (the graph is instantiated there): what a new batch shape costs
    beyond the Python pass that records it"""
    end_s = 0.0

    def capture_end(self):
        t0 = time.perf_counter()
        super().capture_end()
        self.end_s = time.perf_counter() - t0


def record_graph(graph, run, stream, pool=None):
    """record `run()` into `graph` on `stream` (not the current one) WITHOUT a device synchronize: what `with torch.cuda.graph(...)` does
    minus its torch.cuda.synchronize() + empty_cache().  Recording executes nothing, so the capture stream only has to be ordered after
    the work already issued (an event wait); the replay goes to whatever stream is current then.  Returns what run() returns."""
    cur = torch.cuda.current_stream()
    stream.wait_stream(cur)
    with torch.cuda.stream(stream):
        if pool is None:
            graph.capture_begin()
        else:
            graph.capture_begin(pool=pool)
        try:
            out = run()
        finally:
            graph.capture_end()
    cur.wait_stream(stream)
    return out


class TrainStep:
    """One training iteration of train.py:34-48 for a SegmentationModule.

    step(feed) -> (loss, acc) device scalars.  With `graph=True` the step is captured into a hipGraph PER FEED SHAPE: the
    first `warmup_eager` steps run eagerly -- that sizes the workspace, tunes the conv launch plans and warms the allocator --,
    after them a shape is recorded (fwd + bwd + all-reduce + SGD) the first time it is seen (`capture_first_sight`; its launch
    plans are inherited from the same layers at neighbouring sizes, tuner._bucket_lookup) and from then on the batch is copied
    into the graph's static buffers and replayed.  The
    variable-size per-GPU batches of the multi-scale pipeline (BASELINE configs[3]: short side 300...600, long side <= 1000,
    multiples of 8) therefore converge to replays; at most `max_graphs` graphs are kept (least recently used goes first) and
    they share ONE allocator pool, so their activation memory is the maximum over the shapes, not the sum (each graph is
    self-contained: nothing but its (loss, acc) outputs outlives a replay).  `max_graphs` defaults to 512: the ADE20K size list
    under the multi-scale rule yields 467 distinct batch shapes in 20 000 iterations (tools/shape_stream_sim.py), the five most
    frequent cover 39 % of the stream, 16 cover 63 %, 256 cover 98 % -- an LRU of 16 (rounds 2-4) re-captured on HALF of all
    steps, one of 512 never evicts.  What a graph costs beyond the shared pool is its kernel-node list and its static batch
    (a few MB), against 288 GB of HBM."""

    def __init__(self, segmentation_module, lr_encoder=0.02, lr_decoder=0.02, momentum=0.9, weight_decay=1e-4,
                 lr_pow=0.9, max_iters=100000, graph=False, group=None, bucket_bytes=64 << 20, max_graphs=None):
        self.sm = segmentation_module
        enc, dec = segmentation_module.encoder, segmentation_module.decoder
        groups = []
        for net, lr in ((enc, lr_encoder), (dec, lr_decoder)):
            decay, no_decay = group_weight(net)
            groups.append(dict(params=decay, lr=lr, weight_decay=weight_decay, base_lr=lr))
            groups.append(dict(params=no_decay, lr=lr, weight_decay=0.0, base_lr=lr))
        self.opt = FusedSGD(groups, momentum)
        self.lr_pow, self.max_iters = lr_pow, max_iters
        self.iter = 0
        self.group = group
        self.world = world_size(group)
        self.buckets = None
        if self.world > 1:
            dev = next(segmentation_module.parameters()).device
            side = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
            self.buckets = GradientBuckets(list(enc.parameters()) + list(dec.parameters()), bucket_bytes, group, side)
        # h2 path: the split planes of every conv weight are rebuilt by ONE multi-tensor launch after each optimiser
        # step (csrc/weights_prep.hip) instead of 5 small launches per conv inside forward/backward
        self._conv_weights = [m.weight for m in segmentation_module.modules() if type(m) is Conv2d]      # not the grouped ones
        self._weights_ready = False
        self.use_graph = graph
        self.max_graphs = int(os.environ.get('SEMSEG_TRAIN_GRAPHS', max_graphs or 512))
        self._graphs = collections.OrderedDict()      # feed_key -> (graph, static feed, (loss, acc))
        self._seen = {}                               # feed_key -> eager steps run at this shape
        self._provisional = {}                        # feed_key -> a first-sight graph captured without all its launch plans (see below)
        self._pool = None
        self._capture_stream = None
        self._graph = None                            # the graph of the most recent replay (bench.py reports the launch mode)
        self.warmup_eager = 2
        # a NEW batch shape is captured the first time it is seen (after the warm-up steps), not run eagerly once and captured at its
        # second sight: the Python pass that records the step costs about what the eager pass costs, so a shape that comes back (most do:
        # TrainStep docstring) pays one pass instead of two.  A capture cannot time launch plans: a geometry met inside it without a plan to
        # inherit (tuner.stats['missed_capturing']) makes the graph provisional -- replayed this once, then the old order (eager pass
        # that times the plans, capture at the next sight).  SEMSEG_CAPTURE_FIRST_SIGHT=0: the old order for every shape.
        self.capture_first_sight = os.environ.get('SEMSEG_CAPTURE_FIRST_SIGHT', '1') != '0'
        self.timeline = None                          # scaling_model.TimelineProbe: timestamp markers inside the (captured) step
        self._time_next_pass = False                  # graph=False: the next eager pass keeps the plain launches and times the plans a batched pass missed
        self._capture_errors = {}                     # feed_key -> repr of the capture refusal that sent the shape to the old order (logged once)
        self.stats = {'eager': 0, 'captured': 0, 'replayed': 0, 'evicted': 0, 'provisional': 0, 'capture_failed': 0, 'capture_host_s': 0.0, 'instantiate_host_s': 0.0,
                      'eager_host_s': 0.0}

    def adjust_learning_rate(self):
        """train.py:130-139 poly schedule"""
        scale = (1.0 - float(self.iter) / self.max_iters) ** self.lr_pow
        for i, g in enumerate(self.opt.groups):
            self.opt.set_lr(i, g['base_lr'] * scale)

    def _prepare_weights(self):
        if ops.CONV_MODE == 'h2' and ops.FUSE:
            ops.prepare_conv_weights(self._conv_weights)
            self._weights_ready = True

    def _eager(self, feed, recording=False):
        """one step issued on the current stream.  recording: this pass is being RECORDED into a hipGraph (it cannot time launch
        plans anyway), so independent sub-networks go out as side-by-side launches (ops.batch_branches).  An eager pass of a
        graph-mode step is where the tuner times new geometries: it keeps the plain launches.  Without graphs every pass is batched,
        except the one after a batched pass that met a geometry without a launch plan (nothing can be timed while launches are only
        recorded): that one runs the plain launches and times them -- the provisional rule of the graph path."""
        from . import tuner
        batched = recording or (not self.use_graph and not self._time_next_pass)
        missed = tuner.stats['missed_capturing']
        with ops.batch_branches(batched):
            out = self._eager_pass(feed)
        if not recording:
            self._time_next_pass = batched and tuner.stats['missed_capturing'] != missed
        return out

    def _eager_pass(self, feed):
        if not self._weights_ready:
            self._prepare_weights()
        tl = self.timeline
        if tl is not None:
            tl.mark('step_begin')
        self.opt.zero_grad()
        loss, acc = self.sm(feed)
        if self.buckets is not None:
            self.buckets.prepare()            # hooks launch each bucket's all-reduce as backward completes it
        if tl is not None:
            tl.mark('fwd_end')
            tl.arm()                          # its hooks mark each gradient bucket when backward completes it
        # the split weight gradients of the backward pass are summed, and the small ones computed, by batched launches: ONE flush after
        # backward on a single rank, one per gradient bucket (from the hook that completes it, before its all-reduce) on a rank of
        # a data-parallel job (ops.defer_wgrad_reduces)
        with ops.defer_wgrad_reduces(flush_at_buckets=self.buckets is not None and self.buckets.flushes_deferred,
                                     into_sgd=self.buckets is None), \
                ops.defer_fork_sums():        # ... and the gradient sums at the forks are formed by the BN kernels that consume them
            loss.backward()
        if tl is not None:
            tl.armed = False
            tl.mark('bwd_end')
        scale = 1.0
        if self.buckets is not None:
            self.buckets.finish()
            scale = 1.0 / self.world          # loss.mean() over replicas (train.py:42)
        self.opt.step(grad_scale=scale)           # one pass: sums the deferred gradient slabs, updates, leaves max|w| for the planes
        ops.finish_leftover_slabs()
        self._prepare_weights()               # planes of the UPDATED weights, for the next step
        if tl is not None:
            tl.mark('step_end')
        return loss.detach(), acc.detach()

    def _host_state(self):
        """what a recording pass changes on the HOST (it executes nothing on the device): restored before the eager retry of a
        refused first-sight recording, so that the retry is the step the caller asked for, not the one after it"""
        from . import tuner
        return dict(opt_steps=self.opt.steps, opt_state=set(self.opt.state.keys()), weights_ready=self._weights_ready,
                    wplanes=dict(ops._WPLANES), dropout=dict(ops._DROPOUT_STATE), tuner_stats=dict(tuner.stats),
                    captured=self.stats['captured'], evicted=self.stats['evicted'])

    def _restore_host_state(self, host):
        from . import tuner
        self.opt.steps = host['opt_steps']
        for p in [p for p in self.opt.state if p not in host['opt_state']]:
            del self.opt.state[p]                     # a momentum buffer created inside the refused recording lives in its pool
        self._weights_ready = host['weights_ready']
        ops._WPLANES.clear()
        ops._WPLANES.update(host['wplanes'])          # plane buffers first allocated inside the recording went with its pool
        ops._DROPOUT_STATE.clear()
        ops._DROPOUT_STATE.update(host['dropout'])
        ops._PENDING_WGRADS[:] = []
        ops._PENDING_SLABS[:] = []
        ops._FWD_USES.clear()
        ops._ADDENDS.clear()
        ops._SLABS_FOR_SGD.clear()                    # the recorded SGD kernel that would have summed / flagged them never ran
        ops._ABSMAX_FRESH.clear()
        tuner.stats.clear()
        tuner.stats.update(host['tuner_stats'])
        self.stats['captured'] = host['captured']
        if self.buckets is not None:
            self.buckets.armed = False

    def launch_mode(self):
        """'eager' | 'graph' (the whole step is ONE hipGraph) | 'segmented' (SegmentedStep: graphs between the collectives).
        world > 1: the RCCL all-reduces can be captured with the rest of the step -- a world-1 RCCL all-reduce survives capture +
        replay on this stack (tools/probes/rccl_graph_probe.py) -- but that has not run on a multi-GPU box, so it is opt-in
        (SEMSEG_DDP_GRAPH=1, set by bench.py after its self-test); the default at world > 1 is the segmented executor, which
        keeps the collectives out of the graphs.  SEMSEG_DDP_SEGMENTED=0: eager launches.  SEMSEG_FORCE_SYNC_PATH=1 (tests):
        the segmented executor on a single rank."""
        if not self.use_graph:
            return 'eager'
        if self.world == 1 and not ops._SYNC_GROUP['force']:
            return 'graph'
        if os.environ.get('SEMSEG_DDP_GRAPH', '0') == '1':
            return 'graph'
        return 'segmented' if os.environ.get('SEMSEG_DDP_SEGMENTED', '1') != '0' else 'eager'

    def step(self, feed):
        if self.world > 1:
            from . import comm
            comm.peer_check(self.group)       # a host-mapped word, no device sync: raises once a peer exchange has timed out
        self.adjust_learning_rate()
        self.iter += 1
        mode = self.launch_mode()
        if mode == 'eager':
            self.stats['eager'] += 1
            return self._eager(feed)
        key = feed_key(feed)
        rec = self._graphs.get(key)
        if rec is None:
            seen = self._seen.get(key, 0)
            first_sight = seen < 1 and self.capture_first_sight and key not in self._provisional
            if self.opt.steps < self.warmup_eager or (seen < 1 and not first_sight):
                self._seen[key] = seen + 1
                self.stats['eager'] += 1
                t0 = time.perf_counter()
                out = self._eager(feed)
                self.stats['eager_host_s'] += time.perf_counter() - t0      # host time of issuing the step (no device sync)
                return out
            from . import tuner
            missed = tuner.stats['missed_capturing']
            t0 = time.perf_counter()
            host = self._host_state()
            try:
                rec = self._capture(key, feed, mode)
            except RuntimeError as e:
                # ONLY a refusal that is about the capture itself (something on this shape's path cannot be recorded before it has run
                # once) is answered by the old order; an out-of-memory error, a kernel or argument error from _native.check, anything
                # else propagates -- the eager pass would hit it again, or worse, would not
                from . import _native
                if not first_sight or not _native.is_capture_error(e):
                    raise
                if key not in self._capture_errors:
                    import sys
                    self._capture_errors[key] = repr(e)
                    print('mit_semseg.engine: recording the step at first sight failed for batch shape %s (%r); this shape runs '
                          'eagerly once and is recorded at its next sight' % (key, e), file=sys.stderr)
                # nothing of a recording executes, so the step is still to be done -- eagerly, from the host state the recording
                # pass started with (it has advanced the optimiser's step count, created records, armed the buckets)
                self._restore_host_state(host)
                self._graphs.pop(key, None)
                self._provisional[key] = None
                self.stats['capture_failed'] += 1
                self._seen[key] = 1
                self.stats['eager'] += 1
                return self._eager(feed)
            self.stats['capture_host_s'] += time.perf_counter() - t0        # the capture pass + graph instantiation
            if first_sight and tuner.stats['missed_capturing'] != missed:
                # provisional: replayed this once; the next sight runs eagerly (times the missing plans), the one after captures for good.
                # The graph object outlives its replay (it is dropped when the shape is captured again, many steps later)
                self._provisional[key] = self._graphs.pop(key)
                self.stats['provisional'] += 1
            else:
                self._provisional.pop(key, None)
        else:
            self._graphs.move_to_end(key)
        graph, static, out = rec
        for k, v in static.items():
            v.copy_(feed[k])
        graph.replay()
        self._graph = graph
        self.stats['replayed'] += 1
        return out

    def _capture(self, key, feed, mode='graph'):
        """capture records but does not execute: the caller replays right away"""
        while len(self._graphs) >= max(1, self.max_graphs):
            self._graphs.popitem(last=False)          # least recently used shape: its graph and static buffers are released
            self.stats['evicted'] += 1
        static = {k: v.clone() for k, v in feed.items() if torch.is_tensor(v)}
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()
        if mode == 'segmented':
            graph = SegmentedStep()
            out = graph.capture(lambda: self._eager(static, recording=True), self._pool)
        else:
            # NOT `with torch.cuda.graph(...)`: its __enter__ is torch.cuda.synchronize() + empty_cache(), which drains the queue of
            # replays the host has run ahead of and leaves the device idle for the whole recording pass -- on the variable-size stream
            # (a new shape every few steps at the start of a run) that serialises ~40 ms of host work per new shape with the device.
            # Recording executes nothing: the capture stream only has to be ordered after the work already issued (an event wait, no
            # host sync), and the replay goes to the current stream as before.  The graphs' private pool needs no room made for it.
            graph = _TimedGraph()
            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream()
            out = record_graph(graph, lambda: self._eager(static, recording=True), self._capture_stream, self._pool)
            self.stats['instantiate_host_s'] += graph.end_s    # of capture_host_s: hipStreamEndCapture + hipGraphInstantiate
        rec = self._graphs[key] = (graph, static, out)
        self.stats['captured'] += 1
        return rec

    def flush(self):
        """kept for callers of the multi-step-graph API of round 1 (every step now launches when it is called)"""
        return None


class InferenceGraph:
    """segmentation_module(feed, segSize=...) of the inference branch (models.py:480-484, eval.py:66-71) captured into one
    hipGraph per (input shape, segSize, role) and replayed: an eager forward of R50dilated+PPM is ~350 launches and host-bound
    (4.0 ms per 512x512 image, tools/bench_infer.py), the replay is bound by its kernels.

        run = InferenceGraph(segmentation_module)
        prob = run(img, segSize=(H, W))                     # [N, num_class, H, W] probabilities; valid until the next call
        scores = run.multi_scale([img_300, img_375, ...], (H, W))      # eval.py:59-71: mean of the per-scale probabilities

    `multi_scale` keeps ONE score buffer per segSize; the fused head kernel of the first scale writes `p / n` into it, the
    others add theirs (ops.head_output) -- no full-resolution temporaries, no scale / add passes."""

    def __init__(self, segmentation_module, max_graphs=8):
        self.sm = segmentation_module.eval()
        self.max_graphs = max_graphs
        self._graphs = {}           # (img shape, segSize, accumulate, weight) -> (graph, static image, output)
        self._scores = {}           # (N, segSize) -> score buffer of multi_scale
        self._capture_stream = None

    def _eager(self, img, segSize, head=None):
        with torch.no_grad():
            if head is None:
                return self.sm({'img_data': img}, segSize=segSize)
            with ops.head_output(*head):
                return self.sm({'img_data': img}, segSize=segSize)

    def _run(self, img, segSize, head=None):
        key = (tuple(img.shape), segSize, None if head is None else (head[0].data_ptr(), head[1], head[2]))
        rec = self._graphs.get(key)
        if rec is None:
            if len(self._graphs) >= self.max_graphs:
                return self._eager(img, segSize, head)     # multi-scale evaluation with many distinct sizes: stay eager
            if head is not None and head[2]:
                keep = head[0].clone()                     # the warm-up call below must not count twice
            self._eager(img, segSize, head)                # first call: tunes the conv plans, builds the weight planes
            if head is not None and head[2]:
                head[0].copy_(keep)
            static = img.clone()
            graph = torch.cuda.CUDAGraph()
            if self._capture_stream is None:
                self._capture_stream = torch.cuda.Stream()
            out = record_graph(graph, lambda: self._eager(static, segSize, head), self._capture_stream)
            rec = self._graphs[key] = (graph, static, out)
        graph, static, out = rec
        static.copy_(img)
        graph.replay()
        return out

    def __call__(self, img, segSize):
        return self._run(img, (int(segSize[0]), int(segSize[1])))

    def multi_scale(self, imgs, segSize):
        """scores = sum_i softmax(upsample(logits(imgs[i]))) / len(imgs) at segSize (eval.py:59-71, test.py:66-78); the returned
        buffer is reused by the next call with the same segSize"""
        segSize = (int(segSize[0]), int(segSize[1]))
        n = int(imgs[0].shape[0])
        dev = imgs[0].device
        num_class = self._num_class(dev)
        key = (n, segSize)
        buf = self._scores.get(key)
        if buf is None:
            if len(self._scores) >= 4 * self.max_graphs:
                self._scores.clear()
            buf = self._scores[key] = ops.empty_nhwc(n, num_class, segSize[0], segSize[1], dev)
        w = 1.0 / len(imgs)
        for i, img in enumerate(imgs):
            self._run(img, segSize, (buf, w, i > 0))
        return buf

    def _num_class(self, dev):
        """output channels of the decoder's classifier (its last convolution; the deep-supervision twin has the same width)"""
        return [m.out_channels for m in self.sm.decoder.modules() if hasattr(m, 'out_channels')][-1]


def evaluate(segmentation_module, loader, num_class, device=None, tally=None, use_graph=True, on_item=None, group=None,
             reduce=True):
    """The evaluation loop of the reference (eval.py:40-105) with the arithmetic on the device: per item of a
    `ValDataset`-like iterable ({'img_data': [one tensor per scale], 'seg_label': [1,H,W]}) the softmax scores of every
    scale are produced at the label map's size and averaged in the reference's order (`scores = scores + scores_tmp / n`,
    eval.py:59-71), `pred = argmax` (first maximum, eval.py:73) and the tallies of `accuracy()` / `intersectionAndUnion()`
    (eval.py:81-86, utils.py:128-156) accumulate in a `MetricTally`.
    Returns (pixel accuracy, per-class IoU, mean IoU, tally) -- what eval.py:98-105 prints.
    `on_item(item, pred)` (optional) sees every prediction (visualisation hook, eval.py:88-94).
    Several processes (one per GPU, eval_multipro.py:122-169 shards the file list with start_idx / end_idx): every rank runs
    this loop over ITS shard and the integer tallies are summed over `group` at the end (`reduce`), so each rank returns the
    metrics of the whole set -- order-independent integer sums, identical to a single-process run."""
    from . import utils
    segmentation_module.eval()
    run = InferenceGraph(segmentation_module) if use_graph else None
    if device is None:
        device = next(segmentation_module.parameters()).device
    for item in loader:
        if isinstance(item, (list, tuple)):
            item = item[0]                                    # user_scattered_collate batches of one (eval.py:51)
        label = item['seg_label'][0].to(device)
        seg_size = (int(label.shape[0]), int(label.shape[1]))
        imgs = [img.to(device) for img in item['img_data']]
        if run is not None:
            scores = run.multi_scale(imgs, seg_size)          # p_0 / n, then += p_i / n in the fused head kernel
        else:
            scores = None
            for i, img in enumerate(imgs):
                ctx = ops.head_output(scores, 1.0 / len(imgs), i > 0)
                with torch.no_grad(), ctx:
                    s = segmentation_module({'img_data': img}, segSize=seg_size)
                if ctx.used:
                    scores = s                                    # weighted + accumulated by the fused head kernel
                else:                                             # a decoder that is not this build's: eval.py:70 literally
                    s = s / len(imgs)
                    scores = s if scores is None else scores + s
        pred, tally = utils.segmentation_metrics(scores, label, tally)
        if on_item is not None:
            on_item(item, pred)
    if tally is None:
        tally = utils.MetricTally(num_class, device)
    if reduce and world_size(group) > 1:
        import torch.distributed as dist
        dist.all_reduce(tally.counts, op=dist.ReduceOp.SUM, group=group)
    acc, iou, miou = tally.summary()
    return acc, iou, miou, tally
