"""Training-step engine: train.py:34-48 semantics (zero_grad -> forward -> backward -> 2x SGD with the
poly LR of train.py:130-139) executed as HIP kernels, optionally captured ONCE into a hipGraph and
replayed (the whole step is capture-safe: no host sync, no allocation outside torch's graph pool, the
learning rate lives in device memory)."""
import os

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

    def __init__(self, groups, momentum=0.9, lr_slots=1):
        """groups: list of dict(params=[...], lr=float, weight_decay=float).  `lr_slots`: device-resident learning rates
        per group -- a hipGraph that holds several consecutive steps reads slot i in its i-th step."""
        self.groups = groups
        self.momentum = momentum
        self.state = {}
        self.steps = 0
        self.lr_slot = 0
        for g in groups:
            dev = g['params'][0].device
            g['lr_t'] = torch.full((lr_slots,), float(g['lr']), device=dev, dtype=torch.float32)

    def set_lr(self, group_index, lr, slot=0):
        g = self.groups[group_index]
        g['lr'] = lr
        g['lr_t'][slot:slot + 1].fill_(lr)

    def zero_grad(self):
        for g in self.groups:
            for p in g['params']:
                p.grad = None

    def step(self, grad_scale=1.0):
        first = self.steps == 0
        for g in self.groups:
            ps = [p for p in g['params'] if p.grad is not None]
            if not ps:
                continue
            bufs = []
            for p in ps:
                b = self.state.get(p)
                if b is None:
                    b = torch.empty_like(p.grad)
                    self.state[p] = b
                bufs.append(b)
            ops.sgd_step(ps, [p.grad for p in ps], bufs, first, [g['weight_decay']] * len(ps),
                         g['lr_t'][self.lr_slot:self.lr_slot + 1], self.momentum, grad_scale)
        self.steps += 1


class TrainStep:
    """One training iteration of train.py:34-48 for a SegmentationModule.

    step(feed) -> (loss, acc) device scalars.  With `graph=True` the first `warmup_eager` calls run eagerly
    (they size the workspace and the allocator pools), then fwd+bwd+all-reduce+SGD is captured into a
    hipGraph; later calls copy the batch into static buffers and replay."""

    def __init__(self, segmentation_module, lr_encoder=0.02, lr_decoder=0.02, momentum=0.9, weight_decay=1e-4,
                 lr_pow=0.9, max_iters=100000, graph=False, group=None, bucket_bytes=64 << 20, graph_steps=None):
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
        # hipGraph replay.  `graph_steps` consecutive training steps are captured into ONE graph: every hipGraphLaunch
        # costs ~1 ms during which the GPU idles (profiles/r1i_trace_gaps_graph.txt: kernels are back to back inside a
        # replay, the only holes are between replays), so S steps per launch amortise it S-fold.  step() then returns
        # the (loss, acc) device scalars of its step, which hold their values once the S-th call of the group has
        # launched the replay.  SEMSEG_GRAPH_STEPS overrides.
        self.graph_steps = max(1, int(os.environ.get('SEMSEG_GRAPH_STEPS', graph_steps or 1))) if graph else 1
        self._graph = None
        self._static = None           # graph_steps feed dicts
        self._out = None              # graph_steps (loss, acc)
        self.warmup_eager = 2
        if self.graph_steps > 1:
            self.opt = FusedSGD(groups, momentum, lr_slots=self.graph_steps)

    def adjust_learning_rate(self, slot=0):
        """train.py:130-139 poly schedule"""
        scale = (1.0 - float(self.iter) / self.max_iters) ** self.lr_pow
        for i, g in enumerate(self.opt.groups):
            self.opt.set_lr(i, g['base_lr'] * scale, slot)

    def _prepare_weights(self):
        if ops.CONV_MODE == 'h2' and ops.FUSE:
            ops.prepare_conv_weights(self._conv_weights)
            self._weights_ready = True

    def _eager(self, feed):
        if not self._weights_ready:
            self._prepare_weights()
        self.opt.zero_grad()
        loss, acc = self.sm(feed)
        if self.buckets is not None:
            self.buckets.prepare()            # hooks launch each bucket's all-reduce as backward completes it
        ops.side_wgrad_begin(loss.device)     # weight gradients run on a second stream, off backward's critical path
        loss.backward()
        ops.side_wgrad_join()
        scale = 1.0
        if self.buckets is not None:
            self.buckets.finish()
            scale = 1.0 / self.world          # loss.mean() over replicas (train.py:42)
        self.opt.step(grad_scale=scale)
        self._prepare_weights()               # planes of the UPDATED weights, for the next step
        return loss.detach(), acc.detach()

    def step(self, feed):
        # world > 1: the RCCL all-reduces (SyncBN statistics on the compute stream, gradient buckets on the side stream)
        # can be captured too -- a world-1 RCCL all-reduce survives capture + replay on this stack
        # (tools/probes/rccl_graph_probe.py) -- but the N > 1 capture has not run on a multi-GPU box yet, so it is
        # opt-in (SEMSEG_DDP_GRAPH=1) and the default data-parallel step launches eagerly.
        if not self.use_graph or (self.world > 1 and os.environ.get('SEMSEG_DDP_GRAPH', '0') != '1'):
            self.adjust_learning_rate()
            self.iter += 1
            return self._eager(feed)
        if self._graph is None and self.opt.steps < self.warmup_eager:
            self.adjust_learning_rate()
            self.iter += 1
            return self._eager(feed)
        S = self.graph_steps
        if self._graph is None:
            # capture S consecutive steps; capture records but does not execute, the replays below run them
            self._static = [{k: v.clone() for k, v in feed.items() if torch.is_tensor(v)} for _ in range(S)]
            self._graph = torch.cuda.CUDAGraph()
            self._out = []
            with torch.cuda.graph(self._graph):
                for i in range(S):
                    self.opt.lr_slot = i
                    self._out.append(self._eager(self._static[i]))
            self.opt.lr_slot = 0
            self._sub = 0
        i = self._sub
        self.adjust_learning_rate(slot=i)
        self.iter += 1
        for k, v in self._static[i].items():
            v.copy_(feed[k])
        self._sub = (i + 1) % S
        if self._sub == 0:
            self._graph.replay()
        return self._out[i]

    def flush(self):
        """graph_steps > 1: a group of steps that has not been replayed yet (fewer than S calls since the last replay) is
        run eagerly from the staged batches.  Call before reading results / saving weights."""
        if self._graph is None or self.graph_steps == 1 or self._sub == 0:
            return
        n, self._sub = self._sub, 0
        for i in range(n):
            self.opt.lr_slot = i
            loss, acc = self._eager(self._static[i])
            self._out[i][0].copy_(loss)
            self._out[i][1].copy_(acc)
        self.opt.lr_slot = 0


class InferenceGraph:
    """segmentation_module(feed, segSize=...) of the inference branch (models.py:480-484, eval.py:66-71) captured into one
    hipGraph per (input shape, segSize) and replayed: an eager forward of R50dilated+PPM is ~350 launches and host-bound
    (4.0 ms per 512x512 image, tools/bench_infer.py), the replay is bound by its kernels.

        run = InferenceGraph(segmentation_module)
        prob = run(img, segSize=(H, W))        # [N, num_class, H, W] probabilities; valid until the next call
    """

    def __init__(self, segmentation_module, max_graphs=8):
        self.sm = segmentation_module.eval()
        self.max_graphs = max_graphs
        self._graphs = {}           # (img shape, segSize) -> (graph, static image, static output)

    def _eager(self, img, segSize):
        with torch.no_grad():
            return self.sm({'img_data': img}, segSize=segSize)

    def __call__(self, img, segSize):
        segSize = (int(segSize[0]), int(segSize[1]))
        key = (tuple(img.shape), segSize)
        rec = self._graphs.get(key)
        if rec is None:
            if len(self._graphs) >= self.max_graphs:
                return self._eager(img, segSize)          # multi-scale evaluation with many distinct sizes: stay eager
            self._eager(img, segSize)                      # first call: tunes the conv plans, builds the weight planes
            static = img.clone()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._eager(static, segSize)
            rec = self._graphs[key] = (graph, static, out)
        graph, static, out = rec
        static.copy_(img)
        graph.replay()
        return out


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
        imgs = item['img_data']
        scores = None
        for img in imgs:
            img = img.to(device)
            if run is not None:
                s = run(img, seg_size)
            else:
                with torch.no_grad():
                    s = segmentation_module({'img_data': img}, segSize=seg_size)
            s = s / len(imgs)                                 # a new tensor: the graph's output buffer is reused next call
            scores = s if scores is None else scores + s      # 0 + s == s exactly: same sums as the reference's zeros start
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
