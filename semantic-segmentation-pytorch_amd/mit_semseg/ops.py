"""Autograd operators of the MI355X hot path.  Every op is a thin torch.autograd.Function around the
C ABI of libsemseg_hip.so (include/semseg_hip.h): torch supplies device memory, the current HIP
stream and the autograd tape -- all arithmetic runs in the hand-written gfx950 kernels.

Tensor convention: tensors keep the reference's LOGICAL shape [N,C,H,W] but live in NHWC memory
(torch channels_last strides, possibly a channel slice of a wider buffer).  `as_nhwc` returns the
pixel stride `ld` the kernels take; anything else (e.g. the NCHW-contiguous input image of
train.py:41) is converted once by the nchw_to_nhwc kernel.

There is no CPU / eager fallback: non-CUDA tensors raise.
"""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _native
from . import tuner

vp = ctypes.c_void_p


def _p(t):
    return vp(t.data_ptr()) if t is not None else vp(0)


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _st():
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Stream object
    (8 us, 222 calls per eager step = 1.9 ms of host time, tools/probes/eager_cpu_profile.py); the raw getter is ~0.3 us."""
    if _raw_stream is not None:
        return vp(_raw_stream(torch.cuda.current_device()))
    return vp(torch.cuda.current_stream().cuda_stream)


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError('mit_semseg (MI355X build): tensors must live on a HIP device; '
                               'there is no CPU fallback for the native hot path')


# ------------------------------------------------------------------------------------------------
# workspace: one persistent scratch buffer per device (split-K / split-M slabs, BN partial sums)
# ------------------------------------------------------------------------------------------------
_WS = {}
_WS_RETIRED = []        # superseded scratch buffers: captured hipGraphs hold their raw addresses, so they are never freed
_WS_MIN = 64 << 20


# Independent sub-networks on separate HIP streams (HRNet's parallel branches, hrnet.py:225-227): the launch-bound kernels of the
# low-resolution branches overlap with the high-resolution branch, in eager launches and -- captured as parallel chains -- in the
# step's hipGraph.  Autograd runs every backward node on the stream of its forward, so backward overlaps the same way.
BRANCH_STREAMS = os.environ.get('SEMSEG_BRANCH_STREAMS', '1') != '0'       # '2': in no_grad (inference) passes as well
_BRANCH_POOL = {}       # device index -> [torch.cuda.Stream]
_BRANCH_TAG = {}        # raw stream handle -> workspace tag


def _branch_streams(device, count):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    pool = _BRANCH_POOL.setdefault(idx, [])
    while len(pool) < count:
        st = torch.cuda.Stream(device=device)
        _BRANCH_TAG[st.cuda_stream] = 'branch%d' % len(pool)
        pool.append(st)
    return pool[:count]


# Independent sub-networks as SIDE-BY-SIDE launches (round 6; csrc/batch.h): the branches run the same sequence of conv / BN
# launches on different geometries, each a few blocks and ~10 us long, so the step is their number (2 726 per HRNetV2 step) whatever
# stream they sit on.  Inside `batch_branches()` (TrainStep, around forward + backward of a pass that does not time launch plans)
# run_branches executes the Python of every branch with the library RECORDING its launches, and the records at the same position of
# every branch leave as ONE launch (semseg_batch_end) -- in forward and, through ONE autograd node for all branches (BranchesFn),
# in backward.  Same kernels, same order inside a branch: bit-identical.  SEMSEG_BATCH_BRANCHES=0 disables.
BATCH_BRANCHES = os.environ.get('SEMSEG_BATCH_BRANCHES', '1') != '0'
_BATCH = [False]


class batch_branches:
    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = _BATCH[0]
        _BATCH[0] = self.enabled and BATCH_BRANCHES and CONV_MODE == 'h2' and FUSE
        return self

    def __exit__(self, *exc):
        _BATCH[0] = self.prev
        return False


_TENSOR_RECORDS = ('_semseg_planes', '_semseg_absmax', '_semseg_planes_only')


def _copy_records(src, dst):
    """the split planes / bound scalars / planes-only mark that travel with tensor `src`, for its alias `dst` (same storage)"""
    for name in _TENSOR_RECORDS:
        rec = getattr(src, name, None)
        if rec is None:
            continue
        if name == '_semseg_planes':
            if rec[4] == src._version and rec[5] == src.data_ptr() == dst.data_ptr():
                attach_planes(dst, rec[0], rec[1], rec[2], rec[3])
        elif name == '_semseg_absmax':
            if rec[1] == src._version and rec[2] == src.data_ptr() == dst.data_ptr():
                attach_absmax(dst, rec[0])
        else:
            dst._semseg_planes_only = rec


def _fork_streams(main, streams):
    """the branch streams carry no launch of a batched scope -- everything leaves on `main` -- but the Python of branch i runs with
    streams[i] current so that the caching allocator keeps the branches' temporaries in pools of their own (a block freed by one
    branch must not be handed to another while both are only RECORDED: inside a branch the recorded order is the issue order, across
    branches it is not); forked from / joined to `main` like the side streams of the old form so that a graph capture covers them"""
    fork = torch.cuda.Event()
    fork.record(main)
    for st in streams:
        st.wait_event(fork)


def _join_streams(main, streams):
    for st in streams:
        done = torch.cuda.Event()
        done.record(st)
        main.wait_event(done)


# While launches are only RECORDED, a kernel torch itself launches inside a branch -- autograd adding two gradients that meet at a tensor
# with two consumers (use ops.fork), a .contiguous() / .clone() / arithmetic on tensors -- would run AT ONCE, ahead of the recorded
# launches that produce its operands.  The models of this package keep such operations out of their branches; SEMSEG_BATCH_CHECK=1 (and
# the GPU tests of the scopes) verifies it: every torch operator dispatched inside a scope must be one that launches nothing
# (allocation, views, metadata), anything else raises.
BATCH_CHECK = os.environ.get('SEMSEG_BATCH_CHECK', '0') == '1'
_LAUNCHLESS = ('empty', 'empty_like', 'empty_strided', 'detach', 'detach_', 'alias', 'view', '_unsafe_view', 'reshape', '_reshape_alias', 'permute',
               'transpose', 't', 'as_strided', 'slice', 'select', 'narrow', 'unsqueeze', 'squeeze', 'expand', 'unbind', 'split', 'chunk',
               'view_as', 'is_same_size', 'stride', 'size', 'sym_size', 'sym_stride', 'sym_numel', 'sym_storage_offset', 'numel', 'dim',
               'is_contiguous', 'is_pinned', 'record_stream', 'lift_fresh', 'set_', 'requires_grad_', '_version', 'data_ptr', 'is_leaf')


def _no_torch_kernels():
    """a TorchDispatchMode that lets only launch-free operators through (BATCH_CHECK)"""
    from torch.utils._python_dispatch import TorchDispatchMode

    class NoTorchKernels(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = getattr(func, '__name__', str(func)).split('.')[0]
            if name not in _LAUNCHLESS:
                raise RuntimeError('torch operator %s ran inside a side-by-side scope (ops.run_branches): its kernel would overtake the '
                                   'recorded launches that produce its operands -- keep it out of the branch (ops.fork for a tensor with '
                                   'two consumers, the native ops for arithmetic)' % (func,))
            return func(*args, **(kwargs or {}))
    return NoTorchKernels()


class _Recording:
    """the library records the launches of the C-ABI calls made inside (semseg_batch_begin ... _end on `main`)"""

    def __init__(self, n, main, streams):
        self.n, self.main, self.streams = n, main, streams

    def __enter__(self):
        L = _native.lib()
        _fork_streams(self.main, self.streams)
        _native.check(L.semseg_batch_begin(self.n, vp(self.main.cuda_stream)), 'batch_begin')
        _native.RECORDING[0] = True
        self.prev_timing = tuner.NO_TIMING[0]
        tuner.NO_TIMING[0] = True
        return self

    def branch(self, i):
        _native.lib().semseg_batch_branch(i)
        if not BATCH_CHECK:
            return torch.cuda.stream(self.streams[i])
        import contextlib
        stack = contextlib.ExitStack()
        stack.enter_context(torch.cuda.stream(self.streams[i]))
        stack.enter_context(_no_torch_kernels())
        return stack

    def __exit__(self, et, ev, tb):
        L = _native.lib()
        _native.RECORDING[0] = False
        tuner.NO_TIMING[0] = self.prev_timing
        if et is not None:
            L.semseg_batch_abort()
            _join_streams(self.main, self.streams)
            return False
        rc = L.semseg_batch_end()
        _join_streams(self.main, self.streams)
        _native.check(rc, 'batch_end')
        return False


class Branch:
    """a branch of run_branches that is not an nn.Module: `fn(x)` plus the modules whose parameters it uses (their gradients are
    produced inside the branches' node, BranchesFn.backward)"""

    def __init__(self, fn, modules=()):
        self.fn, self.modules = fn, tuple(modules)

    def __call__(self, x):
        return self.fn(x)

    def parameters(self):
        for m in self.modules:
            yield from m.parameters()


def _branch_params(f):
    get = getattr(f, 'parameters', None)
    return [p for p in get() if p.requires_grad] if get is not None else []


class BranchesFn(Function):
    """[f(x) for f, x in zip(fns, xs)] for independent sub-networks as ONE autograd node whose forward and backward record the
    launches of all branches and issue them position by position (see BATCH_BRANCHES above).  The branches' own autograd graphs
    live inside the node: forward builds them on detached inputs, backward runs them one after the other (torch.autograd.grad on the
    calling thread) while the library records.  A branch takes one tensor or a list of tensors (holder['counts'])."""

    @staticmethod
    def forward(ctx, holder, *flat):
        fns, counts = holder['fns'], holder['counts']
        n = len(fns)
        dev = flat[0].device
        main = torch.cuda.current_stream(dev)
        streams = _branch_streams(dev, n)
        ctx.set_materialize_grads(False)
        inner_x, inner_y = [], []
        with _Recording(n, main, streams) as rec:
            k = 0
            for i in range(n):
                m = counts[i] if counts[i] is not None else 1
                with rec.branch(i), torch.enable_grad():
                    xi = []
                    for t, need in zip(flat[k:k + m], ctx.needs_input_grad[1 + k:1 + k + m]):
                        d = t.detach()
                        _copy_records(t, d)
                        d.requires_grad_(bool(need))
                        xi.append(d)
                    yi = fns[i](xi if counts[i] is not None else xi[0])
                k += m
                inner_x.append(xi)
                inner_y.append(yi)
        ctx.inner = (fns, inner_x, inner_y, streams)
        holder['outs'] = inner_y
        return tuple(y.detach() for y in inner_y)

    @staticmethod
    @once_differentiable
    def backward(ctx, *gys):
        fns, inner_x, inner_y, streams = ctx.inner
        ctx.inner = None
        n = len(fns)
        dev = inner_y[0].device
        main = torch.cuda.current_stream(dev)
        gxs = [[None] * len(xi) for xi in inner_x]
        param_grads = []
        with _Recording(n, main, streams) as rec:
            for i in range(n):
                if gys[i] is None or not inner_y[i].requires_grad:
                    continue
                params = _branch_params(fns[i])
                wanted = [k for k, t in enumerate(inner_x[i]) if t.requires_grad]
                inputs = [inner_x[i][k] for k in wanted] + params
                if not inputs:
                    continue
                with rec.branch(i):
                    grads = torch.autograd.grad([inner_y[i]], inputs, [gys[i]], allow_unused=True)
                for k, g in zip(wanted, grads):
                    gxs[i][k] = g
                param_grads.extend((p, g) for p, g in zip(params, grads[len(wanted):]) if g is not None)
        # what AccumulateGrad does, after the scope has closed (an accumulation into an existing gradient is a torch kernel): the
        # returned buffer BECOMES .grad -- the adoption the deferred weight gradients rely on (_may_defer / flush_wgrad_reduces)
        for p, g in param_grads:
            if p.grad is None:
                p.grad = g
            else:
                p.grad = p.grad + g
            # ... and then runs the parameter's post-accumulate-grad hooks (gradient buckets, the timeline probe of bench.py's
            # scaling model: they mark when a parameter's gradient is complete)
            hooks = getattr(p, '_post_accumulate_grad_hooks', None)
            if hooks:
                for hook in list(hooks.values()):
                    hook(p)
        return (None,) + tuple(g for row in gxs for g in row)


def _run_branches_batched(fns, args):
    counts = [len(a) if isinstance(a, (list, tuple)) else None for a in args]
    flat = [t for a in args for t in (a if isinstance(a, (list, tuple)) else (a,))]
    holder = {'fns': list(fns), 'counts': counts}
    outs = BranchesFn.apply(holder, *flat)
    for o, y in zip(outs, holder.pop('outs')):
        _copy_records(y, o)                             # the planes / bound the last op of the branch left on its output
    return list(outs)


def _batchable(fns, args):
    if not (_BATCH[0] and torch.is_grad_enabled() and 1 < len(fns) <= 16 and len(fns) == len(args)):
        return False
    dev, any_grad = None, False
    for f, a in zip(fns, args):
        if not callable(f):
            return False
        ts = a if isinstance(a, (list, tuple)) else (a,)
        if not ts:
            return False
        for t in ts:
            if not (torch.is_tensor(t) and t.is_cuda and t.dim() == 4 and t.dtype == torch.float32):
                return False
            if dev is not None and t.device != dev:
                return False
            dev = t.device
            any_grad = any_grad or t.requires_grad
        any_grad = any_grad or bool(_branch_params(f))
    if not any_grad:
        return False
    if _sync_active() or _SEGMENTS is not None or _native.lib().semseg_batch_active():
        return False                                    # SyncBN exchanges keep ONE issue order; a scope inside a scope stays sequential
    return torch.cuda.current_stream(dev).cuda_stream not in _BRANCH_TAG


def batch_unit():
    """the current branch of an open side-by-side scope enters its next unit: launches of different branches pair up unit by unit
    (ConvBNActFn does this itself; a branch made of other operators marks its steps with it)"""
    _native.next_unit()


def run_branches(fns, args, side_streams=True):
    """[f(a) for f, a in zip(fns, args)] for independent sub-networks: as side-by-side launches inside `batch_branches()`
    (BranchesFn), else with f_1 ... f_n-1 on side streams (f_0, the largest, stays on the current stream), joined before
    returning (side_streams=False: one after the other on the current stream).  Off when SyncBN is active (the ranks must issue
    their exchanges in ONE order) and during a segmented capture."""
    if _batchable(fns, args):
        return _run_branches_batched(fns, args)
    if not side_streams:
        return [f(a) for f, a in zip(fns, args)]
    x0 = args[0]
    while isinstance(x0, (list, tuple)):                  # an argument may be a list of tensors (every branch reads all of them)
        x0 = x0[0]
    if not (BRANCH_STREAMS and len(fns) > 1 and torch.is_tensor(x0) and x0.is_cuda) or _sync_active() or _SEGMENTS is not None:
        return [f(a) for f, a in zip(fns, args)]
    if not torch.is_grad_enabled() and os.environ.get('SEMSEG_BRANCH_STREAMS', '1') != '2':
        return [f(a) for f, a in zip(fns, args)]         # measured on the training step only (a forked hipGraph is submitted node by node)
    main = torch.cuda.current_stream(x0.device)
    if main.cuda_stream in _BRANCH_TAG:                  # nested use: stay sequential on this branch's stream
        return [f(a) for f, a in zip(fns, args)]
    pool = _branch_streams(x0.device, len(fns) - 1)
    _presplit_shared(args)
    fork = torch.cuda.Event()
    fork.record(main)
    outs = [None] * len(fns)
    for i in range(1, len(fns)):
        pool[i - 1].wait_event(fork)
        _crosses_to(args[i], pool[i - 1])                # allocated on this stream, read (and saved for backward) on the branch's
        with torch.cuda.stream(pool[i - 1]):
            outs[i] = fns[i](args[i])
    outs[0] = fns[0](args[0])
    for i in range(1, len(fns)):
        done = torch.cuda.Event()
        done.record(pool[i - 1])
        main.wait_event(done)
        _crosses_to(outs[i], main)                       # allocated on the branch stream, consumed on this one
    return outs


def _presplit_shared(args):
    """A tensor that several branches read gets its split planes HERE, before the fork: input_planes() leaves the planes of a
    conv input on the tensor for its next consumer, and a branch on another stream must not pick up planes whose split kernel
    it is not ordered after."""
    if not FUSE or CONV_MODE not in SCHEMES:
        return
    count, first = {}, {}
    for a in args:
        for t in (a if isinstance(a, (list, tuple)) else (a,)):
            if torch.is_tensor(t) and t.is_cuda and t.dim() == 4:
                count[id(t)] = count.get(id(t), 0) + 1
                first[id(t)] = t
    for k, c in count.items():
        if c > 1:
            input_planes(first[k], CONV_MODE)


def _crosses_to(t, stream):
    """tensor `t` -- and the split planes / bound scalars that travel with it -- is used on `stream`, which is not the stream it
    was allocated on: tell the caching allocator, or the block could be handed out again while that stream still reads it"""
    if isinstance(t, (list, tuple)):
        for e in t:
            _crosses_to(e, stream)
        return
    if not torch.is_tensor(t) or not t.is_cuda:
        return
    t.record_stream(stream)
    rec = getattr(t, '_semseg_planes', None)
    if rec is not None and torch.is_tensor(rec[0]):
        rec[0].record_stream(stream)
    rec = getattr(t, '_semseg_absmax', None)
    if rec is not None:
        for b in rec[0]:
            if torch.is_tensor(b):
                b.record_stream(stream)


def workspace(nbytes, device, tag=''):
    """`tag`: kernels running concurrently on different streams need disjoint scratch (the branch streams of run_branches
    get their own).  A buffer that is outgrown stays allocated (`_WS_RETIRED`): a hipGraph captured earlier (TrainStep, one
    InferenceGraph per shape) has its address baked into split-K / BN-partial launches, and handing the block back to the
    caching allocator would let a replay scribble over whatever tensor gets it next.  Growth is geometric (x1.5), so the
    retired blocks add up to at most twice the live one."""
    if _BRANCH_TAG and device.type == 'cuda':           # a branch stream of run_branches: its own scratch
        tag = _BRANCH_TAG.get(torch.cuda.current_stream(device).cuda_stream, tag)
    key = (device.type, device.index if (device.index is not None or device.type != 'cuda') else torch.cuda.current_device(), tag)
    t = _WS.get(key)
    if t is None or t.numel() < nbytes:
        size = max(_WS_MIN, int(nbytes * 1.5))
        if t is not None:
            _WS_RETIRED.append(t)
        t = torch.empty(size, dtype=torch.uint8, device=device)
        _WS[key] = t
    return t


# ------------------------------------------------------------------------------------------------
# layout helpers
# ------------------------------------------------------------------------------------------------
def nhwc_ld(x):
    """Pixel stride (floats) if logical-NCHW tensor `x` is NHWC-addressable, else None."""
    n, c, h, w = x.shape
    sn, sc, sh, sw = x.stride()
    if c > 1 and sc != 1:
        return None
    if w > 1:
        ld = sw
    elif h > 1:
        ld = sh // max(w, 1)
    elif n > 1:
        ld = sn // max(h * w, 1)
    else:
        ld = c
    if ld < c:
        return None
    if w > 1 and sw != ld:
        return None
    if h > 1 and sh != w * ld:
        return None
    if n > 1 and sn != h * w * ld:
        return None
    return ld


def empty_nhwc(n, c, h, w, device, dtype=torch.float32):
    """New dense NHWC buffer, returned as its logical-NCHW view."""
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def _refuse_planes_only(x):
    if getattr(x, '_semseg_planes_only', False):
        raise RuntimeError('this BN output was written as h2 planes only (conv_bn_act(..., planes_only=True)): its fp32 values do '
                           'not exist; its one consumer must be a convolution of the h2 path')


def as_nhwc_of(x):
    """as_nhwc of the detached tensor.  The planes-only mark is a Python attribute of the tensor OBJECT and detach() makes a
    new object, so it is read here, before the detach (every Function.forward takes its fp32 operands through this)."""
    _refuse_planes_only(x)
    return as_nhwc(x.detach())


def as_nhwc(x):
    """Returns (tensor, ld): `tensor` is logical NCHW over NHWC memory."""
    if _ADDENDS:                               # a gradient whose second addend is still pending (defer_fork_sums): sum it now
        x = _materialize_sum(x)
    return _as_nhwc_plain(x)


def _as_nhwc_plain(x):
    _require_cuda(x)
    if x.dtype != torch.float32:
        raise RuntimeError('native hot path computes in fp32, got %s' % x.dtype)
    _refuse_planes_only(x)
    ld = nhwc_ld(x)
    if ld is not None:
        return x, ld
    n, c, h, w = x.shape
    if x.is_contiguous():                      # NCHW -> NHWC on the device
        out = empty_nhwc(n, c, h, w, x.device)
        _native.check(_native.lib().semseg_nchw_to_nhwc(_p(x), _p(out), n, c, h * w, _st()), 'nchw_to_nhwc')
        return out, c
    out = empty_nhwc(n, c, h, w, x.device)
    out.copy_(x)                               # exotic strides: torch glue copy (never on the hot path)
    return out, c


def krsc(w):
    """[K,C,R,S] parameter -> (tensor whose memory is KRSC, same logical shape)."""
    if w.permute(0, 2, 3, 1).is_contiguous():
        return w
    return w.contiguous(memory_format=torch.channels_last) if (w.shape[2] > 1 or w.shape[3] > 1) else \
        w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


# ------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------
# 'h2': 2-way fp16 split, 3 MFMA products (default); 's3': 3-way bf16 split, 6 products; 'f32': exact fp32 MFMA kernels
CONV_MODE = os.environ.get('SEMSEG_CONV', 'h2')
if CONV_MODE not in ('h2', 's3', 'f32'):
    raise RuntimeError("SEMSEG_CONV must be 'h2', 's3' or 'f32', got %r" % CONV_MODE)


class SplitScheme:
    """Entry points of one operand-split convolution family of the C ABI (csrc/conv_split.hip)."""

    def __init__(self, name, split, tag):
        self.name, self._split, self._tag = name, split, tag

    def fn(self, L, what):
        if what in ('bytes', 'split'):
            return getattr(L, 'semseg_%s%s' % (self._split, '_bytes' if what == 'bytes' else ''))
        if what in ('workspace_bytes', 'set_plan'):
            return getattr(L, 'semseg_conv2d_%s_%s' % (self._tag, what))
        return getattr(L, 'semseg_conv2d_%s_%s' % (what, self._tag))          # fwd / dgrad / wgrad

    def split(self, t, rows, ch, ld):
        """fp32 rows [rows][ld] (first `ch` columns) -> 16-bit planes of this scheme."""
        L = _native.lib()
        out = torch.empty(self.fn(L, 'bytes')(rows, ch), dtype=torch.uint8, device=t.device)
        _native.check(self.fn(L, 'split')(_p(t), ld, _p(out), rows, ch, _st()), self._split)
        return out


SCHEMES = {'s3': SplitScheme('s3', 'split3', 's3'), 'h2': SplitScheme('h2', 'split_h2', 'h2')}


def split3(t, rows, ch, ld):
    return SCHEMES['s3'].split(t, rows, ch, ld)


class Conv2dFn(Function):
    """nn.Conv2d forward/backward (square stride/pad/dilation), exact-fp32 MFMA implicit-GEMM kernels
    (SEMSEG_CONV=f32; csrc/conv_igemm.hip, conv_wgrad.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, dil):
        L = _native.lib()
        x, x_ld = as_nhwc_of(x)
        w = krsc(weight.detach())
        _require_cuda(w, bias)
        n, c, h, wd = x.shape
        k, c2, r, s = w.shape
        if c2 != c:
            raise RuntimeError('conv2d: input has %d channels, weight expects %d' % (c, c2))
        oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
        y = empty_nhwc(n, k, oh, ow, x.device)
        wsb = L.semseg_conv2d_workspace_bytes(n, h, wd, c, k, r, s, stride, pad, dil)
        ws = workspace(wsb, x.device)
        _native.check(L.semseg_conv2d_fwd(_p(x), x_ld, _p(w), _p(bias.detach() if bias is not None else None),
                                          _p(y), k, n, h, wd, c, k, r, s, stride, pad, dil,
                                          _p(ws), ws.numel(), _st()), 'conv2d_fwd')
        ctx.save_for_backward(x, w)
        ctx.geom = (x_ld, stride, pad, dil, bias is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        L = _native.lib()
        x, w = ctx.saved_tensors
        x_ld, stride, pad, dil, has_bias = ctx.geom
        n, c, h, wd = x.shape
        k, _, r, s = w.shape
        dy, dy_ld = as_nhwc(dy)
        ws = workspace(L.semseg_conv2d_workspace_bytes(n, h, wd, c, k, r, s, stride, pad, dil), x.device)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = torch.empty((c, r, s, k), device=x.device, dtype=torch.float32)
            _native.check(L.semseg_weight_krsc_to_crsk(_p(w), _p(wt), k, r * s, c, _st()), 'weight_transpose')
            dx = empty_nhwc(n, c, h, wd, x.device)
            _native.check(L.semseg_conv2d_dgrad(_p(dy), dy_ld, _p(wt), _p(dx), c, n, h, wd, c, k, r, s,
                                                stride, pad, dil, _p(ws), ws.numel(), _st()), 'conv2d_dgrad')
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            dwb = torch.empty((k, r, s, c), device=x.device, dtype=torch.float32)
            if has_bias and ctx.needs_input_grad[2]:
                db = torch.empty((k,), device=x.device, dtype=torch.float32)
            _native.check(L.semseg_conv2d_wgrad(_p(x), x_ld, _p(dy), dy_ld, _p(dwb), _p(db), n, h, wd, c, k, r, s,
                                                stride, pad, dil, _p(ws), ws.numel(), _st()), 'conv2d_wgrad')
            dw = dwb.permute(0, 3, 1, 2)
        return dx, dw, db, None, None, None


# SEMSEG_FUSE=0 turns the plane hand-over between producers and consumers off (every conv then splits its own operands)
FUSE = os.environ.get('SEMSEG_FUSE', '1') != '0'


def attach_planes(t, buf, scheme, rows, ch):
    """Remember the split planes of activation `t` on the tensor object: the next conv that consumes this very object
    (unchanged: same version counter, same storage) reads them instead of splitting again."""
    t._semseg_planes = (buf, scheme, rows, ch, t._version, t.data_ptr())


def planes_of(t, scheme, rows, ch):
    rec = getattr(t, '_semseg_planes', None) if FUSE else None
    if rec is not None and rec[1:4] == (scheme, rows, ch) and rec[4] == t._version and rec[5] == t.data_ptr():
        return rec[0]
    return None


def share_planes(src, dst):
    """`dst` aliases `src` (same storage, e.g. the two outputs of fork()): a consumer that split one of them leaves the planes
    to the other"""
    rec = getattr(src, '_semseg_planes', None)
    if rec is not None and rec[4] == src._version and rec[5] == src.data_ptr() == dst.data_ptr():
        attach_planes(dst, rec[0], rec[1], rec[2], rec[3])


def input_planes(x, scheme):
    """Split planes of conv input `x` (logical NCHW): handed over by the producer (BN kernels of the fused path), left by
    an earlier consumer of the same tensor (block inputs feed two convs), or computed now."""
    n, c, h, w = x.shape
    xp = planes_of(x, scheme, n * h * w, c)
    if xp is None:
        xn, ld = as_nhwc_of(x)
        bounds = bounds_of(x) if scheme == 'h2' else None
        if bounds is not None and len(bounds) <= 8:
            # an upper bound of max|x| travels with the tensor (BN outputs, pooling / concat / sums of them): the exponent comes
            # from it and the absmax pass over x is skipped -- one launch instead of two (h2_exponent accepts any upper bound)
            L = _native.lib()
            xp = torch.empty(L.semseg_split_h2_bytes(n * h * w, c), dtype=torch.uint8, device=xn.device)
            bp = (vp * len(bounds))(*[b.data_ptr() for b in bounds])
            _native.check(L.semseg_split_h2_bounds(_p(xn), ld, _p(xp), n * h * w, c, bp, len(bounds), _st()), 'split_h2_bounds')
        else:
            xp = SCHEMES[scheme].split(xn, n * h * w, c, ld)
        if FUSE:
            attach_planes(x, xp, scheme, n * h * w, c)
    return xp


def bound_sum(bounds):
    """one device scalar >= sum of the given bound scalars: the bound of |t1 + t2 + ...| from bounds of the terms"""
    bounds = list(bounds)
    if len(bounds) == 1:
        return bounds[0]
    if not 1 <= len(bounds) <= 8:
        raise ValueError('bound_sum takes 1 ... 8 bound scalars')
    out = torch.empty((1,), device=bounds[0].device, dtype=torch.float32)
    bp = (vp * len(bounds))(*[b.data_ptr() for b in bounds])
    _native.check(_native.lib().semseg_bound_sum(bp, len(bounds), _p(out), _st()), 'bound_sum')
    return out


def attach_absmax(t, bounds):
    """`bounds`: one 1-element device tensor, or a tuple of them: max|t| <= max(bounds) (BN kernels of the fused path;
    a concat carries the bounds of its inputs)."""
    if torch.is_tensor(bounds):
        bounds = (bounds,)
    t._semseg_absmax = (tuple(bounds), t._version, t.data_ptr())


def bounds_of(t):
    """tuple of device scalars bounding max|t|, or None"""
    rec = getattr(t, '_semseg_absmax', None) if (FUSE and t is not None) else None
    if rec is not None and rec[1] == t._version and rec[2] == t.data_ptr():
        return rec[0]
    return None


def absmax_of(t):
    """the single device scalar bounding max|t| (None if unknown or if the bound is a set of several scalars)"""
    b = bounds_of(t)
    return b[0] if (b is not None and len(b) == 1) else None


# Winograd F(2x2, 3x3) for the forward pass (and weight gradient) of 3x3 stride-1 convs with at least this many input channels
# (csrc/winograd.hip).  Measured per layer on 2 x 64 x 64 maps (profiles/r4_winograd_midsize_probe.txt: every piece timed alone,
# the batched GEMM on its tuned tile form): 4096 ch 0.55 vs 0.79 ms direct, 1024 ch 136 vs 210 us, 512 ch (layer4's dilated 3x3
# convs) 88 + 7 (statistics sweep) vs 123 us, 256 ch 41 + 7 vs 48 us (break-even: the transforms are HBM-bound passes, V is 4x the
# input) -> threshold 512.  Round 2 had it at 1024: the GEMM then ran on the 8-wave 256 x 256 tile (66 us at 512 channels
# against 56 us on the 16-wave form the tuner now pins as plan pass 3).  SEMSEG_WINOGRAD=0 disables.
WINOGRAD = os.environ.get('SEMSEG_WINOGRAD', '1') != '0'
WINOGRAD_MIN_C = int(os.environ.get('SEMSEG_WINOGRAD_MIN_C', '512'))
# the BN statistics of a conv -> BN pair gathered in the conv's GEMM epilogue instead of by a sweep over its output (ConvBNActFn);
# SEMSEG_EPILOGUE_STATS=0: the separate statistics kernel (A/B switch, tests/test_gpu_models.py SWITCH_CASES)
EPILOGUE_STATS = os.environ.get('SEMSEG_EPILOGUE_STATS', '1') != '0'
# the weight gradient of the same layers in the Winograd domain (dU[f] = dM[f]^T V[f], V kept from the forward pass):
# in-box A/B (gpurun wg1) 16.50 -> 16.02 ms per step.  SEMSEG_WINOGRAD_WGRAD=0 disables.
WINOGRAD_WGRAD = os.environ.get('SEMSEG_WINOGRAD_WGRAD', '1') != '0'
# ... from this many input channels on: the batched weight-gradient GEMM has only K x C / 256^2 x 16 output tiles, 64 for a
# 512 -> 512 layer, and splitting the 2048-tile reduction 8 ways to fill the chip costs a 134 MB slab reduce -- measured no gain
# over the direct weight gradient there (gpurun r4g), so the 512-channel layers run Winograd forward + direct weight gradient
WINOGRAD_WGRAD_MIN_C = 1024
# the data gradient of the same layers in the Winograd domain: dx = conv3x3(dz, flipped + transposed weights) through the same
# input transform (from the planes of dz) / batched GEMM / output transform; U' = G g' G^T comes from the weight preparation.
# conv_last of the PPM head (4096 <- 512 @ 64 x 64): ~0.6 vs 0.79 ms for the direct kernel.  SEMSEG_WINOGRAD_DGRAD=0 disables.
WINOGRAD_DGRAD = os.environ.get('SEMSEG_WINOGRAD_DGRAD', '1') != '0'


def _wino_dgrad_eligible(k):
    return WINOGRAD_DGRAD and k >= 256           # the reduction of the data-gradient GEMM runs over the filters


def _wino_eligible(k, c, r, s):
    return WINOGRAD and FUSE and CONV_MODE == 'h2' and r == 3 and s == 3 and c % 4 == 0 and k % 4 == 0 and c >= WINOGRAD_MIN_C


# ---- conv weights: split planes prepared for ALL convs in one multi-tensor launch (engine calls it after the SGD step)
_WPLANES = {}       # id(param) -> (weakref, version, data_ptr, krsc planes, crsk planes)


def prepare_conv_weights(weights):
    """h2 split planes (KRSC for fwd, CRSK for dgrad) of every 4-D conv weight in `weights`, csrc/weights_prep.hip.
    The planes stay valid until the parameter changes through torch (version counter) -- the fused SGD kernel updates
    parameters behind torch's back, so the engine calls this again right after it."""
    import weakref
    L = _native.lib()
    todo = []
    for w in weights:
        if w.dim() != 4 or not w.permute(0, 2, 3, 1).is_contiguous():
            continue
        _require_cuda(w)
        k, c, r, s = w.shape
        rec = _WPLANES.get(id(w))
        if rec is None or rec[0]() is not w or rec[2] != w.data_ptr():
            kb = torch.empty(L.semseg_split_h2_bytes(k * r * s, c), dtype=torch.uint8, device=w.device)
            cb = torch.empty(L.semseg_split_h2_bytes(c * r * s, k), dtype=torch.uint8, device=w.device)
            ub = torch.empty(L.semseg_split_h2_bytes(16 * k, c), dtype=torch.uint8, device=w.device) \
                if _wino_eligible(k, c, r, s) else None
            utb = torch.empty(L.semseg_split_h2_bytes(16 * c, k), dtype=torch.uint8, device=w.device) \
                if (_wino_eligible(k, c, r, s) and _wino_dgrad_eligible(k)) else None
            rec = (weakref.ref(w), w._version, w.data_ptr(), kb, cb, ub, utb)
            _ABSMAX_FRESH.pop(id(w), None)               # new plane buffers: their slots have not been written
        else:
            rec = (rec[0], w._version, rec[2], rec[3], rec[4], rec[5], rec[6])
        _WPLANES[id(w)] = rec
        todo.append((w, rec))
    if not todo:
        return 0
    arr = (_native.WPrepTensor * len(todo))()
    for i, (w, rec) in enumerate(todo):
        k, c, r, s = w.shape
        arr[i].w, arr[i].krsc, arr[i].crsk = w.data_ptr(), rec[3].data_ptr(), rec[4].data_ptr()
        arr[i].K, arr[i].T, arr[i].C = k, r * s, c
        arr[i].wino = rec[5].data_ptr() if rec[5] is not None else None
        arr[i].wino_t = rec[6].data_ptr() if rec[6] is not None else None
    # every tensor's partial |w| maxima are in place (the fused SGD kernel wrote them with the update): no absmax pass
    # (and nothing touched the weight through torch since: the version counter the flag was taken at)
    have = SGD_FUSED and all(_ABSMAX_FRESH.get(id(w), None) == w._version for w, _ in todo)
    for w, _ in todo:
        _ABSMAX_FRESH.pop(id(w), None)
    _native.check(L.semseg_weights_prepare_h2_after_sgd(arr, len(todo), 1 if have else 0, _st()), 'weights_prepare_h2')
    return len(todo)


def weight_wino(w):
    """Winograd-transformed planes of `w` prepared for this exact parameter state, else None"""
    rec = _WPLANES.get(id(w)) if FUSE else None
    if rec is not None and rec[0]() is w and rec[1] == w._version and rec[2] == w.data_ptr():
        return rec[5]
    return None


def weight_wino_t(w):
    """planes of the Winograd-transformed weights of the DATA GRADIENT (U', rows (f, c), channels k) for this exact parameter
    state, else None"""
    rec = _WPLANES.get(id(w)) if FUSE else None
    if rec is not None and rec[0]() is w and rec[1] == w._version and rec[2] == w.data_ptr():
        return rec[6]
    return None


def weight_planes(w, scheme):
    """(KRSC planes, CRSK planes) prepared by prepare_conv_weights for this exact parameter state, else (None, None)."""
    if not (FUSE and scheme == 'h2'):
        return None, None
    rec = _WPLANES.get(id(w))
    if rec is not None and rec[0]() is w and rec[1] == w._version and rec[2] == w.data_ptr():
        return rec[3], rec[4]
    if not torch.is_grad_enabled() and w.dim() == 4 and w.permute(0, 2, 3, 1).is_contiguous():
        # inference (eval.py / test.py run under no_grad): the weights do not change between calls, so their planes are
        # built on first use and kept (2 launches per conv and per forward otherwise); any torch-side update bumps the
        # version counter and the fused SGD kernel invalidates the record (sgd_step)
        prepare_conv_weights([w])
        rec = _WPLANES[id(w)]
        return rec[3], rec[4]
    return None, None


def _weight_crsk_planes(L, sch, w, dev):
    k, c, r, s = w.shape
    wt = torch.empty((c, r, s, k), device=dev, dtype=torch.float32)
    _native.check(L.semseg_weight_krsc_to_crsk(_p(w), _p(wt), k, r * s, c, _st()), 'weight_transpose')
    return sch.split(wt, c * r * s, k, k)


class Conv2dSplitFn(Function):
    """nn.Conv2d forward/backward on the 16-bit MFMA with fp32-class accuracy: every operand is split into 16-bit
    planes once (h2: 2 x fp16 with a per-tensor power-of-two scale, 3 products; s3: 3 x bf16, 6 products) and the
    significant partial products are accumulated in fp32 (csrc/conv_split.hip).  The planes of the input (`xp`) are
    kept for the weight gradient; the split of dy is shared by the data and weight gradients; `wp` / `wtp` are the
    prepared weight planes (None: split here)."""

    @staticmethod
    def forward(ctx, x, weight, bias, xp, wp, wtp, stride, pad, dil, scheme):
        L = _native.lib()
        sch = SCHEMES[scheme]
        w = krsc(weight.detach())
        _require_cuda(w, bias)
        n, c, h, wd = x.shape
        k, c2, r, s = w.shape
        if c2 != c:
            raise RuntimeError('conv2d: input has %d channels, weight expects %d' % (c, c2))
        oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
        geom = (n, h, wd, c, k, r, s, stride, pad, dil)
        y = empty_nhwc(n, k, oh, ow, x.device)
        wsp = wp if wp is not None else sch.split(w, k * r * s, c, c)
        b = bias.detach() if bias is not None else None

        def launch():
            ws = workspace(sch.fn(L, 'workspace_bytes')(*geom), x.device)
            _native.check(sch.fn(L, 'fwd')(_p(xp), _p(wsp), _p(b), _p(y), k, *geom, _p(ws), ws.numel(), _st()),
                          'conv2d_fwd_' + scheme)
        tuner.ensure(scheme, 0, geom, launch)
        launch()
        ctx.save_for_backward(xp, w, wtp)
        ctx.geom = geom
        ctx.has_bias = bias is not None
        ctx.scheme = scheme
        ctx.w_param = _note_weight_use(weight, ctx.needs_input_grad[1])
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        L = _native.lib()
        xs, w, wtp = ctx.saved_tensors
        geom = ctx.geom
        scheme = ctx.scheme
        sch = SCHEMES[scheme]
        n, h, wd, c, k, r, s, stride, pad, dil = geom
        oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
        dev = w.device
        dy, dy_ld = as_nhwc(dy)
        dys = sch.split(dy, n * oh * ow, k, dy_ld)
        dx, dw = _split_conv_grads(L, sch, scheme, geom, xs, dys, w, wtp, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                   param=ctx.w_param)
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty((k,), device=dev, dtype=torch.float32)
            ws = workspace(sch.fn(L, 'workspace_bytes')(*geom), dev)
            _native.check(L.semseg_bias_grad(_p(dy), dy_ld, _p(db), n * oh * ow, k, _p(ws), ws.numel(), _st()), 'bias_grad')
        return dx, dw, db, None, None, None, None, None, None, None


# Weight gradients whose launch plan splits the pixels leave per-chunk partial sums (slabs) that a reduce launch adds up -- 58
# launches of 4 - 7 us per configs[1] step, 270 on HRNetV2, all at the launch floor.  Inside `defer_wgrad_reduces()` (TrainStep: one
# rank, gradients zeroed before backward, nothing reads a weight gradient before backward has returned) the slabs stay where they are
# and ONE multi-tensor launch sums all of them after backward (flush_wgrad_reduces): same slab order per tensor, bit-identical sums.
# Not under SyncBN / gradient buckets (their hooks read a gradient as soon as autograd has accumulated it) and never for plain
# autograd callers (a second backward would accumulate into an unreduced gradient).  SEMSEG_DEFER_WGRAD_REDUCE=0 disables.
DEFER_WGRAD_REDUCE = os.environ.get('SEMSEG_DEFER_WGRAD_REDUCE', '1') != '0'
# ... and the small weight gradients themselves wait too: a geometry whose launch plan is the register-staged 64 x 64 tile (2 - 60
# blocks on 256 CUs; 233 such launches per HRNetV2 step, 32 on configs[1]) keeps its operand planes alive until backward has returned,
# and flush_wgrad_reduces runs the blocks of up to 24 of them side by side in one launch (semseg_conv2d_wgrad_multi_h2: the blocks
# of the per-layer launches unchanged, bit-identical slabs).  SEMSEG_DEFER_WGRAD_LAUNCH=0 launches them where autograd reaches them.
DEFER_WGRAD_LAUNCH = os.environ.get('SEMSEG_DEFER_WGRAD_LAUNCH', '1') != '0'
# ... and a small weight gradient (K, C <= 128) whose own plan is another tile joins them on the 64 x 64 tile with the same number of
# blocks (csrc/conv_split.hip wgrad_member_plan): its plan was timed with the chip to itself, the batched launch shares it.  HRNetV2's
# 96-channel branch: 19.29 -> 18.36 ms per step.  SEMSEG_WGRAD_MEMBER_PLAN=0: every weight gradient on its own plan.
WGRAD_MEMBER_PLAN = os.environ.get('SEMSEG_WGRAD_MEMBER_PLAN', '1') != '0'
# ONE pass over the weights after backward (round 6): the fused SGD kernel sums the slabs of the deferred weight gradients itself and
# leaves the maxima of the updated weights for the weight preparation (csrc/head.hip sgd_fused_kernel).  SEMSEG_SGD_FUSED=0: the
# reduce launch, the plain SGD kernel and the absmax pass of the weight preparation, as before.
SGD_FUSED = os.environ.get('SEMSEG_SGD_FUSED', '1') != '0'
_SLABS_FOR_SGD = {}          # id(parameter) -> (slab tensor, gradient buffer, numel, splits, parameter): summed by sgd_step
_ABSMAX_FRESH = {}           # id(conv weight) -> its torch version when the fused SGD kernel wrote the partial |w| maxima of its planes
_DEFER = [False]
_PENDING_SLABS = []          # (slab tensor, gradient buffer, numel, splits, parameter, producing stream)
_PENDING_WGRADS = []         # (x planes, dy planes, slab tensor, gradient buffer, geometry, parameter, producing stream)
# forward uses of each leaf weight since the last flush, keyed by id(): a weight used at two sites of one graph has its second
# gradient ADDED to the first by autograd -- which would read the first while it is still unreduced -- so only a weight with
# exactly one recorded use is deferred (no record, e.g. a flush between forward and backward: not deferred either)
_FWD_USES = {}


def _note_weight_use(weight, wanted=True):
    """called by the forward of every convolution node (`wanted` = ctx.needs_input_grad of the weight: autograd runs forward()
    with grad mode off, so that -- not torch.is_grad_enabled() -- says whether a backward will come); returns the parameter to
    hand to backward when its gradient is one autograd only accumulates (see _is_leaf_weight), else None"""
    if not (wanted and _is_leaf_weight(weight)):
        return None
    _FWD_USES[id(weight)] = _FWD_USES.get(id(weight), 0) + 1
    return weight


def _may_defer(param):
    """backward-time half of the deferral contract: the gradient buffer this node returns must BECOME param.grad (autograd's
    AccumulateGrad steals it only if .grad is empty and the strides match the parameter's) and nothing else may add to it"""
    return bool(param is not None and param.grad is None and _FWD_USES.get(id(param), 0) == 1)


class defer_wgrad_reduces:
    """flush_at_buckets: the caller runs gradient buckets (parallel.GradientBuckets) that call flush_wgrad_reduces(mid_backward=True)
    from the hook that completes a bucket, BEFORE the bucket is staged and its all-reduce launched -- so a rank gets the batched /
    deferred weight gradients too, bucket by bucket, and no bucket ever travels with an unfinished gradient in it."""

    def __init__(self, flush_at_buckets=False, into_sgd=False):
        """into_sgd: the caller runs ops.sgd_step on every parameter right after backward and nothing reads a gradient in between
        (TrainStep on one rank): the slabs of a deferred gradient are then summed by the SGD kernel itself (semseg_sgd_step_fused:
        same slab order, same bits, the sum left in .grad) instead of by a reduce launch of their own; whatever sgd_step did not
        take is reduced by finish_leftover_slabs()."""
        self.flush_at_buckets = flush_at_buckets
        self.into_sgd = bool(into_sgd) and SGD_FUSED and not flush_at_buckets

    def __enter__(self):
        self.prev = _DEFER[0]
        _DEFER[0] = DEFER_WGRAD_REDUCE and CONV_MODE == 'h2' and (self.flush_at_buckets or not _sync_active())
        return self

    def __exit__(self, *exc):
        _DEFER[0] = self.prev
        if exc and exc[0] is not None:           # backward raised: nothing of the half-built lists is launched, the caller's
            _PENDING_WGRADS[:] = []              # exception is the one that propagates
            _PENDING_SLABS[:] = []
            _FWD_USES.clear()
            return False
        if self.into_sgd:
            flush_wgrad_reduces(into_sgd=True)
        else:
            flush_wgrad_reduces()
        return False


def deferring():
    return _DEFER[0]


def flush_wgrad_reduces(mid_backward=False, into_sgd=False):
    """sum the slabs of every weight gradient deferred since the last flush, one launch per 64 tensors (on the current stream: after
    backward() has returned autograd has made it wait for the streams the gradients were produced on).  mid_backward: called from a
    gradient-bucket hook while backward is still running -- the current stream is first made to wait for every OTHER stream a pending
    operand was produced on (the branch streams of run_branches), and the forward-use counts stay (the rest of the graph has not run)."""
    cur = torch.cuda.current_stream() if torch.cuda.is_available() else None
    if mid_backward and cur is not None:
        for rec in list(_PENDING_WGRADS) + list(_PENDING_SLABS):
            st = rec[-1]
            if st is not None and st != cur:
                cur.wait_stream(st)
    if _PENDING_WGRADS:
        probs, _PENDING_WGRADS[:] = list(_PENDING_WGRADS), []
        parr = (_native.WgradProblem * len(probs))()
        on_dev = probs[0][2].is_cuda
        for i, (xs, dys, slabs, out, geom, _, _) in enumerate(probs):
            q = parr[i]
            q.xs, q.dys, q.slabs, q.slabs_bytes = xs.data_ptr(), dys.data_ptr(), slabs.data_ptr(), slabs.numel() * 4
            q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.dil = geom
            if on_dev:                       # planes produced on a branch stream, read on this one
                xs.record_stream(cur)
                dys.record_stream(cur)
        member = _native.lib().semseg_conv2d_wgrad_member_plan(1 if WGRAD_MEMBER_PLAN else 0)      # the plans the problems were queued under
        try:
            _native.check(_native.lib().semseg_conv2d_wgrad_multi_h2(parr, len(probs), _st()), 'conv2d_wgrad_multi_h2')
        finally:
            _native.lib().semseg_conv2d_wgrad_member_plan(member)
        for i, (xs, dys, slabs, out, geom, param, st) in enumerate(probs):
            _PENDING_SLABS.append((slabs, out, geom[4] * geom[5] * geom[6] * geom[3], int(parr[i].splits), param, st))
    if not mid_backward:
        _FWD_USES.clear()
    if not _PENDING_SLABS:
        return
    items, _PENDING_SLABS[:] = list(_PENDING_SLABS), []
    if mid_backward:
        # called from the hook that completed ONE gradient bucket: autograd orders a weight's AccumulateGrad against those of its
        # sibling BN parameters only by an unspecified tie-break, so an entry whose parameter has not received its gradient yet
        # belongs to a bucket that is still open -- it stays pending for the flush of that bucket (or the final one), instead
        # of failing the adoption check below before autograd had its chance
        later = [it for it in items if it[4] is not None and it[4].grad is None]
        if later:
            items = [it for it in items if not (it[4] is not None and it[4].grad is None)]
            _PENDING_SLABS.extend(later)
            if not items:
                return
    for slabs, out, numel, splits, param, _ in items:
        # the contract checked where it is cheap (once per eager step / capture pass): the buffer being completed below IS
        # the parameter's gradient.  Anything else -- autograd copied or accumulated the unreduced buffer -- is garbage already.
        if param is not None and (param.grad is None or param.grad.data_ptr() != out.data_ptr()):
            raise RuntimeError('deferred weight gradient of a %s parameter did not become its .grad (layout %s, grad %s): '
                               'set SEMSEG_DEFER_WGRAD_REDUCE=0 or keep conv weights as KRSC leaves used once per step'
                               % (tuple(param.shape), param.stride(), 'missing' if param.grad is None else 'copied'))
    if into_sgd and not mid_backward:
        # the gradients of leaf parameters wait for the SGD kernel (sgd_step); a buffer without a parameter behind it is summed now
        for it in items:
            if it[4] is not None:
                _SLABS_FOR_SGD[id(it[4])] = it[:5]
        items = [it for it in items if it[4] is None]
        if not items:
            return
    arr = (_native.SlabTensor * len(items))()
    on_dev = items[0][0].is_cuda
    for i, (slabs, out, numel, splits, _, _) in enumerate(items):
        arr[i].slabs, arr[i].out, arr[i].numel, arr[i].splits = slabs.data_ptr(), out.data_ptr(), numel, splits
        if on_dev:                           # slabs / gradient allocated on a branch stream, summed on this one
            slabs.record_stream(cur)
            out.record_stream(cur)
    _native.check(_native.lib().semseg_reduce_slabs_multi(arr, len(items), _st()), 'reduce_slabs_multi')


def _is_leaf_weight(weight):
    """a weight whose gradient autograd only ACCUMULATES (into .grad, after this node) -- the one case in which the weight gradient
    may be completed after backward() has returned.  A weight computed from parameters (GroupedConv2d's block-diagonal expansion)
    has its gradient read by the next autograd node right away."""
    if not (weight.requires_grad and weight.is_leaf and weight.grad_fn is None):
        return False
    # ... and whose memory is KRSC (layers.Conv2d keeps it so; any layout when R = S = 1): only then do the strides of the
    # returned [K, C, R, S] view of the KRSC gradient buffer match the parameter's and AccumulateGrad adopts the buffer itself.
    # A KCRS-contiguous nn.Parameter would get a COPY, taken before the deferred launches have run.
    return bool(weight.dim() == 4 and weight.permute(0, 2, 3, 1).is_contiguous())


def _producer_stream(t):
    return torch.cuda.current_stream(t.device) if t.is_cuda else None


def _split_conv_grads(L, sch, scheme, geom, xs, dys, w, wtp, need_dx, need_dw, param=None):
    """Data and weight gradient of a split convolution from the planes of the input (xs) and of dy (dys).  param: the leaf
    parameter behind w (_note_weight_use) or None -- inside defer_wgrad_reduces() the gradient of a parameter that passes
    _may_defer may be finished by flush_wgrad_reduces."""
    n, h, wd, c, k, r, s, stride, pad, dil = geom
    dev = w.device
    dx = dw = None
    if need_dw:
        dwb = torch.empty((k, r, s, c), device=dev, dtype=torch.float32)

        def launch_w():
            ws = workspace(sch.fn(L, 'workspace_bytes')(*geom), dev)
            _native.check(sch.fn(L, 'wgrad')(_p(xs), _p(dys), _p(dwb), *geom, _p(ws), ws.numel(), _st()),
                          'conv2d_wgrad_' + scheme)
        tuner.ensure(scheme, 2, geom, launch_w)        # candidates are timed WITH their reduce: what a plan costs either way
        if _DEFER[0] and scheme == 'h2' and _may_defer(param):
            # member mode: a small weight gradient whose own plan is an LDS-DMA tile is read as the 64 x 64 tile the batched launch
            # takes (same number of blocks) -- for these calls only, never while the tuner times a plan (tuner.ensure above)
            member = L.semseg_conv2d_wgrad_member_plan(1 if (DEFER_WGRAD_LAUNCH and WGRAD_MEMBER_PLAN) else 0)
            try:
                nbytes = L.semseg_conv2d_wgrad_slabs_bytes(*geom)
                slabs = torch.empty((max(16, nbytes) + 3) // 4, device=dev, dtype=torch.float32)
                if DEFER_WGRAD_LAUNCH and L.semseg_conv2d_wgrad_tile_h2(*geom) == 1:
                    _PENDING_WGRADS.append((xs, dys, slabs, dwb, tuple(geom), param, _producer_stream(dwb)))
                else:
                    splits = ctypes.c_int(0)
                    _native.check(L.semseg_conv2d_wgrad_slabs_h2(_p(xs), _p(dys), _p(slabs), slabs.numel() * 4, ctypes.byref(splits),
                                                                 *geom, _st()), 'conv2d_wgrad_slabs_h2')
                    _PENDING_SLABS.append((slabs, dwb, k * r * s * c, int(splits.value), param, _producer_stream(dwb)))
            finally:
                L.semseg_conv2d_wgrad_member_plan(member)
        else:
            launch_w()
        dw = dwb.permute(0, 3, 1, 2)
    if need_dx:
        wts = wtp if wtp is not None else _weight_crsk_planes(L, sch, w, dev)
        dx = empty_nhwc(n, c, h, wd, dev)

        def launch_d():
            ws = workspace(sch.fn(L, 'workspace_bytes')(*geom), dev)
            _native.check(sch.fn(L, 'dgrad')(_p(dys), _p(wts), _p(dx), c, *geom, _p(ws), ws.numel(), _st()),
                          'conv2d_dgrad_' + scheme)
        tuner.ensure(scheme, 1, geom, launch_d)
        launch_d()
    return dx, dw


# dedicated depthwise 3x3 kernels (csrc/depthwise.hip) instead of the block-diagonal dense expansion of
# models.layers.GroupedConv2d (960-fold redundant MFMA work for mobilenet.py:48's widest layer).  Default since they passed
# float64 parity on the MI355X (tests/test_gpu_depthwise.py) and the MobileNetV2 goldens; SEMSEG_DEPTHWISE_DIRECT=0 restores
# the dense expansion.
DEPTHWISE_DIRECT = os.environ.get('SEMSEG_DEPTHWISE_DIRECT', '1') == '1'


class DepthwiseConv3x3Fn(Function):
    """nn.Conv2d(C, C, 3, stride, padding, dilation, groups=C, bias=False) (mobilenet.py:48,60): weight [C, 1, 3, 3]."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, dil):
        L = _native.lib()
        x, x_ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        if tuple(weight.shape) != (c, 1, 3, 3) or c % 4:
            raise RuntimeError('depthwise3x3: expected a [C, 1, 3, 3] weight with C %% 4 == 0, got %s for C = %d'
                               % (tuple(weight.shape), c))
        _require_cuda(weight)
        wt = weight.detach().reshape(c, 9).t().contiguous()                  # tap-major [9][C]
        oh, ow = conv_out_size(h, 3, stride, pad, dil), conv_out_size(w, 3, stride, pad, dil)
        y = empty_nhwc(n, c, oh, ow, x.device)
        _native.check(L.semseg_depthwise3x3_fwd(_p(x), x_ld, _p(wt), _p(y), c, n, h, w, c, stride, pad, dil, _st()),
                      'depthwise3x3_fwd')
        ctx.save_for_backward(x, wt)
        ctx.geom = (n, h, w, c, stride, pad, dil)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        L = _native.lib()
        x, wt = ctx.saved_tensors
        n, h, w, c, stride, pad, dil = ctx.geom
        _, x_ld = as_nhwc(x)
        dy, dy_ld = as_nhwc(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(n, c, h, w, x.device)
            _native.check(L.semseg_depthwise3x3_dgrad(_p(dy), dy_ld, _p(wt), _p(dx), c, n, h, w, c, stride, pad, dil, _st()),
                          'depthwise3x3_dgrad')
        if ctx.needs_input_grad[1]:
            dwt = torch.empty((9, c), device=x.device, dtype=torch.float32)
            ws = workspace(L.semseg_depthwise3x3_workspace_bytes(n, h, w, c, stride, pad, dil), x.device)
            _native.check(L.semseg_depthwise3x3_wgrad(_p(x), x_ld, _p(dy), dy_ld, _p(dwt), n, h, w, c, stride, pad, dil, _p(ws),
                                                      ws.numel(), _st()), 'depthwise3x3_wgrad')
            dw = dwt.t().reshape(c, 1, 3, 3)
        return dx, dw, None, None, None


def depthwise_conv3x3(x, weight, stride=1, padding=1, dilation=1):
    _require_cuda(x)
    return DepthwiseConv3x3Fn.apply(x, weight, int(stride), int(padding), int(dilation))


GROUPED_DIRECT = os.environ.get('SEMSEG_GROUPED_DIRECT', '1') == '1'       # csrc/grouped.hip; 0: dense expansion


class GroupedConv3x3Fn(Function):
    """nn.Conv2d(C, K, 3, stride, padding, dilation, groups=g, bias=False) (resnext.py:30-31): weight [K, C/g, 3, 3]."""

    @staticmethod
    def forward(ctx, x, weight, groups, stride, pad, dil):
        L = _native.lib()
        x, x_ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        k, cg = int(weight.shape[0]), int(weight.shape[1])
        if tuple(weight.shape[2:]) != (3, 3) or cg * groups != c or k % groups or cg % 4 or (k // groups) % 4:
            raise RuntimeError('grouped3x3: weight %s does not fit C = %d, groups = %d (channels per group must be multiples of 4)'
                               % (tuple(weight.shape), c, groups))
        _require_cuda(weight)
        wt = weight.detach().permute(0, 2, 3, 1).reshape(k, 9, cg).contiguous()
        oh, ow = conv_out_size(h, 3, stride, pad, dil), conv_out_size(w, 3, stride, pad, dil)
        y = empty_nhwc(n, k, oh, ow, x.device)
        _native.check(L.semseg_grouped3x3_fwd(_p(x), x_ld, _p(wt), _p(y), k, n, h, w, c, k, groups, stride, pad, dil, _st()),
                      'grouped3x3_fwd')
        ctx.save_for_backward(x, wt)
        ctx.geom = (n, h, w, c, k, groups, stride, pad, dil)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        L = _native.lib()
        x, wt = ctx.saved_tensors
        n, h, w, c, k, groups, stride, pad, dil = ctx.geom
        _, x_ld = as_nhwc(x)
        dy, dy_ld = as_nhwc(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(n, c, h, w, x.device)
            _native.check(L.semseg_grouped3x3_dgrad(_p(dy), dy_ld, _p(wt), _p(dx), c, n, h, w, c, k, groups, stride, pad, dil,
                                                    _st()), 'grouped3x3_dgrad')
        if ctx.needs_input_grad[1]:
            cg = c // groups
            dwt = torch.empty((k, 9, cg), device=x.device, dtype=torch.float32)
            ws = workspace(L.semseg_grouped3x3_workspace_bytes(n, h, w, c, k, groups, stride, pad, dil), x.device)
            _native.check(L.semseg_grouped3x3_wgrad(_p(x), x_ld, _p(dy), dy_ld, _p(dwt), n, h, w, c, k, groups, stride, pad, dil,
                                                    _p(ws), ws.numel(), _st()), 'grouped3x3_wgrad')
            dw = dwt.reshape(k, 3, 3, cg).permute(0, 3, 1, 2)
        return dx, dw, None, None, None, None


def grouped_conv3x3(x, weight, groups, stride=1, padding=1, dilation=1):
    _require_cuda(x)
    return GroupedConv3x3Fn.apply(x, weight, int(groups), int(stride), int(padding), int(dilation))


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1):
    if CONV_MODE == 'f32':
        return Conv2dFn.apply(x, weight, bias, int(stride), int(padding), int(dilation))
    _require_cuda(x)
    xp = input_planes(x, CONV_MODE)
    wp, wtp = weight_planes(weight, CONV_MODE)
    return Conv2dSplitFn.apply(x, weight, bias, xp, wp, wtp, int(stride), int(padding), int(dilation), CONV_MODE)


# ------------------------------------------------------------------------------------------------
# batch norm (+ residual add + ReLU), optional cross-rank statistics
# ------------------------------------------------------------------------------------------------
_SYNC_GROUP = {'group': None, 'enabled': False, 'force': os.environ.get('SEMSEG_FORCE_SYNC_PATH', '0') == '1'}
# engine.SegmentedStep while its capture pass runs: a collective then ENDS the hipGraph segment being captured, is recorded and
# the next segment begins -- at replay the collectives are issued eagerly between the segment launches
_SEGMENTS = None
PEER_FUSED = os.environ.get('SEMSEG_PEER_FUSED', '1') != '0'


def set_sync_bn_group(group, enabled=True):
    """Enable SyncBN: statistics are all-reduced (RCCL) over `group` between bn_stats and bn_finalize --
    the one-process-per-GPU replacement of reference batchnorm.py:63-117 / comm.py."""
    _SYNC_GROUP['group'] = group
    _SYNC_GROUP['enabled'] = enabled


def _sync_active():
    """True when BN statistics have to be all-reduced across ranks (SyncBN on a world of more than one rank)."""
    if _SYNC_GROUP['force']:          # tests: the unfused SyncBN kernel sequence (and segment boundaries) on a single rank
        return True
    if not _SYNC_GROUP['enabled']:
        return False
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(_SYNC_GROUP['group']) > 1


def _sync_peer(doubles):
    """the peer-exchange context (csrc/peer.hip) that serves the SyncBN group, if it is up and carries `doubles` per exchange;
    SEMSEG_PEER_FUSED=0 keeps the exchange a kernel of its own between the unfused BN entry points"""
    if not PEER_FUSED or not _SYNC_GROUP['enabled']:
        return None
    from . import comm
    return comm.peer_handle(_SYNC_GROUP['group'], doubles)


def allreduce_sum(buf, group=None, channel='sync'):
    """sum of `buf` over the ranks of `group`, in place, on the current stream: the native RCCL communicator of `channel` when
    one is up (mit_semseg.comm: semseg_comm_allreduce_sum through the C ABI), else torch.distributed"""
    from . import comm
    if comm.allreduce_sum(buf, group, channel):
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)


def _maybe_allreduce(buf):
    if _SEGMENTS is not None:
        if _sync_active():
            from . import comm
            if comm.peer_active(_SYNC_GROUP['group']) and comm.peer_allreduce_sum(buf, _SYNC_GROUP['group']):
                return          # the peer exchange is an ordinary kernel: it stays INSIDE the segment being captured
            _SEGMENTS.collective('allreduce', buf, _SYNC_GROUP['group'])
        return
    if _SYNC_GROUP['enabled']:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(_SYNC_GROUP['group']) > 1:
            allreduce_sum(buf, _SYNC_GROUP['group'])


class BatchNormActFn(Function):
    """y = act(BN(z) [+ residual]); training: batch statistics + running-stat EMA; eval: running stats."""

    @staticmethod
    def forward(ctx, z, gamma, beta, running_mean, running_var, residual, training, momentum, eps, relu, nbt):
        L = _native.lib()
        z, z_ld = as_nhwc_of(z)
        n, c, h, w = z.shape
        if z_ld != c:
            z = z.contiguous(memory_format=torch.channels_last)
        _require_cuda(gamma, beta, running_mean, running_var)
        P = n * h * w
        dev = z.device
        coef = torch.empty((4, c), device=dev, dtype=torch.float32)   # mean, invstd, scale, shift
        stats = None
        g, b = gamma.detach(), beta.detach()
        if training:
            if P <= 1:
                raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                                 % str(list(z.shape)))
            stats = torch.empty((2 * c + 1,), device=dev, dtype=torch.float64)
            ws = workspace(L.semseg_bn_workspace_bytes(P, c), dev)
            _native.check(L.semseg_bn_stats(_p(z), P, c, _p(stats), _p(ws), ws.numel(), _st()), 'bn_stats')
            _maybe_allreduce(stats)
            _native.check(L.semseg_bn_finalize(_p(stats), c, _p(g), _p(b), _p(running_mean), _p(running_var), _p(nbt),
                                               float(momentum), float(eps), _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                               _p(coef[3]), _st()), 'bn_finalize')
        else:
            _native.check(L.semseg_bn_eval_coeffs(_p(g), _p(b), _p(running_mean), _p(running_var), float(eps), c,
                                                  _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _st()),
                          'bn_eval_coeffs')
        res, res_ld = (None, 0)
        if residual is not None:
            res, res_ld = as_nhwc_of(residual)
        y = empty_nhwc(n, c, h, w, dev)
        _native.check(L.semseg_bn_apply(_p(z), _p(coef[2]), _p(coef[3]), _p(res), res_ld, int(relu), _p(y), c, P, c,
                                        _st()), 'bn_apply')
        ctx.save_for_backward(z, y if relu else None, coef, g, stats)
        ctx.cfg = (bool(training), bool(relu), residual is not None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        L = _native.lib()
        z, y, coef, gamma, stats = ctx.saved_tensors
        training, relu, has_res = ctx.cfg
        n, c, h, w = z.shape
        P = n * h * w
        dev = z.device
        dy, dy_ld = as_nhwc(dy)
        need_param = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        sums = torch.empty((2 * c,), device=dev, dtype=torch.float64)
        dgamma = torch.empty((c,), device=dev, dtype=torch.float32)
        dbeta = torch.empty((c,), device=dev, dtype=torch.float32)
        if training or need_param:
            ws = workspace(L.semseg_bn_workspace_bytes(P, c), dev)
            _native.check(L.semseg_bn_bwd_reduce(_p(dy), dy_ld, _p(y), c, _p(z), _p(coef[0]), _p(coef[1]), int(relu),
                                                 P, c, _p(sums), _p(dgamma), _p(dbeta), _p(ws), ws.numel(), _st()),
                          'bn_bwd_reduce')
            if training:
                _maybe_allreduce(sums)
        dz = empty_nhwc(n, c, h, w, dev)
        dres = empty_nhwc(n, c, h, w, dev) if (has_res and ctx.needs_input_grad[5]) else None
        count = stats[2 * c:] if stats is not None else None
        _native.check(L.semseg_bn_bwd_apply(_p(dy), dy_ld, _p(y), c, _p(z), _p(coef[0]), _p(coef[1]), _p(gamma),
                                            _p(sums), _p(count), int(training), int(relu), _p(dz), _p(dres), P, c,
                                            _st()), 'bn_bwd_apply')
        return (dz, dgamma if ctx.needs_input_grad[1] else None, dbeta if ctx.needs_input_grad[2] else None,
                None, None, dres, None, None, None, None, None)


def batch_norm_act(z, gamma, beta, running_mean, running_var, residual=None, training=False, momentum=0.1,
                   eps=1e-5, relu=False, num_batches_tracked=None):
    """`num_batches_tracked` (int64 0-dim buffer) is incremented by the finalize kernel in training mode."""
    return BatchNormActFn.apply(z, gamma, beta, running_mean, running_var, residual, bool(training),
                                float(momentum), float(eps), bool(relu), num_batches_tracked)


def _winograd_fwd(L, x, bounds, u_planes, z, geom):
    """z = conv3x3(x) (stride 1, pad == dil) by Winograd F(2x2, 3x3) on h2 planes: input transform -> one batched GEMM
    launch over the 16 frequencies -> output transform (csrc/winograd.hip)."""
    n, h, wd, c, k, r, s, stride, pad, dil = geom
    x, x_ld = as_nhwc(x)
    dev = x.device
    tiles = L.semseg_winograd_tiles(n, h, wd, dil)
    v = torch.empty(L.semseg_split_h2_bytes(16 * tiles, c), dtype=torch.uint8, device=dev)
    m = torch.empty((16 * tiles, k), dtype=torch.float32, device=dev)
    nb = len(bounds)
    bp = (vp * nb)(*[b.data_ptr() for b in bounds])
    _native.check(L.semseg_winograd_input_h2(_p(x), x_ld, bp, nb, _p(v), n, h, wd, c, dil, _st()), 'winograd_input_h2')

    def gemm():
        _native.check(L.semseg_winograd_gemm_h2(_p(v), _p(u_planes), _p(m), tiles, c, k, _st()), 'winograd_gemm_h2')
    tuner.ensure_winograd_gemm(tiles, c, k, gemm)
    gemm()
    _native.check(L.semseg_winograd_output(_p(m), _p(z), k, n, h, wd, k, dil, _st()), 'winograd_output')
    return v


def _winograd_wgrad(L, v, dzp, geom):
    """dw of the same convolution from the forward's V planes and the h2 planes of dz: dM = A dz A^T (planes), one batched
    launch dU[f] = dM[f]^T V[f], dw = G^T dU G.  Returns dw as [K, C, 3, 3] (a view of the KRSC buffer)."""
    n, h, wd, c, k, r, s, stride, pad, dil = geom
    dev = v.device
    tiles = L.semseg_winograd_tiles(n, h, wd, dil)
    dm = torch.empty(L.semseg_split_h2_bytes(16 * tiles, k), dtype=torch.uint8, device=dev)
    du = torch.empty((16, k, c), dtype=torch.float32, device=dev)
    dwb = torch.empty((k, r, s, c), device=dev, dtype=torch.float32)
    ws = workspace(L.semseg_winograd_wgrad_workspace_bytes(tiles, c, k), dev)
    _native.check(L.semseg_winograd_dm_h2(_p(dzp), _p(dm), n, h, wd, k, dil, _st()), 'winograd_dm_h2')
    _native.check(L.semseg_winograd_wgrad_gemm_h2(_p(v), _p(dm), _p(du), tiles, c, k, _p(ws), ws.numel(), _st()),
                  'winograd_wgrad_gemm_h2')
    _native.check(L.semseg_winograd_dg(_p(du), _p(dwb), k, c, _st()), 'winograd_dg')
    return dwb.permute(0, 3, 1, 2)


# The GEMM over the 16 frequencies and the output transform of the Winograd data gradient as ONE kernel
# (semseg_winograd_gemm_output_h2, csrc/conv_split.hip wino_fused_kernel): no fp32 intermediate M (16 x tiles x C floats written
# and read back by the unfused pair).  Which form runs is measured per geometry (tuner.choose); SEMSEG_WINOGRAD_FUSED=0 keeps
# the batched GEMM + output transform everywhere.
WINOGRAD_FUSED = os.environ.get('SEMSEG_WINOGRAD_FUSED', '1') != '0'
WINOGRAD_FUSED_FORMS = 10         # library forms of the fused kernel: 32-deep k-tiles (8 waves on a 3- / 4- / 5-slot ring, 4 waves on 4 / 5
                                  # slots), 64-deep k-tiles with full-line DMA pieces on a half-tile ring (8 / 4 waves: 5 / 6), round 5: 7 =
                                  # form 5 with its DMA pieces spread behind the MFMA groups, 8 / 9 = 7 / 5 with s_setprio (form 6 spread: measured, never chosen, removed)


def _winograd_dgrad(L, dzp, ut_planes, geom, form=None):
    """dx of a 3x3 stride-1 pad == dil convolution in the Winograd domain, from the h2 planes of dz: the convolution of dz (k
    channels) with the flipped, transposed weights (U' planes, csrc/weights_prep.hip) -- input transform from planes, then
    either the batched GEMM of the forward with the roles of c and k swapped + the output transform into dx (form 0), or both in
    the fused kernel (form 1 + library form).  form None: the fastest on this geometry (timed once, tuner.choose)."""
    n, h, wd, c, k, r, s, stride, pad, dil = geom
    dev = dzp.device
    tiles = L.semseg_winograd_tiles(n, h, wd, dil)
    v = torch.empty(L.semseg_split_h2_bytes(16 * tiles, k), dtype=torch.uint8, device=dev)
    dx = empty_nhwc(n, c, h, wd, dev)
    _native.check(L.semseg_winograd_input_planes_h2(_p(dzp), _p(v), n, h, wd, k, dil, _st()), 'winograd_input_planes_h2')
    box = {}

    def unfused():
        if 'm' not in box:
            box['m'] = torch.empty((16 * tiles, c), dtype=torch.float32, device=dev)
        m = box['m']

        def gemm():
            _native.check(L.semseg_winograd_gemm_h2(_p(v), _p(ut_planes), _p(m), tiles, k, c, _st()), 'winograd_gemm_h2')
        tuner.ensure_winograd_gemm(tiles, k, c, gemm)
        gemm()
        _native.check(L.semseg_winograd_output(_p(m), _p(dx), c, n, h, wd, c, dil, _st()), 'winograd_output')

    def fused(lib_form):
        def run():
            _native.check(L.semseg_winograd_gemm_output_h2(_p(v), _p(ut_planes), _p(dx), c, n, h, wd, k, c, dil, lib_form, _st()),
                          'winograd_gemm_output_h2')
        return run
    cands = [unfused]
    # the fused kernel has one block per 128 tiles x 128 channels for ALL frequencies: it needs enough of them to fill the chip
    if WINOGRAD_FUSED and ((tiles + 127) // 128) * ((c + 127) // 128) >= 96 and (n * h * wd * c) < (1 << 31):
        cands += [fused(i) for i in range(WINOGRAD_FUSED_FORMS)]
    if form is None:
        form = tuner.choose((tiles, n, dil, c, k, 3, 3, 1, h, wd), cands) if len(cands) > 1 else 0
    cands[form if form < len(cands) else 0]()
    box.clear()
    return dx


class ConvBNActFn(Function):
    """y = act(BN_train(conv(x)) [+ residual]) as ONE autograd node on the h2 path (resnet.py:72-92 blocks,
    models.py:160-167 conv3x3_bn_relu, hrnet.py).  Versus Conv2dSplitFn + BatchNormActFn:
      * the BN statistics pass also gathers per-channel min/max of z, which bound |y| rigorously -> the exponent of y's
        split planes is known before y is written and bn_apply emits y's planes itself (no absmax + split passes in the
        next conv);
      * backward: the BN gradient is written ONLY as split planes (its sole consumers are the conv gradients), with the
        exponent bounded from the reduction sums (csrc/bn.hip, second half).
    The planes of y and the |y| bound leave forward through `box` (a dict), not as autograd outputs: autograd would
    materialise a zero gradient of their size in every backward."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, residual, xp, wp, wtp, res_absmax, running_mean, running_var, nbt, cfg, box):
        wut = box.get('wino_t')             # U' planes of the Winograd data gradient (None: direct kernel)
        _native.next_unit()                 # inside a side-by-side scope: the branches pair up their launches unit by unit
        stride, pad, dil, momentum, eps, relu, emit = cfg[:7]
        planes_only = len(cfg) > 7 and cfg[7]
        box['planes_only'] = False
        L = _native.lib()
        sch = SCHEMES['h2']
        w = krsc(weight.detach())
        g, b = gamma.detach(), beta.detach()
        _require_cuda(w, g, b, running_mean, running_var)
        n, c, h, wd = x.shape
        k, c2, r, s = w.shape
        if c2 != c:
            raise RuntimeError('conv2d: input has %d channels, weight expects %d' % (c, c2))
        oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
        P = n * oh * ow
        if P <= 1:
            raise ValueError('Expected more than 1 value per channel when training, got input size %s'
                             % str([n, k, oh, ow]))
        geom = (n, h, wd, c, k, r, s, stride, pad, dil)
        dev = x.device
        z = empty_nhwc(n, k, oh, ow, dev)
        wino = box.get('wino')
        wino_v = None
        if wino is not None:
            wino_v = _winograd_fwd(L, x.detach(), box['x_bounds'], wino, z, geom)
            if not (WINOGRAD_WGRAD and c >= WINOGRAD_WGRAD_MIN_C and ctx.needs_input_grad[1]):
                wino_v = None
        res, res_ld = (None, 0)
        if residual is not None:
            res, res_ld = as_nhwc_of(residual)
        bound_ok = residual is None or res_absmax is not None
        absmax = torch.empty((1,), device=dev, dtype=torch.float32) if bound_ok else None
        yp = torch.empty(L.semseg_split_h2_bytes(P, k), dtype=torch.uint8, device=dev) if (emit and bound_ok) else None
        sync = _sync_active()
        peer = _sync_peer(2 * k + 1) if sync else None
        fused_stats = (not sync or peer is not None) and (yp is not None or absmax is not None)
        bound = None if yp is not None else absmax            # no planes: the bound of |y| comes from the finish kernel itself
        parts = ctypes.c_int(0)
        stats_ws = None
        if wino is None:
            wsp = wp if wp is not None else sch.split(w, k * r * s, c, c)
            if fused_stats and EPILOGUE_STATS:
                # the BN statistics of z are gathered in the GEMM epilogue (one partial row per block row tile; csrc/conv_split.hip
                # gemm_epilogue) when the launch plan of this geometry does not split the reduction: no sweep over z for them
                stats_bytes = L.semseg_conv2d_fwd_stats_bytes(k)

                def launch():
                    # the split-K need follows the plan in force (the tuner pins one candidate after the other around this call):
                    # sized per launch, and the partials' place is remembered from the launch that counts -- the last one
                    conv_bytes = (sch.fn(L, 'workspace_bytes')(*geom) + 255) & ~255
                    ws = workspace(conv_bytes + stats_bytes, dev)
                    base = ws.data_ptr()
                    _native.check(L.semseg_conv2d_fwd_stats_h2(_p(xp), _p(wsp), _p(z), k, *geom, vp(base), conv_bytes,
                                                               vp(base + conv_bytes), stats_bytes, _p(bound), ctypes.byref(parts),
                                                               _st()), 'conv2d_fwd_stats_h2')
                    if parts.value == 0 and tuner.timing():
                        # a candidate that splits the reduction pays the statistics sweep on top: part of what the tuner compares
                        _native.check(L.semseg_bn_stats_mm_partial(_p(z), P, k, vp(base + conv_bytes), stats_bytes, _st()),
                                      'bn_stats_mm_partial')
                    stats_owner[0] = ws          # the tensor the partials live in stays referenced until the finish kernel is issued
                    return base + conv_bytes
                stats_owner = [None]
                tuner.ensure('h2', 0, geom, launch)
                stats_ws = launch()
            else:
                def launch():
                    ws = workspace(sch.fn(L, 'workspace_bytes')(*geom), dev)
                    _native.check(sch.fn(L, 'fwd')(_p(xp), _p(wsp), _p(None), _p(z), k, *geom, _p(ws), ws.numel(), _st()),
                                  'conv2d_fwd_h2')
                tuner.ensure('h2', 0, geom, launch)
                launch()
        stats = torch.empty((2 * k + 1,), device=dev, dtype=torch.float64)
        zmm = torch.empty((2 * k,), device=dev, dtype=torch.float32)
        ws = workspace(L.semseg_bn_mm_workspace_bytes(P, k), dev)
        coef = torch.empty((4, k), device=dev, dtype=torch.float32)   # mean, invstd, scale, shift
        y = empty_nhwc(n, k, oh, ow, dev)
        gate = None
        if fused_stats:
            # one rank: finish + finalize in one kernel; the apply kernel derives the exponent from the per-block bounds.
            # SyncBN over the peer exchange: the same kernel exchanges its sums with the other ranks on the way (csrc/peer_dev.h)
            bb = torch.empty(((k + 15) // 16,), device=dev, dtype=torch.int32)
            tail = (_p(g), _p(b), _p(running_mean), _p(running_var), _p(nbt), float(momentum), float(eps), int(relu),
                    _p(res_absmax), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _p(bb))
            if parts.value > 0:
                _native.check(L.semseg_bn_fwd_finish_fused(vp(stats_ws), L.semseg_conv2d_fwd_stats_bytes(k), parts.value, P, k,
                                                           _p(stats), _p(zmm), *tail, _st(), peer, _p(bound)),
                              'bn_fwd_finish_fused')
            else:
                args = (_p(z), P, k, _p(stats), _p(zmm)) + tail + (_p(ws), ws.numel(), _st())
                if peer is not None:
                    _native.check(L.semseg_bn_fwd_stats_fused_peer(*args, peer, _p(bound)), 'bn_fwd_stats_fused_peer')
                elif bound is not None:
                    _native.check(L.semseg_bn_fwd_stats_fused_bound(*args, _p(bound)), 'bn_fwd_stats_fused_bound')
                else:
                    _native.check(L.semseg_bn_fwd_stats_fused(*args), 'bn_fwd_stats_fused')
            if yp is not None and relu and residual is not None:
                # backward's gate of a BN with residual is (y > 0): leave it as 1 bit per element instead of reading y twice
                gate = torch.empty((P * (k // 8),), device=dev, dtype=torch.uint8)
                _native.check(L.semseg_bn_apply_h2_gate(_p(z), _p(coef[2]), _p(coef[3]), _p(res), res_ld, int(relu), _p(y),
                                                        _p(yp), P, k, _p(bb), _p(absmax), _st(), _p(gate)), 'bn_apply_h2_gate')
            elif yp is not None:
                # planes_only: the caller vouches that the next convolution's planes are the ONLY consumer of y (bn1 / bn2 of a
                # bottleneck): the fp32 copy is not written (4 of the 12 bytes per element this kernel moves)
                skip_y = planes_only and residual is None
                box['planes_only'] = skip_y
                if skip_y and PLANES_ONLY_POISON:         # tests: a reader that slipped past the mark sees NaN, not stale memory
                    y.fill_(float('nan'))
                _native.check(L.semseg_bn_apply_h2(_p(z), _p(coef[2]), _p(coef[3]), _p(res), res_ld, int(relu),
                                                   _p(None) if skip_y else _p(y), _p(yp),
                                                   P, k, _p(bb), _p(absmax), _st()), 'bn_apply_h2')
            else:
                _native.check(L.semseg_bn_apply(_p(z), _p(coef[2]), _p(coef[3]), _p(res), res_ld, int(relu), _p(y), k, P, k,
                                                _st()), 'bn_apply')
        else:
            _native.check(L.semseg_bn_stats_mm(_p(z), P, k, _p(stats), _p(zmm), _p(ws), ws.numel(), _st()), 'bn_stats_mm')
            _maybe_allreduce(stats)
            _native.check(L.semseg_bn_finalize_mm(_p(stats), _p(zmm), k, _p(g), _p(b), _p(running_mean), _p(running_var),
                                                  _p(nbt), float(momentum), float(eps), int(relu), _p(res_absmax),
                                                  _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]), _p(absmax), _p(yp), P,
                                                  _st()), 'bn_finalize_mm')
            if yp is not None:
                _native.check(L.semseg_bn_apply_h2(_p(z), _p(coef[2]), _p(coef[3]), _p(res), res_ld, int(relu), _p(y),
                                                   _p(yp), P, k, _p(None), _p(None), _st()), 'bn_apply_h2')
            else:
                _native.check(L.semseg_bn_apply(_p(z), _p(coef[2]), _p(coef[3]), _p(res), res_ld, int(relu), _p(y), k, P, k,
                                                _st()), 'bn_apply')
        # the ReLU gate of a BN without residual is recomputed from z in backward: y need not be kept for it
        keep_y = relu and residual is not None and gate is None
        ctx.save_for_backward(xp, w, wtp, z, y if keep_y else None, coef, g, stats, zmm, wino_v, gate, wut)
        ctx.geom = geom
        ctx.cfg = (bool(relu), residual is not None)
        ctx.w_param = _note_weight_use(weight, ctx.needs_input_grad[1])
        box['planes'], box['absmax'] = yp, absmax
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        L = _native.lib()
        _native.next_unit()
        sch = SCHEMES['h2']
        xp, w, wtp, z, y, coef, gamma, stats, zmm, wino_v, gate_bits, wut = ctx.saved_tensors
        relu, has_res = ctx.cfg
        geom = ctx.geom
        n, h, wd, c, k, r, s, stride, pad, dil = geom
        oh, ow = conv_out_size(h, r, stride, pad, dil), conv_out_size(wd, s, stride, pad, dil)
        P = n * oh * ow
        dev = w.device
        sync = _sync_active()
        peer = _sync_peer(2 * k + 1) if sync else None
        # the gradient may arrive as TWO addends (a fork's sum left to its consumer, defer_fork_sums): the fused BN backward kernels
        # add them where they read them; the unfused SyncBN path below reads dy through entry points without a second pointer
        dy2 = _take_addend(dy) if (k % 8 == 0 and (not sync or peer is not None)) else None
        dy2_ld = 0
        if dy2 is not None:
            dy, dy_ld = _as_nhwc_plain(dy)                # the pair is formed by the kernels below
            dy2, dy2_ld = _as_nhwc_plain(dy2)
        else:
            dy, dy_ld = as_nhwc(dy)
        sums = torch.empty((2 * k,), device=dev, dtype=torch.float64)
        gmax = torch.empty((k,), device=dev, dtype=torch.float32)
        dgamma = torch.empty((k,), device=dev, dtype=torch.float32)
        dbeta = torch.empty((k,), device=dev, dtype=torch.float32)
        ws = workspace(L.semseg_bn_mm_workspace_bytes(P, k), dev)
        count = stats[2 * k:]
        dzp = torch.empty(L.semseg_split_h2_bytes(P, k), dtype=torch.uint8, device=dev)
        dres = empty_nhwc(n, k, oh, ow, dev) if (has_res and ctx.needs_input_grad[4]) else None
        gate = relu and not has_res                       # ReLU gate from z (forward's own fmaf), y was not saved
        gsc, gsh = (coef[2], coef[3]) if gate else (None, None)
        y_arg, y_ld = (_p(gate_bits), 0) if gate_bits is not None else (_p(y), k)      # y_ld 0: the forward's ReLU bitmask
        if not sync or peer is not None:
            bb = torch.empty(((k + 15) // 16,), device=dev, dtype=torch.int32)
            args = (_p(dy), dy_ld, y_arg, y_ld, _p(z), _p(coef[0]), _p(coef[1]), _p(gsc), _p(gsh), int(relu), P, k, _p(count),
                    _p(zmm), _p(gamma), 1, _p(sums), _p(dgamma), _p(dbeta), _p(bb), _p(ws), ws.numel(), _st())
            if dy2 is not None:
                _native.check(L.semseg_bn_bwd_reduce_fused_sum2(_p(dy), dy_ld, _p(dy2), dy2_ld, *args[2:], peer),
                              'bn_bwd_reduce_fused_sum2')
            elif peer is None:
                _native.check(L.semseg_bn_bwd_reduce_fused(*args), 'bn_bwd_reduce_fused')
            else:
                _native.check(L.semseg_bn_bwd_reduce_fused_peer(*args, peer), 'bn_bwd_reduce_fused_peer')
        else:
            bb = None
            if gate:                                      # the unfused reduce reads y: rebuild the gate tensor once
                y = empty_nhwc(n, k, oh, ow, dev)
                _native.check(L.semseg_bn_apply(_p(z), _p(coef[2]), _p(coef[3]), _p(None), 0, 1, _p(y), k, P, k, _st()),
                              'bn_apply')
                gsc = gsh = None
                y_arg, y_ld = _p(y), k
            _native.check(L.semseg_bn_bwd_reduce_mm(_p(dy), dy_ld, _p(y), k, _p(z), _p(coef[0]), _p(coef[1]), int(relu), P,
                                                    k, _p(sums), _p(gmax), _p(dgamma), _p(dbeta), _p(ws), ws.numel(),
                                                    _st()), 'bn_bwd_reduce_mm')
            _maybe_allreduce(sums)
            _native.check(L.semseg_bn_bwd_bound(_p(sums), _p(count), _p(gmax), _p(zmm), _p(coef[0]), _p(coef[1]), _p(gamma),
                                                k, 1, _p(dzp), P, _st()), 'bn_bwd_bound')
        if dy2 is not None:
            _native.check(L.semseg_bn_bwd_apply_h2_sum2(_p(dy), dy_ld, _p(dy2), dy2_ld, y_arg, y_ld, _p(z), _p(coef[0]), _p(coef[1]),
                                                        _p(gamma), _p(sums), _p(count), 1, int(relu), _p(dzp), _p(dres), P, k,
                                                        _p(gsc), _p(gsh), _p(bb), _st()), 'bn_bwd_apply_h2_sum2')
        else:
            _native.check(L.semseg_bn_bwd_apply_h2(_p(dy), dy_ld, y_arg, y_ld, _p(z), _p(coef[0]), _p(coef[1]), _p(gamma),
                                                   _p(sums), _p(count), 1, int(relu), _p(dzp), _p(dres), P, k, _p(gsc), _p(gsh),
                                                   _p(bb), _st()), 'bn_bwd_apply_h2')
        need_dw = ctx.needs_input_grad[1]
        dw_wino = None
        if wino_v is not None and need_dw:
            dw_wino = _winograd_wgrad(L, wino_v, dzp, geom)
            need_dw = False
        need_dx = ctx.needs_input_grad[0]
        dx_wino = None
        if wut is not None and need_dx:
            dx_wino = _winograd_dgrad(L, dzp, wut, geom)
            need_dx = False
        dx, dw = _split_conv_grads(L, sch, 'h2', geom, xp, dzp, w, wtp, need_dx, need_dw, param=ctx.w_param)
        if dw_wino is not None:
            dw = dw_wino
        if dx_wino is not None:
            dx = dx_wino
        return (dx, dw, dgamma if ctx.needs_input_grad[2] else None, dbeta if ctx.needs_input_grad[3] else None, dres,
                None, None, None, None, None, None, None, None, None)


# A BN output whose only consumer is the next convolution of the h2 path is read as planes, never as fp32: its fp32 copy need not
# be written (semseg_bn_apply_h2 with y == NULL).  The CALLER knows that (models/resnet.py: bn1 / bn2 of a block), says so with
# planes_only=True, and the tensor that comes back raises if anything asks for its fp32 values (as_nhwc).  SEMSEG_PLANES_ONLY=0
# writes every output in full.
PLANES_ONLY = os.environ.get('SEMSEG_PLANES_ONLY', '1') != '0'
PLANES_ONLY_POISON = os.environ.get('SEMSEG_PLANES_ONLY_POISON', '0') == '1'


def reads_fp32_input(conv):
    """True if ops.conv_bn_act would read the fp32 values of this convolution's input (the Winograd forward transforms x itself)"""
    k, c, r, s = conv.weight.shape
    return conv.stride[0] == 1 and conv.padding[0] == conv.dilation[0] and _wino_eligible(k, c, r, s)


def conv_bn_act(x, weight, gamma, beta, running_mean, running_var, num_batches_tracked, residual=None, stride=1,
                padding=0, dilation=1, training=False, momentum=0.1, eps=1e-5, relu=False, planes_only=False):
    """act(BN(conv(x)) + residual) for a bias-free conv.  Training on the h2 path with K % 8 == 0 runs the fused node
    (ConvBNActFn); everything else composes conv2d + batch_norm_act."""
    if not (FUSE and CONV_MODE == 'h2' and training and weight.shape[0] % 8 == 0):
        z = conv2d(x, weight, None, stride, padding, dilation)
        return batch_norm_act(z, gamma, beta, running_mean, running_var, residual=residual, training=training,
                              momentum=momentum, eps=eps, relu=relu, num_batches_tracked=num_batches_tracked)
    _require_cuda(x)
    wp, wtp = weight_planes(weight, 'h2')
    kk, cc, rr, ss = weight.shape
    cfg = (int(stride), int(padding), int(dilation), float(momentum), float(eps), bool(relu), bool(relu),
           bool(planes_only and PLANES_ONLY and relu and residual is None and torch.is_grad_enabled()))
    box = {}
    x_planes_only = getattr(x, '_semseg_planes_only', False)       # then the Winograd forward (it reads fp32 x) is not an option
    if int(stride) == 1 and int(padding) == int(dilation) and _wino_eligible(kk, cc, rr, ss) and not x_planes_only:
        xb = bounds_of(x)
        if xb is not None and len(xb) <= 8:
            box['wino'] = weight_wino(weight)            # None until prepare_conv_weights has run for this weight state
            box['x_bounds'] = xb
        if x.requires_grad and _wino_dgrad_eligible(kk):
            box['wino_t'] = weight_wino_t(weight)        # the data gradient in the Winograd domain needs no bound: dz comes as planes
    n, c, h, w = x.shape
    if box.get('wino') is not None and ((WINOGRAD_WGRAD and c >= WINOGRAD_WGRAD_MIN_C) or not weight.requires_grad):
        xp = planes_of(x, 'h2', n * h * w, c)        # Winograd forward and weight gradient work on V: x needs no planes
    else:
        xp = input_planes(x, 'h2')
    y = ConvBNActFn.apply(x, weight, gamma, beta, residual, xp, wp, wtp, absmax_of(residual), running_mean, running_var,
                          num_batches_tracked, cfg, box)
    yp, absmax = box['planes'], box['absmax']
    if yp is not None:
        attach_planes(y, yp, 'h2', y.shape[0] * y.shape[2] * y.shape[3], y.shape[1])
    if absmax is not None:
        attach_absmax(y, absmax)
    if box.get('planes_only'):
        y._semseg_planes_only = True
    return y


# ------------------------------------------------------------------------------------------------
# elementwise
# ------------------------------------------------------------------------------------------------
class AddActFn(Function):
    @staticmethod
    def forward(ctx, a, b, relu):
        L = _native.lib()
        a, a_ld = as_nhwc_of(a)
        b, b_ld = as_nhwc_of(b)
        n, c, h, w = a.shape
        out = empty_nhwc(n, c, h, w, a.device)
        _native.check(L.semseg_add_act(_p(a), a_ld, _p(b), b_ld, int(relu), _p(out), c, n * h * w, c, _st()), 'add_act')
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        _native.next_unit()
        if not ctx.relu:
            # the same tensor goes to TWO producers: if it is the first addend of a deferred fork sum, form the sum here -- one of
            # the producers might not be a native consumer that looks the pair up (it would read the first addend alone)
            if _ADDENDS:
                dy = _materialize_sum(dy)
            return dy, dy, None
        (out,) = ctx.saved_tensors
        dy, dy_ld = as_nhwc(dy)
        n, c, h, w = out.shape
        dx = empty_nhwc(n, c, h, w, out.device)
        _native.check(_native.lib().semseg_relu_bwd(_p(dy), dy_ld, _p(out), c, _p(dx), c, n * h * w, c, _st()), 'relu_bwd')
        return dx, dx, None


def add_act(a, b, relu=False):
    return AddActFn.apply(a, b, bool(relu))


class ClampMaxFn(Function):
    """y = min(x, cap): the upper clamp of nn.ReLU6 (mobilenet.py:26,34) behind a fused conv -> BN -> ReLU unit."""

    @staticmethod
    def forward(ctx, x, cap):
        x, ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, h, w, x.device)
        _native.check(_native.lib().semseg_clamp_max(_p(x), ld, float(cap), _p(y), c, n * h * w, c, _st()), 'clamp_max')
        ctx.save_for_backward(y)
        ctx.cap = float(cap)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy, ld = as_nhwc(dy)
        n, c, h, w = y.shape
        dx = empty_nhwc(n, c, h, w, y.device)
        _native.check(_native.lib().semseg_clamp_max_bwd(_p(dy), ld, _p(y), c, ctx.cap, _p(dx), c, n * h * w, c, _st()),
                      'clamp_max_bwd')
        return dx, None


def clamp_max(x, cap):
    _require_cuda(x)
    if x.shape[1] % 4:
        raise RuntimeError('clamp_max: channel count must be a multiple of 4, got %d' % x.shape[1])
    return ClampMaxFn.apply(x, float(cap))


# The two gradients that meet at a fork are not added by a launch of their own when the consumer of the sum can read two addends:
# inside `defer_fork_sums()` (TrainStep, around backward) ForkFn.backward hands back its FIRST gradient and records the second in
# _ADDENDS; ConvBNActFn.backward -- the producer of every forked tensor of the ResNet / HRNet blocks -- takes the pair and passes both
# to the BN backward kernels (semseg_bn_bwd_*_sum2: the same fp32 add per element, bit-identical to the materialised sum).  Every
# other native consumer reaches its gradient through as_nhwc(), which adds a pending pair on the spot; a pair that NOBODY took by
# the end of backward (a torch-side consumer saw the first addend alone) raises in __exit__.  SEMSEG_DEFER_FORK_SUMS=0 disables.
DEFER_FORK_SUMS = os.environ.get('SEMSEG_DEFER_FORK_SUMS', '1') != '0'
_FORK_DEFER = [False]
_ADDENDS = {}                # data_ptr of the first addend -> (first, second, [some consumer formed the sum])


class defer_fork_sums:
    def __enter__(self):
        self.prev = _FORK_DEFER[0]
        _FORK_DEFER[0] = DEFER_FORK_SUMS and CONV_MODE == 'h2' and FUSE
        return self

    def __exit__(self, *exc):
        _FORK_DEFER[0] = self.prev
        left = [tuple(rec[0].shape) for rec in _ADDENDS.values() if not rec[2][0]]
        _ADDENDS.clear()
        if left and not (exc and exc[0] is not None):
            raise RuntimeError('%d gradient sum(s) deferred at a fork were never formed (shapes %s): a consumer outside the native '
                               'layer read the first addend alone; set SEMSEG_DEFER_FORK_SUMS=0' % (len(left), left[:4]))
        return False


def _take_addend(g):
    """the second addend of gradient `g` if ForkFn left one pending (the caller then forms g + addend where it reads g), else None.
    The record stays until backward has ended -- a backward that hands its incoming gradient on unchanged (an identity, or the same
    tensor to two producers) leads several consumers to the same pair, and each of them must see the whole sum."""
    if not _ADDENDS or g is None:
        return None
    rec = _ADDENDS.get(g.data_ptr())
    if rec is None or rec[0].shape != g.shape or rec[0].stride() != g.stride():
        return None
    rec[2][0] = True
    if rec[1].is_cuda:
        # the addend is read on the consumer's stream, which need not be the one it was allocated on (branch streams): the caching
        # allocator must not hand its block out again before that read
        rec[1].record_stream(torch.cuda.current_stream(rec[1].device))
    return rec[1]


def _add_nhwc(ga, gb):
    L = _native.lib()
    ga, a_ld = _as_nhwc_plain(ga)              # plain: the pair being summed stays on record (its first addend would resolve again)
    gb, b_ld = _as_nhwc_plain(gb)
    n, c, h, w = ga.shape
    out = empty_nhwc(n, c, h, w, ga.device)
    if c % 4 == 0 and a_ld % 4 == 0 and b_ld % 4 == 0:
        _native.check(L.semseg_add_act(_p(ga), a_ld, _p(gb), b_ld, 0, _p(out), c, n * h * w, c, _st()), 'add_act')
    else:
        _native.check(L.semseg_copy2d(_p(ga), a_ld, _p(out), c, n * h * w, c, 0, _st()), 'copy2d')
        _native.check(L.semseg_copy2d(_p(gb), b_ld, _p(out), c, n * h * w, c, 1, _st()), 'copy2d')
    return out


def _materialize_sum(g):
    other = _take_addend(g) if torch.is_tensor(g) else None
    return g if other is None else _add_nhwc(g, other)


class ForkFn(Function):
    """A tensor with TWO consumers (block input -> first conv + shortcut, encoder map -> two heads): returns two aliases of
    it; backward adds the two gradients with the native add kernel -- otherwise autograd's own accumulation does that sum with
    a torch kernel (288 launches per R50 step).  A consumer that produced no gradient contributes nothing."""

    @staticmethod
    def forward(ctx, x):
        # without this autograd hands backward a full-size ZERO tensor for an alias nobody differentiated through (Resnet.forward
        # forks every stage output when return_feature_maps is set; the PPM / C1 heads use one or two of them), and the branch
        # below would never be taken: a fill plus an add launch per unused alias and step
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    @once_differentiable
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return gb if ga is None else ga
        if _ADDENDS:                                    # a pending pair of an inner fork is summed here: one level is deferred
            ga, gb = _materialize_sum(ga), _materialize_sum(gb)
        if _FORK_DEFER[0] and ga.dtype == torch.float32 and gb.dtype == torch.float32 and ga.shape == gb.shape:
            ga, a_ld = _as_nhwc_plain(ga)
            gb, b_ld = _as_nhwc_plain(gb)
            c = ga.shape[1]
            if c % 8 == 0 and a_ld % 4 == 0 and b_ld % 4 == 0 and ga.data_ptr() % 16 == 0 and gb.data_ptr() % 16 == 0 and \
                    ga.data_ptr() not in _ADDENDS:
                _ADDENDS[ga.data_ptr()] = (ga, gb, [False])
                return ga
        return _add_nhwc(ga, gb)


def fork(x, n=2):
    """`n` aliases of `x` for its `n` consumers (gradients are summed natively, see ForkFn); the split-plane / bound records
    of `x` travel with every alias.  Without autograd (inference) or for tensors that need no gradient `x` itself is returned
    `n` times."""
    if n <= 1:
        return (x,) if n == 1 else ()
    if not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * n
    _require_cuda(x)
    outs = []
    rest = x
    for _ in range(n - 1):
        a, rest = ForkFn.apply(rest)
        outs.append(a)
    outs.append(rest)
    rec, bound = getattr(x, '_semseg_planes', None), bounds_of(x)
    for t in outs:
        if rec is not None and rec[4] == x._version and rec[5] == x.data_ptr():
            attach_planes(t, rec[0], rec[1], rec[2], rec[3])
        if bound is not None:
            attach_absmax(t, bound)
    return tuple(outs)


_DROPOUT_STATE = {}


def dropout_mask(n, c, p, device):
    """[N, C] multipliers of nn.Dropout2d(p) (models.py:460,464): Bernoulli(1-p) / (1-p) per (sample, channel), drawn by a
    counter-based hash on the device (csrc/pool_resize.hip); the launch counter lives in device memory, so a captured launch
    draws a fresh mask at every replay.  Seeded from torch's RNG (torch.manual_seed governs it) on first use per device."""
    key = (device.type, device.index)
    st = _DROPOUT_STATE.get(key)
    if st is None:
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
        st = _DROPOUT_STATE[key] = torch.tensor([seed, 0], dtype=torch.int64, device=device)
    mask = torch.empty((n, c), device=device, dtype=torch.float32)
    _native.check(_native.lib().semseg_dropout_mask(_p(mask), n * c, float(p), _p(st), _st()), 'dropout_mask')
    return mask


class ConcatFn(Function):
    """torch.cat(dim=1) into one NHWC buffer; backward hands out channel-slice VIEWS (no copy)."""

    @staticmethod
    def forward(ctx, *xs):
        L = _native.lib()
        xs = [as_nhwc_of(x) for x in xs]
        n, _, h, w = xs[0][0].shape
        ctot = sum(x.shape[1] for x, _ in xs)
        out = empty_nhwc(n, ctot, h, w, xs[0][0].device)
        off = 0
        base = out.data_ptr()
        for x, ld in xs:
            c = x.shape[1]
            _native.check(L.semseg_copy2d(_p(x), ld, vp(base + 4 * off), ctot, n * h * w, c, 0, _st()), 'copy2d')
            off += c
        ctx.chans = [x.shape[1] for x, _ in xs]
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        dy, _ = as_nhwc(dy)
        outs, off = [], 0
        for c in ctx.chans:
            outs.append(dy[:, off:off + c])
            off += c
        return tuple(outs)


def concat(xs):
    y = ConcatFn.apply(*xs)
    bs = [bounds_of(x) for x in xs]
    if all(b is not None for b in bs):                   # max|cat| <= max over the inputs' bounds
        attach_absmax(y, tuple(t for b in bs for t in b))
    return y


class ScaleNCFn(Function):
    """Dropout2d with an explicit per-(n,c) multiplier (models.py:460,464)."""

    @staticmethod
    def forward(ctx, x, mask):
        x, ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        if ld != c:
            x = x.contiguous(memory_format=torch.channels_last)
        mask = mask.detach().to(device=x.device, dtype=torch.float32).contiguous()
        y = empty_nhwc(n, c, h, w, x.device)
        _native.check(_native.lib().semseg_scale_nc(_p(x), _p(mask), _p(y), n, h * w, c, _st()), 'scale_nc')
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        dy, ld = as_nhwc(dy)
        n, c, h, w = dy.shape
        if ld != c:
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = empty_nhwc(n, c, h, w, dy.device)
        _native.check(_native.lib().semseg_scale_nc(_p(dy), _p(mask), _p(dx), n, h * w, c, _st()), 'scale_nc')
        return dx, None


def scale_nc(x, mask):
    return ScaleNCFn.apply(x, mask)


# ------------------------------------------------------------------------------------------------
# pooling / resize
# ------------------------------------------------------------------------------------------------
class MaxPool3x3s2Fn(Function):
    @staticmethod
    def forward(ctx, x):
        x, ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        if ld != c:
            x = x.contiguous(memory_format=torch.channels_last)
        oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = empty_nhwc(n, c, oh, ow, x.device)
        idx = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.uint8)
        _native.check(_native.lib().semseg_maxpool3x3s2_fwd(_p(x), _p(y), _p(idx), n, h, w, c, _st()), 'maxpool_fwd')
        ctx.save_for_backward(idx)
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        n, c, h, w = ctx.shape
        dy, ld = as_nhwc(dy)
        if ld != c:
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = empty_nhwc(n, c, h, w, dy.device)
        _native.check(_native.lib().semseg_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), n, h, w, c, _st()), 'maxpool_bwd')
        return dx


def max_pool_3x3_s2(x):
    y = MaxPool3x3s2Fn.apply(x)
    bound = bounds_of(x)
    if bound is not None:
        attach_absmax(y, bound)          # max over windows of x: the bound of |x| holds for |y|
    return y


class AdaptiveAvgPoolFn(Function):
    @staticmethod
    def forward(ctx, x, oh, ow):
        x, ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, oh, ow, x.device)
        _native.check(_native.lib().semseg_adaptive_avgpool_fwd(_p(x), ld, _p(y), n, h, w, c, oh, ow, _st()),
                      'adaptive_avgpool_fwd')
        ctx.shape = (n, c, h, w, oh, ow)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        n, c, h, w, oh, ow = ctx.shape
        dy, ld = as_nhwc(dy)
        if ld != c:
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx = empty_nhwc(n, c, h, w, dy.device)
        _native.check(_native.lib().semseg_adaptive_avgpool_bwd(_p(dy), _p(dx), c, 0, n, h, w, c, oh, ow, _st()),
                      'adaptive_avgpool_bwd')
        return dx, None, None


def adaptive_avg_pool(x, size):
    oh, ow = (size, size) if isinstance(size, int) else size
    y = AdaptiveAvgPoolFn.apply(x, int(oh), int(ow))
    bound = bounds_of(x)
    if bound is not None:
        attach_absmax(y, bound)          # an average over a window of x: the bound of |x| holds
    return y


class MultiAdaptiveAvgPoolFn(Function):
    """nn.AdaptiveAvgPool2d of one map to several square grids (the pyramid of PPM / UPerNet): the map is read once,
    the gradient written once (csrc/pool_resize.hip, multipool_*)."""

    @staticmethod
    def forward(ctx, x, sizes):
        L = _native.lib()
        ctx.set_materialize_grads(False)              # backward builds the zero gradient of an unused scale itself (it is tiny)
        x, ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        ns = len(sizes)
        ys = [empty_nhwc(n, c, s, s, x.device) for s in sizes]
        sz = (ctypes.c_int * ns)(*sizes)
        ptrs = (vp * ns)(*[y.data_ptr() for y in ys])
        ws = workspace(L.semseg_adaptive_avgpool_multi_workspace_bytes(n, h, c, sz, ns), x.device)
        _native.check(L.semseg_adaptive_avgpool_multi_fwd(_p(x), ld, n, h, w, c, ns, sz, ptrs, _p(ws), ws.numel(), _st()),
                      'adaptive_avgpool_multi_fwd')
        ctx.shape = (n, c, h, w)
        ctx.sizes = tuple(sizes)
        return tuple(ys)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dys):
        L = _native.lib()
        n, c, h, w = ctx.shape
        sizes = ctx.sizes
        dev = next(d for d in dys if d is not None).device
        gs = []
        for d, s in zip(dys, sizes):
            if d is None:
                d = torch.zeros((n, s, s, c), device=dev).permute(0, 3, 1, 2)
            d, ld = as_nhwc(d)
            if ld != c:
                d = d.contiguous(memory_format=torch.channels_last)
            gs.append(d)
        ns = len(sizes)
        sz = (ctypes.c_int * ns)(*sizes)
        ptrs = (vp * ns)(*[g.data_ptr() for g in gs])
        dx = empty_nhwc(n, c, h, w, dev)
        _native.check(L.semseg_adaptive_avgpool_multi_bwd(ptrs, sz, ns, _p(dx), c, n, h, w, c, _st()),
                      'adaptive_avgpool_multi_bwd')
        return dx, None


def adaptive_avg_pool_multi(x, sizes):
    """[adaptive_avg_pool(x, s) for s in sizes] -- fused when the sizes are square ints, at most 4 scales and 16 column
    bins in total (PPM: 1 + 2 + 3 + 6); anything else pools scale by scale."""
    sizes = list(sizes)
    if 1 < len(sizes) <= 4 and all(isinstance(s, int) and s > 0 for s in sizes) and sum(sizes) <= 16 and x.shape[1] % 4 == 0:
        ys = list(MultiAdaptiveAvgPoolFn.apply(x, tuple(sizes)))
    else:
        return [adaptive_avg_pool(x, s) for s in sizes]
    bound = bounds_of(x)
    if bound is not None:
        for y in ys:
            attach_absmax(y, bound)      # an average over a window of x: the bound of |x| holds (the pyramid convs then split their
    return ys                            # input from the bound: one launch that pairs up across the branches, no absmax pass)


class BilinearFn(Function):
    """F.interpolate(mode='bilinear', align_corners=False); optional fused `+ base` (FPN / HRNet sums)
    and trailing ReLU (hrnet.py:248)."""

    @staticmethod
    def forward(ctx, x, oh, ow, base, relu):
        L = _native.lib()
        x, ld = as_nhwc_of(x)
        n, c, h, w = x.shape
        y = empty_nhwc(n, c, oh, ow, x.device)
        acc = 0
        if base is not None:
            b, b_ld = as_nhwc_of(base)
            _native.check(L.semseg_copy2d(_p(b), b_ld, _p(y), c, n * oh * ow, c, 0, _st()), 'copy2d')
            acc = 1
        _native.check(L.semseg_bilinear_fwd(_p(x), ld, _p(y), c, acc, int(relu), n, h, w, oh, ow, c, _st()),
                      'bilinear_fwd')
        ctx.shape = (n, c, h, w, oh, ow)
        ctx.has_base = base is not None
        ctx.relu = relu
        if relu:
            ctx.save_for_backward(y)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        L = _native.lib()
        _native.next_unit()
        n, c, h, w, oh, ow = ctx.shape
        dy, ld = as_nhwc(dy)
        if ctx.relu:
            (y,) = ctx.saved_tensors
            g = empty_nhwc(n, c, oh, ow, dy.device)
            _native.check(L.semseg_relu_bwd(_p(dy), ld, _p(y), c, _p(g), c, n * oh * ow, c, _st()), 'relu_bwd')
            dy, ld = g, c
        dx = None
        if ctx.needs_input_grad[0]:
            dx = empty_nhwc(n, c, h, w, dy.device)
            _native.check(L.semseg_bilinear_bwd(_p(dy), ld, _p(dx), c, 0, n, h, w, oh, ow, c, _st()), 'bilinear_bwd')
        return dx, None, None, (dy if ctx.has_base else None), None


def interpolate_bilinear(x, size, base=None, relu=False):
    oh, ow = int(size[0]), int(size[1])
    if base is None and not relu and x.shape[2] == oh and x.shape[3] == ow:
        return x                                            # identity resize (scale 1): exact copy in torch too
    y = BilinearFn.apply(x, oh, ow, base, bool(relu))
    if base is None:
        b = bounds_of(x)                                    # convex combination of source pixels (and ReLU): same bound
        if b is not None:
            attach_absmax(y, b)
    return y


# ------------------------------------------------------------------------------------------------
# head
# ------------------------------------------------------------------------------------------------
def _rows(x):
    """logical [N,C,H,W] NHWC-dense tensor -> (dense tensor, P, C)."""
    x, ld = as_nhwc(x)
    n, c, h, w = x.shape
    if ld != c:
        x = x.contiguous(memory_format=torch.channels_last)
    return x, n * h * w, c


class LogSoftmaxFn(Function):
    @staticmethod
    def forward(ctx, z):
        z, P, c = _rows(z.detach())
        out = empty_nhwc(z.shape[0], c, z.shape[2], z.shape[3], z.device)
        _native.check(_native.lib().semseg_log_softmax_fwd(_p(z), _p(out), P, c, _st()), 'log_softmax_fwd')
        ctx.save_for_backward(out)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        g, P, c = _rows(g)
        dz = empty_nhwc(out.shape[0], c, out.shape[2], out.shape[3], out.device)
        _native.check(_native.lib().semseg_log_softmax_bwd(_p(g), _p(out), _p(dz), P, c, _st()), 'log_softmax_bwd')
        return dz


def log_softmax(z):
    return LogSoftmaxFn.apply(z)


def softmax(z):
    """Inference-only softmax over the class axis (models.py:482-484)."""
    z, P, c = _rows(z.detach())
    out = empty_nhwc(z.shape[0], c, z.shape[2], z.shape[3], z.device)
    _native.check(_native.lib().semseg_softmax_fwd(_p(z), _p(out), P, c, _st()), 'softmax_fwd')
    return out


# destination of the inference head (engine.InferenceGraph.multi_scale / engine.evaluate): `out` = the score map to write or to
# accumulate into, `weight` = 1 / number of scales (eval.py:70), `accumulate` = add to what `out` holds
_HEAD = {'out': None, 'weight': 1.0, 'accumulate': False, 'used': False}


class head_output:
    """with ops.head_output(out, weight, accumulate): segmentation_module(feed, segSize=...) -- the decoder's fused
    up-sample + softmax kernel then writes `weight * probabilities` into `out` (or adds them to it) instead of a new tensor"""

    def __init__(self, out, weight=1.0, accumulate=False):
        self.cfg = {'out': out, 'weight': float(weight), 'accumulate': bool(accumulate), 'used': False}
        self.used = False

    def __enter__(self):
        self.prev = dict(_HEAD)
        _HEAD.update(self.cfg)
        return self

    def __exit__(self, *exc):
        self.used = _HEAD['used']          # False: the module inside did not go through ops.upsample_softmax (foreign decoder)
        _HEAD.update(self.prev)


def upsample_softmax(z, size):
    """Inference head (models.py:480-484): softmax over the classes of the logits bilinearly up-sampled to `size`, one fused
    kernel (csrc/pool_resize.hip semseg_upsample_softmax).  Returns [N, C, H, W] probabilities (times the weight of an enclosing
    ops.head_output context, into its buffer)."""
    z, ld = as_nhwc_of(z)
    n, c, h, w = z.shape
    oh, ow = int(size[0]), int(size[1])
    out, weight, acc = _HEAD['out'], _HEAD['weight'], _HEAD['accumulate']
    if out is not None:
        if tuple(out.shape) != (n, c, oh, ow) or nhwc_ld(out) != c or out.dtype != torch.float32 or not out.is_cuda:
            raise RuntimeError('head_output buffer %s does not fit the score map %s' % (tuple(out.shape), (n, c, oh, ow)))
    else:
        out, acc = empty_nhwc(n, c, oh, ow, z.device), False
    _native.check(_native.lib().semseg_upsample_softmax(_p(z), ld, _p(out), c, int(acc), float(weight), n, h, w, oh, ow, c,
                                                        _st()), 'upsample_softmax')
    _HEAD['used'] = True
    return out


class NLLAccFn(Function):
    """nn.NLLLoss(ignore_index) (train.py:154) + pixel_acc (models.py:12-18) in one pass.
    Returns (loss, acc) 0-dim tensors; only `loss` is differentiable."""

    @staticmethod
    def forward(ctx, logp, label, ignore_index):
        logp, P, c = _rows(logp.detach())
        _require_cuda(label)
        label = label.detach().contiguous()
        if label.dtype != torch.int64:
            label = label.long()
        if label.numel() != P:
            raise ValueError('Expected target size %s, got %s' % ([logp.shape[0], logp.shape[2], logp.shape[3]],
                                                                 list(label.shape)))
        out = torch.empty((3,), device=logp.device, dtype=torch.float32)
        ws = workspace(256 * 4 * 8, logp.device)
        _native.check(_native.lib().semseg_nll_acc_fwd(_p(logp), _p(label), int(ignore_index), P, c, _p(out), _p(ws),
                                                       ws.numel(), _st()), 'nll_acc_fwd')
        ctx.save_for_backward(out, label)
        ctx.cfg = (int(ignore_index), tuple(logp.shape))
        loss, acc = out[0].clone(), out[1].clone()      # 0-dim results (4-byte glue copies)
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    @once_differentiable
    def backward(ctx, gloss, gacc):
        out, label = ctx.saved_tensors
        ignore_index, shape = ctx.cfg
        n, c, h, w = shape
        gloss = gloss.contiguous().float()
        dlogp = empty_nhwc(n, c, h, w, out.device)
        _native.check(_native.lib().semseg_nll_bwd(_p(gloss), _p(out), _p(label), ignore_index, _p(dlogp), n * h * w, c,
                                                   _st()), 'nll_bwd')
        return dlogp, None, None


def nll_loss_acc(logp, label, ignore_index=-1):
    return NLLAccFn.apply(logp, label, ignore_index)


# ------------------------------------------------------------------------------------------------
# optimiser
# ------------------------------------------------------------------------------------------------
def sgd_step(params, grads, bufs, first_step, weight_decays, lr_tensor, momentum=0.9, grad_scale=1.0):
    """Fused multi-tensor SGD-momentum (train.py:117-126 semantics).  All lists are parallel; `lr_tensor`
    is a 1-element device tensor (so the poly schedule can change it under hipGraph replay).  `first_step`: one flag per
    tensor (or a single bool for all): the momentum buffer is uninitialised and is SET to the gradient, as torch.optim.SGD
    does the first time a parameter receives one."""
    if isinstance(first_step, (bool, int)):
        first_step = [first_step] * len(params)
    n = len(params)
    L = _native.lib()
    fused = SGD_FUSED and CONV_MODE == 'h2'
    arr = ((_native.SgdTensor2 if fused else _native.SgdTensor) * n)()
    keep = []
    for i, (p, g, b) in enumerate(zip(params, grads, bufs)):
        if p.stride() != g.stride() or p.stride() != b.stride():
            raise RuntimeError('sgd_step: param/grad/momentum strides differ')
        arr[i].param = p.data_ptr()
        arr[i].grad = g.data_ptr()
        arr[i].momentum_buf = b.data_ptr()
        arr[i].numel = p.numel()
        arr[i].weight_decay = float(weight_decays[i])
        arr[i].first_step = 1 if first_step[i] else 0
        if not fused:
            continue
        rec = _SLABS_FOR_SGD.pop(id(p), None)
        if rec is not None:
            slabs, out, numel, splits, _ = rec
            if out.data_ptr() != g.data_ptr() or numel != p.numel():
                raise RuntimeError('sgd_step: the deferred gradient of a %s parameter is not the gradient handed to the optimiser'
                                   % (tuple(p.shape),))
            arr[i].slabs, arr[i].splits = slabs.data_ptr(), splits
            keep.append(slabs)
        wrec = _WPLANES.get(id(p)) if FUSE else None
        if wrec is not None and wrec[0]() is p and wrec[2] == p.data_ptr():
            k, c, r, s2 = p.shape
            arr[i].absmax_slots = L.semseg_weights_absmax_slots(vp(wrec[3].data_ptr()), k, r * s2, c)
            _ABSMAX_FRESH[id(p)] = p._version
    if fused:
        _native.check(L.semseg_sgd_step_fused(arr, n, _p(lr_tensor), float(momentum), float(grad_scale), _st()), 'sgd_step_fused')
    else:
        _native.check(L.semseg_sgd_step(arr, n, _p(lr_tensor), float(momentum), float(grad_scale), _st()), 'sgd_step')
    # the kernel updates the parameters behind torch's back (no version bump): prepared weight planes are stale until
    # prepare_conv_weights runs again (TrainStep does, right after this call)
    for p in params:
        rec = _WPLANES.get(id(p))
        if rec is not None:
            _WPLANES[id(p)] = (rec[0], -1, rec[2], rec[3], rec[4], rec[5], rec[6])


def finish_leftover_slabs():
    """deferred gradients that no sgd_step took (a parameter outside the optimiser's groups): summed by the reduce launch after all"""
    if not _SLABS_FOR_SGD:
        return
    items = list(_SLABS_FOR_SGD.values())
    _SLABS_FOR_SGD.clear()
    arr = (_native.SlabTensor * len(items))()
    for i, (slabs, out, numel, splits, _) in enumerate(items):
        arr[i].slabs, arr[i].out, arr[i].numel, arr[i].splits = slabs.data_ptr(), out.data_ptr(), numel, splits
    _native.check(_native.lib().semseg_reduce_slabs_multi(arr, len(items), _st()), 'reduce_slabs_multi')
