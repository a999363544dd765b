"""Native RCCL communicator of the data-parallel path, through the C ABI (include/semseg_hip.h, csrc/comm.hip:
semseg_comm_init / semseg_comm_allreduce_sum_* / semseg_comm_destroy) -- the replacement of the reference's thread rendezvous
(lib/nn/modules/comm.py:46-131, batchnorm.py:98-117).

Why not torch.distributed for the collectives themselves: a training step of R50dilated+PPM issues 122 SyncBN all-reduces of
a few hundred doubles plus the gradient buckets; through c10d each call costs 30-50 us of host time (work objects, stream
events, Python), through this wrapper one ctypes call enqueues ncclAllReduce on the current HIP stream (~5 us), so the step
stays GPU-bound.  torch.distributed remains the RENDEZVOUS (it carries the 128-byte unique id from rank 0 to the others) and
the fallback transport (gloo in the CPU tests, or when RCCL cannot be bound).

    comm.init(group)          # collective; returns True when the native communicators are up on every rank
    comm.allreduce_sum(t)     # in place on the current stream; False -> caller falls back to torch.distributed

TWO communicators per group: RCCL serialises the operations of ONE communicator in issue order and does not allow two of them in
flight from different streams, but the data-parallel step has exactly that -- latency-critical SyncBN payloads on the compute stream
while a 64 MiB gradient bucket is being reduced on the side stream.  Channel 'sync' serves the compute stream, channel 'bucket' the
gradient buckets (torch.distributed avoids the problem by funnelling every collective of a group through one internal stream, which
would put each SyncBN all-reduce of backward behind the bucket in flight).
"""
import ctypes
import os

import torch

from . import _native

_COMMS = {}          # id(group) -> dict(rank, world, sync=handle, bucket=handle)
CHANNELS = ('sync', 'bucket')


def enabled():
    """SEMSEG_NATIVE_COMM=0 (read when the communicator would be built): collectives stay with torch.distributed"""
    return os.environ.get('SEMSEG_NATIVE_COMM', '1') != '0'


def _key(group):
    return id(group) if group is not None else 0


def active(group=None):
    return _key(group) in _COMMS


def init(group=None, selftest=True):
    """Bring up one RCCL communicator for the ranks of `group` (torch.distributed must be initialised with the nccl backend:
    one process per GPU, current device set).  Collective.  The ranks vote after EVERY stage (id broadcast, communicator init, self-test) and leave
    together on the first failure -- no rank is left alone inside a blocking ncclCommInitRank; on any failure the communicators
    built so far are torn down everywhere and False is returned (torch.distributed then carries the collectives)."""
    import torch.distributed as dist
    if not enabled() or not (dist.is_available() and dist.is_initialized()):
        return False
    if _key(group) in _COMMS:
        return True
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if world <= 1 or dist.get_backend(group) != 'nccl' or not torch.cuda.is_available():
        return False
    L = _native.lib()
    dev = torch.device('cuda', torch.cuda.current_device())
    handles = {}
    ok = bool(L.semseg_comm_available())

    def agree(flag):
        # every stage ends with a vote: a rank that failed must not leave the others inside the next collective (ncclCommInitRank
        # of the second channel blocks until ALL ranks call it), so everybody learns the verdict and leaves the loop together
        f = torch.tensor([1 if flag else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
        return int(f.item()) == 1

    for ch in CHANNELS:
        idbuf = (ctypes.c_ubyte * 128)()
        have = ok
        if have and rank == 0:
            have = L.semseg_comm_unique_id(idbuf) == 0
        # rendezvous over the existing process group: rank 0's id (and whether it has one) to everybody
        t = torch.zeros(129, dtype=torch.uint8, device=dev)
        if rank == 0:
            t[:128] = torch.tensor(list(idbuf), dtype=torch.uint8)
            t[128] = 1 if have else 0
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        host = t.cpu()
        # stage 1: does every rank have the library and rank 0's id?  (vote BEFORE the blocking communicator init)
        ok = agree(ok and int(host[128]) == 1)
        if not ok:
            break
        handle = ctypes.c_void_p()
        ids = (ctypes.c_ubyte * 128)(*host[:128].tolist())
        ok = L.semseg_comm_init(rank, world, ids, ctypes.byref(handle)) == 0
        if handle:
            handles[ch] = handle
        # stage 2: the communicator is up on every rank (and RCCL counts the ranks we expect)
        if ok:
            cnt = ctypes.c_int(0)
            ok = L.semseg_comm_count(handle, ctypes.byref(cnt)) != 0 or cnt.value == world    # old RCCL without CommCount: skip
        ok = agree(ok)
        if not ok:
            break
        if selftest:
            # every rank contributes rank+1: the sum must be world (world+1) / 2 in both payload types
            a = torch.full((257,), float(rank + 1), dtype=torch.float64, device=dev)
            b = torch.full((1031,), float(rank + 1), dtype=torch.float32, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            ok = L.semseg_comm_allreduce_sum_f64(handle, ctypes.c_void_p(a.data_ptr()), a.numel(), st) == 0 and \
                L.semseg_comm_allreduce_sum_f32(handle, ctypes.c_void_p(b.data_ptr()), b.numel(), st) == 0
            torch.cuda.synchronize()
            want = world * (world + 1) / 2.0
            ok = ok and bool((a == want).all()) and bool((b == want).all())
            # stage 3: the known-sum self-test
            ok = agree(ok)
            if not ok:
                break
    if not ok:
        for h in handles.values():
            L.semseg_comm_destroy(h)
        return False
    _COMMS[_key(group)] = dict(rank=rank, world=world, **handles)
    return True


def rccl_ranks(group=None):
    """{channel: ncclCommCount of its communicator} -- what RCCL itself says about the world the collectives run over (None when
    the native communicators are not up)"""
    rec = _COMMS.get(_key(group))
    if rec is None:
        return None
    out = {}
    for ch in CHANNELS:
        cnt = ctypes.c_int(0)
        rc = _native.lib().semseg_comm_count(rec[ch], ctypes.byref(cnt))
        out[ch] = cnt.value if rc == 0 else None
    return out


def peer_world(group=None):
    rec = _PEERS.get(_key(group))
    return rec['world'] if rec is not None else None


def allreduce_sum(buf, group=None, channel='sync'):
    """In-place sum over the ranks on the CURRENT stream: channel 'sync' (the compute stream's SyncBN payloads) through the
    one-node peer exchange when it is up, else -- and channel 'bucket', the gradient buckets on their side stream, always --
    through the RCCL communicator of the channel; returns False if no native communicator serves
    `group` (or the dtype is not one the ABI carries) -- the caller then uses torch.distributed."""
    if channel == 'sync' and _PEERS and peer_allreduce_sum(buf, group):
        return True
    rec = _COMMS.get(_key(group))
    if rec is None or not buf.is_cuda or not buf.is_contiguous():
        return False
    L = _native.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    if buf.dtype == torch.float64:
        fn = L.semseg_comm_allreduce_sum_f64
    elif buf.dtype == torch.float32:
        fn = L.semseg_comm_allreduce_sum_f32
    else:
        return False
    _native.check(fn(rec[channel], ctypes.c_void_p(buf.data_ptr()), buf.numel(), st), 'comm_allreduce_sum')
    return True


# ------------------------------------------------------------------------------------------------
# one-node peer exchange (csrc/peer.hip): the SyncBN payloads as ONE small kernel over xGMI peer stores
# ------------------------------------------------------------------------------------------------
_PEERS = {}          # id(group) -> dict(handle, rank, world, cap)
PEER_MAX_DOUBLES = 2 * 4096 + 1          # [sum, sum^2, n] of a 4096-channel BN


def peer_enabled():
    """The peer exchange is built whenever it can be (one node, world <= semseg_peer_max_world()) and its collective bring-up
    passes -- every rank mapped every inbox, the co-residency vote, and a known-sum self-test over more exchanges than the
    protocol has slots, run on a SHORT timeout (SEMSEG_PEER_SELFTEST_TIMEOUT_S, 10 s) so that a node on which IPC-mapped inboxes
    do not work falls back to RCCL / torch.distributed within seconds, unanimously, instead of training on the slow path by default
    (rounds 2-4: opt-in, only bench.py turned it on; a train.py user got 122 small all-reduces per step).  SEMSEG_PEER=0 opts out."""
    return os.environ.get('SEMSEG_PEER', '1') != '0'


def peer_active(group=None):
    return _key(group) in _PEERS


def peer_handle(group=None, doubles=0):
    """the context (ctypes handle) for the entry points that exchange inside their own kernels
    (semseg_bn_fwd_stats_fused_peer / semseg_bn_bwd_reduce_fused_peer); None if the exchange is not up or the payload does not fit"""
    rec = _PEERS.get(_key(group))
    return rec['handle'] if rec is not None and doubles <= rec['cap'] else None


def _same_host(group):
    import socket
    import torch.distributed as dist
    names = [None] * dist.get_world_size(group)
    dist.all_gather_object(names, socket.gethostname(), group=group)
    return len(set(names)) == 1


def peer_init(group=None, max_doubles=PEER_MAX_DOUBLES, selftest=True):
    """Bring up the peer exchange among the ranks of `group` (all on ONE node, at most semseg_peer_max_world() of them; any
    torch.distributed backend -- it only carries the 64-byte IPC handles).  Collective.  Every rank creates its inbox on its
    current device, the handles travel by all_gather, every rank maps the others' inboxes; a known-sum self-test over more
    exchanges than the protocol has slots; unanimous or torn down everywhere (False: RCCL / torch.distributed then carry the payloads)."""
    import torch.distributed as dist
    if not peer_enabled() or not (dist.is_available() and dist.is_initialized()) or not torch.cuda.is_available():
        return False
    if _key(group) in _PEERS:
        return True
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    L = _native.lib()
    if world <= 1 or world > L.semseg_peer_max_world() or not _same_host(group):
        return False
    timeout_s = float(os.environ.get('SEMSEG_PEER_TIMEOUT_S', '120'))
    selftest_timeout_s = min(timeout_s, float(os.environ.get('SEMSEG_PEER_SELFTEST_TIMEOUT_S', '10')))
    handle = ctypes.c_void_p()
    ok = L.semseg_peer_create(rank, world, max_doubles, selftest_timeout_s if selftest else timeout_s, ctypes.byref(handle)) == 0
    mine = (ctypes.c_ubyte * 64)()
    ok = ok and L.semseg_peer_handle(handle, mine) == 0
    handles = [None] * world
    dist.all_gather_object(handles, bytes(mine) if ok else None, group=group)
    ok = ok and all(h is not None for h in handles)
    if ok:
        for r, h in enumerate(handles):
            if r != rank and L.semseg_peer_attach(handle, r, (ctypes.c_ubyte * 64)(*h)) != 0:
                ok = False
                break
    # the exchanging BN finish kernels wait for their peers INSIDE the kernel: their whole grid -- one block per 16 channels -- must
    # be resident at once on every rank's device, for the widest payload this exchange was created for ((max_doubles - 1) / 2
    # channels).  Checked here, collectively, not inside a step (a rank that failed it there left the others spinning).
    ok = ok and L.semseg_bn_peer_channel_capacity() >= (max_doubles - 1) // 2
    # every rank knows whether every rank could map every inbox BEFORE anybody launches an exchange (an exchange with a rank
    # that is not taking part would only end by its timeout)
    ok = _unanimous(ok, group)
    if ok and selftest:
        dev = torch.device('cuda', torch.cuda.current_device())
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        want = world * (world + 1) / 2.0
        for n in (3, 257, max_doubles, 1025, 64, 2049, 5):       # 7 exchanges > 4 slots: reuse is exercised
            a = torch.full((n,), float(rank + 1), dtype=torch.float64, device=dev)
            a[0] = 1.0 / (rank + 1)                                # an order-sensitive sum: bit-identical on all ranks or not at all
            # every exchange is launched whatever the verdict so far: the other ranks are waiting for this rank's payload
            rc = L.semseg_peer_allreduce_sum_f64(handle, ctypes.c_void_p(a.data_ptr()), n, st)
            torch.cuda.synchronize()
            first = 0.0
            for r in range(world):
                first = first + 1.0 / (r + 1)
            ok = ok and rc == 0 and L.semseg_peer_status(handle) == 0 and bool((a[1:] == want).all()) and float(a[0].item()) == first
        ok = _unanimous(ok, group)
        if ok:
            # ... and the kernels a training step really exchanges in: the fused BN finish kernels (forward statistics, backward
            # sums, the two-addend form) and one of them inside a captured + replayed hipGraph -- still on the short timeout, so a
            # node where the in-kernel exchange does not work falls back within seconds instead of failing inside a step
            try:
                ok = _peer_selftest_fused(L, handle, rank, world, dev)
            except Exception as e:                                     # noqa: BLE001 -- any failure means "not on this node"
                import sys
                print('mit_semseg.comm: peer self-test (fused BN kernels) raised %r on rank %d' % (e, rank), file=sys.stderr)
                ok = False
            ok = _unanimous(ok, group)
    if not ok:                       # the same verdict on every rank (_unanimous): everybody takes this branch together
        torch.cuda.synchronize()
        dist.barrier(group=group)    # nobody unmaps an inbox while a peer's self-test kernel may still store into it
        if handle:
            L.semseg_peer_destroy(handle)
        return False
    if selftest:
        L.semseg_peer_set_timeout(handle, timeout_s)        # the training run waits longer for a straggler than the self-test did
    _PEERS[_key(group)] = dict(handle=handle, rank=rank, world=world, cap=max_doubles)
    return True


def _selftest_rows(rank, P, C, device, dtype=torch.float32):
    """rank's [P, C] matrix of small integers (pixel % 3 + channel % 5 + rank + 1): every sum of them -- and of their squares and
    products -- is exact in fp64 whatever the order, so the exchanged totals have ONE right answer"""
    pix = torch.arange(P, device=device, dtype=dtype).remainder(3)
    ch = torch.arange(C, device=device, dtype=dtype).remainder(5)
    return (pix[:, None] + ch[None, :] + float(rank + 1)).contiguous()


def _peer_selftest_fused(L, handle, rank, world, dev, P=96, C=48):
    """The SyncBN kernels of the training step on a known problem, through THIS exchange: semseg_bn_fwd_stats_fused_peer (twice
    eagerly, then captured into a hipGraph and replayed twice), semseg_bn_bwd_reduce_fused_peer and semseg_bn_bwd_reduce_fused_sum2.
    Every rank launches every exchange whatever its verdict so far (the peers wait for its payload).  True iff every total is the
    known one."""
    vp = ctypes.c_void_p

    def p(t):
        return vp(t.data_ptr()) if t is not None else vp(0)
    f32 = dict(device=dev, dtype=torch.float32)
    z = _selftest_rows(rank, P, C, dev)
    dy = _selftest_rows(rank, P, C, dev) + 2.0
    dy2 = _selftest_rows(world - 1 - rank, P, C, dev)
    zs = [_selftest_rows(r, P, C, 'cpu', torch.float64) for r in range(world)]
    want_sum = sum(t.sum(0) for t in zs)
    want_sq = sum((t * t).sum(0) for t in zs)
    want_dy = sum((t + 2.0).sum(0) for t in zs)
    want_dy12 = want_dy + want_sum                               # sum over ranks of dy2 = sum over ranks of z (the ranks mirrored)
    gamma, beta = torch.ones(C, **f32), torch.zeros(C, **f32)
    coef = torch.empty((4, C), **f32)
    zmm = torch.empty((2 * C,), **f32)
    bb = torch.empty(((C + 15) // 16,), device=dev, dtype=torch.int32)
    ws = torch.empty(int(L.semseg_bn_mm_workspace_bytes(P, C)), device=dev, dtype=torch.uint8)
    stats = torch.zeros((2 * C + 1,), device=dev, dtype=torch.float64)
    sums = torch.zeros((2 * C,), device=dev, dtype=torch.float64)
    dgamma, dbeta = torch.empty(C, **f32), torch.empty(C, **f32)
    ok = [True]

    def fwd():
        st = vp(torch.cuda.current_stream().cuda_stream)
        rc = L.semseg_bn_fwd_stats_fused_peer(p(z), P, C, p(stats), p(zmm), p(gamma), p(beta), vp(0), vp(0), vp(0), 0.1, 1e-5, 0,
                                              vp(0), p(coef[0]), p(coef[1]), p(coef[2]), p(coef[3]), p(bb), p(ws), ws.numel(), st,
                                              handle, vp(0))
        ok[0] = ok[0] and rc == 0

    def fwd_good():
        torch.cuda.synchronize()
        s = stats.cpu()
        mean = (want_sum / float(world * P)).float()
        good = L.semseg_peer_status(handle) == 0 and torch.equal(s[:C], want_sum) and torch.equal(s[C:2 * C], want_sq) and \
            float(s[2 * C]) == float(world * P) and torch.equal(coef[0].cpu(), mean)
        stats.zero_()
        coef.zero_()
        return good

    for _ in range(2):
        fwd()
        ok[0] = fwd_good() and ok[0]
    # the same exchange recorded into a hipGraph and replayed: how TrainStep / SegmentedStep issue it (the kernel reads the
    # exchange's sequence number from device memory when it RUNS, so a replay takes the next slot like an eager launch)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        graph.capture_begin()
        try:
            fwd()
        finally:
            graph.capture_end()
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):
        graph.replay()
        ok[0] = fwd_good() and ok[0]
    # backward sums on the statistics of a last eager forward (mean / invstd / count as a step has them)
    fwd()
    torch.cuda.synchronize()
    st = vp(torch.cuda.current_stream().cuda_stream)
    count = stats[2 * C:]
    xhat = [(t - (want_sum / float(world * P))[None, :]) for t in zs]

    def bwd_good(want_first, addend):
        torch.cuda.synchronize()
        s = sums.cpu()
        var = want_sq / float(world * P) - (want_sum / float(world * P)) ** 2
        second = sum(((t + 2.0 + (a if addend else 0.0)) * x).sum(0) for t, a, x in zip(zs, reversed(zs), xhat)) / (var + 1e-5).sqrt()
        good = L.semseg_peer_status(handle) == 0 and torch.equal(s[:C], want_first) and \
            bool(((s[C:] - second).abs() <= 1e-4 * (1.0 + second.abs())).all())
        sums.zero_()
        return good
    rc = L.semseg_bn_bwd_reduce_fused_peer(p(dy), C, vp(0), C, p(z), p(coef[0]), p(coef[1]), vp(0), vp(0), 0, P, C, p(count), p(zmm),
                                           p(gamma), 1, p(sums), p(dgamma), p(dbeta), p(bb), p(ws), ws.numel(), st, handle)
    ok[0] = bwd_good(want_dy, False) and rc == 0 and ok[0]
    rc = L.semseg_bn_bwd_reduce_fused_sum2(p(dy), C, p(dy2), C, vp(0), C, p(z), p(coef[0]), p(coef[1]), vp(0), vp(0), 0, P, C, p(count),
                                           p(zmm), p(gamma), 1, p(sums), p(dgamma), p(dbeta), p(bb), p(ws), ws.numel(), st, handle)
    ok[0] = bwd_good(want_dy12, True) and rc == 0 and ok[0]
    return bool(ok[0])


def _unanimous(ok, group):
    import torch.distributed as dist
    votes = [None] * dist.get_world_size(group)
    dist.all_gather_object(votes, bool(ok), group=group)
    return all(votes)


def peer_allreduce_sum(buf, group=None):
    """In-place sum of a float64 payload over the ranks, one kernel on the CURRENT stream (capturable); False if the peer exchange
    does not serve `group` or the payload does not fit."""
    rec = _PEERS.get(_key(group))
    if rec is None or buf.dtype != torch.float64 or not buf.is_cuda or not buf.is_contiguous() or buf.numel() > rec['cap']:
        return False
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _native.check(_native.lib().semseg_peer_allreduce_sum_f64(rec['handle'], ctypes.c_void_p(buf.data_ptr()), buf.numel(), st),
                  'peer_allreduce_sum')
    return True


def peer_check(group=None):
    """Raise if an exchange of `group` ever timed out (the kernel then poisoned its result with NaN); reads a host-mapped word,
    no device synchronisation -- call it where the host looks at the loss anyway."""
    rec = _PEERS.get(_key(group))
    if rec is not None and _native.lib().semseg_peer_status(rec['handle']) != 0:
        raise RuntimeError('peer exchange: a rank waited longer than SEMSEG_PEER_TIMEOUT_S for the payload of another rank '
                           '(a rank died, or the ranks no longer issue the same sequence of SyncBN layers)')


def peer_destroy(group=None):
    """Collective: nobody unmaps an inbox another rank may still be writing to."""
    rec = _PEERS.pop(_key(group), None)
    if rec is not None:
        import torch.distributed as dist
        torch.cuda.synchronize()
        if dist.is_available() and dist.is_initialized():
            dist.barrier(group=group)
        _native.lib().semseg_peer_destroy(rec['handle'])


def destroy(group=None):
    rec = _COMMS.pop(_key(group), None)
    if rec is not None:
        for ch in CHANNELS:
            _native.lib().semseg_comm_destroy(rec[ch])


def destroy_all():
    for k in list(_COMMS):
        destroy_key = _COMMS.pop(k)
        for ch in CHANNELS:
            _native.lib().semseg_comm_destroy(destroy_key[ch])
