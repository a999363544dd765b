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
    one process per GPU, current device set).  Collective.  Every rank learns whether ALL ranks succeeded; on any failure the
    communicator is torn down everywhere and False is returned (torch.distributed then carries the collectives)."""
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
        handle = ctypes.c_void_p()
        if int(host[128]) == 1 and ok:
            ids = (ctypes.c_ubyte * 128)(*host[:128].tolist())
            ok = L.semseg_comm_init(rank, world, ids, ctypes.byref(handle)) == 0
        else:
            ok = False
        if handle:
            handles[ch] = handle
        if ok and selftest:
            # every rank contributes rank+1: the sum must be world (world+1) / 2 in both payload types
            a = torch.full((257,), float(rank + 1), dtype=torch.float64, device=dev)
            b = torch.full((1031,), float(rank + 1), dtype=torch.float32, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            ok = L.semseg_comm_allreduce_sum_f64(handle, ctypes.c_void_p(a.data_ptr()), a.numel(), st) == 0 and \
                L.semseg_comm_allreduce_sum_f32(handle, ctypes.c_void_p(b.data_ptr()), b.numel(), st) == 0
            torch.cuda.synchronize()
            want = world * (world + 1) / 2.0
            ok = ok and bool((a == want).all()) and bool((b == want).all())
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) != 1:
        for h in handles.values():
            L.semseg_comm_destroy(h)
        return False
    _COMMS[_key(group)] = dict(rank=rank, world=world, **handles)
    return True


def allreduce_sum(buf, group=None, channel='sync'):
    """In-place sum over the ranks on the CURRENT stream through the communicator of `channel` ('sync': the compute stream's
    SyncBN payloads; 'bucket': the gradient buckets on their side stream); returns False if no native communicator serves
    `group` (or the dtype is not one the ABI carries) -- the caller then uses torch.distributed."""
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
