// One-node peer exchange for the SyncBN payloads: an all-reduce(sum) of a few hundred to a few thousand
// doubles in ONE small kernel, over xGMI peer-to-peer stores -- the replacement of the reference's master/slave queue
// rendezvous (lib/nn/modules/comm.py:46-131, batchnorm.py:98-117) for the 122 latency-bound exchanges of a training step.
//
// Why not RCCL for these: a step of R50dilated+PPM has 61 BN layers x (forward [sum, sum^2, n] + backward [sum g, sum g xhat]).
// An ncclAllReduce of 4 KB costs a kernel launch plus a multi-hop ring / tree protocol (tens of microseconds at 8 ranks) and
// either breaks the step's hipGraph into segments or has to be captured with the library's internals.  The payloads are tiny,
// the node is fully connected (7 xGMI links per GPU), so every rank simply WRITES its payload into every peer's inbox and sums
// what arrives in its own -- one hop, no ring, no host, an ordinary kernel that hipGraph capture treats like any other.
//
// Protocol (the "low latency" idea: data and flag travel in the same 8-byte word, so there is no fence and no cache flush):
//   * inbox (per rank): kSlots x world x 2 cap 64-bit words of UNCACHED device memory, shared with the peers through
//     hipIpcGetMemHandle / hipIpcOpenMemHandle (or directly, for contexts of one process);
//   * exchange number q (device-resident counter, advanced by the kernel itself so a replayed hipGraph keeps counting):
//     slot = q mod kSlots, tag = q + 1 (never 0);
//   * push: double i of my payload -> words (tag<<32 | lo32), (tag<<32 | hi32) at [slot][my rank][2 i] of EVERY peer's inbox
//     (system-scope 8-byte stores: single-copy atomic, written through);
//   * sum: for rank r = 0 .. world-1 IN THAT ORDER on every rank (my own term from my buffer, the others polled from my inbox
//     until both words carry the tag): the result is bit-identical on all ranks, so replicas cannot drift;
//   * a slot is reused kSlots exchanges later; no rank can be more than one exchange ahead of the slowest (it needs the
//     slowest rank's payload of exchange q to leave q), so kSlots >= 2 rules out overwriting unread words.
//   * a rank that waits longer than `timeout_s` raises the context's status (host-mapped word, semseg_peer_status) and
//     poisons its result with NaN instead of hanging the GPU.
#include "peer_dev.h"
#include <string.h>

using namespace semseg_peer;

namespace {

constexpr int kThreads = 1024;

struct Peer {
    PeerArgs a;
    void* own_inbox;
    size_t inbox_bytes;
    unsigned* status_host;
    bool opened[kMaxWorld];
};

__global__ __launch_bounds__(kThreads) void peer_allreduce_f64_kernel(PeerArgs a, double* __restrict__ buf, int n) {
    __shared__ unsigned s_seq;
    if (threadIdx.x == 0) s_seq = *a.seq;
    __syncthreads();
    const unsigned q = s_seq;
    for (int i = threadIdx.x; i < n; i += kThreads) push(a, q, i, buf[i]);
    const long long t0 = wall_clock64();
    bool timed_out = false;
    for (int i = threadIdx.x; i < n && !timed_out; i += kThreads) buf[i] = gather_sum(a, q, i, buf[i], t0, timed_out);
    if (timed_out) atomicOr(a.status, 1u);
    __syncthreads();
    if (threadIdx.x == 0) advance(a, q, 1);
}

}  // namespace

bool semseg_peer::peer_args(void* peer, PeerArgs* out) {
    Peer* p = (Peer*)peer;
    if (!p || !out) return false;
    for (int r = 0; r < p->a.world; ++r)
        if (!p->a.inbox[r]) return false;
    *out = p->a;
    return true;
}

extern "C" int semseg_peer_max_world(void) { return kMaxWorld; }

// the timeout of the exchanges launched FROM NOW ON (kernels take the context by value at launch; a captured graph keeps the
// value of its capture): comm.peer_init runs its self-test on a short one and raises it for the training run
extern "C" int semseg_peer_set_timeout(void* peer, double timeout_s) {
    if (!peer || !(timeout_s > 0.0)) return SEMSEG_EINVAL;
    static_cast<Peer*>(peer)->a.timeout_ticks = (long long)(timeout_s * 1e8);
    return 0;
}

extern "C" int semseg_peer_create(int rank, int world, int max_doubles, double timeout_s, void** peer_out) {
    if (!peer_out || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || max_doubles < 1 || !(timeout_s > 0.0))
        return SEMSEG_EINVAL;
    Peer* p = new Peer();
    memset(p, 0, sizeof(*p));
    p->a.rank = rank;
    p->a.world = world;
    p->a.cap = max_doubles;
    p->a.timeout_ticks = (long long)(timeout_s * 1e8);
    p->inbox_bytes = (size_t)kSlots * world * 2 * (size_t)max_doubles * sizeof(unsigned long long);
    // uncached (else fine-grained) device memory: remote stores land in memory, local polls read memory -- never a stale L2 line
    if (hipExtMallocWithFlags(&p->own_inbox, p->inbox_bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags(&p->own_inbox, p->inbox_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            delete p;
            return SEMSEG_ECOMM;
        }
    }
    bool ok = hipMemset(p->own_inbox, 0, p->inbox_bytes) == hipSuccess;
    ok = ok && hipMalloc((void**)&p->a.seq, 2 * sizeof(unsigned)) == hipSuccess && hipMemset(p->a.seq, 0, 2 * sizeof(unsigned)) == hipSuccess;
    if (ok) p->a.done = p->a.seq + 1;
    ok = ok && hipHostMalloc((void**)&p->status_host, sizeof(unsigned), hipHostMallocMapped) == hipSuccess;
    if (ok) {
        *p->status_host = 0;
        ok = hipHostGetDevicePointer((void**)&p->a.status, p->status_host, 0) == hipSuccess;
    }
    ok = ok && hipDeviceSynchronize() == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        if (p->own_inbox) (void)hipFree(p->own_inbox);
        if (p->a.seq) (void)hipFree(p->a.seq);
        if (p->status_host) (void)hipHostFree(p->status_host);
        delete p;
        return SEMSEG_ECOMM;
    }
    p->a.inbox[rank] = (unsigned long long*)p->own_inbox;
    *peer_out = p;
    return 0;
}

extern "C" int semseg_peer_handle(void* peer, void* handle64) {
    if (!peer || !handle64) return SEMSEG_EINVAL;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is the 64-byte handle the header documents");
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, ((Peer*)peer)->own_inbox) != hipSuccess) {
        (void)hipGetLastError();
        return SEMSEG_ECOMM;
    }
    memcpy(handle64, &h, sizeof(h));
    return 0;
}

extern "C" int semseg_peer_attach(void* peer, int src_rank, const void* handle64) {
    Peer* p = (Peer*)peer;
    if (!p || !handle64 || src_rank < 0 || src_rank >= p->a.world || src_rank == p->a.rank || p->a.inbox[src_rank]) return SEMSEG_EINVAL;
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    void* q = nullptr;
    if (hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !q) {
        (void)hipGetLastError();
        return SEMSEG_ECOMM;
    }
    p->a.inbox[src_rank] = (unsigned long long*)q;
    p->opened[src_rank] = true;
    return 0;
}

// contexts of ONE process (one per device, or two on one device in the tests): the inbox pointer is shared directly
extern "C" int semseg_peer_attach_local(void* peer, int src_rank, void* other_peer) {
    Peer* p = (Peer*)peer;
    Peer* o = (Peer*)other_peer;
    if (!p || !o || src_rank < 0 || src_rank >= p->a.world || src_rank == p->a.rank || o->a.rank != src_rank ||
        o->a.world != p->a.world || o->a.cap != p->a.cap || p->a.inbox[src_rank])
        return SEMSEG_EINVAL;
    p->a.inbox[src_rank] = (unsigned long long*)o->own_inbox;
    return 0;
}

extern "C" int semseg_peer_allreduce_sum_f64(void* peer, double* buf, size_t count, void* stream) {
    Peer* p = (Peer*)peer;
    if (!p || (!buf && count) || count > (size_t)p->a.cap) return SEMSEG_EINVAL;
    for (int r = 0; r < p->a.world; ++r)
        if (!p->a.inbox[r]) return SEMSEG_EINVAL;           // a peer was never attached
    if (count == 0) return 0;
    hipLaunchKernelGGL(peer_allreduce_f64_kernel, dim3(1), dim3(kThreads), 0, (hipStream_t)stream, p->a, buf, (int)count);
    SEMSEG_LAUNCH_CHECK();
    return 0;
}

extern "C" int semseg_peer_status(void* peer) {
    Peer* p = (Peer*)peer;
    if (!p) return SEMSEG_EINVAL;
    return *(volatile unsigned*)p->status_host ? SEMSEG_ECOMM : 0;
}

extern "C" int semseg_peer_destroy(void* peer) {
    Peer* p = (Peer*)peer;
    if (!p) return 0;
    for (int r = 0; r < p->a.world; ++r)
        if (p->opened[r]) (void)hipIpcCloseMemHandle(p->a.inbox[r]);
    (void)hipFree(p->own_inbox);
    (void)hipFree(p->a.seq);
    (void)hipHostFree(p->status_host);
    delete p;
    return 0;
}
