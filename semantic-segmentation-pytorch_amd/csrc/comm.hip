// Collective entry points of the C ABI (comm_init / allreduce_sum / comm_destroy): the one-process-per-GPU
// replacement of the reference's thread rendezvous (lib/nn/modules/comm.py:46-131, batchnorm.py:98-117) -- the SyncBN
// statistics and the gradient buckets are summed by RCCL all-reduces over xGMI, enqueued on the caller's HIP stream
// (stream-ordered with the kernels around them, capturable into a hipGraph).
//
// RCCL is bound at RUN time (dlopen of the librccl the process already holds -- PyTorch-ROCm ships one -- else the
// system's): libsemseg_hip.so itself has no link-time dependency on it, so it loads on hosts without RCCL and every other
// entry point keeps working there; semseg_comm_* then return SEMSEG_ECOMM.
#include "common.h"
#include <dlfcn.h>
#include <mutex>
#include <string.h>

namespace {

// the subset of rccl.h this file needs (ABI-stable since NCCL 2.0)
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7, ncclFloat64 = 8 };     // ncclDataType_t
enum { ncclSum = 0 };                           // ncclRedOp_t

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*GetVersion)(int*) = nullptr;
    int (*CommCount)(ncclComm_t, int*) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        // RTLD_NOLOAD first: bind to the copy the process already mapped (torch's), never a second RCCL next to it
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle)
        for (const char* n : names) {
            g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.handle) break;
        }
    if (!g_rccl.handle) return;
#define BIND(field, sym) *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, sym)
    BIND(GetUniqueId, "ncclGetUniqueId");
    BIND(CommInitRank, "ncclCommInitRank");
    BIND(CommDestroy, "ncclCommDestroy");
    BIND(AllReduce, "ncclAllReduce");
    BIND(GroupStart, "ncclGroupStart");
    BIND(GroupEnd, "ncclGroupEnd");
    BIND(GetVersion, "ncclGetVersion");
    BIND(CommCount, "ncclCommCount");
#undef BIND
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.GroupStart &&
                g_rccl.GroupEnd;
}

const Rccl* rccl() {
    std::call_once(g_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

struct Comm {
    ncclComm_t comm;
    int rank, world;
};

}  // namespace

extern "C" int semseg_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int semseg_comm_version(void) {
    const Rccl* r = rccl();
    int v = 0;
    if (!r || !r->GetVersion || r->GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

extern "C" int semseg_comm_unique_id(void* id128) {
    const Rccl* r = rccl();
    if (!r) return SEMSEG_ECOMM;
    if (!id128) return SEMSEG_EINVAL;
    ncclUniqueId id;
    const int rc = r->GetUniqueId(&id);
    if (rc != ncclSuccess) return SEMSEG_ECOMM;
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int semseg_comm_init(int rank, int world, const void* id128, void** comm_out) {
    const Rccl* r = rccl();
    if (!r) return SEMSEG_ECOMM;
    if (!id128 || !comm_out || world < 1 || rank < 0 || rank >= world) return SEMSEG_EINVAL;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    if (r->CommInitRank(&c, world, id, rank) != ncclSuccess || !c) return SEMSEG_ECOMM;
    *comm_out = new Comm{c, rank, world};
    return 0;
}

// the number of ranks RCCL itself reports for the communicator (ncclCommCount): the evidence bench.py prints for "these
// collectives ran over N ranks of RCCL"
extern "C" int semseg_comm_count(void* comm, int* count_out) {
    const Rccl* r = rccl();
    if (!r || !r->CommCount) return SEMSEG_ECOMM;
    if (!comm || !count_out) return SEMSEG_EINVAL;
    return r->CommCount(((Comm*)comm)->comm, count_out) == ncclSuccess ? 0 : SEMSEG_ECOMM;
}

template <int DTYPE>
static int allreduce(void* comm, void* buf, size_t count, void* stream) {
    const Rccl* r = rccl();
    if (!r) return SEMSEG_ECOMM;
    if (!comm || (!buf && count)) return SEMSEG_EINVAL;
    if (count == 0) return 0;
    Comm* c = (Comm*)comm;
    return r->AllReduce(buf, buf, count, DTYPE, ncclSum, c->comm, (hipStream_t)stream) == ncclSuccess ? 0 : SEMSEG_ECOMM;
}

extern "C" int semseg_comm_allreduce_sum_f32(void* comm, float* buf, size_t count, void* stream) {
    return allreduce<ncclFloat32>(comm, buf, count, stream);
}

extern "C" int semseg_comm_allreduce_sum_f64(void* comm, double* buf, size_t count, void* stream) {
    return allreduce<ncclFloat64>(comm, buf, count, stream);
}

// several small payloads in ONE RCCL group (independent SyncBN layers: the PPM branches, HRNet's parallel branches):
// one launch / one ring traversal instead of `n`
extern "C" int semseg_comm_allreduce_sum_f64_multi(void* comm, double* const* bufs, const size_t* counts, int n, void* stream) {
    const Rccl* r = rccl();
    if (!r) return SEMSEG_ECOMM;
    if (!comm || n < 0 || (n && (!bufs || !counts))) return SEMSEG_EINVAL;
    Comm* c = (Comm*)comm;
    if (r->GroupStart() != ncclSuccess) return SEMSEG_ECOMM;
    int bad = 0;
    for (int i = 0; i < n; ++i)
        if (counts[i] && r->AllReduce(bufs[i], bufs[i], counts[i], ncclFloat64, ncclSum, c->comm, (hipStream_t)stream) != ncclSuccess)
            bad = 1;
    if (r->GroupEnd() != ncclSuccess) bad = 1;
    return bad ? SEMSEG_ECOMM : 0;
}

extern "C" int semseg_comm_destroy(void* comm) {
    const Rccl* r = rccl();
    if (!comm) return 0;
    Comm* c = (Comm*)comm;
    int rc = 0;
    if (r && r->CommDestroy(c->comm) != ncclSuccess) rc = SEMSEG_ECOMM;
    delete c;
    return rc;
}
