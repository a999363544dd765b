// Launches of INDEPENDENT problems side by side (round 6; HRNetV2's parallel branches, hrnet.py:225-227, and anything else the host
// declares independent with ops.run_branches).
//
// The four branches of a HighResolutionModule run the same sequence of conv / BN launches on four geometries, each launch a few
// blocks and ~10 us long: 2 726 launches per training step, 27 ms of kernel time at 1.03 kernels in flight
// (profiles/r6_bench_trace_concurrency_cfg4.txt).  While the host has a "branches" scope open (semseg_batch_begin), a launch made
// through launch_body<BODY>() is RECORDED -- (kernel family, grid, argument block), under the ordinal of the C-ABI call it came
// from -- instead of issued; all the host logic of the entry points (launch plans, out-parameters, workspace layout) still runs at
// call time.  semseg_batch_end() zips the recorded sequences of the branches: the records that sit at the same position (same
// ordinal, same kernel instantiation) leave as ONE launch of many_kernel<BODY>, whose block b finds its problem j in a table in the
// kernel arguments (first[j] <= b < first[j + 1]) and runs block b - first[j] of that problem's own grid through the UNCHANGED
// kernel body -- same tiles, same summation order, bit-identical results (the form of wgrad_multi_kernel, conv_split.hip).  Records
// without a partner leave as the plain single launch.  Order inside a branch is kept; branches do not depend on each other.
//
// A kernel takes part by being written as a BODY type:
//     struct my_body { static constexpr int THREADS = 256;
//                      static __device__ __forceinline__ void run(const U3 blockIdx, const U3 gridDim, <arguments>) {...} };
// (the two leading parameters shadow the built-ins, so a kernel body moves into run() unchanged) and launched with
//     semseg_batch::launch_body<my_body>(grid, smem_bytes, stream, arguments...).
// Every OTHER launch of the library made while a scope is open first flushes what has been recorded (common.h redefines
// hipLaunchKernelGGL accordingly): correct for any kernel, batched only for the converted ones.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

namespace semseg_batch {

constexpr int kMaxBranches = 16;     // branches of one scope (HRNetV2's stage-4 exchange: 12 independent conv chains)
constexpr int kMaxGroup = 4;         // problems of one many_kernel launch (4 argument blocks stay far below the 4 KB argument segment)
constexpr int kMaxArgBytes = 480;    // argument block of one record

struct U3 {                          // blockIdx / gridDim as plain data (what run() receives under the names of the built-ins)
    unsigned x, y, z;
};

struct Record;
typedef int (*GroupLaunch)(const Record* const* recs, int n, hipStream_t st);

struct Record {
    GroupLaunch launch;              // issues n >= 1 records of this kernel instantiation as one launch; identifies the instantiation
    int op;                          // ordinal of the C-ABI call inside its branch (semseg_batch_next_op)
    int max_group;                   // 1: this instantiation has no many-problem form (BODY::MULTI == false): recorded, issued alone
    const char* name;                // __PRETTY_FUNCTION__ of the issuing function: names the body in SEMSEG_BATCH_DEBUG traces
    int cost;                        // how long ONE block of this launch runs, in the launch site's own unit (hint_cost; 0: unknown):
                                     // the problems of a group are laid out longest blocks first (the hardware starts blocks in order)
    U3 grid;
    unsigned smem;
    alignas(16) unsigned char args[kMaxArgBytes];
};

// csrc/batch.hip
bool recording();                                // a scope is open in this process
Record* new_record();                            // appended to the current branch
int flush_recorded();                            // everything recorded so far leaves on the scope's stream; the scope stays open
hipStream_t direct_stream(hipStream_t requested, const char* what = nullptr);   // for a launch that is NOT recorded: flush, then the scope's stream (no scope: `requested`)
void hint_cost(int cost);                        // for the NEXT launch_body of this thread's current branch (a GEMM: k-tiles per block)

// ---- argument blocks ---------------------------------------------------------------------------------------------------------
template <class... A>
struct Pack;
template <>
struct Pack<> {
    static __host__ __device__ __forceinline__ Pack make() { return Pack(); }
};
template <class H, class... T>
struct Pack<H, T...> {
    H head;
    Pack<T...> tail;
    static __host__ __device__ __forceinline__ Pack make(H h, T... t) {
        Pack p;
        p.head = h;
        p.tail = Pack<T...>::make(t...);
        return p;
    }
};

template <class BODY, class... B>
__device__ __forceinline__ void apply(const U3 bi, const U3 gd, const Pack<>&, B... bound) {
    BODY::run(bi, gd, bound...);
}
template <class BODY, class H, class... T, class... B>
__device__ __forceinline__ void apply(const U3 bi, const U3 gd, const Pack<H, T...>& p, B... bound) {
    apply<BODY>(bi, gd, p.tail, bound..., p.head);
}

// ---- the two kernels of a body -------------------------------------------------------------------------------------------------
template <class BODY, class... A>
__global__ __launch_bounds__(BODY::THREADS) void one_kernel(A... a) {
    BODY::run(U3{blockIdx.x, blockIdx.y, blockIdx.z}, U3{gridDim.x, gridDim.y, gridDim.z}, a...);
}

template <class P>
struct Table {
    int n;
    int first[kMaxGroup + 1];        // multiples of 8: block b of a problem keeps the XCD (b % 8) its own launch would give it
    U3 grid[kMaxGroup];
    P p[kMaxGroup];
};

template <class BODY, class... A>
__global__ __launch_bounds__(BODY::THREADS) void many_kernel(const Table<Pack<A...>> t) {
    const int b = blockIdx.x;
    int j = 0;
    while (j + 1 < t.n && b >= t.first[j + 1]) ++j;          // block-uniform: scalar loads from the argument segment
    const U3 g = t.grid[j];
    const unsigned l = (unsigned)(b - t.first[j]);
    if (l >= g.x * g.y * g.z) return;                        // the padding blocks between two problems
    U3 bi;
    bi.x = l % g.x;
    const unsigned r = l / g.x;
    bi.y = r % g.y;
    bi.z = r / g.y;
    apply<BODY>(bi, g, t.p[j]);
}

// > 64 KiB of dynamic LDS is a property of (kernel, device) that has to be asked for once
struct SmemAttr {
    size_t set[32] = {};
    int ensure(const void* kernel, size_t smem) {
        if (smem <= 48 * 1024) return 0;
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = -1;
        if (dev >= 0 && set[dev] >= smem) return 0;
        hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
        if (dev >= 0) set[dev] = smem;
        return 0;
    }
};

template <class BODY>
struct body_multi {      // a body may opt out of the many-problem form (`static constexpr bool MULTI = false`): its records are issued one by one
    template <class B>
    static constexpr auto test(int) -> decltype(B::MULTI) { return B::MULTI; }
    template <class B>
    static constexpr bool test(...) { return true; }
    static constexpr bool value = test<BODY>(0);
};

template <class BODY, class... A>
struct Issue {
    // peel the pack back into a parameter list
    template <class... B>
    static int one(dim3 grid, size_t smem, hipStream_t st, const Pack<>&, B... bound) {
        static SmemAttr attr;
        if (int e = attr.ensure((const void*)one_kernel<BODY, A...>, smem)) return e;
        one_kernel<BODY, A...><<<grid, dim3(BODY::THREADS), smem, st>>>(bound...);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? 0 : (int)e;
    }
    template <class H, class... T, class... B>
    static int one(dim3 grid, size_t smem, hipStream_t st, const Pack<H, T...>& p, B... bound) {
        return one(grid, smem, st, p.tail, bound..., p.head);
    }

    static int group(const Record* const* recs, int n, hipStream_t st) {
        typedef Pack<A...> P;
        static_assert(sizeof(P) <= kMaxArgBytes, "argument block of a recorded launch");
        static_assert(std::is_trivially_copyable<P>::value, "kernel arguments are plain data");
        if (n == 1) {
            P p;
            memcpy(&p, recs[0]->args, sizeof(P));
            return one(dim3(recs[0]->grid.x, recs[0]->grid.y, recs[0]->grid.z), recs[0]->smem, st, p);
        }
        if constexpr (body_multi<BODY>::value) {
            if (n > kMaxGroup) return -1;
            Table<P> t;
            memset(&t, 0, sizeof(t));
            t.n = n;
            long blocks = 0;
            size_t smem = 0;
            for (int i = 0; i < n; ++i) {
                memcpy(&t.p[i], recs[i]->args, sizeof(P));
                t.grid[i] = recs[i]->grid;
                t.first[i] = (int)blocks;
                blocks += (long)recs[i]->grid.x * recs[i]->grid.y * recs[i]->grid.z;
                blocks = (blocks + 7) & ~7L;
                if (recs[i]->smem > smem) smem = recs[i]->smem;
            }
            if (blocks >= (1L << 31)) return -1;
            t.first[n] = (int)blocks;
            static SmemAttr attr;
            if (int e = attr.ensure((const void*)many_kernel<BODY, A...>, smem)) return e;
            many_kernel<BODY, A...><<<dim3((unsigned)blocks), dim3(BODY::THREADS), smem, st>>>(t);
            const hipError_t e = hipGetLastError();
            return e == hipSuccess ? 0 : (int)e;
        } else {
            return -1;               // the zip never groups records of max_group 1
        }
    }
    static const char* name() { return __PRETTY_FUNCTION__; }
};

// The launch of a kernel written as a BODY: recorded while a scope is open, issued at once otherwise.  Returns 0 or a hipError_t.
template <class BODY, class... A>
static inline int launch_body(dim3 grid, size_t smem, hipStream_t st, A... a) {
    typedef Pack<A...> P;
    if (recording()) {
        if constexpr (sizeof(P) <= kMaxArgBytes) {
            Record* r = new_record();
            if (r) {
                r->launch = &Issue<BODY, A...>::group;
                r->max_group = body_multi<BODY>::value ? kMaxGroup : 1;
                r->name = Issue<BODY, A...>::name();
                r->grid = U3{grid.x, grid.y, grid.z};
                r->smem = (unsigned)smem;
                const P p = P::make(a...);
                memcpy(r->args, &p, sizeof(P));
                return 0;
            }
        }
        st = direct_stream(st, Issue<BODY, A...>::name());
    }
    return Issue<BODY, A...>::one(grid, smem, st, P::make(a...));
}

template <class T>
struct unparen;
template <class T>
struct unparen<void(T)> {
    typedef T type;
};

}   // namespace semseg_batch

// SEMSEG_LAUNCH_BODY((body_type<..>), grid, smem_bytes, stream, arguments...) inside a function that returns int: a launch error returns
#define SEMSEG_LAUNCH_BODY(BODY, grid, smem, st, ...)                                                                              \
    do {                                                                                                                           \
        if (int e__ = semseg_batch::launch_body<typename semseg_batch::unparen<void BODY>::type>((grid), (smem), (st), __VA_ARGS__)) \
            return e__;                                                                                                            \
    } while (0)
