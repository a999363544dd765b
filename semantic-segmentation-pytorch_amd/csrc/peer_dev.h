// Device side of the one-node peer exchange (csrc/peer.hip; protocol described there): shared by the stand-alone
// all-reduce kernel and by the BN finish kernels that exchange their per-channel sums in place (csrc/bn.hip).
#pragma once
#include "common.h"

namespace semseg_peer {

constexpr int kMaxWorld = 8;        // one node
constexpr int kSlots = 4;

struct PeerArgs {                   // passed by value to the kernels (baked into a captured graph: set up before capture)
    unsigned long long* inbox[kMaxWorld];     // inbox[r]: rank r's inbox in THIS process' address space
    unsigned* seq;                            // device: exchange counter of this rank
    unsigned* done;                           // device: blocks of the running exchange kernel that have finished
    unsigned* status;                         // host-mapped: != 0 after a timeout
    long long timeout_ticks;                  // wall_clock64 ticks (100 MHz)
    int rank, world, cap;
};

// host: the launch arguments of a context created by semseg_peer_create, false unless every peer is attached
__attribute__((visibility("hidden"))) bool peer_args(void* peer, PeerArgs* out);

__device__ __forceinline__ void st_sys(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned tag_of(unsigned q) { return q % 0xFFFFFFFEu + 1u; }
__device__ __forceinline__ size_t slot_base(const PeerArgs& a, unsigned q) {
    return (size_t)(q % kSlots) * a.world * 2 * (size_t)a.cap;
}

// element `idx` of my payload -> every peer's inbox (nearest-rank-first rotation spreads the first stores over the links)
__device__ __forceinline__ void push(const PeerArgs& a, unsigned q, int idx, double v) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned long long hi_tag = (unsigned long long)tag_of(q) << 32;
    const unsigned long long w0 = hi_tag | (bits & 0xFFFFFFFFull), w1 = hi_tag | (bits >> 32);
    const size_t at = slot_base(a, q) + (size_t)a.rank * 2 * (size_t)a.cap + 2 * (size_t)idx;
    for (int p = 1; p < a.world; ++p) {
        int r = a.rank + p;
        if (r >= a.world) r -= a.world;
        st_sys(a.inbox[r] + at, w0);
        st_sys(a.inbox[r] + at + 1, w1);
    }
}

// sum over the ranks of element `idx`, in rank order 0 .. world-1 on every rank (my own term is `own`; the words of all peers
// are requested together and re-polled until they carry the tag).  Past the deadline: NaN and `timed_out`.
__device__ __forceinline__ double gather_sum(const PeerArgs& a, unsigned q, int idx, double own, long long t0, bool& timed_out) {
    const unsigned tag = tag_of(q);
    const unsigned long long* mine = a.inbox[a.rank] + slot_base(a, q) + 2 * (size_t)idx;
    const size_t lane_words = 2 * (size_t)a.cap;
    unsigned long long w0[kMaxWorld], w1[kMaxWorld];
    unsigned pending = 0;
#pragma unroll
    for (int r = 0; r < kMaxWorld; ++r)
        if (r < a.world && r != a.rank) pending |= 1u << r;
    while (pending) {
#pragma unroll
        for (int r = 0; r < kMaxWorld; ++r)
            if (pending >> r & 1u) {
                w0[r] = ld_sys(mine + (size_t)r * lane_words);
                w1[r] = ld_sys(mine + (size_t)r * lane_words + 1);
            }
#pragma unroll
        for (int r = 0; r < kMaxWorld; ++r)
            if ((pending >> r & 1u) && (unsigned)(w0[r] >> 32) == tag && (unsigned)(w1[r] >> 32) == tag) pending &= ~(1u << r);
        if (pending && wall_clock64() - t0 > a.timeout_ticks) {
            timed_out = true;
            return __longlong_as_double(0x7FF8000000000000ll);
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int r = 0; r < kMaxWorld; ++r)
        if (r < a.world) {
            const double v = r == a.rank ? own : __longlong_as_double((long long)((w1[r] << 32) | (w0[r] & 0xFFFFFFFFull)));
            acc = r == 0 ? v : acc + v;
        }
    return acc;
}

// End of an exchange kernel, ONE thread per block, after the block's last gather: the last block to arrive advances the exchange
// counter.  Every block read the counter when it started, so nobody can see the new value before the next kernel; the ticket is
// a relaxed device-scope atomic (ordering only, no data travels through it: no fence, no cache write-back).
__device__ __forceinline__ void advance(const PeerArgs& a, unsigned q, unsigned nblocks) {
    if (nblocks == 1) {
        *a.seq = q + 1;
        return;
    }
    const unsigned t = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == nblocks - 1) {
        __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *a.seq = q + 1;
    }
}

}  // namespace semseg_peer
