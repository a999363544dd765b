// The recorder behind semseg_batch::launch_body (csrc/batch.h): one process-wide "branches" scope, the lists of recorded launches
// per branch, and the zip that issues them position by position.  C ABI: include/semseg_hip.h, "side-by-side launches".
#include "common.h"
#include "batch.h"
#include <mutex>
#include <vector>

namespace semseg_batch {

struct State {
    bool active = false;
    int branches = 0, cur = 0;
    hipStream_t stream = nullptr;        // every launch of the scope -- recorded or not -- leaves on this stream
    int error = 0;                       // first error of a flush made on behalf of a direct launch (reported by end)
    std::vector<Record> list[kMaxBranches];
    int op[kMaxBranches] = {};
    long recorded = 0, launches = 0, problems = 0, scopes = 0;
};
static State g;
static std::recursive_mutex g_mu;

bool recording() { return g.active; }

Record* new_record() {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return nullptr;
    std::vector<Record>& v = g.list[g.cur];
    v.emplace_back();
    Record* r = &v.back();
    r->op = g.op[g.cur];
    ++g.recorded;
    return r;
}

void count_launch(int problems) {
    ++g.launches;
    g.problems += problems;
}

// Zip: every branch's list is in issue order with non-decreasing op ordinals.  Walk the ordinals upwards; inside one ordinal (the
// launches of one C-ABI call: e.g. GEMM [+ split-K reduce]) match the branches' records front to front -- the head of the first
// branch that still has records of this ordinal names the kernel instantiation, every other branch whose head is the same
// instantiation joins the group (up to kMaxGroup), the group leaves as one launch.  Branch order inside a group = branch index.
int flush_recorded() {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return 0;
    const hipStream_t st = g.stream;
    size_t cur[kMaxBranches] = {};
    const int nb = g.branches;
    int rc = 0;
    for (;;) {
        int op = -1;
        for (int b = 0; b < nb; ++b)
            if (cur[b] < g.list[b].size() && (op < 0 || g.list[b][cur[b]].op < op)) op = g.list[b][cur[b]].op;
        if (op < 0) break;
        for (;;) {
            const Record* grp[kMaxGroup];
            int members[kMaxGroup];
            int n = 0;
            GroupLaunch fn = nullptr;
            for (int b = 0; b < nb && n < kMaxGroup; ++b) {
                if (cur[b] >= g.list[b].size()) continue;
                const Record& r = g.list[b][cur[b]];
                if (r.op != op) continue;
                if (!fn) fn = r.launch;
                if (r.launch != fn) continue;
                grp[n] = &r;
                members[n] = b;
                ++n;
            }
            if (n == 0) break;
            if (!rc) rc = fn(grp, n, st);
            for (int i = 0; i < n; ++i) ++cur[members[i]];
        }
    }
    for (int b = 0; b < nb; ++b) g.list[b].clear();
    return rc;
}

hipStream_t direct_stream(hipStream_t requested) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return requested;
    const int rc = flush_recorded();
    if (rc && !g.error) g.error = rc;
    return g.stream;
}

}   // namespace semseg_batch

using namespace semseg_batch;

extern "C" int semseg_batch_begin(int branches, void* stream) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g.active || branches < 1 || branches > kMaxBranches) return SEMSEG_EINVAL;
    g.active = true;
    g.branches = branches;
    g.cur = 0;
    g.stream = (hipStream_t)stream;
    g.error = 0;
    for (int b = 0; b < kMaxBranches; ++b) {
        g.list[b].clear();
        g.op[b] = 0;
    }
    ++g.scopes;
    return 0;
}

extern "C" int semseg_batch_branch(int index) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active || index < 0 || index >= g.branches) return SEMSEG_EINVAL;
    g.cur = index;
    return 0;
}

extern "C" int semseg_batch_next_op(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return SEMSEG_EINVAL;
    ++g.op[g.cur];
    return 0;
}

extern "C" int semseg_batch_flush(void) { return flush_recorded(); }

extern "C" int semseg_batch_end(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return SEMSEG_EINVAL;
    const int rc = flush_recorded();
    g.active = false;
    return g.error ? g.error : rc;
}

extern "C" int semseg_batch_abort(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    for (int b = 0; b < kMaxBranches; ++b) g.list[b].clear();
    g.active = false;
    return 0;
}

extern "C" int semseg_batch_active(void) { return g.active ? 1 : 0; }

// counters since the process started: [0] scopes opened, [1] launches recorded, [2] launches issued through launch_body (recorded
// or not), [3] problems those launches carried (== [2] when nothing was merged)
extern "C" int semseg_batch_stats(long long* out4) {
    if (!out4) return SEMSEG_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    out4[0] = g.scopes;
    out4[1] = g.recorded;
    out4[2] = g.launches;
    out4[3] = g.problems;
    return 0;
}
