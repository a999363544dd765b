// The recorder behind semseg_batch::launch_body (csrc/batch.h): one process-wide "branches" scope, the lists of recorded launches
// per branch, and the zip that issues them position by position.  C ABI: include/semseg_hip.h, "side-by-side launches".
#include "common.h"
#include "batch.h"
#include <mutex>
#include <string>
#include <vector>
#include <stdio.h>
#include <stdlib.h>

namespace semseg_batch {

struct State {
    bool active = false;
    int branches = 0, cur = 0;
    hipStream_t stream = nullptr;        // every launch of the scope -- recorded or not -- leaves on this stream
    int next_cost = 0;
    int error = 0;                       // first error of a flush made on behalf of a direct launch (reported by end)
    std::vector<Record> list[kMaxBranches];
    int op[kMaxBranches] = {};
    long recorded = 0, scopes = 0, flushed_launches = 0, forced = 0;      // semseg_batch_stats
};
static State g;
static std::recursive_mutex g_mu;
static const bool g_debug = [] { const char* e = getenv("SEMSEG_BATCH_DEBUG"); return e && atoi(e) != 0; }();

// the body's name out of __PRETTY_FUNCTION__ of Issue<BODY, A...>::name()
static std::string short_name(const char* pretty) {
    if (!pretty) return "?";
    std::string s(pretty);
    const size_t a = s.find("BODY = ");
    if (a == std::string::npos) return s.substr(0, 80);
    size_t b = s.find(", A = ", a);
    if (b == std::string::npos) b = s.find(']', a);
    return s.substr(a + 7, b == std::string::npos ? 80 : b - a - 7);
}

bool recording() { return g.active; }

Record* new_record() {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return nullptr;
    std::vector<Record>& v = g.list[g.cur];
    v.emplace_back();
    Record* r = &v.back();
    r->op = g.op[g.cur];
    r->cost = g.next_cost;
    g.next_cost = 0;
    ++g.recorded;
    return r;
}

void hint_cost(int cost) {
    if (g.active) g.next_cost = cost;
}

// Zip: every branch's list is in issue order with non-decreasing op ordinals.  Walk the ordinals upwards; inside one ordinal (the
// launches of one C-ABI call: e.g. GEMM [+ split-K reduce], or statistics sweep + finish) match the branches' records front to
// front: one branch's head names the kernel instantiation that leaves next (see below which), every other branch whose head is the
// same instantiation joins the group (up to kMaxGroup / the instantiation's own limit), the group leaves as one launch, its problems
// ordered longest blocks first.  Records of a branch never overtake each other; branches do not depend on each other.
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
            // which kernel leaves next: the head of some branch's records of this ordinal -- preferably one that no other branch
            // still has FURTHER BACK in its list (a statistics sweep that only the split-K branches launch goes first, so that the
            // finish kernels behind it meet the other branches' finish kernels)
            int cap = kMaxGroup;
            for (int pass = 0; pass < 2 && !fn; ++pass) {
                for (int b = 0; b < nb && !fn; ++b) {
                    if (cur[b] >= g.list[b].size() || g.list[b][cur[b]].op != op) continue;
                    const GroupLaunch cand = g.list[b][cur[b]].launch;
                    bool later = false;
                    for (int o = 0; o < nb && !later && pass == 0; ++o)
                        for (size_t k = cur[o] + 1; k < g.list[o].size() && g.list[o][k].op == op; ++k)
                            if (g.list[o][k].launch == cand && g.list[o][cur[o]].launch != cand) { later = true; break; }
                    if (!later) {
                        fn = cand;
                        cap = g.list[b][cur[b]].max_group < kMaxGroup ? g.list[b][cur[b]].max_group : kMaxGroup;
                    }
                }
            }
            for (int b = 0; b < nb && n < cap; ++b) {
                if (cur[b] >= g.list[b].size()) continue;
                const Record& r = g.list[b][cur[b]];
                if (r.op != op || r.launch != fn) continue;
                grp[n] = &r;
                members[n] = b;
                ++n;
            }
            if (n == 0) break;
            if (g_debug) {
                fprintf(stderr, "[semseg_batch] op %d: %d x %s (branches", op, n, short_name(grp[0]->name).c_str());
                for (int i = 0; i < n; ++i) fprintf(stderr, " %d", members[i]);
                fprintf(stderr, ")\n");
            }
            g.flushed_launches += 1;
            for (int i = 1; i < n; ++i)            // longest blocks first (stable insertion sort on the launch sites' hints)
                for (int k = i; k > 0 && grp[k]->cost > grp[k - 1]->cost; --k) {
                    const Record* t = grp[k]; grp[k] = grp[k - 1]; grp[k - 1] = t;
                }
            if (!rc) rc = fn(grp, n, st);
            for (int i = 0; i < n; ++i) ++cur[members[i]];
        }
    }
    for (int b = 0; b < nb; ++b) g.list[b].clear();
    return rc;
}

hipStream_t direct_stream(hipStream_t requested, const char* what) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return requested;
    ++g.forced;
    if (g_debug) fprintf(stderr, "[semseg_batch] branch %d op %d: direct launch of %s forces a flush\n", g.cur, g.op[g.cur], what ? short_name(what).c_str() : "?");
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

// the host enters the next UNIT of the current branch (one autograd node: conv -> BN -> ReLU forward, or its backward): ordinals
// restart at a multiple of 256, so a branch that made a call more or less inside a unit (a layout conversion, a gradient sum formed
// on the spot) is back in step with the others at the next unit
extern "C" int semseg_batch_next_unit(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g.active) return SEMSEG_EINVAL;
    g.op[g.cur] = ((g.op[g.cur] >> 8) + 1) << 8;
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

// counters since the process started: [0] scopes opened, [1] launches recorded, [2] launches the zip issued for them, [3] direct
// launches made inside a scope (each forced a flush of what had been recorded: an unconverted kernel on a branch's path)
extern "C" int semseg_batch_stats(long long* out4) {
    if (!out4) return SEMSEG_EINVAL;
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    out4[0] = g.scopes;
    out4[1] = g.recorded;
    out4[2] = g.flushed_launches;
    out4[3] = g.forced;
    return 0;
}
