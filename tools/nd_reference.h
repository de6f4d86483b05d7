// The sequential, stack-driven nested dissection of round 4's first form of csrc/slu_analyse.h, kept as the
// CHECKER of the in-place / threaded one (tools/nd_compare.cpp, tests/test_direct_analysis.py): both must give
// the same elimination order and the same supernode boundaries on every graph.  Test infrastructure only.
#pragma once
#include "../nodal_amd/csrc/slu_analyse.h"

namespace slu {

// Nested dissection by level structures.  order: vertices in elimination order; sn_start: supernode
// boundaries in that order.
inline void nested_dissection_ref(int64_t n, const Graph &g, std::vector<int32_t> &order, std::vector<int32_t> &sn_start) {
    order.clear();
    order.reserve((size_t)n);
    sn_start.assign(1, 0);
    auto emit = [&](const std::vector<int32_t> &vs) {
        if (vs.empty()) return;
        order.insert(order.end(), vs.begin(), vs.end());
        sn_start.push_back((int32_t)order.size());
    };
    // hubs: set aside, eliminated last
    const double avg = n > 0 ? (double)g.adj.size() / (double)n : 0.0;
    const int64_t hub_bar = std::max<int64_t>(64, (int64_t)(20.0 * avg));
    std::vector<int32_t> hubs, rest;
    std::vector<int32_t> tag((size_t)n, -1);  // task a vertex currently belongs to (-2: hub / done)
    for (int64_t v = 0; v < n; ++v) {
        if (g.ptr[(size_t)v + 1] - g.ptr[(size_t)v] > hub_bar) {
            hubs.push_back((int32_t)v);
            tag[(size_t)v] = -2;
        } else {
            rest.push_back((int32_t)v);
        }
    }
    struct Task {
        std::vector<int32_t> vs;
        bool separator;   // emit as it is
        bool connected;   // known to be one component
    };
    std::vector<Task> stack;
    stack.push_back(Task{std::move(rest), false, false});
    std::vector<int32_t> lvl((size_t)n, -1), queue;
    int32_t next_id = 0;
    int32_t base = 0;  // (levels are made unique per search by a growing base: no clearing pass)
    while (!stack.empty()) {
        Task t = std::move(stack.back());
        stack.pop_back();
        if (t.vs.empty()) continue;
        if (t.separator) {
            emit(t.vs);
            continue;
        }
        const int32_t id = next_id++;
        for (int32_t v : t.vs) tag[(size_t)v] = id;
        auto bfs = [&](int32_t root, int32_t mark_base) -> int32_t {  // levels lvl[v] = mark_base + depth; returns last vertex
            queue.clear();
            queue.push_back(root);
            lvl[(size_t)root] = mark_base;
            size_t head = 0;
            while (head < queue.size()) {
                const int32_t v = queue[head++];
                for (int64_t e = g.ptr[(size_t)v]; e < g.ptr[(size_t)v + 1]; ++e) {
                    const int32_t u = g.adj[(size_t)e];
                    if (tag[(size_t)u] != id || lvl[(size_t)u] >= mark_base) continue;
                    lvl[(size_t)u] = lvl[(size_t)v] + 1;
                    queue.push_back(u);
                }
            }
            return queue.back();
        };
        if (base > (1 << 30)) {
            std::fill(lvl.begin(), lvl.end(), -1);
            base = 0;
        }
        if (!t.connected) {
            // split into connected components first
            const int32_t b0 = base;
            base += (int32_t)t.vs.size() + 2;
            std::vector<Task> comps;
            for (int32_t v : t.vs) {
                if (lvl[(size_t)v] >= b0) continue;
                (void)bfs(v, b0);
                comps.push_back(Task{queue, false, true});
            }
            if (comps.size() > 1) {
                for (auto &c : comps) stack.push_back(std::move(c));
                continue;
            }
        }
        if ((int64_t)t.vs.size() <= LEAF) {
            emit(t.vs);
            continue;
        }
        // pseudo-peripheral vertex: two sweeps
        int32_t b1 = base;
        base += (int32_t)t.vs.size() + 2;
        int32_t far = bfs(t.vs[0], b1);
        b1 = base;
        base += (int32_t)t.vs.size() + 2;
        far = bfs(far, b1);
        const int32_t b2 = base;
        base += (int32_t)t.vs.size() + 2;
        (void)bfs(far, b2);
        const int32_t nlev = lvl[(size_t)queue.back()] - b2 + 1;
        if (nlev < 3) {  // a clique-like piece: one dense supernode
            emit(t.vs);
            continue;
        }
        std::vector<int64_t> count((size_t)nlev, 0);
        for (int32_t v : queue) ++count[(size_t)(lvl[(size_t)v] - b2)];
        const int64_t total = (int64_t)queue.size();
        int32_t best = -1;
        double best_cost = 1e300;
        int64_t below = count[0];
        for (int32_t m = 1; m + 1 < nlev; ++m) {
            const double frac = (double)below / (double)total;
            // small separator, balanced halves: size * (1 + penalty for imbalance)
            const double imb = std::fabs(frac + 0.5 * (double)count[(size_t)m] / (double)total - 0.5);
            const double cost = (double)count[(size_t)m] * (1.0 + 8.0 * imb * imb * 4.0) + (imb > 0.3 ? 1e9 * imb : 0.0);
            if (cost < best_cost) { best_cost = cost; best = m; }
            below += count[(size_t)m];
        }
        Task A{{}, false, false}, B{{}, false, false}, S{{}, true, true};
        for (int32_t v : queue) {
            const int32_t l = lvl[(size_t)v] - b2;
            if (l < best) A.vs.push_back(v);
            else if (l > best) B.vs.push_back(v);
            else {
                bool up = false;
                for (int64_t e = g.ptr[(size_t)v]; e < g.ptr[(size_t)v + 1] && !up; ++e) {
                    const int32_t u = g.adj[(size_t)e];
                    up = tag[(size_t)u] == id && lvl[(size_t)u] - b2 == best + 1;
                }
                (up ? S.vs : A.vs).push_back(v);
            }
        }
        if (S.vs.empty() || A.vs.empty() || B.vs.empty()) {  // (cannot happen on a connected piece with >= 3 levels)
            emit(t.vs);
            continue;
        }
        for (int32_t v : S.vs) tag[(size_t)v] = -2;
        stack.push_back(std::move(S));
        stack.push_back(std::move(B));
        stack.push_back(std::move(A));
    }
    // a hub supernode of thousands of vertices would be one huge dense pivot block: chunks of 256 instead
    for (size_t k = 0; k < hubs.size(); k += 256) {
        std::vector<int32_t> part(hubs.begin() + (long)k, hubs.begin() + (long)std::min(hubs.size(), k + 256));
        emit(part);
    }
}


}  // namespace slu
