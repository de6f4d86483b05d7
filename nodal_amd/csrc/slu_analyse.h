// Host-side analysis of the sparse direct route (sparse_direct.hip): row matching, nested dissection,
// symbolic factorisation over supernodes.  Plain C++ (no HIP): also compiled by tools/slu_host_check.cpp,
// which runs the numeric phase on the host with the same structures and compares with SuperLU.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace slu {

constexpr int LEAF = 32;          // pieces of at most this many vertices are not dissected further


// rmatch[j] = row matched to column j (a perfect matching on the non-zero entries), false if none exists
inline bool row_matching(int64_t n, const int32_t *indptr, const int32_t *indices, const double *data,
                  std::vector<int32_t> &rmatch) {
    std::vector<int32_t> cmatch((size_t)n, -1);
    rmatch.assign((size_t)n, -1);
    // 1. the diagonal where it carries weight (node rows: the sum of the conductances)
    for (int64_t i = 0; i < n; ++i) {
        double rowmax = 0.0, diag = 0.0;
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const double v = std::fabs(data[e]);
            if (v > rowmax) rowmax = v;
            if (indices[e] == (int32_t)i) diag = v;
        }
        if (diag > 0.0 && diag >= 0.01 * rowmax) {
            cmatch[(size_t)i] = (int32_t)i;
            rmatch[(size_t)i] = (int32_t)i;
        }
    }
    // 2. the largest entry in a free column
    for (int64_t i = 0; i < n; ++i) {
        if (cmatch[(size_t)i] >= 0) continue;
        double best = 0.0;
        int32_t bj = -1;
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const double v = std::fabs(data[e]);
            const int32_t j = indices[e];
            if (v > best && rmatch[(size_t)j] < 0) { best = v; bj = j; }
        }
        if (bj >= 0) {
            cmatch[(size_t)i] = bj;
            rmatch[(size_t)bj] = (int32_t)i;
        }
    }
    // 3. SHORTEST augmenting paths (breadth-first) for what is left: a branch row takes the incidence entry of
    // one of its lead nodes and that node's row the branch column -- two rows move.  (Depth-first search walks
    // through the whole network first -- paths of a thousand rows on a 40 x 40 grid -- and every node row on
    // the path trades its dominant diagonal for an off-diagonal -g.)
    std::vector<int32_t> stamp((size_t)n, -1), via((size_t)n, -1), queue;
    for (int64_t r0 = 0; r0 < n; ++r0) {
        if (cmatch[(size_t)r0] >= 0) continue;
        queue.assign(1, (int32_t)r0);
        bool found = false;
        int32_t end_col = -1;
        for (size_t head = 0; head < queue.size() && !found; ++head) {
            const int32_t i = queue[head];
            for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
                const int32_t j = indices[e];
                if (data[e] == 0.0 || stamp[(size_t)j] == (int32_t)r0) continue;
                stamp[(size_t)j] = (int32_t)r0;
                via[(size_t)j] = i;
                if (rmatch[(size_t)j] < 0) {
                    end_col = j;
                    found = true;
                    break;
                }
                queue.push_back(rmatch[(size_t)j]);
            }
        }
        if (!found) return false;
        if (getenv("SLU_DEBUG")) fprintf(stderr, "  augment row %d: %zu rows visited, ends in column %d\n", (int)r0, queue.size(), end_col);
        // flip the path: column end_col <- via row, that row's old column <- its via row, ...
        int32_t j = end_col;
        while (true) {
            const int32_t i = via[(size_t)j];
            const int32_t old = cmatch[(size_t)i];
            cmatch[(size_t)i] = j;
            rmatch[(size_t)j] = i;
            if (i == (int32_t)r0) break;
            j = old;
        }
    }
    return true;
}

struct Graph {
    std::vector<int64_t> ptr;
    std::vector<int32_t> adj;
};

// vertices = columns; edge p -- l iff (P A)(p, l) != 0 or (P A)(l, p) != 0, p != l
inline void symmetrised_graph(int64_t n, const int32_t *indptr, const int32_t *indices, const std::vector<int32_t> &rmatch,
                       Graph &g) {
    std::vector<int32_t> cmatch((size_t)n);
    for (int64_t j = 0; j < n; ++j) cmatch[(size_t)rmatch[(size_t)j]] = (int32_t)j;
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t p = cmatch[(size_t)i];
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int32_t l = indices[e];
            if (l == p) continue;
            ++cnt[(size_t)p + 1];
            ++cnt[(size_t)l + 1];
        }
    }
    for (int64_t v = 0; v < n; ++v) cnt[(size_t)v + 1] += cnt[(size_t)v];
    std::vector<int32_t> raw((size_t)cnt[(size_t)n]);
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t p = cmatch[(size_t)i];
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int32_t l = indices[e];
            if (l == p) continue;
            raw[(size_t)fill[(size_t)p]++] = l;
            raw[(size_t)fill[(size_t)l]++] = p;
        }
    }
    g.ptr.assign((size_t)n + 1, 0);
    g.adj.clear();
    g.adj.reserve(raw.size());
    for (int64_t v = 0; v < n; ++v) {
        auto b = raw.begin() + cnt[(size_t)v], e = raw.begin() + cnt[(size_t)v + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        g.adj.insert(g.adj.end(), b, e);
        g.ptr[(size_t)v + 1] = (int64_t)g.adj.size();
    }
}

// Nested dissection by level structures.  order: vertices in elimination order; sn_start: supernode
// boundaries in that order.
inline void nested_dissection(int64_t n, const Graph &g, std::vector<int32_t> &order, std::vector<int32_t> &sn_start) {
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

struct Symbolic {
    std::vector<int32_t> rowof, colof, newrow;
    std::vector<int32_t> sn_start;
    std::vector<int64_t> struct_ptr;
    std::vector<int32_t> struct_idx, cmap;
    std::vector<int32_t> parent, level;
    std::vector<int64_t> front_off, vec_off;
    std::vector<int32_t> lvl_ptr, lvl_sn, lvl_maxdim;
    std::vector<int32_t> child_ptr, child_idx;
    std::vector<int64_t> dest;
    int32_t max_dim = 0;
};

// false: structurally singular (no perfect matching)
inline bool analyse(int64_t n, const int32_t *indptr, const int32_t *indices, const double *data, Symbolic &S,
             bool trace) {
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](auto t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    std::vector<int32_t> rmatch;
    if (!row_matching(n, indptr, indices, data, rmatch)) return false;
    const double t_match = ms_since(t0);
    Graph g;
    symmetrised_graph(n, indptr, indices, rmatch, g);
    const double t_graph = ms_since(t0);
    std::vector<int32_t> order;
    nested_dissection(n, g, order, S.sn_start);
    const double t_nd = ms_since(t0);
    const int32_t nsn = (int32_t)S.sn_start.size() - 1;
    std::vector<int32_t> newpos((size_t)n), sn_of((size_t)n);
    for (int64_t k = 0; k < n; ++k) newpos[(size_t)order[(size_t)k]] = (int32_t)k;
    for (int32_t t = 0; t < nsn; ++t)
        for (int32_t k = S.sn_start[(size_t)t]; k < S.sn_start[(size_t)t + 1]; ++k) sn_of[(size_t)k] = t;
    S.colof.assign(order.begin(), order.end());
    S.rowof.resize((size_t)n);
    S.newrow.resize((size_t)n);
    for (int64_t k = 0; k < n; ++k) {
        S.rowof[(size_t)k] = rmatch[(size_t)order[(size_t)k]];
        S.newrow[(size_t)S.rowof[(size_t)k]] = (int32_t)k;
    }
    // symbolic factorisation over the supernodes
    S.struct_ptr.assign((size_t)nsn + 1, 0);
    S.struct_idx.clear();
    S.parent.assign((size_t)nsn, -1);
    S.level.assign((size_t)nsn, 0);
    std::vector<int32_t> first_child((size_t)nsn, -1), next_sibling((size_t)nsn, -1), last_child((size_t)nsn, -1);
    std::vector<int32_t> stamp((size_t)n, -1), list;
    for (int32_t t = 0; t < nsn; ++t) {
        const int32_t end = S.sn_start[(size_t)t + 1];
        list.clear();
        for (int32_t k = S.sn_start[(size_t)t]; k < end; ++k) {
            const int32_t v = order[(size_t)k];
            for (int64_t e = g.ptr[(size_t)v]; e < g.ptr[(size_t)v + 1]; ++e) {
                const int32_t p = newpos[(size_t)g.adj[(size_t)e]];
                if (p >= end && stamp[(size_t)p] != t) {
                    stamp[(size_t)p] = t;
                    list.push_back(p);
                }
            }
        }
        for (int32_t c = first_child[(size_t)t]; c >= 0; c = next_sibling[(size_t)c]) {
            for (int64_t q = S.struct_ptr[(size_t)c]; q < S.struct_ptr[(size_t)c + 1]; ++q) {
                const int32_t p = S.struct_idx[(size_t)q];
                if (p >= end && stamp[(size_t)p] != t) {
                    stamp[(size_t)p] = t;
                    list.push_back(p);
                }
            }
            S.level[(size_t)t] = std::max(S.level[(size_t)t], S.level[(size_t)c] + 1);
        }
        std::sort(list.begin(), list.end());
        S.struct_idx.insert(S.struct_idx.end(), list.begin(), list.end());
        S.struct_ptr[(size_t)t + 1] = (int64_t)S.struct_idx.size();
        if (!list.empty()) {
            const int32_t p = sn_of[(size_t)list[0]];
            S.parent[(size_t)t] = p;
            if (last_child[(size_t)p] < 0) first_child[(size_t)p] = t;
            else next_sibling[(size_t)last_child[(size_t)p]] = t;
            last_child[(size_t)p] = t;
        }
    }
    // fronts, vectors, levels
    S.front_off.assign((size_t)nsn + 1, 0);
    S.vec_off.assign((size_t)nsn + 1, 0);
    int32_t nlev = 0;
    S.max_dim = 0;
    for (int32_t t = 0; t < nsn; ++t) {
        const int64_t sz = S.sn_start[(size_t)t + 1] - S.sn_start[(size_t)t];
        const int64_t dim = sz + (S.struct_ptr[(size_t)t + 1] - S.struct_ptr[(size_t)t]);
        S.front_off[(size_t)t + 1] = S.front_off[(size_t)t] + dim * dim;
        S.vec_off[(size_t)t + 1] = S.vec_off[(size_t)t] + dim + sz;  // (+ s words: scratch of the row interchanges)
        nlev = std::max(nlev, S.level[(size_t)t] + 1);
        S.max_dim = std::max<int32_t>(S.max_dim, (int32_t)dim);
    }
    S.lvl_ptr.assign((size_t)nlev + 1, 0);
    for (int32_t t = 0; t < nsn; ++t) ++S.lvl_ptr[(size_t)S.level[(size_t)t] + 1];
    for (int32_t l = 0; l < nlev; ++l) S.lvl_ptr[(size_t)l + 1] += S.lvl_ptr[(size_t)l];
    S.lvl_sn.resize((size_t)nsn);
    S.lvl_maxdim.assign((size_t)nlev, 0);
    {
        std::vector<int32_t> fill(S.lvl_ptr.begin(), S.lvl_ptr.end() - 1);
        for (int32_t t = 0; t < nsn; ++t) {
            S.lvl_sn[(size_t)fill[(size_t)S.level[(size_t)t]]++] = t;
            const int32_t dim = (int32_t)((S.sn_start[(size_t)t + 1] - S.sn_start[(size_t)t]) +
                                          (S.struct_ptr[(size_t)t + 1] - S.struct_ptr[(size_t)t]));
            S.lvl_maxdim[(size_t)S.level[(size_t)t]] = std::max(S.lvl_maxdim[(size_t)S.level[(size_t)t]], dim);
        }
    }
    // children lists and their index maps into the parents' fronts
    S.child_ptr.assign((size_t)nsn + 1, 0);
    for (int32_t t = 0; t < nsn; ++t)
        if (S.parent[(size_t)t] >= 0) ++S.child_ptr[(size_t)S.parent[(size_t)t] + 1];
    for (int32_t t = 0; t < nsn; ++t) S.child_ptr[(size_t)t + 1] += S.child_ptr[(size_t)t];
    S.child_idx.resize((size_t)S.child_ptr[(size_t)nsn]);
    {
        std::vector<int32_t> fill(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (int32_t t = 0; t < nsn; ++t)
            if (S.parent[(size_t)t] >= 0) S.child_idx[(size_t)fill[(size_t)S.parent[(size_t)t]]++] = t;  // ascending: a fixed order
    }
    auto local_in = [&](int32_t t, int32_t p) -> int32_t {  // local index of permuted position p in front t
        const int32_t start = S.sn_start[(size_t)t], end = S.sn_start[(size_t)t + 1];
        if (p < end) return p - start;
        const auto b = S.struct_idx.begin() + S.struct_ptr[(size_t)t], e = S.struct_idx.begin() + S.struct_ptr[(size_t)t + 1];
        const auto it = std::lower_bound(b, e, p);
        return (it != e && *it == p) ? (end - start) + (int32_t)(it - b) : -1;
    };
    S.cmap.assign(S.struct_idx.size(), -1);
    for (int32_t c = 0; c < nsn; ++c) {
        const int32_t p = S.parent[(size_t)c];
        if (p < 0) continue;
        for (int64_t q = S.struct_ptr[(size_t)c]; q < S.struct_ptr[(size_t)c + 1]; ++q)
            S.cmap[(size_t)q] = local_in(p, S.struct_idx[(size_t)q]);
    }
    // destination of every CSR entry
    const int64_t nnz = indptr[n];
    S.dest.assign((size_t)nnz, -1);
    for (int64_t i = 0; i < n; ++i) {
        const int32_t kr = S.newrow[(size_t)i];
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int32_t kc = newpos[(size_t)indices[e]];
            const int32_t t = sn_of[(size_t)std::min(kr, kc)];
            const int32_t lr = local_in(t, kr), lc = local_in(t, kc);
            if (lr < 0 || lc < 0) return false;  // (cannot happen: every entry is an edge of the graph)
            const int64_t dim = (S.sn_start[(size_t)t + 1] - S.sn_start[(size_t)t]) +
                                (S.struct_ptr[(size_t)t + 1] - S.struct_ptr[(size_t)t]);
            S.dest[(size_t)e] = S.front_off[(size_t)t] + lr + (int64_t)lc * dim;
        }
    }
    if (trace)
        fprintf(stderr,
                "[direct] analysis: matching %.1f ms, graph %.1f, dissection %.1f, symbolic %.1f; %d supernodes, "
                "%d levels, largest front %d, fronts %.3f GB\n",
                t_match, t_graph - t_match, t_nd - t_graph, ms_since(t0) - t_nd, nsn, nlev, S.max_dim,
                (double)S.front_off[(size_t)nsn] * 8e-9);
    return true;
}


}  // namespace slu
