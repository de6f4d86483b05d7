// Host-side analysis of the sparse direct route (sparse_direct.hip): row matching, nested dissection,
// symbolic factorisation over supernodes.  Plain C++ (no HIP): also compiled by tools/slu_host_check.cpp,
// which runs the numeric phase on the host with the same structures and compares with SuperLU.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <thread>
#include <vector>

namespace slu {

constexpr int LEAF = 32;          // pieces of at most this many vertices are not dissected further

// f(lo, hi) over [0, n) in contiguous chunks on up to 16 host threads (the calling one included); chunks of at
// least `min_chunk` items, one chunk = a plain call.  The loops handed to it write disjoint ranges.
template <class F>
inline void parallel_chunks(int64_t n, int64_t min_chunk, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = getenv("NODAL_HOST_THREADS")) hw = (unsigned)std::max(1, atoi(e));
    const int64_t T = std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), 16, n / std::max<int64_t>(1, min_chunk)}));
    if (T <= 1) {
        f((int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)T - 1);
    for (int64_t c = 1; c < T; ++c) th.emplace_back(f, n * c / T, n * (c + 1) / T);
    f((int64_t)0, n / T);
    for (auto &t : th) t.join();
}


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
    // each vertex's list sorted and made unique in place (independent: threads), then packed
    g.ptr.assign((size_t)n + 1, 0);
    parallel_chunks(n, 50000, [&](int64_t lo, int64_t hi) {
        for (int64_t v = lo; v < hi; ++v) {
            auto b = raw.begin() + cnt[(size_t)v], e = raw.begin() + cnt[(size_t)v + 1];
            std::sort(b, e);
            g.ptr[(size_t)v + 1] = (int64_t)(std::unique(b, e) - b);
        }
    });
    for (int64_t v = 0; v < n; ++v) g.ptr[(size_t)v + 1] += g.ptr[(size_t)v];
    g.adj.resize((size_t)g.ptr[(size_t)n]);
    parallel_chunks(n, 50000, [&](int64_t lo, int64_t hi) {
        for (int64_t v = lo; v < hi; ++v)
            std::copy(raw.begin() + cnt[(size_t)v], raw.begin() + cnt[(size_t)v] + (g.ptr[(size_t)v + 1] - g.ptr[(size_t)v]),
                      g.adj.begin() + g.ptr[(size_t)v]);
    });
}

// Nested dissection by level structures.  order: vertices in elimination order; sn_start: supernode
// boundaries in that order.
//
// A piece is split into its connected components; a connected piece of more than LEAF vertices gets a
// pseudo-peripheral root (two sweeps), a level structure from it, and the level that balances best becomes the
// separator S between A (the levels below, plus the separator's vertices without a neighbour above) and B; the
// order is order(A), order(B), S.  The halves are independent: above nd_par_min() vertices the first one goes to another
// thread (a few levels deep: 16 threads at most), the result is the sequential one's whatever the schedule --
// every piece owns its vertices' `tag` and `lvl` words, and a piece's level marks start above every mark its
// ancestors left on those vertices.
inline int64_t nd_par_min() {  // pieces below this are not worth a thread (NODAL_ND_PAR: the tests thread small graphs)
    static const int64_t m = getenv("NODAL_ND_PAR") ? atoll(getenv("NODAL_ND_PAR")) : 20000;
    return m;
}
inline int nd_par_depth() { static const int d = getenv("NODAL_ND_DEPTH") ? atoi(getenv("NODAL_ND_DEPTH")) : 4; return d; }
//
// In place: a piece is a span [lo, hi) of ONE permutation array, its breadth-first queue the same span of ONE
// scratch array; the split writes A, B, S back into the span in that order, so the finished array IS the
// elimination order and a supernode boundary is a flag at its first position.  (A vector per piece and level --
// 45 n words at 1e6 vertices -- cost more in page faults than the searches did on a cold process.)

struct NdState {
    const Graph &g;
    // (relaxed atomics: a piece reads the tags of its vertices' neighbours, which another thread may be re-tagging
    // for its own piece -- with an id that is never this piece's, old or new)
    std::vector<std::atomic<int32_t>> tag;  // piece a vertex currently belongs to (-2: hub / separator / done)
    std::vector<int32_t> lvl, perm, scratch;
    std::vector<uint8_t> cls, snflag;
    std::atomic<int32_t> next_id{0};
    NdState(const Graph &gr, int64_t n)
        : g(gr), tag((size_t)n), lvl((size_t)n, -1), perm((size_t)n), scratch((size_t)n), cls((size_t)n), snflag((size_t)n + 1, 0) {}
};

// levels lvl[v] = mark_base + depth over the piece `id` from `root`; the visited vertices go to q[0 ..) in
// breadth-first order; returns their number
inline int64_t nd_bfs(NdState &st, int32_t id, int32_t root, int32_t mark_base, int32_t *q) {
    const Graph &g = st.g;
    int64_t head = 0, tail = 0;
    q[tail++] = root;
    st.lvl[(size_t)root] = mark_base;
    while (head < tail) {
        const int32_t v = q[head++];
        for (int64_t e = g.ptr[(size_t)v]; e < g.ptr[(size_t)v + 1]; ++e) {
            const int32_t u = g.adj[(size_t)e];
            if (st.tag[(size_t)u].load(std::memory_order_relaxed) != id || st.lvl[(size_t)u] >= mark_base) continue;
            st.lvl[(size_t)u] = st.lvl[(size_t)v] + 1;
            q[tail++] = u;
        }
    }
    return tail;
}

// `base`: above every mark on the piece's vertices.  `connected`: known to be one component.
inline void nd_dissect(NdState &st, int64_t lo, int64_t hi, bool connected, int32_t base, int depth) {
    if (hi <= lo) return;
    const Graph &g = st.g;
    const int64_t m = hi - lo;
    int32_t *vs = st.perm.data() + lo, *q = st.scratch.data() + lo;
    const int32_t id = st.next_id.fetch_add(1, std::memory_order_relaxed);
    for (int64_t k = 0; k < m; ++k) st.tag[(size_t)vs[k]].store(id, std::memory_order_relaxed);
    const int32_t span = (int32_t)m + 2;
    // The first sweep (from the piece's first vertex) doubles as the connectivity test: if it reaches every vertex
    // the piece is one component and the sweep is the first of the pseudo-peripheral search.
    bool swept = false;
    if (!connected) {
        swept = nd_bfs(st, id, vs[0], base, q) == m;
        base += span;
        connected = swept;
    }
    if (!connected) {
        const int32_t b0 = base;
        base += span;
        std::vector<int64_t> starts;  // of the components inside q, in the order they were found
        int64_t filled = 0;
        for (int64_t k = 0; k < m; ++k) {
            if (st.lvl[(size_t)vs[k]] >= b0) continue;
            starts.push_back(filled);
            filled += nd_bfs(st, id, vs[k], b0, q + filled);
        }
        // (the order of the sequential, stack-driven form this one replaced: the component found last comes
        // first; each in breadth-first order)
        starts.push_back(filled);
        int64_t at = 0;
        std::vector<int64_t> bounds(1, 0);
        for (size_t c = starts.size() - 1; c-- > 0;) {
            std::copy(q + starts[c], q + starts[c + 1], vs + at);
            at += starts[c + 1] - starts[c];
            bounds.push_back(at);
        }
        for (size_t c = 0; c + 1 < bounds.size(); ++c) nd_dissect(st, lo + bounds[c], lo + bounds[c + 1], true, base, depth);
        return;
    }
    if (m <= LEAF) {
        st.snflag[(size_t)lo] = 1;
        return;
    }
    // pseudo-peripheral vertex: two sweeps
    if (!swept) {
        (void)nd_bfs(st, id, vs[0], base, q);
        base += span;
    }
    (void)nd_bfs(st, id, q[m - 1], base, q);
    base += span;
    const int32_t b2 = base;
    (void)nd_bfs(st, id, q[m - 1], b2, q);
    base += span;
    const int32_t nlev = st.lvl[(size_t)q[m - 1]] - b2 + 1;
    if (nlev < 3) {  // a clique-like piece: one dense supernode
        st.snflag[(size_t)lo] = 1;
        return;
    }
    std::vector<int64_t> count((size_t)nlev, 0);
    for (int64_t k = 0; k < m; ++k) ++count[(size_t)(st.lvl[(size_t)q[k]] - b2)];
    int32_t best = -1;
    double best_cost = 1e300;
    int64_t below = count[0];
    for (int32_t l = 1; l + 1 < nlev; ++l) {
        const double frac = (double)below / (double)m;
        // small separator, balanced halves: size * (1 + penalty for imbalance)
        const double imb = std::fabs(frac + 0.5 * (double)count[(size_t)l] / (double)m - 0.5);
        const double cost = (double)count[(size_t)l] * (1.0 + 8.0 * imb * imb * 4.0) + (imb > 0.3 ? 1e9 * imb : 0.0);
        if (cost < best_cost) { best_cost = cost; best = l; }
        below += count[(size_t)l];
    }
    // classes: 0 = A (the levels below the separator's, and its vertices without a neighbour above), 1 = B, 2 = S
    uint8_t *cls = st.cls.data() + lo;
    int64_t nA = 0, nB = 0, nS = 0;
    for (int64_t k = 0; k < m; ++k) {
        const int32_t v = q[k];
        const int32_t l = st.lvl[(size_t)v] - b2;
        uint8_t c = 0;
        if (l > best) c = 1;
        else if (l == best) {
            bool up = false;
            for (int64_t e = g.ptr[(size_t)v]; e < g.ptr[(size_t)v + 1] && !up; ++e) {
                const int32_t u = g.adj[(size_t)e];
                up = st.tag[(size_t)u].load(std::memory_order_relaxed) == id && st.lvl[(size_t)u] - b2 == best + 1;
            }
            c = up ? 2 : 0;
        }
        cls[k] = c;
        nA += c == 0;
        nB += c == 1;
        nS += c == 2;
    }
    if (nS == 0 || nA == 0 || nB == 0) {  // (cannot happen on a connected piece with >= 3 levels)
        st.snflag[(size_t)lo] = 1;
        return;
    }
    int64_t at[3] = {0, nA, nA + nB};
    for (int64_t k = 0; k < m; ++k) vs[at[cls[k]]++] = q[k];  // (each class in breadth-first order)
    for (int64_t k = nA + nB; k < m; ++k) st.tag[(size_t)vs[k]].store(-2, std::memory_order_relaxed);
    st.snflag[(size_t)(lo + nA + nB)] = 1;  // the separator: one supernode, after both halves
    if (depth < nd_par_depth() && nA >= nd_par_min() && nB >= nd_par_min()) {
        auto fa = std::async(std::launch::async, [&st, lo, nA, base, depth]() { nd_dissect(st, lo, lo + nA, false, base, depth + 1); });
        nd_dissect(st, lo + nA, lo + nA + nB, false, base, depth + 1);
        fa.get();
    } else {
        nd_dissect(st, lo, lo + nA, false, base, depth + 1);
        nd_dissect(st, lo + nA, lo + nA + nB, false, base, depth + 1);
    }
}

// `skip` (optional): vertices that are not part of the graph any more (peel_low_degree below has ordered them);
// `order` then holds the others only.
inline void nested_dissection(int64_t n, const Graph &g, std::vector<int32_t> &order, std::vector<int32_t> &sn_start,
                              const std::vector<uint8_t> *skip = nullptr) {
    // hubs: set aside, eliminated last
    const double avg = n > 0 ? (double)g.adj.size() / (double)n : 0.0;
    const int64_t hub_bar = std::max<int64_t>(64, (int64_t)(20.0 * avg));
    NdState st(g, n);
    int64_t nrest = 0, nhubs = 0;
    for (int64_t v = 0; v < n; ++v) {  // perm: the rest in front (ascending), the hubs behind them (ascending)
        if (skip && (*skip)[(size_t)v]) {
            st.tag[(size_t)v].store(-2, std::memory_order_relaxed);
            continue;
        }
        if (g.ptr[(size_t)v + 1] - g.ptr[(size_t)v] > hub_bar) {
            st.scratch[(size_t)nhubs++] = (int32_t)v;
            st.tag[(size_t)v].store(-2, std::memory_order_relaxed);
        } else {
            st.perm[(size_t)nrest++] = (int32_t)v;
            st.tag[(size_t)v].store(-1, std::memory_order_relaxed);
        }
    }
    std::copy(st.scratch.begin(), st.scratch.begin() + nhubs, st.perm.begin() + nrest);
    const int64_t m = nrest + nhubs;  // (= n without a skip list)
    nd_dissect(st, 0, nrest, false, 0, 0);
    // a hub supernode of thousands of vertices would be one huge dense pivot block: chunks of 256 instead
    for (int64_t k = nrest; k < m; k += 256) st.snflag[(size_t)k] = 1;
    order = std::move(st.perm);
    order.resize((size_t)m);
    sn_start.clear();
    for (int64_t k = 0; k < m; ++k)
        if (st.snflag[(size_t)k]) sn_start.push_back((int32_t)k);
    sn_start.push_back((int32_t)m);
    if (m == 0) sn_start.assign(1, 0);
}

// Tree-like parts first.  Level structures dissect a branching tree badly (half of its vertices sit in the last
// level: a 2944-wide front for a binary tree of 20 000 nodes), while eliminating vertices with at most two
// neighbours costs no fill to speak of: a leaf none, a vertex of a chain one edge between its neighbours.  Rounds
// of INDEPENDENT such vertices (no two adjacent: a round is one level of the assembly tree, fronts of dimension
// <= 3; the highest hash wins among adjacent candidates) are ordered first, as lowdeg.hip eliminates them
// numerically on passive networks; what is left -- with the fill edges -- goes to the dissection.  Only when the first
// round takes a sixteenth of the graph (trees, chains, wires; a grid has its four corners); rounds go on while they
// take half a percent of what is left.
inline uint32_t peel_hash(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
inline void peel_low_degree(int64_t n, const Graph &g, std::vector<int32_t> &peeled, std::vector<uint8_t> &gone,
                            Graph &core) {
    peeled.clear();
    {   // (cheap way out: fewer than a sixteenth of the vertices have two neighbours or fewer to begin with)
        int64_t low = 0;
        for (int64_t v = 0; v < n; ++v) low += g.ptr[(size_t)v + 1] - g.ptr[(size_t)v] <= 2;
        if (low * 16 < n) return;
    }
    gone.assign((size_t)n, 0);
    std::vector<std::vector<int32_t>> extra((size_t)n);  // fill edges (both directions)
    std::vector<int32_t> nb;                              // scratch: current neighbours of a vertex
    auto neighbours = [&](int32_t u, std::vector<int32_t> &out) {  // remaining neighbours, each once, at most 3 wanted
        out.clear();
        auto take = [&](int32_t w) {
            if (gone[(size_t)w] || w == u) return;
            for (int32_t x : out)
                if (x == w) return;
            out.push_back(w);
        };
        for (int64_t e = g.ptr[(size_t)u]; e < g.ptr[(size_t)u + 1] && out.size() < 3; ++e) take(g.adj[(size_t)e]);
        for (size_t e = 0; e < extra[(size_t)u].size() && out.size() < 3; ++e) take(extra[(size_t)u][e]);
    };
    std::vector<int32_t> cand, next_cand, chosen;
    std::vector<uint8_t> is_cand((size_t)n, 0), queued((size_t)n, 0);
    for (int64_t v = 0; v < n; ++v) {
        neighbours((int32_t)v, nb);
        if (nb.size() <= 2) { cand.push_back((int32_t)v); is_cand[(size_t)v] = 1; }
    }
    int64_t remaining = n;
    for (int round = 0; round < 64 && !cand.empty(); ++round) {
        // independent set among the candidates: the highest (hash, id) among adjacent candidates
        chosen.clear();
        for (int32_t v : cand) {
            neighbours(v, nb);
            if (nb.size() > 2) { is_cand[(size_t)v] = 0; continue; }  // (its degree grew: a fill edge)
            const uint64_t pv = ((uint64_t)peel_hash((uint32_t)v) << 32) | (uint32_t)v;
            bool best = true;
            for (int32_t w : nb)
                if (is_cand[(size_t)w] && (((uint64_t)peel_hash((uint32_t)w) << 32) | (uint32_t)w) > pv) best = false;
            if (best) chosen.push_back(v);
        }
        if (round == 0 && (int64_t)chosen.size() * 16 < n) break;
        if ((int64_t)chosen.size() * 200 < remaining) break;
        next_cand.clear();
        for (int32_t v : cand)
            if (is_cand[(size_t)v]) queued[(size_t)v] = 0;
        for (int32_t v : chosen) {
            neighbours(v, nb);
            gone[(size_t)v] = 1;
            is_cand[(size_t)v] = 0;
            peeled.push_back(v);
            if (nb.size() == 2) {  // the fill edge, unless it is there already
                int32_t a = nb[0], b = nb[1];
                // (look the edge up from the end with the shorter lists: a hub collecting thousands of chains would
                // otherwise be walked once per chain)
                if ((g.ptr[(size_t)a + 1] - g.ptr[(size_t)a]) + (int64_t)extra[(size_t)a].size() >
                    (g.ptr[(size_t)b + 1] - g.ptr[(size_t)b]) + (int64_t)extra[(size_t)b].size())
                    std::swap(a, b);
                bool have = false;
                for (int64_t e = g.ptr[(size_t)a]; e < g.ptr[(size_t)a + 1] && !have; ++e) have = g.adj[(size_t)e] == b;
                for (size_t e = 0; e < extra[(size_t)a].size() && !have; ++e) have = extra[(size_t)a][e] == b;
                if (!have) {
                    extra[(size_t)a].push_back(b);
                    extra[(size_t)b].push_back(a);
                }
            }
            for (int32_t w : nb)
                if (!queued[(size_t)w]) { queued[(size_t)w] = 1; next_cand.push_back(w); }
        }
        remaining -= (int64_t)chosen.size();
        // candidates of the next round: the old ones that were not chosen, and the neighbours of the chosen ones
        for (int32_t v : cand)
            if (is_cand[(size_t)v] && !gone[(size_t)v] && !queued[(size_t)v]) { queued[(size_t)v] = 1; next_cand.push_back(v); }
        cand.clear();
        for (int32_t v : next_cand) {
            queued[(size_t)v] = 0;
            if (gone[(size_t)v]) continue;
            neighbours(v, nb);
            is_cand[(size_t)v] = nb.size() <= 2;
            if (is_cand[(size_t)v]) cand.push_back(v);
        }
        std::sort(cand.begin(), cand.end());  // (a fixed order whatever the history)
    }
    if (peeled.empty()) return;
    // what is left, with the fill edges
    core.ptr.assign((size_t)n + 1, 0);
    core.adj.clear();
    std::vector<int32_t> list;
    for (int64_t u = 0; u < n; ++u) {
        if (!gone[(size_t)u]) {
            list.clear();
            for (int64_t e = g.ptr[(size_t)u]; e < g.ptr[(size_t)u + 1]; ++e)
                if (!gone[(size_t)g.adj[(size_t)e]]) list.push_back(g.adj[(size_t)e]);
            for (int32_t w : extra[(size_t)u])
                if (!gone[(size_t)w]) list.push_back(w);
            std::sort(list.begin(), list.end());
            list.erase(std::unique(list.begin(), list.end()), list.end());
            core.adj.insert(core.adj.end(), list.begin(), list.end());
        }
        core.ptr[(size_t)u + 1] = (int64_t)core.adj.size();
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
    {
        std::vector<int32_t> peeled;
        std::vector<uint8_t> gone;
        Graph core;
        static const bool peel = !(getenv("NODAL_DIRECT_PEEL") && atoi(getenv("NODAL_DIRECT_PEEL")) == 0);
        if (peel) peel_low_degree(n, g, peeled, gone, core);
        if (peeled.empty()) {
            nested_dissection(n, g, order, S.sn_start);
        } else {
            std::vector<int32_t> rest_order, rest_sn;
            nested_dissection(n, core, rest_order, rest_sn, &gone);
            order = peeled;  // every peeled vertex a supernode of its own
            order.insert(order.end(), rest_order.begin(), rest_order.end());
            S.sn_start.resize(peeled.size());
            for (size_t k = 0; k < peeled.size(); ++k) S.sn_start[k] = (int32_t)k;
            for (int32_t b : rest_sn) S.sn_start.push_back((int32_t)peeled.size() + b);
            if (trace)
                fprintf(stderr, "[direct] %zu of %lld vertices with at most two neighbours ordered first\n", peeled.size(),
                        (long long)n);
        }
    }
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
    const double t_sf = ms_since(t0);
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
    const double t_lv = ms_since(t0);
    S.cmap.assign(S.struct_idx.size(), -1);
    parallel_chunks(nsn, 2000, [&](int64_t lo, int64_t hi) {
        for (int64_t c = lo; c < hi; ++c) {
            const int32_t p = S.parent[(size_t)c];
            if (p < 0) continue;
            for (int64_t q = S.struct_ptr[(size_t)c]; q < S.struct_ptr[(size_t)c + 1]; ++q)
                S.cmap[(size_t)q] = local_in(p, S.struct_idx[(size_t)q]);
        }
    });
    const double t_cm = ms_since(t0);
    // destination of every CSR entry
    const int64_t nnz = indptr[n];
    S.dest.assign((size_t)nnz, -1);
    std::atomic<bool> misplaced{false};
    parallel_chunks(n, 50000, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const int32_t kr = S.newrow[(size_t)i];
            for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
                const int32_t kc = newpos[(size_t)indices[e]];
                const int32_t t = sn_of[(size_t)std::min(kr, kc)];
                const int32_t lr = local_in(t, kr), lc = local_in(t, kc);
                if (lr < 0 || lc < 0) {  // (cannot happen: every entry is an edge of the graph)
                    misplaced.store(true, std::memory_order_relaxed);
                    continue;
                }
                const int64_t dim = (S.sn_start[(size_t)t + 1] - S.sn_start[(size_t)t]) +
                                    (S.struct_ptr[(size_t)t + 1] - S.struct_ptr[(size_t)t]);
                S.dest[(size_t)e] = S.front_off[(size_t)t] + lr + (int64_t)lc * dim;
            }
        }
    });
    if (misplaced.load()) return false;
    if (trace)
        fprintf(stderr,
                "[direct] analysis: matching %.1f ms, graph %.1f, dissection %.1f, symbolic %.1f (structures %.1f, "
                "levels %.1f, child maps %.1f, entry map %.1f); %d supernodes, %d levels, largest front %d, fronts %.3f GB\n",
                t_match, t_graph - t_match, t_nd - t_graph, ms_since(t0) - t_nd, t_sf - t_nd, t_lv - t_sf, t_cm - t_lv,
                ms_since(t0) - t_cm, nsn, nlev, S.max_dim,
                (double)S.front_off[(size_t)nsn] * 8e-9);
    return true;
}


}  // namespace slu
