// Presolve for large general MNA systems: eliminate every branch equation exactly.
//
// A voltage-defined branch m (E, VCVS, CCVS) with leads a, b states
//     e_a - e_b = cst + gain (e_c - e_d)
// so one of its leads -- the pivot p -- is not an unknown at all:
//     e_p = e_q + cst' + gain' (e_c - e_d),      q = the other lead (or ground).
// Substituting that into every component attached to p rewrites the netlist into an
// equivalent one WITHOUT the branch: a resistor p--j becomes a resistor q--j, a
// current source cst'/r and a transconductance gain'/r; p's KCL equation is merged
// into q's (or dropped when q is ground), which removes the branch current i_m from
// the system.  A CCCS is a transconductance outright.  The result is a netlist of
// R / A / GM stamps with B' = 0 on K' = K - #pivots nodes: a (nearly) symmetric
// M-matrix, exactly what the multigrid of amg.hip is good at, instead of the
// saddle-point system whose ~1e4 voltage sources the node-block preconditioner of
// sparse_general.hip cannot see (360 GMRES iterations on config 5).
// Afterwards e_p follows from its expression and i_m from the ORIGINAL KCL row of p.
//
// Supported pattern: the voltage-defined branches form a forest on their lead nodes (a loop of
// sources ends the presolve: the system is singular); every tree is rooted at ground if it
// touches ground and every other node of it is a pivot, defined through the branch to its
// parent.  Chains resolve by substitution (stacked sources add up, a control node that is
// itself a pivot is replaced by its expression, for the branch-less dependent sources (CCCS)
// too) as long as every pivot ends with at most one control term on surviving nodes and no
// substituted control carries a term itself.  A tree with a pivot that breaks these rules
// (cascaded / self-controlled / stacked dependent sources) is not eliminated: its branches stay
// in the reduced system as E / VCVS rows with their branch unknowns (B' > 0) -- everything else
// goes as before, and the reduced system takes the general solver.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>

#include "ctx.h"

namespace {

constexpr int TB = 256;

struct Expr {
    int p, q;       // pivot node, base node (-1 = ground)
    double cst;     // constant
    int c, d;       // control nodes of the single term (-1 = ground / none)
    double g;       // gain of the term (0 = no term)
};

inline unsigned grid_for(int64_t n) {
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > 4096 ? 4096 : g);
}

// ---- recovery kernels -------------------------------------------------------------------

// potentials of the surviving nodes
__global__ __launch_bounds__(TB) void scatter_nodes(int K, const int32_t *__restrict__ newidx,
                                                    const double *__restrict__ y,
                                                    double *__restrict__ x) {
    for (int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x; j < K; j += (int64_t)gridDim.x * TB) {
        const int t = newidx[j];
        if (t >= 0) x[j] = y[t];
    }
}

// eliminated nodes from their expressions (bases and controls are surviving nodes)
__global__ __launch_bounds__(TB) void eval_pivots(int ne, const int32_t *__restrict__ p,
                                                  const int32_t *__restrict__ q,
                                                  const double *__restrict__ cst,
                                                  const int32_t *__restrict__ c,
                                                  const int32_t *__restrict__ d,
                                                  const double *__restrict__ g,
                                                  double *__restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < ne; i += (int64_t)gridDim.x * TB) {
        double v = cst[i];
        if (q[i] >= 0) v += x[q[i]];
        if (g[i] != 0.0) v += g[i] * ((c[i] >= 0 ? x[c[i]] : 0.0) - (d[i] >= 0 ? x[d[i]] : 0.0));
        x[p[i]] = v;
    }
}

// currents of the branches that stayed in the reduced system: its unknowns Kr + k'
__global__ __launch_bounds__(TB) void copy_kept(int nkept, const int32_t *__restrict__ kept_kk, int K, int Kr,
                                                const double *__restrict__ y, double *__restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < nkept; i += (int64_t)gridDim.x * TB)
        x[K + kept_kk[i]] = y[Kr + i];
}

// branch currents from the ORIGINAL system.  pass 0: branches whose own row defines
// them (CCCS: row K+k has a unit diagonal); pass 1 + h: voltage-defined branches whose pivot
// sits at height h of its source tree, from the KCL row of the pivot node (the currents of
// the branches hanging below it and of the CCCS outputs in that row are known by then).
__global__ __launch_bounds__(TB) void recover_currents(int pass, int K, int B,
                                                       const int32_t *__restrict__ level_of,
                                                       const int32_t *__restrict__ row_of,
                                                       const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices,
                                                       const double *__restrict__ data,
                                                       const double *__restrict__ rhs,
                                                       double *__restrict__ x) {
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < B; k += (int64_t)gridDim.x * TB) {
        if (level_of[k] != pass) continue;
        const int row = row_of[k];  // K + k for pass 0 branches, the pivot node otherwise
        const int col = K + (int)k;
        double s = rhs[row], coef = 0.0;
        for (int32_t e = indptr[row]; e < indptr[row + 1]; ++e) {
            const int j = indices[e];
            if (j == col) coef = data[e];
            else s = fma(-data[e], x[j], s);
        }
        x[col] = s / coef;
    }
}

}  // namespace

// A voltage-defined branch the presolve leaves IN the reduced system (its tree of sources has a loop, or
// one of its pivots cannot be expressed: two control terms, a control that carries a term itself ...):
// e_a - e_b = value [+ value (e_c - e_d) for a VCVS; a CCVS becomes a VCVS on its driver's nodes].
struct KeptBranch {
    int kk;       // branch number in the original system
    int type;     // NODAL_T_E or NODAL_T_VCVS
    double value;
    int a, b, c, d;
};

struct PresolvePlan {
    bool ok = false;
    std::vector<Expr> exprs;
    std::vector<KeptBranch> kept;  // in branch order: the reduced system's branch k' is kept[k']
    std::vector<int32_t> pivots;   // sorted pivot nodes
    std::vector<int32_t> row_of;   // B: row that determines each branch current
    std::vector<int32_t> level_of; // B: pass of the current recovery (0 = own row, 1 + height in the source tree)
    int max_level = 0;
    int32_t Kr = 0;
    int32_t newidx(int node) const {  // surviving node -> reduced index (pivots are sorted)
        if (node < 0) return -1;
        return node - (int32_t)(std::lower_bound(pivots.begin(), pivots.end(), node) - pivots.begin());
    }
};

// ---- device side of the rewrite ---------------------------------------------------------

__global__ __launch_bounds__(TB) void mark_pivots(int np, const int32_t *__restrict__ pivots,
                                                  uint32_t *__restrict__ survive) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < np; i += (int64_t)gridDim.x * TB)
        survive[pivots[i]] = 0u;
}
__global__ __launch_bounds__(TB) void fill_u32(uint32_t *p, uint32_t v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        p[i] = v;
}
// newidx[j] = reduced index of node j, -1 for pivots (scan = exclusive scan of survive)
__global__ __launch_bounds__(TB) void make_newidx(int K, const uint32_t *__restrict__ survive,
                                                  const uint32_t *__restrict__ scan,
                                                  int32_t *__restrict__ newidx) {
    for (int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x; j < K; j += (int64_t)gridDim.x * TB)
        newidx[j] = survive[j] ? (int32_t)scan[j] : -1;
}

// the rewritten rows arrive with the original system's node numbers (ground = -1 stays)
// A rewritten component as the host writes it (one 32-byte row: ONE copy brings them all, where eight arrays were
// eight copies of 10-14 us each) and the kernel that places the rows behind the kept components of the reduced
// table, renumbering their nodes (numbers of the ORIGINAL system) on the way.
struct ExtraRow {
    double value;
    int32_t a, b, c, d, k;
    int32_t type;
};
static_assert(sizeof(ExtraRow) == 32, "ExtraRow is one 32-byte row");
__global__ __launch_bounds__(TB) void place_extras(int64_t nx, const ExtraRow *__restrict__ rows,
                                                   const int32_t *__restrict__ newidx, uint8_t *__restrict__ type,
                                                   double *__restrict__ value, int32_t *__restrict__ a,
                                                   int32_t *__restrict__ b, int32_t *__restrict__ c,
                                                   int32_t *__restrict__ d, int32_t *__restrict__ drv,
                                                   int32_t *__restrict__ k) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < nx; i += (int64_t)gridDim.x * TB) {
        const ExtraRow r = rows[i];
        type[i] = (uint8_t)r.type;
        value[i] = r.value;
        a[i] = r.a >= 0 ? newidx[r.a] : -1;
        b[i] = r.b >= 0 ? newidx[r.b] : -1;
        c[i] = r.c >= 0 ? newidx[r.c] : -1;
        d[i] = r.d >= 0 ? newidx[r.d] : -1;
        drv[i] = -1;
        k[i] = r.k;
    }
}

struct DevTable {
    const uint8_t *type;
    const double *value;
    const int32_t *a, *b, *c, *d, *drv;
};
struct DevTableOut {
    uint8_t *type;
    double *value;
    int32_t *a, *b, *c, *d, *drv, *k;
};

// keep[i] = 1: component survives unchanged (up to node renumbering);
// hit[i]  = 1: it touches an eliminated node and is rewritten by the host;
// (a lead OR a control node; the control's expression is substituted); branch components
// are dropped.  (*invalid is kept for patterns the host cannot rewrite.)
__global__ __launch_bounds__(TB) void classify(DevTable t, int64_t nc,
                                               const int32_t *__restrict__ newidx,
                                               uint32_t *__restrict__ keep,
                                               uint32_t *__restrict__ hit,
                                               int32_t *__restrict__ invalid) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < nc; i += (int64_t)gridDim.x * TB) {
        const int ty = t.type[i];
        uint32_t kp = 0, ht = 0;
        if (ty == NODAL_T_R || ty == NODAL_T_A || ty == NODAL_T_CCCS || ty == NODAL_T_GM) {
            const int a = t.a[i], b = t.b[i];
            const bool touched = (a >= 0 && newidx[a] < 0) || (b >= 0 && newidx[b] < 0);
            bool control_eliminated = false;
            if (ty == NODAL_T_CCCS || ty == NODAL_T_GM) {
                const int c = t.c[i], d = t.d[i];
                control_eliminated = (c >= 0 && newidx[c] < 0) || (d >= 0 && newidx[d] < 0);
            }
            if (touched || control_eliminated) ht = 1; else kp = 1;  // the host substitutes
        }
        keep[i] = kp;
        hit[i] = ht;
    }
}

__global__ __launch_bounds__(TB) void compact(DevTable t, int64_t nc,
                                              const int32_t *__restrict__ newidx,
                                              const uint32_t *__restrict__ keep,
                                              const uint32_t *__restrict__ keep_pos,
                                              const uint32_t *__restrict__ hit,
                                              const uint32_t *__restrict__ hit_pos,
                                              DevTableOut o, int32_t *__restrict__ hit_list) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < nc; i += (int64_t)gridDim.x * TB) {
        if (hit[i]) hit_list[hit_pos[i]] = (int32_t)i;
        if (!keep[i]) continue;
        const uint32_t p = keep_pos[i];
        int ty = t.type[i];
        double v = t.value[i];
        auto nid = [&](int node) { return node < 0 ? -1 : newidx[node]; };
        if (ty == NODAL_T_CCCS) {  // a transconductance outright
            v = v / t.value[t.drv[i]];
            ty = NODAL_T_GM;
        }
        o.type[p] = (uint8_t)ty;
        o.value[p] = v;
        o.a[p] = nid(t.a[i]);
        o.b[p] = nid(t.b[i]);
        const bool ctl = ty == NODAL_T_GM;
        o.c[p] = ctl ? nid(t.c[i]) : -1;
        o.d[p] = ctl ? nid(t.d[i]) : -1;
        o.drv[p] = -1;
        o.k[p] = -1;
    }
}

// Host analysis: pivots and expressions from the (few) branch components.
static void presolve_plan(const nodal_ctx *h, const double *value, PresolvePlan &plan) {
    static const bool trace_plan = getenv("NODAL_TRACE") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace_plan)
            fprintf(stderr, "[presolve]   plan: %s at %.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count());
    };
    const HostTable &t = h->host;
    const int64_t nc = h->ncomp;
    const int K = h->K, B = h->B;
    plan.ok = false;
    plan.exprs.clear();
    plan.kept.clear();
    plan.row_of.assign(B, -1);
    plan.level_of.assign(B, 0);
    plan.max_level = 0;
    std::vector<char> seen_k(B, 0);

    // ---- the voltage-defined branches: e_a - e_b = cst + gain (e_c - e_d) ----
    struct Raw { int kk, a, b, c, d; double cst, gain; };
    std::vector<Raw> raws;
    std::vector<int64_t> cccs_rows;
    const uint8_t *ty_arr = t.type.data();
    // (the rows with a branch unknown were listed once at upload: 2e4 of the 2e6 rows of config 5)
    std::vector<int64_t> scanned;
    const std::vector<int64_t> *rows = &t.branch_rows;
    if (rows->empty() && B > 0) {  // a table built on the device (batch.hip) has no list
        for (int64_t i = 0; i < nc; ++i)
            if (ty_arr[i] >= NODAL_T_E && ty_arr[i] <= NODAL_T_CCCS) scanned.push_back(i);
        rows = &scanned;
    }
    for (const int64_t i : *rows) {
        const int ty = ty_arr[i];
        if (ty < NODAL_T_E || ty > NODAL_T_CCCS) continue;
        const int kk = t.k[i];
        if (kk < 0 || kk >= B || seen_k[kk]) return;  // duplicated names: not handled
        seen_k[kk] = 1;
        if (ty == NODAL_T_CCCS) {
            if (t.drv[i] < 0) return;
            plan.row_of[kk] = K + kk;  // its own row defines the current (level 0)
            cccs_rows.push_back(i);
            continue;
        }
        Raw r{kk, t.a[i], t.b[i], -1, -1, 0.0, 0.0};
        if (r.a == r.b) return;
        if (ty == NODAL_T_E) r.cst = value[i];
        else {
            r.c = t.c[i];
            r.d = t.d[i];
            if (r.c == r.d) r.c = r.d = -1;
            else if (ty == NODAL_T_VCVS) r.gain = value[i];
            else {
                if (t.drv[i] < 0) return;
                r.gain = -value[i] / value[t.drv[i]];
            }
        }
        raws.push_back(r);
    }
    for (int kk = 0; kk < B; ++kk)
        if (!seen_k[kk]) return;

    // ---- orientation: the branches form a graph on their lead nodes; every connected
    // component must be a tree, rooted at ground if it touches ground.  Each non-root node
    // is a pivot, defined through the branch to its parent (chains of sources included).
    std::vector<int32_t> ids;  // compact node ids; ground (-1) first
    ids.push_back(-1);
    for (const Raw &r : raws) {
        if (r.a >= 0) ids.push_back(r.a);
        if (r.b >= 0) ids.push_back(r.b);
    }
    lap("branches listed");
    std::sort(ids.begin() + 1, ids.end());
    ids.erase(std::unique(ids.begin() + 1, ids.end()), ids.end());
    const int nn = (int)ids.size();
    // node -> id through a direct map (K entries kept with the host table, -1 between uses: the binary searches
    // it replaces were most of the plan's 1.3 ms on config 5's 3e4 lead nodes)
    std::vector<int32_t> &slot = t.node_slot;
    if ((int)slot.size() != K) slot.assign((size_t)K, -1);
    struct SlotGuard {
        std::vector<int32_t> &slot;
        const std::vector<int32_t> &nodes;
        ~SlotGuard() {
            for (size_t i = 1; i < nodes.size(); ++i) slot[(size_t)nodes[i]] = -1;
        }
    } slot_guard{slot, ids};
    for (int i = 1; i < nn; ++i) slot[(size_t)ids[i]] = i;
    lap("lead nodes sorted");
    auto id_of = [&](int node) { return node < 0 ? 0 : (int)slot[(size_t)node]; };
    // adjacency in CSR form (two flat arrays: a vector per node costs an allocation per node, 1 ms at
    // 3e4 lead nodes): neighbours of id u are adj_nb / adj_raw [adj_start[u], adj_start[u + 1])
    std::vector<int> adj_start(nn + 1, 0), adj_nb(2 * raws.size()), adj_raw(2 * raws.size()), lead_a(raws.size()),
        lead_b(raws.size());
    for (int m = 0; m < (int)raws.size(); ++m) {
        lead_a[m] = id_of(raws[m].a);
        lead_b[m] = id_of(raws[m].b);
        ++adj_start[lead_a[m] + 1];
        ++adj_start[lead_b[m] + 1];
    }
    for (int u = 0; u < nn; ++u) adj_start[u + 1] += adj_start[u];
    {
        std::vector<int> fill(adj_start.begin(), adj_start.end() - 1);
        for (int m = 0; m < (int)raws.size(); ++m) {  // (in branch order, as the per-node lists were)
            adj_nb[fill[lead_a[m]]] = lead_b[m];
            adj_raw[fill[lead_a[m]]++] = m;
            adj_nb[fill[lead_b[m]]] = lead_a[m];
            adj_raw[fill[lead_b[m]]++] = m;
        }
    }
    lap("adjacency");
    // Ground is a fixed potential, not a junction: the sub-trees hanging off it are independent.  Every
    // connected piece of the branch graph WITHOUT ground is a "tree" of its own (comp), whose branches are
    // eliminated together or stay together (bad: set below when one of its pivots cannot be expressed; a
    // loop -- among its own nodes or through a second connection to ground -- ends the presolve).
    std::vector<int> parent(nn, -1), via(nn, -1), comp(nn, -1), order;
    std::vector<char> visited(nn, 0), bad;
    order.reserve(nn);
    auto grow = [&](int start) {  // BFS of one tree from `start` (already labelled), never through ground
        const int cid = comp[start];
        size_t head = order.size();
        order.push_back(start);
        while (head < order.size()) {
            const int u = order[head++];
            for (int e = adj_start[u]; e < adj_start[u + 1]; ++e) {
                const int v = adj_nb[e], m = adj_raw[e];
                if (m == via[u]) continue;
                if (v == 0 || visited[v]) {  // a second way to ground, or a loop
                    bad[cid] = 1;
                    continue;
                }
                visited[v] = 1;
                comp[v] = cid;
                parent[v] = u;
                via[v] = m;
                order.push_back(v);
            }
        }
    };
    visited[0] = 1;
    for (int e = adj_start[0]; e < adj_start[1]; ++e) {  // the trees rooted at ground
        const int v = adj_nb[e], m = adj_raw[e];
        if (visited[v]) {  // (v == 0: a branch from ground to ground was refused above)
            if (v != 0) bad[comp[v]] = 1;  // its tree reaches ground twice
            continue;
        }
        visited[v] = 1;
        comp[v] = (int)bad.size();
        bad.push_back(0);
        parent[v] = 0;
        via[v] = m;
        grow(v);
    }
    for (int root = 1; root < nn; ++root) {  // floating trees: the root survives
        if (visited[root] || adj_start[root] == adj_start[root + 1]) continue;
        visited[root] = 1;
        comp[root] = (int)bad.size();
        bad.push_back(0);
        grow(root);
    }
    lap("trees oriented");
    // A loop of voltage-defined branches makes the system singular whatever the values (its circulating
    // current appears in no other equation): no presolve; the caller's verdicts / full-system solve decide.
    for (const char b : bad)
        if (b) return;

    // ---- resolution: every pivot in terms of SURVIVING nodes, one control term at most.  A pivot that
    // cannot be expressed spoils its tree (all of its branches stay); the others are resolved again, until
    // nothing changes (a pass per level of such a cascade: normally one).
    struct Res { int base; double cst; int c, d; double g; };  // base / c / d are node numbers
    std::vector<Res> res(nn);
    std::vector<char> state(nn, 0);  // 0 = open, 1 = in progress, 2 = done
    auto is_pivot_id = [&](int id) { return id > 0 && via[id] >= 0 && !bad[comp[id]]; };
    // id of a node that is a lead of some branch, -1 for ground and for plain surviving nodes
    auto lead_id = [&](int node) { return node < 0 ? -1 : (int)slot[(size_t)node]; };
    auto comp_of_raw = [&](int m) { return comp[lead_a[m] ? lead_a[m] : lead_b[m]]; };
    std::vector<int> stack;
    bool changed = true;
    int passes = 0;
    while (changed) {
        changed = false;
        ++passes;
        std::fill(state.begin(), state.end(), 0);
        auto spoil = [&](int id) {
            if (!bad[comp[id]]) {
                bad[comp[id]] = 1;
                changed = true;
            }
        };
        auto resolve = [&](int start) {
            stack.clear();
            stack.push_back(start);
            while (!stack.empty()) {
                const int v = stack.back();
                if (state[v] == 2) { stack.pop_back(); continue; }
                if (!is_pivot_id(v)) {
                    res[v] = Res{ids[v], 0.0, -1, -1, 0.0};
                    state[v] = 2;
                    stack.pop_back();
                    continue;
                }
                const Raw &r = raws[via[v]];
                int deps[3] = {parent[v], -1, -1};
                if (r.gain != 0.0) {
                    deps[1] = lead_id(r.c);
                    deps[2] = lead_id(r.d);
                }
                bool ready = true, broken = false;
                state[v] = 1;
                for (int dpd : deps) {
                    if (dpd < 0 || state[dpd] == 2) continue;
                    if (state[dpd] == 1) { broken = true; break; }  // circular definition
                    stack.push_back(dpd);
                    ready = false;
                }
                Res out{ids[v], 0.0, -1, -1, 0.0};
                if (!broken && !ready) continue;  // stays "in progress"; revisited when the dependencies are done
                if (!broken) {
                    const double sign = ids[v] == r.a ? 1.0 : -1.0;
                    out = res[parent[v]];
                    out.cst += sign * r.cst;
                    if (r.gain != 0.0) {
                        auto ctl = [&](int node) -> Res {
                            if (node < 0) return Res{-1, 0.0, -1, -1, 0.0};
                            const int id = lead_id(node);
                            return id >= 0 ? res[id] : Res{node, 0.0, -1, -1, 0.0};
                        };
                        const Res rc = ctl(r.c), rd = ctl(r.d);
                        if (rc.g != 0.0 || rd.g != 0.0) broken = true;  // nested control terms
                        const double g = sign * r.gain;
                        out.cst += g * (rc.cst - rd.cst);
                        if (!broken && rc.base != rd.base) {
                            if (out.g != 0.0) broken = true;  // two control terms
                            out.c = rc.base;
                            out.d = rd.base;
                            out.g = g;
                        }
                    }
                    // a pivot that (after resolution) still controls itself cannot be substituted
                    if (!broken && out.g != 0.0 && (out.c == ids[v] || out.d == ids[v])) broken = true;
                }
                if (broken) {  // its tree stays: the node survives (the pass is repeated without the tree)
                    spoil(v);
                    out = Res{ids[v], 0.0, -1, -1, 0.0};
                    while (!stack.empty() && stack.back() != v) stack.pop_back();
                }
                res[v] = out;
                state[v] = 2;
                stack.pop_back();
            }
        };
        for (int v : order)
            if (state[v] != 2) resolve(v);
        if (changed) continue;
        // a CCCS (a transconductance after the rewrite) or a kept VCVS controlled by a pivot whose own
        // expression carries a control term -- or, for the kept VCVS, by any pivot -- cannot be written as
        // one row: that pivot's tree stays as well
        for (const int64_t i : cccs_rows)
            for (const int node : {t.c[i], t.d[i]}) {
                const int id = lead_id(node);
                if (id > 0 && is_pivot_id(id) && res[id].g != 0.0) spoil(id);
            }
        for (int m = 0; m < (int)raws.size(); ++m) {
            if (!bad[comp_of_raw(m)] || raws[m].gain == 0.0) continue;
            for (const int node : {raws[m].c, raws[m].d}) {
                const int id = lead_id(node);
                if (id > 0 && is_pivot_id(id)) spoil(id);
            }
        }
    }
    lap("pivots resolved");

    // ---- the plan: expressions, rows and levels for the current recovery ----
    std::vector<int> height(nn, 0);
    for (size_t i = order.size(); i-- > 0;) {
        const int v = order[i];
        if (is_pivot_id(v) && parent[v] >= 0) height[parent[v]] = std::max(height[parent[v]], height[v] + 1);
    }
    std::vector<int32_t> taken;
    for (int v : order) {
        if (!is_pivot_id(v)) continue;
        const Res &r = res[v];
        plan.exprs.push_back(Expr{ids[v], r.base, r.cst, r.c, r.d, r.g});
        taken.push_back(ids[v]);
        const int kk = raws[via[v]].kk;
        plan.row_of[kk] = ids[v];
        plan.level_of[kk] = 1 + height[v];
        plan.max_level = std::max(plan.max_level, 1 + height[v]);
    }
    if (taken.empty()) return;  // nothing can be eliminated: the full-system solve
    static const bool keep_allowed = !(getenv("NODAL_PRESOLVE_KEEP") && atoi(getenv("NODAL_PRESOLVE_KEEP")) == 0);
    if (!keep_allowed)  // (round 2's behaviour, for comparisons: one stubborn source ends the presolve)
        for (const char b : bad)
            if (b) return;
    for (int m = 0; m < (int)raws.size(); ++m) {  // the branches that stay (raws are in branch order)
        if (!bad[comp_of_raw(m)]) continue;
        const Raw &r = raws[m];
        plan.kept.push_back(KeptBranch{r.kk, r.gain != 0.0 ? NODAL_T_VCVS : NODAL_T_E, r.gain != 0.0 ? r.gain : r.cst,
                                       r.a, r.b, r.gain != 0.0 ? r.c : -1, r.gain != 0.0 ? r.d : -1});
        plan.row_of[r.kk] = K + r.kk;  // (its current is an unknown of the reduced system: copied back)
        plan.level_of[r.kk] = -1;
    }
    for (int kk = 0; kk < B; ++kk)
        if (plan.row_of[kk] < 0) return;
    std::sort(taken.begin(), taken.end());
    // bases and controls must be surviving nodes by construction
    auto is_taken = [&](int node) {
        const int id = lead_id(node);
        return id >= 0 && is_pivot_id(id);
    };
    for (const Expr &e : plan.exprs)
        if (is_taken(e.q) || is_taken(e.c) || is_taken(e.d)) return;
    plan.pivots = taken;
    plan.Kr = K - (int32_t)taken.size();
    plan.ok = true;
    if (trace_plan && !plan.kept.empty())
        fprintf(stderr, "[presolve]   plan: %d of %d voltage-defined branches stay in the reduced system (%d passes)\n",
                (int)plan.kept.size(), (int)raws.size(), passes);
    lap("done");
}

// Rewrite of the components that touch an eliminated node (host; they are few).
// (arrays in the handle's page-locked arena: they go up to the device as they are; at most four rows per hit)
struct Extras {
    ExtraRow *rows = nullptr;
    int64_t n = 0, cap = 0;
};
static bool rewrite_hits(const nodal_ctx *h, const double *value, const PresolvePlan &plan,
                         const int32_t *hits_begin, int64_t nhits, Extras &x) {
    const HostTable &t = h->host;
    // pivot node -> its expression through the host table's direct map (see presolve_plan)
    std::vector<int32_t> &slot = t.node_slot;
    if ((int)slot.size() != h->K) slot.assign((size_t)h->K, -1);
    struct SlotGuard {
        std::vector<int32_t> &slot;
        const std::vector<Expr> &exprs;
        ~SlotGuard() {
            for (const Expr &e : exprs) slot[(size_t)e.p] = -1;
        }
    } slot_guard{slot, plan.exprs};
    for (size_t i = 0; i < plan.exprs.size(); ++i) slot[(size_t)plan.exprs[i].p] = (int32_t)i;
    auto expr_of = [&](int node) -> const Expr * {
        if (node < 0) return nullptr;
        const int32_t at = slot[(size_t)node];
        return at >= 0 ? &plan.exprs[(size_t)at] : nullptr;
    };
    bool overflow = false;
    auto emit = [&](int ty, double v, int a, int b, int c, int d, int k = -1) {
        if (x.n >= x.cap) { overflow = true; return; }
        // (node numbers of the ORIGINAL system: place_extras renumbers them on the device, where the map is)
        x.rows[x.n++] = ExtraRow{v, a, b, c, d, k, ty};
    };
    struct Side { int base; double cst; int c, d; double g; };
    auto side = [&](int node) {
        if (const Expr *e = expr_of(node)) return Side{e->q, e->cst, e->c, e->d, e->g};
        return Side{node, 0.0, -1, -1, 0.0};
    };
    for (int64_t hh = 0; hh < nhits; ++hh) {
        const int32_t i = hits_begin[hh];
        const int ty = t.type[i];
        const double v = value[i];
        if (ty == NODAL_T_R) {
            const Side sx = side(t.a[i]), sy = side(t.b[i]);
            if (sx.base == sy.base) continue;  // both leads collapse onto one node
            const double g = 1.0 / v;
            emit(NODAL_T_R, v, sx.base, sy.base, -1, -1);
            // current g (e_x - e_y) flows from x's super-node to y's: constant part ...
            const double cstd = sx.cst - sy.cst;
            if (cstd != 0.0) emit(NODAL_T_A, g * cstd, sy.base, sx.base, -1, -1);
            // ... and the controlled parts
            if (sx.g != 0.0) emit(NODAL_T_GM, g * sx.g, sx.base, sy.base, sx.c, sx.d);
            if (sy.g != 0.0) emit(NODAL_T_GM, -g * sy.g, sx.base, sy.base, sy.c, sy.d);
        } else if (ty == NODAL_T_A) {
            const Side sa = side(t.a[i]), sb = side(t.b[i]);
            if (sa.base != sb.base) emit(NODAL_T_A, v, sa.base, sb.base, -1, -1);
        } else if (ty == NODAL_T_CCCS || ty == NODAL_T_GM) {
            // gm (e_c - e_d) flowing a -> b, with eliminated leads AND controls substituted
            const double gm = ty == NODAL_T_CCCS ? v / value[t.drv[i]] : v;
            const Side sa = side(t.a[i]), sb = side(t.b[i]), rc = side(t.c[i]), rd = side(t.d[i]);
            if (rc.g != 0.0 || rd.g != 0.0) return false;  // a control with its own control term
            if (sa.base == sb.base) continue;
            const double cst = gm * (rc.cst - rd.cst);
            if (cst != 0.0) emit(NODAL_T_A, cst, sb.base, sa.base, -1, -1);
            if (rc.base != rd.base) emit(NODAL_T_GM, gm, sa.base, sb.base, rc.base, rd.base);
        }
    }
    // the branches that stay: their leads and controls are surviving nodes (presolve_plan saw to that)
    for (size_t kp = 0; kp < plan.kept.size(); ++kp) {
        const KeptBranch &kb = plan.kept[kp];
        if (expr_of(kb.a) || expr_of(kb.b) || expr_of(kb.c) || expr_of(kb.d)) return false;
        emit(kb.type, kb.value, kb.a, kb.b, kb.c, kb.d, (int)kp);
    }
    return !overflow;
}

// Build the reduced component table directly in the child context's device arrays.
static int presolve_build_reduced(nodal_ctx *h, nodal_ctx *r, const double *value_host,
                                  const double *value_dev, const PresolvePlan &plan, bool *ok) {
    *ok = false;
    hipStream_t st = h->stream;
    const int64_t nc = h->ncomp;
    const int K = h->K;
    const int np = (int)plan.pivots.size();
    // scratch layout in work3: survive/scan [K+1] | keep [nc+1] | hit [nc+1] | pivots | flags | scan tmp
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_surv = 0, o_scan = o_surv + al((size_t)(K + 1) * 4);
    const size_t o_keep = o_scan + al((size_t)(K + 1) * 4), o_kpos = o_keep + al((size_t)(nc + 1) * 4);
    const size_t o_hit = o_kpos + al((size_t)(nc + 1) * 4), o_hpos = o_hit + al((size_t)(nc + 1) * 4);
    const size_t o_piv = o_hpos + al((size_t)(nc + 1) * 4), o_flag = o_piv + al((size_t)np * 4 + 4);
    const size_t o_tmp = o_flag + 256;
    NODAL_HIP_TRY(h, h->work3.reserve(o_tmp + scan_tmp_bytes(std::max<int64_t>(nc, K) + 1)));
    char *w = h->work3.as<char>();
    uint32_t *survive = reinterpret_cast<uint32_t *>(w + o_surv), *sscan = reinterpret_cast<uint32_t *>(w + o_scan);
    uint32_t *keep = reinterpret_cast<uint32_t *>(w + o_keep), *kpos = reinterpret_cast<uint32_t *>(w + o_kpos);
    uint32_t *hit = reinterpret_cast<uint32_t *>(w + o_hit), *hpos = reinterpret_cast<uint32_t *>(w + o_hpos);
    int32_t *d_piv = reinterpret_cast<int32_t *>(w + o_piv);
    int32_t *flags = reinterpret_cast<int32_t *>(w + o_flag);  // [0] invalid
    void *tmp = w + o_tmp;
    NODAL_HIP_TRY(h, h->ps_newidx.reserve((size_t)K * 4 + 64));
    int32_t *newidx = h->ps_newidx.as<int32_t>();

    // (host-built pieces go through the handle's page-locked arena: a copy from pageable memory is staged by
    // the runtime, 30-60 us apiece -- there were a dozen of them per solve)
    if (np) {
        void *stage = nodal_pinned_arena(h, (size_t)np * 4);
        if (stage) memcpy(stage, plan.pivots.data(), (size_t)np * 4);
        NODAL_HIP_TRY(h, hipMemcpyAsync(d_piv, stage ? stage : (const void *)plan.pivots.data(), (size_t)np * 4,
                                        hipMemcpyHostToDevice, st));
    }
    NODAL_HIP_TRY(h, hipMemsetAsync(flags, 0, 16, st));
    fill_u32<<<grid_for(K + 1), TB, 0, st>>>(survive, 1u, K);
    NODAL_HIP_TRY(h, hipMemsetAsync(survive + K, 0, 4, st));
    mark_pivots<<<grid_for(np), TB, 0, st>>>(np, d_piv, survive);
    NODAL_TRY(scan_exclusive_u32(h, survive, sscan, (int64_t)K + 1, nullptr, tmp));
    make_newidx<<<grid_for(K), TB, 0, st>>>(K, survive, sscan, newidx);
    DevTable t{h->type.as<uint8_t>(), value_dev, h->a.as<int32_t>(), h->b.as<int32_t>(),
               h->c.as<int32_t>(), h->d.as<int32_t>(), h->drv.as<int32_t>()};
    classify<<<grid_for(nc), TB, 0, st>>>(t, nc, newidx, keep, hit, flags);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipMemsetAsync(keep + nc, 0, 4, st));
    NODAL_HIP_TRY(h, hipMemsetAsync(hit + nc, 0, 4, st));
    // (the scans leave their totals next to the flag: one read-back of three words)
    NODAL_TRY(scan_exclusive_u32(h, keep, kpos, nc + 1, reinterpret_cast<uint32_t *>(flags) + 1, tmp));
    NODAL_TRY(scan_exclusive_u32(h, hit, hpos, nc + 1, reinterpret_cast<uint32_t *>(flags) + 2, tmp));
    int32_t back[3] = {0, 0, 0};  // invalid, components kept, components hit
    NODAL_TRY(nodal_read_words(h, back, flags, 12));
    if (back[0]) return NODAL_OK;  // a dependent source is controlled by a pivot node
    const int64_t nkeep = (uint32_t)back[1], nhit = (uint32_t)back[2];
    static const bool trace_build = getenv("NODAL_TRACE") != nullptr;
    const auto tb0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace_build)
            fprintf(stderr, "[presolve]   rewrite: %s at %.3f ms (%lld kept, %lld hit)\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count(),
                    (long long)nkeep, (long long)nhit);
    };
    // worst case 4 rewritten components per hit, and the branches that stay
    const int64_t cap = nkeep + 4 * nhit + (int64_t)plan.kept.size() + 16;
    NODAL_HIP_TRY(h, r->type.reserve((size_t)cap + 16));
    NODAL_HIP_TRY(h, r->value.reserve((size_t)cap * 8 + 16));
    DevBuf *icols[] = {&r->a, &r->b, &r->c, &r->d, &r->drv, &r->k};
    for (DevBuf *b : icols) NODAL_HIP_TRY(h, b->reserve((size_t)cap * 4 + 16));
    NODAL_HIP_TRY(h, h->ps_hits.reserve((size_t)nhit * 4 + 64));
    DevTableOut o{r->type.as<uint8_t>(), r->value.as<double>(), r->a.as<int32_t>(), r->b.as<int32_t>(),
                  r->c.as<int32_t>(), r->d.as<int32_t>(), r->drv.as<int32_t>(), r->k.as<int32_t>()};
    compact<<<grid_for(nc), TB, 0, st>>>(t, nc, newidx, keep, kpos, hit, hpos, o, h->ps_hits.as<int32_t>());
    NODAL_HIP_TRY(h, hipGetLastError());
    // arena: hits [nhit] | rows [cx]   (cx = 4 rows per hit at most + the branches that stay)
    const int64_t cx = 4 * nhit + (int64_t)plan.kept.size() + 16;
    const size_t a_hits = (((size_t)nhit * 4) + 63) & ~(size_t)63;
    char *arena = static_cast<char *>(nodal_pinned_arena(h, a_hits + (size_t)cx * sizeof(ExtraRow) + 64));
    if (!arena) return nodal_fail(h, NODAL_E_HIP, "presolve: no page-locked staging memory");
    int32_t *hits = reinterpret_cast<int32_t *>(arena);
    Extras x;
    x.rows = reinterpret_cast<ExtraRow *>(arena + a_hits);
    x.cap = cx;
    NODAL_HIP_TRY(h, h->ps_stage.reserve((size_t)cx * sizeof(ExtraRow) + 64));  // (before the wait: a growing buffer is filled)
    if (nhit) NODAL_HIP_TRY(h, hipMemcpyAsync(hits, h->ps_hits.p, (size_t)nhit * 4, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    lap("components compacted, hits on the host");
    if (!rewrite_hits(h, value_host, plan, hits, nhit, x)) return NODAL_OK;  // not expressible: fall back
    const int64_t nx = x.n;
    lap("hits rewritten");
    if (nx) {
        NODAL_HIP_TRY(h, hipMemcpyAsync(h->ps_stage.p, x.rows, (size_t)nx * sizeof(ExtraRow), hipMemcpyHostToDevice, st));
        place_extras<<<grid_for(nx), TB, 0, st>>>(nx, h->ps_stage.as<ExtraRow>(), newidx, o.type + nkeep, o.value + nkeep,
                                                 o.a + nkeep, o.b + nkeep, o.c + nkeep, o.d + nkeep, o.drv + nkeep,
                                                 o.k + nkeep);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    lap("extras on their way");
    // Fingerprint of the reduced netlist's TOPOLOGY: the parent's (struct_epoch: a value sweep on an
    // assembled topology keeps it, a fresh symbolic assembly does not), the pivots, and the integer columns
    // of the rewritten rows (which rows get emitted depends on values: a zero constant emits no source).
    // Unchanged => the reduced system's stamping lists stand, and with them (same struct_epoch of the
    // reduced context) the symbolic part of its multigrid hierarchy: only values are redone.
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    mix(h->table_epoch); mix(h->struct_epoch); mix((uint64_t)np); mix((uint64_t)nkeep); mix((uint64_t)nx); mix((uint64_t)plan.Kr);
    mix((uint64_t)plan.kept.size());
    for (int32_t pv : plan.pivots) mix((uint64_t)(uint32_t)pv);
    for (int64_t i = 0; i < nx; ++i) {  // (the host reads its own rows while the copy and the placing kernel run)
        const ExtraRow &e = x.rows[(size_t)i];
        mix((uint64_t)(uint8_t)e.type);
        mix((uint64_t)(uint32_t)e.a << 32 | (uint32_t)e.b);
        mix((uint64_t)(uint32_t)e.c << 32 | (uint32_t)e.d);
        mix((uint64_t)(uint32_t)e.k);
    }
    if (nx) NODAL_WAIT_STREAM(h, st);  // (the arena is free again)
    const int32_t nkept = (int32_t)plan.kept.size();
    const bool same_topology = r->have_symbolic && r->reduced_key == key && r->ncomp == nkeep + nx && r->K == plan.Kr &&
                               r->B == nkept;
    r->ncomp = nkeep + nx;
    r->K = plan.Kr;
    r->B = nkept;
    r->n = plan.Kr + nkept;
    r->batch = 0;
    r->have_table = true;
    r->have_numeric = r->have_x = false;
    if (!same_topology) {
        ++r->table_epoch;
        r->have_symbolic = false;
        r->reduced_key = key;
    }
    lap("fingerprint");
    *ok = true;
    return NODAL_OK;
}

// y (reduced potentials, device) -> x (full unknown vector of h, device)
static int presolve_recover(nodal_ctx *h, const PresolvePlan &plan, const double *y) {
    hipStream_t st = h->stream;
    const int K = h->K, B = h->B;
    const int ne = (int)plan.exprs.size();
    // the small recovery tables: one image in the page-locked arena, laid out like the device block, ONE copy
    const size_t a4k = ((size_t)K * 4 + 255) & ~(size_t)255, a4e = ((size_t)ne * 4 + 255) & ~(size_t)255;
    const size_t a8e = ((size_t)ne * 8 + 255) & ~(size_t)255, a4b = ((size_t)B * 4 + 255) & ~(size_t)255;
    const int nkept = (int)plan.kept.size();
    const size_t a4kp = ((size_t)nkept * 4 + 255) & ~(size_t)255;
    const size_t image = 4 * a4e + 2 * a8e + 2 * a4b + a4kp;
    NODAL_HIP_TRY(h, h->ps_buf.reserve(a4k + image + 256));
    char *w = h->ps_buf.as<char>();
    const int32_t *d_new = h->ps_newidx.as<int32_t>();  // built by presolve_build_reduced
    int32_t *d_p = reinterpret_cast<int32_t *>(w + a4k);
    int32_t *d_q = reinterpret_cast<int32_t *>(w + a4k + a4e);
    int32_t *d_c = reinterpret_cast<int32_t *>(w + a4k + 2 * a4e);
    int32_t *d_d = reinterpret_cast<int32_t *>(w + a4k + 3 * a4e);
    double *d_cst = reinterpret_cast<double *>(w + a4k + 4 * a4e);
    double *d_g = reinterpret_cast<double *>(w + a4k + 4 * a4e + a8e);
    int32_t *d_row = reinterpret_cast<int32_t *>(w + a4k + 4 * a4e + 2 * a8e);
    int32_t *d_level = reinterpret_cast<int32_t *>(w + a4k + 4 * a4e + 2 * a8e + a4b);
    int32_t *d_kept = reinterpret_cast<int32_t *>(w + a4k + 4 * a4e + 2 * a8e + 2 * a4b);
    char *img = static_cast<char *>(nodal_pinned_arena(h, image + 64));
    if (!img) return nodal_fail(h, NODAL_E_HIP, "presolve: no page-locked staging memory");
    {
        int32_t *p = reinterpret_cast<int32_t *>(img), *q = reinterpret_cast<int32_t *>(img + a4e);
        int32_t *c = reinterpret_cast<int32_t *>(img + 2 * a4e), *d = reinterpret_cast<int32_t *>(img + 3 * a4e);
        double *cst = reinterpret_cast<double *>(img + 4 * a4e), *g = reinterpret_cast<double *>(img + 4 * a4e + a8e);
        for (int i = 0; i < ne; ++i) {
            const Expr &e = plan.exprs[i];
            p[i] = e.p; q[i] = e.q; c[i] = e.c; d[i] = e.d; cst[i] = e.cst; g[i] = e.g;
        }
        memcpy(img + 4 * a4e + 2 * a8e, plan.row_of.data(), (size_t)B * 4);
        memcpy(img + 4 * a4e + 2 * a8e + a4b, plan.level_of.data(), (size_t)B * 4);
        int32_t *kk = reinterpret_cast<int32_t *>(img + 4 * a4e + 2 * a8e + 2 * a4b);
        for (int i = 0; i < nkept; ++i) kk[i] = plan.kept[(size_t)i].kk;
    }
    NODAL_HIP_TRY(h, hipMemcpyAsync(w + a4k, img, image, hipMemcpyHostToDevice, st));
    double *x = h->x.as<double>();
    scatter_nodes<<<grid_for(K), TB, 0, st>>>(K, d_new, y, x);
    if (ne) eval_pivots<<<grid_for(ne), TB, 0, st>>>(ne, d_p, d_q, d_cst, d_c, d_d, d_g, x);
    if (nkept) copy_kept<<<grid_for(nkept), TB, 0, st>>>(nkept, d_kept, K, plan.Kr, y, x);
    for (int pass = 0; pass <= plan.max_level; ++pass)
        recover_currents<<<grid_for(B), TB, 0, st>>>(pass, K, B, d_level, d_row, h->indptr.as<int32_t>(),
                                                    h->indices.as<int32_t>(), h->data.as<double>(),
                                                    h->rhs.as<double>(), x);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_WAIT_STREAM(h, st);  // (the arena is free again)
    return NODAL_OK;
}

// Try the presolve route for the system of `h` (B > 0).  Returns NODAL_OK with
// *done = true when x was produced and verified; *done = false means "not applicable"
// (the caller falls back to the full-system Krylov solve).
// The plan is host work on the host's copy of the table (0.7 ms at config 5's 2e4 branches) and needs nothing the
// device computes: stamp_numeric calls presolve_plan_ahead between enqueueing its kernels and waiting for their status
// words, so that the host plans while the device stamps the original system.  The key: the table, the member, and the
// values the plan reads (the branch rows' and their drivers').
namespace {
struct PlanCache {
    PresolvePlan plan;
    uint64_t key = 0;
    bool valid = false;
};
const double *plan_values(const nodal_ctx *h) {
    if (h->batch > 0) return h->host.values_batch.empty() ? nullptr : h->host.values_batch.data() + (size_t)h->member * h->ncomp;
    return h->host.value.data();
}
uint64_t plan_key(const nodal_ctx *h, const double *value) {
    uint64_t key = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
    mix(h->table_epoch); mix((uint64_t)h->member); mix((uint64_t)h->ncomp); mix((uint64_t)h->B);
    for (const int64_t i : h->host.branch_rows) {
        uint64_t bits;
        memcpy(&bits, &value[i], 8);
        mix(bits);
        const int32_t dr = h->host.drv.empty() ? -1 : h->host.drv[(size_t)i];
        if (dr >= 0) { memcpy(&bits, &value[dr], 8); mix(bits); }
    }
    return key;
}
}  // namespace

void presolve_plan_ahead(nodal_ctx *h) {
    if (h->B == 0 || h->host.type.empty() || h->host.branch_rows.empty()) return;
    const double *value = plan_values(h);
    if (!value) return;
    PlanCache *c = static_cast<PlanCache *>(h->ps_plan);
    if (!c) h->ps_plan = c = new PlanCache();
    const uint64_t key = plan_key(h, value);
    if (c->valid && c->key == key) return;
    c->valid = false;
    presolve_plan(h, value, c->plan);
    c->key = key;
    c->valid = true;
}
void presolve_free_plan(nodal_ctx *h) {
    delete static_cast<PlanCache *>(h->ps_plan);
    h->ps_plan = nullptr;
}

int presolve_solve(nodal_ctx *h, bool *done, int32_t *info, int32_t *iters, double *resid,
                   bool dense_child) {
    *done = false;
    if (h->B == 0 || h->host.type.empty()) return NODAL_OK;
    const double *value = h->host.value.data();
    if (h->batch > 0) {
        if (h->host.values_batch.empty()) return NODAL_OK;
        value = h->host.values_batch.data() + (size_t)h->member * h->ncomp;
    }
    const bool trace = getenv("NODAL_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    PresolvePlan local_plan;
    PlanCache *pc = static_cast<PlanCache *>(h->ps_plan);
    const bool ahead = pc && pc->valid && !h->host.branch_rows.empty() && pc->key == plan_key(h, value);
    if (!ahead) presolve_plan(h, value, local_plan);
    const PresolvePlan &plan = ahead ? pc->plan : local_plan;  // (planned while the device stamped: presolve_plan_ahead)
    if (!plan.ok) return NODAL_OK;
    const auto t1 = now();

    if (!h->reduced) {
        h->reduced = new nodal_ctx();
        h->reduced->device = h->device;
        h->reduced->stream = h->stream;  // shared: one ordered timeline
        h->reduced->stream2 = h->stream2;
        for (int i = 0; i < 4; ++i) h->reduced->ev[i] = h->ev[i];
        h->reduced->ev_la[0] = h->ev_la[0];
        h->reduced->ev_la[1] = h->ev_la[1];
        h->reduced->stream3 = h->stream3;
        for (int i = 0; i < 6; ++i) h->reduced->ev_bi[i] = h->ev_bi[i];
        h->reduced->dense_blockinv = h->dense_blockinv;
        h->reduced->gj_scalar = h->gj_scalar;
        h->reduced->owns_streams = false;
        h->reduced->stream_owner = h->stream_owner ? h->stream_owner : h;
        h->reduced->keep_host_table = false;
    }
    nodal_ctx *r = h->reduced;
    const double *value_dev =
        h->batch > 0 ? h->values_batch.as<double>() + (int64_t)h->member * h->ncomp : h->value.as<double>();
    bool built = false;
    NODAL_TRY(presolve_build_reduced(h, r, value, value_dev, plan, &built));
    if (!built) return NODAL_OK;
    const auto t2 = now();
    int s = r->have_symbolic ? NODAL_OK : stamp_symbolic(r);  // (kept: same topology as the last solve's)
    if (s == NODAL_OK) s = stamp_numeric(r, 0, nullptr);
    if (s != NODAL_OK) { h->err = r->err; return s; }
    const auto t3 = now();
    int32_t rinfo = 0;
    if (dense_child) {
        // The dense path wants a direct solve.  A passive reduced network is SPD: block
        // elimination.  With transconductance stamps left (dependent sources) the reduced
        // matrix is still a conductance matrix plus a few off-diagonal terms, which the same
        // pivot-free elimination almost always handles -- tried optimistically: a zero pivot
        // or a residual above the acceptance bar hands the ORIGINAL system to the pivoted LU.
        if (r->n == 0) return NODAL_OK;
        *iters = 0;
        *resid = 0.0;
        r->optimistic_nopivot = !r->passive_network && r->B == 0;  // (branches that stayed: the pivoted LU)
        s = dense_prepare(r);
        if (s == NODAL_OK) s = dense_factor_solve(r, &rinfo);
        if (s == NODAL_OK && rinfo > 0 && r->optimistic_nopivot) {
            if (trace) fprintf(stderr, "[presolve] pivot-free elimination of the reduced system hit a zero pivot\n");
            return NODAL_OK;
        }
    } else {
        s = sparse_solve(r, NODAL_SPARSE_AUTO, &rinfo, iters, resid);
        // with branches left in it the reduced system is a general one: should its iteration not converge,
        // the original system takes the caller's full route (which has the dense rescue), not an error
        if (s == NODAL_E_UNSUPPORTED && r->B > 0) {
            if (trace) fprintf(stderr, "[presolve] the reduced system (%d branches kept) did not converge: full-system route\n", (int)r->B);
            return NODAL_OK;
        }
    }
    if (s != NODAL_OK) { h->err = r->err; return s; }
    const auto t4 = now();
    if (trace)
        fprintf(stderr, "[presolve] plan %.2f ms, upload %.2f ms, assemble %.2f ms, solve %.2f ms "
                        "(%d iterations, n' = %d, passive %d)\n",
                ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), *iters, plan.Kr, (int)r->passive_network);
    h->amg_levels = r->amg_levels;
    h->kern_ms = r->kern_ms;
    h->kern_launches = r->kern_launches;
    h->kern_alg = r->kern_alg;
    if (rinfo > 0) {  // reduced system singular => so is the original
        *info = rinfo;
        *done = true;
        return NODAL_OK;
    }
    NODAL_TRY(presolve_recover(h, plan, r->x.as<double>()));
    // accept only if the ORIGINAL system is satisfied
    h->have_x = true;
    double scaled = 0.0;
    NODAL_TRY(sparse_residual(h, &scaled));
    h->have_x = false;
    if (trace)
        fprintf(stderr, "[presolve] %s: scaled residual of the original system %.3e, %d pivots, %d recovery passes\n",
                scaled <= 1e-11 ? "accepted" : "rejected", scaled, (int)plan.pivots.size(), plan.max_level + 1);
    if (!(scaled <= 1e-11)) return NODAL_OK;  // fall back to the full-system solve
    *info = 0;
    *done = true;
    return NODAL_OK;
}
