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
// Supported pattern (everything else falls back to the full-system GMRES): every
// voltage-defined branch gets a distinct pivot among its non-ground leads, and no
// pivot node is the other lead or a control node of any dependent source.
#include <algorithm>
#include <chrono>
#include <cstdlib>

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

// branch currents from the ORIGINAL system.  pass 0: branches whose own row defines
// them (CCCS: row K+k has a unit diagonal); pass 1: voltage-defined branches from the
// KCL row of their pivot node (all other branch currents in that row are known then).
__global__ __launch_bounds__(TB) void recover_currents(int pass, int K, int B,
                                                       const int32_t *__restrict__ row_of,
                                                       const int32_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ indices,
                                                       const double *__restrict__ data,
                                                       const double *__restrict__ rhs,
                                                       double *__restrict__ x) {
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < B; k += (int64_t)gridDim.x * TB) {
        const int row = row_of[k];  // K + k for pass 0 branches, pivot node for pass 1 ones
        const bool own = row >= K;
        if (own != (pass == 0)) continue;
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

struct PresolvePlan {
    bool ok = false;
    std::vector<Expr> exprs;
    std::vector<int32_t> newidx;   // K
    std::vector<int32_t> row_of;   // B: row that determines each branch current
    // reduced component table
    std::vector<uint8_t> type;
    std::vector<double> value;
    std::vector<int32_t> a, b, c, d, drv, k;
    int32_t Kr = 0;
};

// Host analysis + rewrite.  `value` = the member's component values (host).
static void presolve_plan(const nodal_ctx *h, const double *value, PresolvePlan &plan) {
    const HostTable &t = h->host;
    const int64_t nc = h->ncomp;
    const int K = h->K, B = h->B;
    plan.ok = false;
    std::vector<int32_t> expr_of(K, -1);
    std::vector<char> seen_k(B, 0);
    plan.exprs.clear();
    plan.row_of.assign(B, -1);
    // 1. pivots
    for (int64_t i = 0; i < nc; ++i) {
        const int ty = t.type[i];
        if (ty < NODAL_T_E || ty > NODAL_T_CCCS) continue;
        const int kk = t.k[i];
        if (kk < 0 || kk >= B || seen_k[kk]) return;  // duplicated names: not handled
        seen_k[kk] = 1;
        if (ty == NODAL_T_CCCS) {
            if (t.drv[i] < 0) return;
            plan.row_of[kk] = K + kk;
            continue;
        }
        const int a = t.a[i], b = t.b[i];
        if (a == b) return;
        double cst = 0.0, gain = 0.0;
        int c = -1, d = -1;
        if (ty == NODAL_T_E) cst = value[i];
        else {
            c = t.c[i];
            d = t.d[i];
            if (c == d) { c = d = -1; }
            else if (ty == NODAL_T_VCVS) gain = value[i];
            else {
                if (t.drv[i] < 0) return;
                gain = -value[i] / value[t.drv[i]];
            }
        }
        int p = -1, q = -1;
        double sign = 1.0;
        if (a >= 0 && expr_of[a] < 0 && a != c && a != d) { p = a; q = b; }
        else if (b >= 0 && expr_of[b] < 0 && b != c && b != d) { p = b; q = a; sign = -1.0; }
        else return;
        expr_of[p] = (int32_t)plan.exprs.size();
        plan.exprs.push_back(Expr{p, q, sign * cst, c, d, (c < 0 && d < 0) ? 0.0 : sign * gain});
        plan.row_of[kk] = p;
    }
    for (int kk = 0; kk < B; ++kk)
        if (plan.row_of[kk] < 0) return;
    // 2. no pivot may serve as a base or a control node
    for (const Expr &e : plan.exprs) {
        if (e.q >= 0 && expr_of[e.q] >= 0) return;
        if (e.c >= 0 && expr_of[e.c] >= 0) return;
        if (e.d >= 0 && expr_of[e.d] >= 0) return;
    }
    for (int64_t i = 0; i < nc; ++i) {
        const int ty = t.type[i];
        if (ty == NODAL_T_CCCS || ty == NODAL_T_GM) {
            if (t.c[i] >= 0 && expr_of[t.c[i]] >= 0) return;
            if (t.d[i] >= 0 && expr_of[t.d[i]] >= 0) return;
        }
    }
    // 3. renumber the surviving nodes
    plan.newidx.assign(K, -1);
    int32_t Kr = 0;
    for (int j = 0; j < K; ++j)
        if (expr_of[j] < 0) plan.newidx[j] = Kr++;
    plan.Kr = Kr;
    auto nid = [&](int node) { return node < 0 ? -1 : plan.newidx[node]; };
    // 4. rewrite
    auto &T = plan;
    T.type.clear(); T.value.clear(); T.a.clear(); T.b.clear(); T.c.clear(); T.d.clear();
    T.drv.clear(); T.k.clear();
    const size_t guess = (size_t)nc + 8 * plan.exprs.size() + 16;
    T.type.reserve(guess); T.value.reserve(guess); T.a.reserve(guess); T.b.reserve(guess);
    T.c.reserve(guess); T.d.reserve(guess); T.drv.reserve(guess); T.k.reserve(guess);
    auto emit = [&](int ty, double v, int a, int b, int c, int d) {
        T.type.push_back((uint8_t)ty); T.value.push_back(v);
        T.a.push_back(nid(a)); T.b.push_back(nid(b)); T.c.push_back(nid(c)); T.d.push_back(nid(d));
        T.drv.push_back(-1); T.k.push_back(-1);
    };
    struct Side { int base; double cst; int c, d; double g; };
    auto side = [&](int node) {
        if (node >= 0 && expr_of[node] >= 0) {
            const Expr &e = plan.exprs[expr_of[node]];
            return Side{e.q, e.cst, e.c, e.d, e.g};
        }
        return Side{node, 0.0, -1, -1, 0.0};
    };
    for (int64_t i = 0; i < nc; ++i) {
        const int ty = t.type[i];
        const double v = value[i];
        if (ty == NODAL_T_R) {
            const int x = t.a[i], y = t.b[i];
            const bool ex = x >= 0 && expr_of[x] >= 0, ey = y >= 0 && expr_of[y] >= 0;
            if (!ex && !ey) { emit(NODAL_T_R, v, x, y, -1, -1); continue; }
            const Side sx = side(x), sy = side(y);
            const double g = 1.0 / v;
            if (sx.base != sy.base) emit(NODAL_T_R, v, sx.base, sy.base, -1, -1);
            // current g (e_x - e_y) flows from x's super-node to y's: constant part ...
            const double cstd = sx.cst - sy.cst;
            if (cstd != 0.0 && sx.base != sy.base) emit(NODAL_T_A, g * cstd, sy.base, sx.base, -1, -1);
            // ... and the controlled parts
            if (sx.g != 0.0 && sx.base != sy.base) emit(NODAL_T_GM, g * sx.g, sx.base, sy.base, sx.c, sx.d);
            if (sy.g != 0.0 && sx.base != sy.base) emit(NODAL_T_GM, -g * sy.g, sx.base, sy.base, sy.c, sy.d);
        } else if (ty == NODAL_T_A) {
            const Side sa = side(t.a[i]), sb = side(t.b[i]);
            if (sa.base != sb.base) emit(NODAL_T_A, v, sa.base, sb.base, -1, -1);
        } else if (ty == NODAL_T_CCCS || ty == NODAL_T_GM) {
            const double gm = ty == NODAL_T_CCCS ? v / value[t.drv[i]] : v;
            const Side sa = side(t.a[i]), sb = side(t.b[i]);
            if (sa.base != sb.base) emit(NODAL_T_GM, gm, sa.base, sb.base, t.c[i], t.d[i]);
        }
        // E / VCVS / CCVS: encoded in the expressions
    }
    plan.ok = true;
}

// y (reduced potentials, device) -> x (full unknown vector of h, device)
static int presolve_recover(nodal_ctx *h, const PresolvePlan &plan, const double *y) {
    hipStream_t st = h->stream;
    const int K = h->K, B = h->B;
    const int ne = (int)plan.exprs.size();
    // upload the small recovery tables
    std::vector<int32_t> p(ne), q(ne), c(ne), d(ne);
    std::vector<double> cst(ne), g(ne);
    for (int i = 0; i < ne; ++i) {
        const Expr &e = plan.exprs[i];
        p[i] = e.p; q[i] = e.q; c[i] = e.c; d[i] = e.d; cst[i] = e.cst; g[i] = e.g;
    }
    const size_t a4k = ((size_t)K * 4 + 255) & ~(size_t)255, a4e = ((size_t)ne * 4 + 255) & ~(size_t)255;
    const size_t a8e = ((size_t)ne * 8 + 255) & ~(size_t)255, a4b = ((size_t)B * 4 + 255) & ~(size_t)255;
    NODAL_HIP_TRY(h, h->ps_buf.reserve(a4k + 4 * a4e + 2 * a8e + a4b + 256));
    char *w = h->ps_buf.as<char>();
    int32_t *d_new = reinterpret_cast<int32_t *>(w);
    int32_t *d_p = reinterpret_cast<int32_t *>(w + a4k);
    int32_t *d_q = reinterpret_cast<int32_t *>(w + a4k + a4e);
    int32_t *d_c = reinterpret_cast<int32_t *>(w + a4k + 2 * a4e);
    int32_t *d_d = reinterpret_cast<int32_t *>(w + a4k + 3 * a4e);
    double *d_cst = reinterpret_cast<double *>(w + a4k + 4 * a4e);
    double *d_g = reinterpret_cast<double *>(w + a4k + 4 * a4e + a8e);
    int32_t *d_row = reinterpret_cast<int32_t *>(w + a4k + 4 * a4e + 2 * a8e);
    auto up = [&](void *dst, const void *src, size_t bytes) {
        return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) : hipSuccess;
    };
    NODAL_HIP_TRY(h, up(d_new, plan.newidx.data(), (size_t)K * 4));
    NODAL_HIP_TRY(h, up(d_p, p.data(), (size_t)ne * 4));
    NODAL_HIP_TRY(h, up(d_q, q.data(), (size_t)ne * 4));
    NODAL_HIP_TRY(h, up(d_c, c.data(), (size_t)ne * 4));
    NODAL_HIP_TRY(h, up(d_d, d.data(), (size_t)ne * 4));
    NODAL_HIP_TRY(h, up(d_cst, cst.data(), (size_t)ne * 8));
    NODAL_HIP_TRY(h, up(d_g, g.data(), (size_t)ne * 8));
    NODAL_HIP_TRY(h, up(d_row, plan.row_of.data(), (size_t)B * 4));
    double *x = h->x.as<double>();
    scatter_nodes<<<grid_for(K), TB, 0, st>>>(K, d_new, y, x);
    if (ne) eval_pivots<<<grid_for(ne), TB, 0, st>>>(ne, d_p, d_q, d_cst, d_c, d_d, d_g, x);
    for (int pass = 0; pass < 2; ++pass)
        recover_currents<<<grid_for(B), TB, 0, st>>>(pass, K, B, d_row, h->indptr.as<int32_t>(),
                                                    h->indices.as<int32_t>(), h->data.as<double>(),
                                                    h->rhs.as<double>(), x);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipStreamSynchronize(st));  // host vectors above go out of scope
    return NODAL_OK;
}

int nodal_upload_internal(nodal_ctx *h, int64_t ncomp, const uint8_t *type, const double *value,
                          const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *d,
                          const int32_t *drv, const int32_t *k, int32_t K, int32_t B);  // api.hip

// Try the presolve route for the system of `h` (B > 0).  Returns NODAL_OK with
// *done = true when x was produced and verified; *done = false means "not applicable"
// (the caller falls back to the full-system Krylov solve).
int presolve_solve(nodal_ctx *h, bool *done, int32_t *info, int32_t *iters, double *resid) {
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
    PresolvePlan plan;
    presolve_plan(h, value, plan);
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
        h->reduced->owns_streams = false;
        h->reduced->keep_host_table = false;
    }
    nodal_ctx *r = h->reduced;
    NODAL_TRY(nodal_upload_internal(r, (int64_t)plan.type.size(), plan.type.data(), plan.value.data(),
                                    plan.a.data(), plan.b.data(), plan.c.data(), plan.d.data(),
                                    plan.drv.data(), plan.k.data(), plan.Kr, 0));
    const auto t2 = now();
    int s = stamp_symbolic(r);
    if (s == NODAL_OK) s = stamp_numeric(r, 0, nullptr);
    if (s != NODAL_OK) { h->err = r->err; return s; }
    const auto t3 = now();
    int32_t rinfo = 0;
    s = sparse_solve(r, NODAL_SPARSE_AUTO, &rinfo, iters, resid);
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
    if (!(scaled <= 1e-11)) return NODAL_OK;  // fall back to the full-system solve
    *info = 0;
    *done = true;
    return NODAL_OK;
}
