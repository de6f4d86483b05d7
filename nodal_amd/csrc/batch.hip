// Batch entry points of the C ABI (SURVEY.md section 8b "batch variants taking batch x n
// outputs"; BASELINE.json config 4: a value sweep on one topology).
//
// The reference has no batch API: a sweep is a Python loop of `Circuit(netlist, sparse)` +
// `.solve()` (reference nodal/nodal.py:306-336), one circuit after the other.  Here the
// members [first, first + count) of the value table uploaded by nodal_upload_values become
// ONE block-diagonal system, built on the device from the single topology in HBM: circuits
// that share the ground node do not couple (the ground row is eliminated), so member m owns
// the unknowns x[m K : (m+1) K] and x[count K + m B : count K + (m+1) B] of a netlist with
// count * ncomp rows.  One symbolic phase, one numeric fold, one multigrid setup and one
// Krylov iteration serve the whole shard and keep the GPU full (a single 1e4-node circuit
// occupies a few CUs).  If the block system cannot be solved as a whole -- one member is
// singular, has a zero resistance or a stamp collision -- the members are solved one by
// one on the parent context, so that only the offending members report it, as a loop over
// the reference would.
#include <cmath>
#include <vector>

#include "ctx.h"

namespace {

constexpr int TB = 256;

inline unsigned grid_for(int64_t n) {
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > 8192 ? 8192 : g);
}

// row (m, c) of the block-diagonal table = row c of the topology, shifted into member m's index ranges
__global__ __launch_bounds__(TB) void replicate_rows(
    int64_t ncomp, int32_t count, int32_t first, int32_t K, int32_t B,
    const uint8_t *__restrict__ type, const double *__restrict__ values,  // values: [batch][ncomp]
    const int32_t *__restrict__ a, const int32_t *__restrict__ b, const int32_t *__restrict__ c,
    const int32_t *__restrict__ d, const int32_t *__restrict__ drv, const int32_t *__restrict__ k,
    const double *__restrict__ src_scale,  // (null, or [count]: independent sources of member m divided by it)
    uint8_t *__restrict__ otype, double *__restrict__ ovalue, int32_t *__restrict__ oa,
    int32_t *__restrict__ ob, int32_t *__restrict__ oc, int32_t *__restrict__ od,
    int32_t *__restrict__ odrv, int32_t *__restrict__ ok) {
    const int64_t total = ncomp * count;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int32_t m = (int32_t)(i / ncomp);
        const int64_t r = i - (int64_t)m * ncomp;
        auto node = [&](int32_t v) { return v >= 0 ? v + m * K : -1; };
        const uint8_t t = type[r];
        const double v = values[(int64_t)(first + m) * ncomp + r];
        otype[i] = t;
        ovalue[i] = (src_scale && (t == NODAL_T_A || t == NODAL_T_E)) ? v / src_scale[m] : v;  // (a power of two: exact)
        oa[i] = node(a[r]);
        ob[i] = node(b[r]);
        oc[i] = node(c[r]);
        od[i] = node(d[r]);
        const int32_t dr = drv[r], kk = k[r];
        odrv[i] = dr >= 0 ? (int32_t)(dr + (int64_t)m * ncomp) : -1;
        ok[i] = kk >= 0 ? kk + m * B : -1;
    }
}

// block-diagonal unknown vector -> [count][K + B], member m's part times its right-hand-side scale
__global__ __launch_bounds__(TB) void split_members(int32_t count, int32_t K, int32_t B,
                                                    const double *__restrict__ x,
                                                    const double *__restrict__ scale,
                                                    double *__restrict__ out) {
    const int64_t n = (int64_t)K + B, total = n * count;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int64_t m = i / n, j = i - m * n;
        out[i] = scale[m] * (j < K ? x[m * K + j] : x[(int64_t)count * K + m * B + (j - K)]);
    }
}

// ---- per-member equilibration of the right-hand side -------------------------------------
// The block system is solved by ONE Krylov iteration with ONE stopping test (|r| <= tol |b| over
// the whole shard).  A member whose sources are orders of magnitude weaker than the others' would
// be left far less converged than a solve of its own -- the reference's loop of per-circuit direct
// solves treats every member alike -- so every member's right-hand side is scaled to [1, 2) by a
// power of two (exact in floating point; the systems are linear and independent: x_m = s_m x~_m)
// before the joint solve.  Then the one test bounds every member's relative residual.
// A block WITH branch unknowns is scaled at its sources instead (source_scales below): its presolve
// (presolve.hip) rebuilds the reduced system from the component VALUES, host and device copies, so the scale
// has to be in the table -- the independent sources A and E of member m divided by s_m, nothing else: the
// systems are linear in them -- or the presolved answer would be checked against a right-hand side it was
// never computed for (and rejected: every such sweep then paid for the plan, the reduced build, the reduced
// solve AND the full-system route).
__device__ __forceinline__ int64_t member_of_row(int64_t i, int32_t count, int32_t K, int32_t B) {
    const int64_t nodes = (int64_t)count * K;
    return i < nodes ? i / K : (i - nodes) / B;
}
// absmax[m] = max |rhs_i| over member m's rows (non-negative doubles order like their bit patterns:
// an integer atomicMax is exact and order-independent)
__global__ __launch_bounds__(TB) void member_absmax(int64_t n, int32_t count, int32_t K, int32_t B,
                                                    const double *__restrict__ rhs,
                                                    unsigned long long *__restrict__ absmax) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double v = fabs(rhs[i]);
        if (v > 0.0 && v == v) atomicMax(&absmax[member_of_row(i, count, K, B)], (unsigned long long)__double_as_longlong(v));
    }
}
// absmax[m] (bit pattern) -> scale[m] = 2^floor(log2 absmax) (1 for an all-zero or non-finite member)
__global__ __launch_bounds__(TB) void member_scales(int32_t count, const unsigned long long *__restrict__ absmax,
                                                    double *__restrict__ scale) {
    for (int32_t m = blockIdx.x * TB + threadIdx.x; m < count; m += gridDim.x * TB) {
        const double v = __longlong_as_double((long long)absmax[m]);
        int e = 0;
        double s = 1.0;
        if (v > 0.0 && v < 1.0 / 0.0) {
            (void)frexp(v, &e);          // v = f 2^e, f in [0.5, 1)
            s = ldexp(1.0, e - 1);       // 2^(e-1) <= v < 2^e
            if (!(s > 0.0)) s = 1.0;     // (denormal range: leave it)
        }
        scale[m] = s;
    }
}
__global__ __launch_bounds__(TB) void member_scale_rhs(int64_t n, int32_t count, int32_t K, int32_t B,
                                                       const double *__restrict__ scale, double *__restrict__ rhs) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        rhs[i] /= scale[member_of_row(i, count, K, B)];  // (a power of two: exact)
}

__global__ __launch_bounds__(TB) void fill_nan_rows(double *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        out[i] = __builtin_nan("");
}


// ---- the block system's symbolic phase, from one member's ------------------------------
// Without branch unknowns member m's rows, entries and contributions are the topology's, shifted:
// row r -> m n + r, entry e -> m nnz + e, contribution p -> m C + p, component i -> m ncomp + i.
// Grouping the block table from scratch (stamp_symbolic on count x ncomp rows) finds exactly
// this -- the per-row sort keys (column, component, slot) shift monotonically -- at count times
// the cost.

// dst[m * per + j] = src[j] + m * step   (src[j] < 0 stays negative when `keep_negative`)
__global__ __launch_bounds__(TB) void shift_copies_i32(int64_t per, int32_t count, const int32_t *__restrict__ src,
                                                       int64_t step, bool keep_negative,
                                                       int32_t *__restrict__ dst) {
    const int64_t total = per * count;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < total; i += (int64_t)gridDim.x * TB) {
        const int64_t m = i / per, j = i - m * per;
        const int32_t v = src[j];
        dst[i] = (keep_negative && v < 0) ? v : (int32_t)(v + m * step);
    }
}
__global__ void set_i32(int32_t *p, int32_t v) { *p = v; }

int replicate_symbolic(nodal_ctx *h, nodal_ctx *c, int32_t count) {
    hipStream_t st = h->stream;
    const int64_t n = h->n, nnz = h->nnz, C = h->ncontrib, nr = h->nrhs, Cr = h->nrhs_contrib;
    if (nnz * count >= (1ll << 31) - 2 || C * count >= (1ll << 31) - 2)
        return nodal_fail(h, NODAL_E_UNSUPPORTED, "run_batch: shard too large for 32-bit indices; split it");
    ++c->struct_epoch;
    // (rows a low-degree elimination could take: the member's count, times the members -- lowdeg.hip looks before it
    // launches its selection)
    if (h->low_rows >= 0 && h->low_rows_epoch == h->struct_epoch) {
        c->low_rows = h->low_rows * count;
        c->low_rows_epoch = c->struct_epoch;
    }
    c->nnz = nnz * count;
    c->ncontrib = C * count;
    c->nrhs = nr * count;
    c->nrhs_contrib = Cr * count;
    NODAL_HIP_TRY(h, c->indices.reserve((size_t)c->nnz * 4 + 4));
    NODAL_HIP_TRY(h, c->rowidx.reserve((size_t)c->nnz * 4 + 4));
    NODAL_HIP_TRY(h, c->cptr.reserve((size_t)(c->nnz + 1) * 4));
    NODAL_HIP_TRY(h, c->contrib.reserve((size_t)c->ncontrib * 4 + 4));
    NODAL_HIP_TRY(h, c->indptr.reserve((size_t)(c->n + 1) * 4));
    NODAL_HIP_TRY(h, c->diag_pos.reserve((size_t)c->n * 4 + 4));
    NODAL_HIP_TRY(h, c->rhs_row.reserve((size_t)c->nrhs * 4 + 4));
    NODAL_HIP_TRY(h, c->rhs_cptr.reserve((size_t)(c->nrhs + 1) * 4));
    NODAL_HIP_TRY(h, c->rhs_contrib.reserve((size_t)c->nrhs_contrib * 4 + 4));
    auto shift = [&](int64_t per, const DevBuf &src, int64_t step, bool keep_negative, DevBuf &dst) {
        if (per > 0)
            shift_copies_i32<<<grid_for(per * count), TB, 0, st>>>(per, count, src.as<int32_t>(), step, keep_negative,
                                                                   dst.as<int32_t>());
    };
    shift(nnz, h->indices, n, false, c->indices);
    shift(nnz, h->rowidx, n, false, c->rowidx);
    shift(nnz, h->cptr, C, false, c->cptr);
    set_i32<<<1, 1, 0, st>>>(c->cptr.as<int32_t>() + c->nnz, (int32_t)c->ncontrib);
    shift(C, h->contrib, (int64_t)h->ncomp << 3, false, c->contrib);  // (u32 comp << 3 | slot)
    shift(n, h->indptr, nnz, false, c->indptr);
    set_i32<<<1, 1, 0, st>>>(c->indptr.as<int32_t>() + c->n, (int32_t)c->nnz);
    shift(n, h->diag_pos, nnz, true, c->diag_pos);
    shift(nr, h->rhs_row, n, false, c->rhs_row);
    shift(nr, h->rhs_cptr, Cr, false, c->rhs_cptr);
    set_i32<<<1, 1, 0, st>>>(c->rhs_cptr.as<int32_t>() + c->nrhs, (int32_t)c->nrhs_contrib);
    shift(Cr, h->rhs_contrib, (int64_t)h->ncomp << 3, false, c->rhs_contrib);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, c->data.reserve((size_t)c->nnz * 8 + 8));
    NODAL_HIP_TRY(h, c->rhs.reserve((size_t)c->n * 8 + 8));
    NODAL_HIP_TRY(h, c->x.reserve((size_t)c->n * 8 + 8));
    NODAL_HIP_TRY(h, c->status.reserve(64));
    c->have_symbolic = true;
    return NODAL_OK;
}

nodal_ctx *block_child(nodal_ctx *h) {
    if (!h->blocksys) {
        nodal_ctx *c = new nodal_ctx();
        c->device = h->device;
        c->stream = h->stream;  // shared: one ordered timeline
        c->stream2 = h->stream2;
        c->stream3 = h->stream3;
        for (int i = 0; i < 4; ++i) c->ev[i] = h->ev[i];
        for (int i = 0; i < 2; ++i) c->ev_la[i] = h->ev_la[i];
        for (int i = 0; i < 6; ++i) c->ev_bi[i] = h->ev_bi[i];
        c->owns_streams = false;
        c->stream_owner = h->stream_owner ? h->stream_owner : h;
        h->blocksys = c;
    }
    nodal_ctx *c = h->blocksys;
    c->dense_blockinv = h->dense_blockinv;
    c->gj_scalar = h->gj_scalar;
    c->use_graphs = h->use_graphs;
    c->use_presolve = h->use_presolve;
    return c;
}

// host copy of the block table for the presolve (only systems with branch equations keep one)
void replicate_host(const nodal_ctx *h, nodal_ctx *c, int32_t first, int32_t count, const double *src_scale) {
    const HostTable &s = h->host;
    HostTable &t = c->host;
    const int64_t nc = h->ncomp, total = nc * count;
    t.type.resize(total); t.value.resize(total);
    t.a.resize(total); t.b.resize(total); t.c.resize(total); t.d.resize(total);
    t.drv.resize(total); t.k.resize(total);
    t.values_batch.clear();
    t.branch_rows.clear();
    for (int32_t m = 0; m < count; ++m)
        for (const int64_t r : s.branch_rows) t.branch_rows.push_back((int64_t)m * nc + r);
    for (int32_t m = 0; m < count; ++m) {
        const double *vals = s.values_batch.data() + (size_t)(first + m) * nc;
        for (int64_t r = 0; r < nc; ++r) {
            const int64_t i = (int64_t)m * nc + r;
            auto node = [&](int32_t v) { return v >= 0 ? v + m * h->K : -1; };
            t.type[i] = s.type[r];
            t.value[i] = (src_scale && (s.type[r] == NODAL_T_A || s.type[r] == NODAL_T_E)) ? vals[r] / src_scale[m] : vals[r];
            t.a[i] = node(s.a[r]); t.b[i] = node(s.b[r]); t.c[i] = node(s.c[r]); t.d[i] = node(s.d[r]);
            t.drv[i] = s.drv[r] >= 0 ? (int32_t)(s.drv[r] + (int64_t)m * nc) : -1;
            t.k[i] = s.k[r] >= 0 ? s.k[r] + m * h->B : -1;
        }
    }
}

// s_m = 2^floor(log2 max|v|) over member m's independent sources (1 if it has none or they are all zero / not
// finite): the same [1, 2) normalisation as member_scales, from the host copy of the value table
void source_scales(const nodal_ctx *h, int32_t first, int32_t count, std::vector<double> &out) {
    const HostTable &s = h->host;
    const int64_t nc = h->ncomp;
    out.assign((size_t)count, 1.0);
    for (int32_t m = 0; m < count; ++m) {
        const double *vals = s.values_batch.data() + (size_t)(first + m) * nc;
        double mx = 0.0;
        for (int64_t r = 0; r < nc; ++r)
            if (s.type[r] == NODAL_T_A || s.type[r] == NODAL_T_E) {
                const double v = fabs(vals[r]);
                if (v > mx && v < 1.0 / 0.0) mx = v;
            }
        if (mx > 0.0) {
            int e = 0;
            (void)frexp(mx, &e);
            const double sc = ldexp(1.0, e - 1);
            if (sc > 0.0 && sc < 1.0 / 0.0) out[(size_t)m] = sc;
        }
    }
}

double elapsed(hipEvent_t a, hipEvent_t b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) != hipSuccess) return 0.0;
    return ms;
}

// members one by one on the parent context: info[m] = solver info (> 0 singular: row of NaNs),
// or -status for a member whose assembly fails (zero resistance, stamp collision)
int solve_members_one_by_one(nodal_ctx *h, int32_t first, int32_t count, int32_t *info_out, double *out) {
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    if (!h->have_symbolic) NODAL_TRY(stamp_symbolic(h));
    for (int32_t m = 0; m < count; ++m) {
        int32_t inf = 0, it = 0;
        double rs = 0.0;
        int64_t bad = -1;
        int s = stamp_numeric(h, first + m, &bad);
        if (s == NODAL_E_ZERO_RESISTANCE || s == NODAL_E_STAMP_COLLISION) {
            inf = -s;
        } else if (s != NODAL_OK) {
            return s;
        } else {
            s = sparse_solve(h, NODAL_SPARSE_AUTO, &inf, &it, &rs);
            if (s != NODAL_OK) return s;
        }
        if (info_out) info_out[m] = inf;
        if (n > 0) {
            if (inf != 0) fill_nan_rows<<<grid_for(n), TB, 0, st>>>(out + (int64_t)m * n, n);
            else NODAL_HIP_TRY(h, hipMemcpyAsync(out + (int64_t)m * n, h->x.p, (size_t)n * 8,
                                                 hipMemcpyDeviceToDevice, st));
            NODAL_HIP_TRY(h, hipGetLastError());
        }
    }
    return NODAL_OK;
}

}  // namespace

void nodal_free_block_child(nodal_ctx *h) {
    if (h->blocksys) {
        nodal_free_buffers(h->blocksys);
        delete h->blocksys;
        h->blocksys = nullptr;
    }
    h->batch_x.release();
    for (auto &e : h->ev_batch)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
}

extern "C" {

int nodal_run_batch(nodal_handle h, int32_t first, int32_t count, int32_t reuse_symbolic,
                    double *x_out, int32_t *info_out) {
    if (!h || first < 0 || count < 1) return NODAL_E_INVALID;
    if (!h->have_table) return nodal_fail(h, NODAL_E_INVALID, "upload_components not called");
    if (h->batch < 1 || (int64_t)first + count > h->batch)
        return nodal_fail(h, NODAL_E_INVALID, "run_batch: members outside the uploaded value table");
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(h->device);
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
    FillStreamScope fill(h->stream);
    if (h->hung) return NODAL_E_HIP;
    nodal_poison_scratch(h);
    hipStream_t st = h->stream;
    const int64_t n = h->n, ncomp = h->ncomp;
    h->batch_count = 0;
    h->last_batch_block = false;
    for (auto &e : h->ev_batch)  // (the solvers record h->ev[2..3] around their dominant kernel)
        if (!e) NODAL_HIP_TRY(h, hipEventCreate(&e));
    hipEvent_t *ev = h->ev_batch;
    NODAL_HIP_TRY(h, h->batch_x.reserve((size_t)n * count * 8 + 64));
    double *out = h->batch_x.as<double>();
    if ((int64_t)count * h->K >= (1ll << 31) - 2 || (int64_t)count * ncomp >= (1ll << 29))
        return nodal_fail(h, NODAL_E_UNSUPPORTED, "run_batch: shard too large for 32-bit indices; split it");

    bool whole = count > 1 && n > 0;
    int s = NODAL_OK;
    if (whole) {
        nodal_ctx *c = block_child(h);
        const int64_t total = ncomp * count;
        const bool same_shape = c->have_table && c->ncomp == total && c->K == (int32_t)((int64_t)count * h->K) &&
                                c->B == (int32_t)((int64_t)count * h->B) && c->block_epoch == h->table_epoch;
        NODAL_HIP_TRY(h, hipEventRecord(ev[0], st));
        DevBuf *dst[] = {&c->type, &c->value, &c->a, &c->b, &c->c, &c->d, &c->drv, &c->k};
        const size_t width[] = {1, 8, 4, 4, 4, 4, 4, 4};
        for (int i = 0; i < 8; ++i) NODAL_HIP_TRY(h, dst[i]->reserve((size_t)total * width[i] + 16));
        // per-member scales: [count] bit patterns of max |rhs| (device route) | [count] doubles
        NODAL_HIP_TRY(h, h->batch_scale.reserve((size_t)count * 16 + 64));
        unsigned long long *absmax = h->batch_scale.as<unsigned long long>();
        double *scale = reinterpret_cast<double *>(absmax + count);
        // a block with branch unknowns and a host copy of its values (what its presolve works from) is
        // scaled at its sources, in the table; every other block at its assembled right-hand side
        const bool source_scaled = h->B > 0 && !h->host.type.empty() && !h->host.values_batch.empty();
        std::vector<double> sc_host;
        if (source_scaled) {
            source_scales(h, first, count, sc_host);
            NODAL_HIP_TRY(h, hipMemcpyAsync(scale, sc_host.data(), (size_t)count * 8, hipMemcpyHostToDevice, st));
            NODAL_WAIT_STREAM(h, st);  // (pageable source: gone when this scope is left early)
        }
        replicate_rows<<<grid_for(total), TB, 0, st>>>(
            ncomp, count, first, h->K, h->B, h->type.as<uint8_t>(), h->values_batch.as<double>(),
            h->a.as<int32_t>(), h->b.as<int32_t>(), h->c.as<int32_t>(), h->d.as<int32_t>(),
            h->drv.as<int32_t>(), h->k.as<int32_t>(), source_scaled ? scale : nullptr, c->type.as<uint8_t>(),
            c->value.as<double>(), c->a.as<int32_t>(), c->b.as<int32_t>(), c->c.as<int32_t>(), c->d.as<int32_t>(),
            c->drv.as<int32_t>(), c->k.as<int32_t>());
        NODAL_HIP_TRY(h, hipGetLastError());
        c->ncomp = total;
        c->K = (int32_t)((int64_t)count * h->K);
        c->B = (int32_t)((int64_t)count * h->B);
        c->n = (int64_t)c->K + c->B;
        c->batch = 0;
        if (!same_shape) ++c->table_epoch;  // (another topology or member count: nothing cached about the block table holds)
        c->have_table = true;
        c->have_numeric = c->have_x = false;
        if (!(reuse_symbolic && same_shape && c->have_symbolic)) c->have_symbolic = false;
        c->block_epoch = h->table_epoch;
        c->keep_host_table = source_scaled;
        if (c->keep_host_table) replicate_host(h, c, first, count, sc_host.data());
        else c->host = HostTable();
        if (!c->have_symbolic) {
            if (h->B == 0) {  // one member's grouping, shifted (the parent's symbolic phase is redone too unless kept)
                if (!(reuse_symbolic && h->have_symbolic)) s = stamp_symbolic(h);
                if (s == NODAL_OK) s = replicate_symbolic(h, c, count);
            } else {
                s = stamp_symbolic(c);
            }
        }
        NODAL_HIP_TRY(h, hipEventRecord(ev[1], st));
        int32_t inf = 0, it = 0;
        double rs = 0.0;
        int64_t bad = -1;
        if (s == NODAL_OK) s = stamp_numeric(c, 0, &bad);
        if (s == NODAL_OK && !source_scaled) {  // every member's right-hand side to [1, 2): one stopping test serves them all
            NODAL_HIP_TRY(h, hipMemsetAsync(absmax, 0, (size_t)count * 8, st));
            member_absmax<<<grid_for(c->n), TB, 0, st>>>(c->n, count, h->K, h->B, c->rhs.as<double>(), absmax);
            member_scales<<<grid_for(count), TB, 0, st>>>(count, absmax, scale);
            member_scale_rhs<<<grid_for(c->n), TB, 0, st>>>(c->n, count, h->K, h->B, scale, c->rhs.as<double>());
            NODAL_HIP_TRY(h, hipGetLastError());
        }
        NODAL_HIP_TRY(h, hipEventRecord(ev[2], st));
        if (s == NODAL_OK) {
            c->amg_levels = 0;
            s = sparse_solve(c, NODAL_SPARSE_AUTO, &inf, &it, &rs);
        }
        NODAL_HIP_TRY(h, hipEventRecord(ev[3], st));
        if (s == NODAL_OK && inf == 0) {
            split_members<<<grid_for(n * count), TB, 0, st>>>(count, h->K, h->B, c->x.as<double>(), scale, out);
            NODAL_HIP_TRY(h, hipGetLastError());
            if (info_out)
                for (int32_t m = 0; m < count; ++m) info_out[m] = 0;
            h->last_iterations = it;
            h->last_relres = rs;
            h->amg_levels = c->amg_levels;
            h->kern_ms = c->kern_ms;
            h->kern_launches = c->kern_launches;
            h->kern_alg = c->kern_alg;
            h->last_batch_block = true;
        } else if (s == NODAL_OK || s == NODAL_E_ZERO_RESISTANCE || s == NODAL_E_STAMP_COLLISION ||
                   s == NODAL_E_UNSUPPORTED || s == NODAL_E_SINGULAR) {
            whole = false;  // some member spoils the block: find out which, one by one
        } else {
            h->err = c->err;
            return s;
        }
        NODAL_WAIT_EVENT(h, ev[3], st);
        h->ms[0] = elapsed(ev[0], ev[1]);
        h->ms[1] = elapsed(ev[1], ev[2]);
        h->ms[2] = elapsed(ev[2], ev[3]);
    }
    if (!whole) NODAL_TRY(solve_members_one_by_one(h, first, count, info_out, out));
    h->batch_count = count;
    if (x_out && n > 0)
        NODAL_HIP_TRY(h, hipMemcpyAsync(x_out, out, (size_t)n * count * 8, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    return NODAL_OK;
}

int nodal_batch_x_device(nodal_handle h, void *device_dst, int64_t capacity_bytes) {
    if (!h || !device_dst) return NODAL_E_INVALID;
    if (h->batch_count < 1) return nodal_fail(h, NODAL_E_INVALID, "run_batch not called");
    const size_t bytes = (size_t)h->n * h->batch_count * 8;
    if ((size_t)capacity_bytes < bytes) return nodal_fail(h, NODAL_E_INVALID, "batch_x_device: destination too small");
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(h->device);
    hipError_t e = bytes ? hipMemcpyAsync(device_dst, h->batch_x.p, bytes, hipMemcpyDeviceToDevice, h->stream)
                         : hipSuccess;
    const int ws = e == hipSuccess ? nodal_wait_stream(h, h->stream, NODAL_SITE) : NODAL_OK;
    (void)hipSetDevice(prev);
    NODAL_HIP_TRY(h, e);
    return ws;
}

int nodal_x_device(nodal_handle h, void *device_dst, int64_t capacity_bytes) {
    if (!h || !device_dst) return NODAL_E_INVALID;
    if (h->hung) return NODAL_E_HIP;
    if (!h->have_x) return nodal_fail(h, NODAL_E_INVALID, "x_device: no solution on the handle");
    const size_t bytes = (size_t)h->n * 8;
    if ((size_t)capacity_bytes < bytes) return nodal_fail(h, NODAL_E_INVALID, "x_device: destination too small");
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(h->device);
    hipError_t e = bytes ? hipMemcpyAsync(device_dst, h->x.p, bytes, hipMemcpyDeviceToDevice, h->stream) : hipSuccess;
    const int ws = e == hipSuccess ? nodal_wait_stream(h, h->stream, NODAL_SITE) : NODAL_OK;
    (void)hipSetDevice(prev);
    NODAL_HIP_TRY(h, e);
    return ws;
}

}  // extern "C"
