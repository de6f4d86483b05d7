// Exact elimination of nodes with one or two neighbours (passive networks, sparse path).
//
// Chain-like parts of a network -- resistor ladders, long wires, trees -- are what the
// aggregation multigrid of amg.hip handles worst (piecewise constants along a wire correct
// little: DESIGN.md section 8).  They are also what Gaussian elimination handles best: a
// node with <= 2 neighbours is eliminated without any fill (two edges become one).  One
// ROUND removes an independent set F of such nodes at once:
//
//     [ D_F   A_FC ] [x_F]   [b_F]        S   = A_CC - A_CF D_F^-1 A_FC      (D_F diagonal)
//     [ A_CF  A_CC ] [x_C] = [b_C]   =>   S x_C = b_C - A_CF D_F^-1 b_F
//                                         x_F  = D_F^-1 (b_F - A_FC x_C)
//
// S is again a symmetric M-matrix (a passive network on the kept nodes: the star-mesh
// transformation of every eliminated node), so the round's result is just another context
// (`h->lowdeg`, matrix only, no component table) that goes through sparse_solve again:
// further rounds while enough nodes can be removed (sparse_solve sets the bar), then the dense
// direct solve or the multigrid CG.  A ladder of 1e5 sections is solved exactly in 7 rounds
// and a small dense solve (2.7 ms, 1.1 ms when repeated) instead of 208 CG iterations (31 ms).
//
// Everything is deterministic: F is chosen by a fixed priority (odd index first, then a hash:
// a candidate is taken if it beats every candidate neighbour), the kept nodes keep their
// relative order, and S is grouped by group.h like every other matrix here (contributions
// summed in a fixed order, S bitwise symmetric).
//
// The reference hands G to SuperLU (scipy spsolve, reference nodal/nodal.py:325; dense:
// np.linalg.solve, nodal/nodal.py:327), whose column ordering (COLAMD) eliminates exactly
// these nodes first; this file is the data-parallel counterpart.  The topology-dependent part
// of a round (F, the pattern of S) is cached per context, see lowdeg_solve.
#include <chrono>

#include "ctx.h"
#include "group.h"

namespace {

using grp::TB;
using grp::grid_for;

struct View {
    int64_t n, nnz;
    const int32_t *indptr, *indices, *rowidx, *diag_pos;
    const double *data;
};

// Total order on the nodes: odd indices first, a hash among equals.  Netlists number the nodes
// of a wire consecutively more often than not (first appearance in the file, and a round keeps
// the order of the nodes it keeps), and then every other node of the wire is taken -- half of
// it per round instead of the third a purely random order gives.
__device__ __forceinline__ uint64_t node_priority(uint32_t i) {
    uint64_t h = (uint64_t)i * 0x9E3779B97F4A7C15ull;
    h ^= h >> 31;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    return ((uint64_t)(i & 1u) << 63) | ((h >> 1) & 0x7FFFFFFF00000000ull) | i;  // distinct for distinct nodes
}

// 1 or 2 neighbours and a positive diagonal
__device__ __forceinline__ bool is_candidate(const View &A, int i) {
    const int32_t dp = A.diag_pos[i];
    if (dp < 0) return false;
    const int off = A.indptr[i + 1] - A.indptr[i] - 1;
    return off >= 1 && off <= 2 && A.data[dp] > 0.0;
}

// keep[i] = 0 for the nodes of the independent set F, 1 otherwise; *count += |F|
__global__ __launch_bounds__(TB) void select_nodes(View A, uint32_t *__restrict__ keep,
                                                   uint32_t *__restrict__ count) {
    uint32_t mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        bool take = is_candidate(A, (int)i);
        if (take) {
            const uint64_t pi = node_priority((uint32_t)i);
            for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
                const int j = A.indices[e];
                if (j != (int)i && is_candidate(A, j) && node_priority((uint32_t)j) > pi) take = false;
            }
        }
        keep[i] = take ? 0u : 1u;
        mine += take ? 1u : 0u;
    }
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && total) atomicAdd(count, total);
}

// newidx[i] = position among the kept nodes, or -1 for an eliminated node
__global__ __launch_bounds__(TB) void finish_newidx(int64_t n, const uint32_t *__restrict__ keep,
                                                    const uint32_t *__restrict__ pos,
                                                    int32_t *__restrict__ newidx) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        newidx[i] = keep[i] ? (int32_t)pos[i] : -1;
}

// Entries of S: a kept-kept entry of A passes through (slot 0); an entry (i, f) towards an
// eliminated node f contributes -a_if a_fk / d_f to (i, k) for each neighbour k of f
// (slot 1 + position of k among the off-diagonal entries of row f; k == i is the diagonal).
struct SchurEntries {
    [[maybe_unused]] static constexpr int SLOTS = 3;
    int64_t nitems;
    const int32_t *indptr, *indices, *rowidx, *newidx;
    template <class F>
    __device__ void for_each(int64_t e, F f) const {
        const int i = rowidx[e], ni = newidx[i];
        if (ni < 0) return;
        const int j = indices[e], nj = newidx[j];
        if (nj >= 0) {
            f(0, ni, nj);
            return;
        }
        int t = 0;
        for (int32_t q = indptr[j]; q < indptr[j + 1]; ++q) {
            const int k = indices[q];
            if (k == j) continue;
            f(1 + t, ni, newidx[k]);  // k is kept: F is an independent set
            ++t;
        }
    }
};

__global__ __launch_bounds__(TB) void schur_values(View A, const int32_t *__restrict__ cptr,
                                                   const uint32_t *__restrict__ contrib,
                                                   double *__restrict__ out, int64_t nent) {
    for (int64_t o = (int64_t)blockIdx.x * TB + threadIdx.x; o < nent; o += (int64_t)gridDim.x * TB) {
        double s = 0.0;
        for (int32_t p = cptr[o]; p < cptr[o + 1]; ++p) {
            const uint32_t c = contrib[p];
            const int64_t e = c >> 3;
            const int slot = (int)(c & 7u);
            const double v = A.data[e];
            if (slot == 0) {
                s += v;
            } else {
                const int f = A.indices[e];
                int t = slot - 1;
                double w = 0.0;
                for (int32_t q = A.indptr[f]; q < A.indptr[f + 1]; ++q) {
                    if (A.indices[q] == f) continue;
                    if (t-- == 0) w = A.data[q];
                }
                s -= v * w / A.data[A.diag_pos[f]];
            }
        }
        out[o] = s;
    }
}

// right-hand side and "has a resistor to ground" flags of the kept nodes
__global__ __launch_bounds__(TB) void reduce_rhs(View A, const int32_t *__restrict__ newidx,
                                                 const double *__restrict__ b,
                                                 const uint8_t *__restrict__ grounded,
                                                 double *__restrict__ bc,
                                                 uint8_t *__restrict__ gc) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        const int ni = newidx[i];
        if (ni < 0) continue;
        double s = b[i];
        uint8_t g = grounded[i];
        for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
            const int f = A.indices[e];
            if (newidx[f] >= 0) continue;
            s -= A.data[e] * b[f] / A.data[A.diag_pos[f]];
            g |= grounded[f];
        }
        bc[ni] = s;
        gc[ni] = g;
    }
}

// A node of the reduced network without neighbours and without a resistor to ground is all
// that is left of a floating sub-network (a chain or tree that touched nothing else): G is
// singular.  Rounding may leave +-1 ulp instead of the exact zero on its diagonal, so the
// verdict is structural.
__global__ __launch_bounds__(TB) void find_floating_leftovers(int64_t n, const int32_t *__restrict__ indptr,
                                                              const uint8_t *__restrict__ grounded,
                                                              uint32_t *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        if (indptr[i + 1] - indptr[i] <= 1 && !grounded[i]) *flag = 1u;
}

__global__ __launch_bounds__(TB) void recover_x(View A, const int32_t *__restrict__ newidx,
                                                const double *__restrict__ b,
                                                const double *__restrict__ xc,
                                                double *__restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        const int ni = newidx[i];
        if (ni >= 0) {
            x[i] = xc[ni];
            continue;
        }
        double s = b[i];
        for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
            const int k = A.indices[e];
            if (k != (int)i) s -= A.data[e] * xc[newidx[k]];
        }
        x[i] = s / A.data[A.diag_pos[i]];
    }
}

// column-major dense copy of a CSR matrix (the panel is zeroed by the caller)
__global__ __launch_bounds__(TB) void scatter_dense(View A, double *__restrict__ G, int64_t ld) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < A.nnz; e += (int64_t)gridDim.x * TB)
        G[(int64_t)A.indices[e] * ld + A.rowidx[e]] = A.data[e];
}

// Connected components of a small network (<= CC_MAX nodes) by min-label hooking and pointer
// jumping in LDS, one workgroup; *flag = 1 if some component has no grounded node.
constexpr int CC_MAX = 4096;
__global__ __launch_bounds__(1024) void small_floating_check(View A, const uint8_t *__restrict__ grounded,
                                                             uint32_t *__restrict__ flag) {
    __shared__ int label[CC_MAX];
    __shared__ int ok[CC_MAX];
    __shared__ int changed;
    const int n = (int)A.n;
    for (int i = threadIdx.x; i < n; i += 1024) {
        label[i] = i;
        ok[i] = 0;
    }
    if (threadIdx.x == 0) changed = 0;
    __syncthreads();
    for (int it = 0; it < 2 * CC_MAX; ++it) {
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int mine = label[i];
            int m = mine;
            for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
                const int lj = label[A.indices[e]];  // may be mid-update: any value read is a valid label
                m = lj < m ? lj : m;
            }
            if (m < mine) {
                label[i] = m;
                changed = 1;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 1024) label[i] = label[label[i]];
        __syncthreads();
        const int again = changed;
        __syncthreads();
        if (!again) break;
        if (threadIdx.x == 0) changed = 0;
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 1024)
        if (grounded[i]) ok[label[i]] = 1;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024)
        if (!ok[label[i]]) *flag = 1u;
}

View view_of(const nodal_ctx *h) {
    View A;
    A.n = h->n;
    A.nnz = h->nnz;
    A.indptr = h->indptr.as<int32_t>();
    A.indices = h->indices.as<int32_t>();
    A.rowidx = h->rowidx.as<int32_t>();
    A.diag_pos = h->diag_pos.as<int32_t>();
    A.data = h->data.as<double>();
    return A;
}

}  // namespace

// u8[n] flags "a resistor joins this node to ground": from the component table, or -- for a
// context produced by an elimination round -- the flags inherited from the parent.
int grounded_flags(nodal_ctx *h, uint8_t *flags_dev) {
    if (!h->csr_only) return stamp_grounded_flags(h, flags_dev);
    NODAL_HIP_TRY(h, hipMemcpyAsync(flags_dev, h->grounded.p, (size_t)h->n, hipMemcpyDeviceToDevice, h->stream));
    return NODAL_OK;
}

// Matrix-only context of at most 4096 unknowns: *floating = 1 if a connected component has no
// resistor to ground (the dense solve that follows cannot tell: elimination without pivoting
// turns the exact zero pivot of such a component into rounding noise).
int csr_small_floating_check(nodal_ctx *h, int32_t *floating) {
    *floating = 0;
    if (!h->csr_only || h->n > CC_MAX) return nodal_fail(h, NODAL_E_INVALID, "small_floating_check: wrong context");
    NODAL_HIP_TRY(h, h->ld_work.reserve(256));
    uint32_t *flag = h->ld_work.as<uint32_t>();
    NODAL_HIP_TRY(h, hipMemsetAsync(flag, 0, 4, h->stream));
    small_floating_check<<<1, 1024, 0, h->stream>>>(view_of(h), h->grounded.as<uint8_t>(), flag);
    NODAL_HIP_TRY(h, hipGetLastError());
    uint32_t f = 0;
    NODAL_TRY(nodal_read_words(h, &f, flag, 4));
    *floating = (int32_t)f;
    return NODAL_OK;
}

// The same test for any small CSR graph (the last level of a multigrid hierarchy): launched on
// the context's stream, *flag_dev (zeroed by the caller) is set if a component is floating.
int csr_floating_check_small(nodal_ctx *h, int64_t n, const int32_t *indptr, const int32_t *indices,
                             const uint8_t *grounded, uint32_t *flag_dev) {
    if (n > CC_MAX) return nodal_fail(h, NODAL_E_INVALID, "csr_floating_check_small: too many nodes");
    View A;
    A.n = n;
    A.nnz = 0;
    A.indptr = indptr;
    A.indices = indices;
    A.rowidx = nullptr;
    A.diag_pos = nullptr;
    A.data = nullptr;
    small_floating_check<<<1, 1024, 0, h->stream>>>(A, grounded, flag_dev);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// G of a matrix-only context as a column-major dense panel
int csr_to_dense(nodal_ctx *h, double *G_dev, int64_t ld) {
    const int64_t n = h->n;
    NODAL_HIP_TRY(h, hipMemsetAsync(G_dev, 0, (size_t)ld * (size_t)n * 8, h->stream));
    if (h->nnz > 0) {
        scatter_dense<<<grid_for(h->nnz), TB, 0, h->stream>>>(view_of(h), G_dev, ld);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    return NODAL_OK;
}

// One elimination round on the passive system of `h`, then the whole solve of what is left
// (recursively through sparse_solve).  *done = false: too few nodes qualify, nothing was
// changed and the caller carries on with its own solver (the bar: n / min_share nodes).
int lowdeg_solve(nodal_ctx *h, int min_share, bool *done, int32_t *info, int32_t *iters, double *resid) {
    *done = false;
    const bool enabled = !(getenv("NODAL_LOWDEG") && atoi(getenv("NODAL_LOWDEG")) == 0);
    const bool trace = getenv("NODAL_TRACE") != nullptr;
    const int64_t n = h->n;
    if (!enabled || n < 2 || h->nnz >= (1ll << 29)) return NODAL_OK;
    hipStream_t st = h->stream;
    const View A = view_of(h);
    const auto t0 = std::chrono::steady_clock::now();
    if (const char *e = getenv("NODAL_LOWDEG_SHARE")) min_share = atoi(e) > 0 ? atoi(e) : min_share;
    const bool cached = h->ld_state != 0 && h->ld_epoch == h->struct_epoch && h->ld_share == min_share &&
                        h->ld_n == n && h->ld_nnz == A.nnz && (h->ld_state == 1 || h->lowdeg);
    auto remember = [&](int state) {
        h->ld_state = state;
        h->ld_epoch = h->struct_epoch;
        h->ld_share = min_share;
        h->ld_n = n;
        h->ld_nnz = A.nnz;
    };
    if (cached && h->ld_state == 1) return NODAL_OK;
    if (cached && h->ld_state == 3) {
        *done = true;
        *info = 1;
        *iters = 0;
        *resid = 0.0;
        return NODAL_OK;
    }
    const bool build = !cached;

    // independent set of low-degree nodes
    const size_t a_keep = grp::align_up((size_t)(n + 1) * 4);
    NODAL_HIP_TRY(h, h->ld_newidx.reserve(a_keep));
    NODAL_HIP_TRY(h, h->ld_work.reserve(2 * a_keep + 256 + scan_tmp_bytes(n + 1)));
    uint32_t *keep = h->ld_work.as<uint32_t>();
    uint32_t *pos = reinterpret_cast<uint32_t *>(h->ld_work.as<char>() + a_keep);
    uint32_t *count = reinterpret_cast<uint32_t *>(h->ld_work.as<char>() + 2 * a_keep);
    void *scan_tmp = h->ld_work.as<char>() + 2 * a_keep + 256;
    int32_t *newidx = h->ld_newidx.as<int32_t>();
    bool slow = false;
    int64_t nk = build ? 0 : h->lowdeg->n;
    if (build && h->low_rows >= 0 && h->low_rows_epoch == h->struct_epoch) {
        // the grouping that built this matrix counted its rows of two or three entries (an upper bound of what the
        // selection below can find): too few for a round to pay -- no launch, no round trip
        const int bar = h->ld_rounds == 0 && min_share > 64 ? 64 : min_share;
        if (h->low_rows * bar < n) {
            h->ld_state = 0;
            remember(1);
            return NODAL_OK;
        }
    }
    if (build) {
        h->ld_state = 0;
        NODAL_HIP_TRY(h, hipMemsetAsync(count, 0, 8, st));
        select_nodes<<<grid_for(n), TB, 0, st>>>(A, keep, count);
        NODAL_HIP_TRY(h, hipGetLastError());
        uint32_t nelim = 0;
        NODAL_TRY(nodal_read_words(h, &nelim, count, 4));
        // low-yield rounds are worth their ~0.3 ms while wires are being shortened (each round takes
        // a third to a half of every wire); a network that keeps yielding a trickle of candidates is cut off
        slow = (int64_t)nelim * 32 < n;
        nk = n - (int64_t)nelim;
        // (the first round needs a real share of such nodes -- 1/64: the four corners of every member of
        // a batch of small grids are 0.3 % and eliminating them only leaves graded links behind, which
        // the smoothed-aggregation hierarchy then declines; later rounds go on at n / min_share)
        const int bar = h->ld_rounds == 0 && min_share > 64 ? 64 : min_share;
        if ((int64_t)nelim * bar < n || nk < 1 || h->ld_rounds >= 48 ||
            (slow && h->ld_slow_rounds >= 12)) {
            remember(1);
            return NODAL_OK;
        }
        NODAL_HIP_TRY(h, hipMemsetAsync(keep + n, 0, 4, st));
        NODAL_TRY(scan_exclusive_u32(h, keep, pos, n + 1, nullptr, scan_tmp));
        finish_newidx<<<grid_for(n), TB, 0, st>>>(n, keep, pos, newidx);
        NODAL_HIP_TRY(h, hipGetLastError());
    }

    // the context of the reduced network: a matrix, a right-hand side and grounded flags
    if (!h->lowdeg) {
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
        c->keep_host_table = false;
        c->csr_only = true;
        c->passive_network = true;
        h->lowdeg = c;
    }
    nodal_ctx *c = h->lowdeg;
    c->dense_blockinv = h->dense_blockinv;
    c->gj_scalar = h->gj_scalar;
    c->use_graphs = h->use_graphs;
    c->amg_min_n = h->amg_min_n;
    c->have_x = false;
    c->have_numeric = false;
    int64_t nent = c->nnz;
    if (build) {
        c->ld_rounds = h->ld_rounds + 1;
        c->ld_slow_rounds = h->ld_slow_rounds + (slow ? 1 : 0);
        c->n = nk;
        c->K = (int32_t)nk;
        c->B = 0;
        ++c->struct_epoch;  // whatever the child had cached about its own matrix is void

        SchurEntries en;
        en.nitems = A.nnz;
        en.indptr = A.indptr;
        en.indices = A.indices;
        en.rowidx = A.rowidx;
        en.newidx = newidx;
        int64_t ncon = 0;
        // (grouping scratch in the child's work buffers: the parent's may hold live data of the caller)
        int64_t low = -1;
        const int bs = grp::build_lists(c, en, nk, &nent, &ncon, c->indices, c->rowidx, c->cptr, c->contrib,
                                        &c->indptr, &c->diag_pos, -1, -1, nullptr, &low);
        c->low_rows = low;  // (what the next round can take at most: it looks before it launches its selection)
        c->low_rows_epoch = c->struct_epoch;
        if (bs != NODAL_OK) {
            h->err = c->err;
            return bs;
        }
        c->nnz = nent;
        c->ncontrib = ncon;
        NODAL_HIP_TRY(h, c->data.reserve((size_t)nent * 8 + 8));
        NODAL_HIP_TRY(h, c->rhs.reserve((size_t)nk * 8 + 8));
        NODAL_HIP_TRY(h, c->x.reserve((size_t)nk * 8 + 8));
        NODAL_HIP_TRY(h, c->grounded.reserve((size_t)nk + 256));
        NODAL_HIP_TRY(h, h->grounded.reserve((size_t)n + 256));
    }
    if (nent > 0) {
        schur_values<<<grid_for(nent), TB, 0, st>>>(A, c->cptr.as<int32_t>(), c->contrib.as<uint32_t>(),
                                                   c->data.as<double>(), nent);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    uint8_t *flags = h->grounded.as<uint8_t>();
    if (!h->csr_only) NODAL_TRY(stamp_grounded_flags(h, flags));  // a child already holds its own
    reduce_rhs<<<grid_for(n), TB, 0, st>>>(A, newidx, h->rhs.as<double>(), flags, c->rhs.as<double>(),
                                          c->grounded.as<uint8_t>());
    NODAL_HIP_TRY(h, hipGetLastError());
    c->have_numeric = true;
    if (build) {
        find_floating_leftovers<<<grid_for(nk), TB, 0, st>>>(nk, c->indptr.as<int32_t>(),
                                                            c->grounded.as<uint8_t>(), count + 1);
        NODAL_HIP_TRY(h, hipGetLastError());
        uint32_t leftover = 0;
        NODAL_TRY(nodal_read_words(h, &leftover, count + 1, 4));
        remember(leftover ? 3 : 2);
        if (leftover) {
            if (trace) fprintf(stderr, "[lowdeg] a floating sub-network collapsed to a single node: singular\n");
            *done = true;
            *info = 1;
            *iters = 0;
            *resid = 0.0;
            return NODAL_OK;
        }
    }
    if (trace && build) {
        NODAL_WAIT_STREAM(h, st);
        fprintf(stderr, "[lowdeg] %lld -> %lld unknowns, %lld -> %lld entries (%.2f ms)\n", (long long)n,
                (long long)nk, (long long)A.nnz, (long long)nent,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }

    int32_t cinfo = 0;
    const int s = sparse_solve(c, NODAL_SPARSE_AUTO, &cinfo, iters, resid);
    if (s != NODAL_OK) {  // let the caller's own solver have a go at the unreduced system
        if (trace) fprintf(stderr, "[lowdeg] reduced solve failed (%s): falling back\n", c->err.c_str());
        return NODAL_OK;
    }
    h->amg_levels = c->amg_levels;
    h->kern_ms = c->kern_ms;
    h->kern_launches = c->kern_launches;
    h->kern_alg = c->kern_alg;
    *done = true;
    *info = cinfo;
    if (cinfo > 0) return NODAL_OK;  // singular (floating sub-network): the caller fills NaNs
    recover_x<<<grid_for(n), TB, 0, st>>>(A, newidx, h->rhs.as<double>(), c->x.as<double>(),
                                         h->x.as<double>());
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}
