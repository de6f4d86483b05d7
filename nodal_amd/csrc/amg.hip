// Aggregation multigrid preconditioner for the sparse SPD path (passive networks:
// G is a weighted graph Laplacian plus ground conductances, an M-matrix).
//
// Replaces -- together with the flexible CG driver below -- what
// scipy.sparse.linalg.spsolve does for the reference (nodal/nodal.py:325) on large
// resistor networks, where Jacobi-preconditioned CG needs ~5.5 sqrt(n) iterations.
//
// Setup (all on the device, deterministic):
//   * aggregation by pairwise matching (Notay's AGMG idea): each free node proposes
//     to its strongest free neighbour, (weight, symmetric edge hash) breaks ties the
//     same way from both ends, mutual proposals are matched; leftovers join the
//     aggregate of their strongest matched neighbour.  PASSES passes per level (the
//     graph of pairs is matched again) give aggregates of ~10 nodes; chain-like levels
//     (fewer than 4 entries per row) use two passes.
//   * coarse matrices A_c = P^T A P (P piecewise constant) are formed by the same
//     grouping pipeline as the stamping (group.h): tuples (agg[row], agg[col]) of the
//     fine entries are bucketed, sorted and summed in a fixed order.
//   * coarsening stops at <= COARSEST_MAX nodes; that matrix is inverted densely.
//   * level objects and scratch are kept across setups (a solve re-runs the setup for new
//     matrix values; re-allocating ~80 buffers cost more than the setup kernels).
// Cycle: unsmoothed aggregation needs the K-cycle for mesh-independent convergence
// (V-cycle iteration counts grow with the number of levels): on every coarse level of at
// least K_MIN_ROWS rows the coarse problem is solved by two steps of flexible CG
// preconditioned by the next level's cycle; smaller levels get a plain V hand-over down to
// the LDS-resident tail, which runs its own two inner steps.  Damped-Jacobi pre- and
// post-smoothing (two sweeps at level 0, one below), fused with the residual /
// prolongation so a level visit costs three kernels.  Networks in which many nodes have
// graded links (resistances over a decade or more, anisotropy) are set up in "contrast mode":
// Jacobi over the aggregates' diagonal blocks on every level above the tail (block_* kernels
// below), free nodes propose only over links of at least a tenth of their strongest one (so
// that aggregates do not cut strong links), two passes per level.
#include <atomic>

#include "group.h"
#include "spmv_stream.h"

namespace {

using grp::grid_for;
using grp::TB;

constexpr int PASSES = 3;
#ifndef NODAL_MATCH_ROUNDS
#define NODAL_MATCH_ROUNDS 5
#endif
constexpr int MATCH_ROUNDS = NODAL_MATCH_ROUNDS;  // propose / confirm rounds per matching pass
constexpr int COARSEST_MAX = 64;
constexpr int MAX_LEVELS = 16;
#ifndef NODAL_OMEGA
#define NODAL_OMEGA 0.85
#endif
constexpr double OMEGA = NODAL_OMEGA;  // damped-Jacobi weight (0.7 / 0.85 / 1.0 measured: see DESIGN.md)
constexpr int DOT_BLOCKS = 128;  // partial sums per dot product
constexpr int64_t K_MIN_ROWS = 32768;  // coarse levels at least this large get the K-cycle
constexpr int64_t SPLIT_PROLONG_MIN = 200000;  // levels this large prolong in a separate pass

struct Csr {
    int64_t n = 0, nnz = 0;
    const int32_t *indptr = nullptr, *indices = nullptr, *rowidx = nullptr;
    const double *data = nullptr;
};

struct Level {
    Csr A;
    DevBuf indptr, indices, rowidx, data, diag_pos, cptr, contrib;  // owned (levels >= 1)
    DevBuf dinv;
    int64_t nc = 0;       // size of the next level
    DevBuf agg;           // i32[n]: node -> aggregate
    DevBuf memptr, mem;   // aggregate -> member nodes (ascending)
    DevBuf binv, boff;  // block smoother: inverses of the aggregates' diagonal blocks (see block_*)
    bool block = false;
    DevBuf vec;           // work vectors, see V_* below
    DevBuf part;          // dot-product partials: 5 x DOT_BLOCKS
    double *v(int which) const { return vec.as<double>() + (int64_t)which * ((A.n + 31) & ~31ll); }
};
enum { V_X = 0, V_R = 1, V_RC = 2, V_C1 = 3, V_C2 = 4, V_V1 = 5, V_V2 = 6, V_R2 = 7, V_T = 8, V_COUNT = 9 };

// ---- LDS-resident tail --------------------------------------------------------
// All levels from `tail` down fit in one CU's LDS (160 KB): their matrices are packed
// once into an image (f64 values, u16 indices) that a single 1024-thread workgroup
// copies into LDS and then runs the whole coarse solve on -- the two flexible-CG
// steps at level `tail` with V-cycles below -- with workgroup barriers instead of
// ~130 kernel launches of 3-6 us each.
constexpr int TAIL_THREADS = 1024;
constexpr int TAIL_MAX_LEVELS = 8;
constexpr int TAIL_LDS_BUDGET = 150 * 1024;

struct TailLevel {
    int n, nnz, nc;
    int o_data, o_dinv, o_indptr, o_memptr, o_indices, o_agg, o_mem;  // image offsets (bytes)
    int o_B, o_X, o_E, o_R;                                            // vectors (bytes)
};
struct TailDesc {
    int nlev;  // levels in the tail, the last one is the dense coarsest level
    TailLevel lv[TAIL_MAX_LEVELS];
    int o_inv, o_C1, o_V1;
    int image_bytes, total_bytes;
};

struct Hierarchy {
    std::vector<Level *> levels;
    DevBuf coarse_inv;  // dense inverse of the last level
    bool coarse_direct = false;
    int tail = -1;      // first level handled by the LDS tail kernel (-1: none)
    int kmax = 1;       // K-cycle (two inner FCG steps) down to this coarse level, plain V hand-over below.
                        // Set per hierarchy: the last coarse level with >= K_MIN_ROWS rows (at least 1).
                        // 1e6 nodes: 1, 4e6: 2 (there 1 needs 108 iterations / 125 ms, 2: 52 / 75 ms).
                        // NODAL_AMG_KMAX overrides.
    int passes0 = PASSES, passes1 = PASSES;  // pairwise matching passes at level 0 / below
    bool passes_forced = false;              // NODAL_AMG_PASSES0/1 given: no per-level adaptation
    bool block_smoother = false;  // aggregate-block Jacobi on the levels above the LDS tail (chosen at setup)
    double theta = 0.0;           // matching: a free node proposes only over links >= theta x its strongest
    int sweeps0 = 2;    // Jacobi sweeps before / after the coarse correction at level 0 (NODAL_AMG_SWEEPS0=1: one)
    TailDesc tdesc;
    DevBuf tail_image;
    // Level objects and scratch buffers outlive a setup: a solve rebuilds the hierarchy for new
    // matrix values, and ~80 hipMalloc / hipFree pairs per setup cost more than the kernels.
    std::vector<Level *> pool;  // every Level ever created; `levels` is the active prefix
    Level scratch[2];           // intermediate pair graphs of the matching passes
    DevBuf pass_map, members_none, members_rows;
    ~Hierarchy() { clear(); }
    static void release_level(Level *l) {
        DevBuf *bufs[] = {&l->indptr, &l->indices, &l->rowidx, &l->data, &l->diag_pos, &l->cptr,
                          &l->contrib, &l->dinv, &l->agg, &l->memptr, &l->mem, &l->vec, &l->part,
                          &l->binv, &l->boff};
        for (DevBuf *b : bufs) b->release();
    }
    Level *take(size_t i) {  // the i-th level object, buffers kept from earlier setups
        while (pool.size() <= i) pool.push_back(new Level());
        pool[i]->nc = 0;
        pool[i]->block = false;
        return pool[i];
    }
    void begin_setup() {
        levels.clear();
        tail = -1;
        coarse_direct = false;
    }
    void clear() {
        for (Level *l : pool) {
            release_level(l);
            delete l;
        }
        pool.clear();
        levels.clear();
        for (Level &t : scratch) release_level(&t);
        pass_map.release();
        members_none.release();
        members_rows.release();
        coarse_inv.release();
        tail_image.release();
        tail = -1;
    }
};

// ---------------------------------------------------------------------------------
// setup kernels
// ---------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t edge_hash(uint32_t a, uint32_t b) {
    const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
    uint64_t h = lo * 0x9E3779B97F4A7C15ull + hi * 0xC2B2AE3D27D4EB4Full;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 32;
    return h;
}

// every free node proposes to its strongest free neighbour
__global__ __launch_bounds__(TB) void match_propose(Csr A, const int32_t *__restrict__ match,
                                                    int32_t *__restrict__ prop, double theta) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        int best = -1;
        if (match[i] < 0) {
            double bw = 0.0;
            uint64_t bh = 0;
            double wmax = 0.0;  // strongest link of the node, taken or not
            if (theta > 0.0)
                for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e)
                    if (A.indices[e] != (int)i) wmax = fmax(wmax, -A.data[e]);
            for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
                const int j = A.indices[e];
                const double w = -A.data[e];
                if (j == i || !(w > 0.0) || match[j] >= 0) continue;
                if (w < theta * wmax) continue;  // rather join the pair of the strong neighbour later
                const uint64_t hh = edge_hash((uint32_t)i, (uint32_t)j);
                if (w > bw || (w == bw && hh > bh)) { bw = w; bh = hh; best = j; }
            }
        }
        prop[i] = best;
    }
}

__global__ __launch_bounds__(TB) void match_confirm(int64_t n, const int32_t *__restrict__ prop,
                                                    int32_t *__restrict__ match) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const int j = prop[i];
        if (j >= 0 && prop[j] == (int)i) match[i] = j;
    }
}

// unmatched nodes join their strongest matched neighbour; leaders are flagged
__global__ __launch_bounds__(TB) void match_join(Csr A, const int32_t *__restrict__ match,
                                                 int32_t *__restrict__ join,
                                                 uint32_t *__restrict__ leader) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        int best = -1;
        if (match[i] < 0) {
            double bw = 0.0;
            uint64_t bh = 0;
            for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
                const int j = A.indices[e];
                const double w = -A.data[e];
                if (j == i || !(w > 0.0) || match[j] < 0) continue;
                const uint64_t hh = edge_hash((uint32_t)i, (uint32_t)j);
                if (w > bw || (w == bw && hh > bh)) { bw = w; bh = hh; best = j; }
            }
        }
        join[i] = best;
        const int m = match[i];
        leader[i] = (m >= 0 ? (int)i < m : best < 0) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(TB) void assign_aggregates(int64_t n, const int32_t *__restrict__ match,
                                                        const int32_t *__restrict__ join,
                                                        const uint32_t *__restrict__ lead_id,
                                                        int32_t *__restrict__ agg) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        int rep = (int)i;
        if (match[i] < 0 && join[i] >= 0) rep = join[i];  // joins a matched pair
        const int m = match[rep];
        if (m >= 0 && m < rep) rep = m;  // the pair's leader is its smaller index
        agg[i] = (int32_t)lead_id[rep];
    }
}

__global__ __launch_bounds__(TB) void compose_map(int64_t n, int32_t *__restrict__ agg,
                                                  const int32_t *__restrict__ next) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        agg[i] = next[agg[i]];
}

// grouping enumerators (group.h)
struct CoarseEntries {  // fine entry e -> (agg[row], agg[col])
    static constexpr int SLOTS = 1;
    static constexpr bool EXACT = true;  // one tuple per fine entry
    const int32_t *rowidx, *indices, *agg;
    int64_t nitems;
    template <class F>
    __device__ void for_each(int64_t e, F f) const {
        f(0, agg[rowidx[e]], agg[indices[e]]);
    }
};
struct Members {  // node i -> (agg[i], 0): one entry per aggregate, members ascending
    static constexpr int SLOTS = 1;
    static constexpr bool EXACT = true;  // one tuple per node
    const int32_t *agg;
    int64_t nitems;
    template <class F>
    __device__ void for_each(int64_t i, F f) const {
        f(0, agg[i], 0);
    }
};

__global__ __launch_bounds__(TB) void sum_groups(const int32_t *__restrict__ cptr,
                                                 const uint32_t *__restrict__ contrib,
                                                 const double *__restrict__ fine,
                                                 double *__restrict__ coarse, int64_t nent) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < nent; e += (int64_t)gridDim.x * TB) {
        double s = 0.0;
        for (int32_t p = cptr[e]; p < cptr[e + 1]; ++p) s += fine[contrib[p] >> 3];
        coarse[e] = s;
    }
}

__global__ __launch_bounds__(TB) void unpack_members(const uint32_t *__restrict__ contrib,
                                                     int32_t *__restrict__ mem, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        mem[i] = (int32_t)(contrib[i] >> 3);
}

__global__ __launch_bounds__(TB) void diag_inverse(Csr A, const int32_t *__restrict__ diag_pos,
                                                   double *__restrict__ dinv,
                                                   double *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        const int32_t e = diag_pos[i];
        const double d = e >= 0 ? A.data[e] : 0.0;
        if (!(d > 0.0)) *flag = 2.0;  // benign race: every writer stores the same value
        dinv[i] = d > 0.0 ? 1.0 / d : 1.0;
    }
}

// dense inverse of the coarsest matrix (n <= COARSEST_MAX) by Gauss-Jordan in LDS;
// SPD, so no pivoting.  A non-positive pivot (singular network) raises the flag -- the symmetric
// positive definite callers then give the hierarchy up -- and the elimination goes on with the row's
// own diagonal entry (1 if that is not positive either) in its place: the inverse of a perturbed matrix,
// every entry of `inv` written.  The general path (sparse_general.hip) ignores the flag: its node block
// may hold an island that only a branch row ties down (a singular diagonal block of a regular system),
// and its preconditioner must still be a finite operator.
__global__ __launch_bounds__(256) void coarsest_inverse(Csr A, double *__restrict__ inv,
                                                        double *__restrict__ flag) {
    __shared__ double M[COARSEST_MAX][2 * COARSEST_MAX + 1];
    __shared__ double dg[COARSEST_MAX];
    const int n = (int)A.n;
    for (int t = threadIdx.x; t < n * 2 * n; t += 256) {
        const int r = t / (2 * n), c = t % (2 * n);
        M[r][c] = (c >= n && c - n == r) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256)
        for (int32_t e = A.indptr[r]; e < A.indptr[r + 1]; ++e) M[r][A.indices[e]] = A.data[e];
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256) dg[r] = M[r][r];
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        double pv = M[k][k];  // uniform: every thread reads the same pivot
        if (!(pv > 0.0) && threadIdx.x == 0) *flag = 3.0;
        // (a pivot that cancelled down to rounding noise -- the last node of a floating island -- would put
        // 1e17 into the inverse: it is replaced like a non-positive one)
        if (!(pv > 1e-13 * dg[k]) || !(pv > 0.0)) pv = dg[k] > 0.0 ? dg[k] : 1.0;
        const double rp = 1.0 / pv;
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * n; c += 256) M[k][c] *= rp;
        __syncthreads();
        for (int t = threadIdx.x; t < n * 2 * n; t += 256) {
            const int r = t / (2 * n), c = t % (2 * n);
            if (r != k && c != k) M[r][c] = fma(-M[r][k], M[k][c], M[r][c]);
        }
        __syncthreads();
        for (int r = threadIdx.x; r < n; r += 256)
            if (r != k) M[r][k] = 0.0;
        __syncthreads();
    }
    for (int t = threadIdx.x; t < n * n; t += 256) inv[t] = M[t / n][n + t % n];
}

// ---------------------------------------------------------------------------------
// cycle kernels
// ---------------------------------------------------------------------------------

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum(double v) {
    __shared__ double ws[TB / 64];
    __syncthreads();
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w) s += ws[w];
    return s;
}

__device__ __forceinline__ double reduce_partials(const double *__restrict__ part, int count) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += TB) s += part[i];
    return block_sum(s);
}

// pre-smoothing from a zero guess, fused with the residual (CSR-stream):
//   x = w D^-1 b ;  r = b - A x          (x_j is recomputed from b_j on the fly)
__global__ __launch_bounds__(TB) void smooth_residual(Csr A, const double *__restrict__ dinv,
                                                      const double *__restrict__ b,
                                                      double *__restrict__ x,
                                                      double *__restrict__ r) {
    stream::for_rows(
        A.indptr, A.indices, A.data, A.n,
        [&](int32_t, int32_t col, double val) { return val * (OMEGA * dinv[col] * b[col]); },
        [&](int64_t i, double sum) {
            const double bi = b[i];
            x[i] = OMEGA * dinv[i] * bi;
            r[i] = bi - sum;
        });
}

// second pre-smoothing sweep (CSR-stream):  x2 = x + w D^-1 r ;  r2 = b - A x2
// (x2_j is recomputed from x_j, r_j on the fly; outputs go to separate vectors)
__global__ __launch_bounds__(TB) void smooth_again(Csr A, const double *__restrict__ dinv,
                                                   const double *__restrict__ b,
                                                   const double *__restrict__ x,
                                                   const double *__restrict__ r,
                                                   double *__restrict__ x2, double *__restrict__ r2) {
    stream::for_rows(
        A.indptr, A.indices, A.data, A.n,
        [&](int32_t, int32_t col, double val) { return val * fma(OMEGA * dinv[col], r[col], x[col]); },
        [&](int64_t i, double sum) {
            x2[i] = fma(OMEGA * dinv[i], r[i], x[i]);
            r2[i] = b[i] - sum;
        });
}

// ---- aggregate-block smoother ---------------------------------------------------------
// B = blockdiag(A restricted to each aggregate).  Point Jacobi cannot damp an error that is
// constant on a strongly coupled cluster INSIDE an aggregate (D is dominated by the strong
// links); B^-1 treats every aggregate exactly.  Binv is stored aggregate by aggregate at
// boff[I], m x m (m = members of I, in the order of the member list), TRANSPOSED: the general
// path's node block is not symmetric.
constexpr uint32_t BLOCK_MAX = 32;  // dense blocks up to this many nodes (what the register kernels build)
__global__ __launch_bounds__(TB) void block_sizes(int64_t nc, const int32_t *__restrict__ memptr,
                                                  uint32_t *__restrict__ sq) {
    for (int64_t I = (int64_t)blockIdx.x * TB + threadIdx.x; I <= nc; I += (int64_t)gridDim.x * TB) {
        const uint32_t m = I < nc ? (uint32_t)(memptr[I + 1] - memptr[I]) : 0u;
        sq[I] = m <= BLOCK_MAX ? m * m : m;  // larger aggregates: their diagonal only
    }
}

// Number of nodes whose links are graded: the strongest one carries more than `share` of the
// diagonal, or is more than `spread` times the weakest one (anisotropy: two strong links and two
// weak ones).  These are the networks on which point Jacobi and plain matching fail (see above).
__global__ __launch_bounds__(TB) void count_dominated(Csr A, double share, double spread,
                                                      uint32_t *__restrict__ count) {
    uint32_t mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        double d = 0.0, mx = 0.0, mn = 1e300;
        for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
            const double v = A.data[e];
            if (A.indices[e] == (int)i) d = v;
            else if (v != 0.0) {
                mx = fmax(mx, fabs(v));
                mn = fmin(mn, fabs(v));
            }
        }
        mine += (d > 0.0 && (mx > share * d || mx > spread * mn)) ? 1u : 0u;
    }
    __shared__ uint32_t total;
    if (threadIdx.x == 0) total = 0;
    __syncthreads();
    if (mine) atomicAdd(&total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && total) atomicAdd(count, total);
}

__global__ __launch_bounds__(TB) void block_stats(int64_t nc, const int32_t *__restrict__ memptr,
                                                  uint32_t *__restrict__ out) {  // [0] m>16, [1] m>32, [2] max m
    for (int64_t I = (int64_t)blockIdx.x * TB + threadIdx.x; I < nc; I += (int64_t)gridDim.x * TB) {
        const uint32_t m = (uint32_t)(memptr[I + 1] - memptr[I]);
        if (m > 16) atomicAdd(&out[0], 1u);
        if (m > 32) atomicAdd(&out[1], 1u);
        atomicMax(&out[2], m);
    }
}

// Blocks of up to W nodes (W = 16: ~95 % of them, W = 32 = BLOCK_MAX: the rest): W lanes per
// aggregate, lane r holds row r in registers, Gauss-Jordan without pivoting (SPD) with the pivot
// row broadcast by shuffles.  Handles the aggregates with LO < m <= W.
template <int W, int LO>
__global__ __launch_bounds__(TB) void block_build_reg(Csr A, int64_t nc, const int32_t *__restrict__ agg,
                                                      const int32_t *__restrict__ memptr,
                                                      const int32_t *__restrict__ mem,
                                                      const uint32_t *__restrict__ boff,
                                                      double *__restrict__ binv, double *__restrict__ flag) {
    const int r = threadIdx.x & (W - 1);
    constexpr int SH = W == 16 ? 4 : 5;
    const int64_t groups = ((int64_t)gridDim.x * TB) >> SH;
    const int64_t rounds = (nc + groups - 1) / groups;  // every lane runs the same number of rounds (shuffles)
    for (int64_t t = 0; t < rounds; ++t) {
        const int64_t I = t * groups + (((int64_t)blockIdx.x * TB + threadIdx.x) >> SH);
        const bool have = I < nc;
        const int32_t p0 = have ? memptr[I] : 0;
        const int m = have ? memptr[I + 1] - p0 : 0;
        const bool ours = m > LO && m <= W;
        if (!__any(ours)) continue;  // wave-uniform: nothing for this wave in this round
        double row[W];
#pragma unroll
        for (int c = 0; c < W; ++c) row[c] = 0.0;
        const bool mine = ours && r < m;
        if (mine) {
            const int i = mem[p0 + r];
            for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
                const int j = A.indices[e];
                if (agg[j] != (int32_t)I) continue;
                int c = 0;
                while (mem[p0 + c] != j) ++c;
                const double v = A.data[e];
#pragma unroll
                for (int cc = 0; cc < W; ++cc) row[cc] = cc == c ? v : row[cc];
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) {
            const bool step = ours && k < m;   // uniform over the W lanes of an aggregate
            const double d = __shfl(row[k], k, W);
            if (step && r == k && !(d > 0.0)) *flag = 1.0;
            const double piv = 1.0 / d;
            if (step && r == k) {
                row[k] = 1.0;
#pragma unroll
                for (int c = 0; c < W; ++c) row[c] *= piv;
            }
            const double f = row[k];
            double pk[W];
#pragma unroll
            for (int c = 0; c < W; ++c) pk[c] = __shfl(row[c], k, W);
            if (step && mine && r != k) {
                row[k] = 0.0;
#pragma unroll
                for (int c = 0; c < W; ++c) row[c] = fma(-f, pk[c], row[c]);
            }
        }
        if (mine) {  // stored transposed (column r of the image = row r of the inverse): see block_apply
            double *B = binv + boff[I] + r;
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (c < m) B[(int64_t)c * m] = row[c];
        }
    }
}

// Aggregates above BLOCK_MAX nodes (a hub that many leftovers joined) keep point Jacobi: only
// the reciprocals of their diagonal entries are stored (a dense inverse would cost m^3).
__global__ __launch_bounds__(TB) void block_build_diag(Csr A, int64_t n, const int32_t *__restrict__ agg,
                                                       const int32_t *__restrict__ memptr,
                                                       const int32_t *__restrict__ mem,
                                                       const uint32_t *__restrict__ boff,
                                                       double *__restrict__ binv, double *__restrict__ flag) {
    for (int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x; p < n; p += (int64_t)gridDim.x * TB) {
        const int i = mem[p];
        const int I = agg[i];
        const int32_t p0 = memptr[I];
        if ((uint32_t)(memptr[I + 1] - p0) <= BLOCK_MAX) continue;
        double d = 0.0;
        for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e)
            if (A.indices[e] == i) d = A.data[e];
        if (!(d > 0.0)) *flag = 1.0;
        binv[boff[I] + (p - p0)] = 1.0 / d;
    }
}

// out = base + w * Binv v   (base == nullptr: out = w * Binv v).  One thread per member
// position: the image holds the inverse TRANSPOSED, so the lanes of an aggregate (row r of
// the inverse each) read consecutive entries, and all of them the same entries of v.
__global__ __launch_bounds__(TB) void block_apply(int64_t n, const int32_t *__restrict__ agg,
                                                  const int32_t *__restrict__ memptr,
                                                  const int32_t *__restrict__ mem,
                                                  const uint32_t *__restrict__ boff,
                                                  const double *__restrict__ binv,
                                                  const double *__restrict__ v,
                                                  const double *__restrict__ base,
                                                  double *__restrict__ out, double w) {
    for (int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x; p < n; p += (int64_t)gridDim.x * TB) {
        const int i = mem[p];
        const int I = agg[i];
        const int32_t p0 = memptr[I];
        const int m = memptr[I + 1] - p0;
        const double *col = binv + boff[I] + (p - p0);
        double s = 0.0;
        if ((uint32_t)m > BLOCK_MAX) s = col[0] * v[i];  // diagonal only
        else
            for (int c = 0; c < m; ++c) s = fma(col[(int64_t)c * m], v[mem[p0 + c]], s);
        out[i] = base ? fma(w, s, base[i]) : w * s;
    }
}

// r = b - A x  (CSR-stream)
__global__ __launch_bounds__(TB) void residual_vec(Csr A, const double *__restrict__ b,
                                                   const double *__restrict__ x,
                                                   double *__restrict__ r) {
    stream::for_rows(
        A.indptr, A.indices, A.data, A.n,
        [&](int32_t, int32_t col, double val) { return val * x[col]; },
        [&](int64_t i, double sum) { r[i] = b[i] - sum; });
}

// rc[I] = sum of r over the members of aggregate I (fixed order).  CSR-stream over the member
// lists: a hub's aggregate can have tens of thousands of members (every leftover spoke joins
// it), which one thread per aggregate would walk alone.
__global__ __launch_bounds__(TB) void restrict_sum(int64_t nc, const int32_t *__restrict__ memptr,
                                                   const int32_t *__restrict__ mem,
                                                   const double *__restrict__ r,
                                                   double *__restrict__ rc) {
    stream::for_rows(
        memptr, mem, static_cast<const double *>(nullptr), nc,
        [&](int32_t, int32_t node, double) { return r[node]; },
        [&](int64_t I, double sum) { rc[I] = sum; });
}

// K-cycle coefficients from the five dot products of the two inner FCG steps
struct KCoef { double s1, s2; };
__device__ __forceinline__ KCoef kcycle_coefficients(const double *__restrict__ part, int nparts) {
    if (nparts == 0) return KCoef{1.0, 0.0};  // plain V-cycle hand-over
    const double rho1 = reduce_partials(part + 0 * DOT_BLOCKS, nparts);
    const double alpha1 = reduce_partials(part + 1 * DOT_BLOCKS, nparts);
    const double gamma = reduce_partials(part + 2 * DOT_BLOCKS, nparts);
    const double beta = reduce_partials(part + 3 * DOT_BLOCKS, nparts);
    const double alpha2 = reduce_partials(part + 4 * DOT_BLOCKS, nparts);
    if (!(rho1 > 0.0)) return KCoef{0.0, 0.0};
    const double rho2 = beta - gamma * gamma / rho1;
    if (!(rho2 > 0.0)) return KCoef{alpha1 / rho1, 0.0};
    return KCoef{alpha1 / rho1 - gamma * alpha2 / (rho1 * rho2), alpha2 / rho2};
}

// coarse correction + post-smoothing, fused (CSR-stream; small and medium levels):
//   x' = x + P (s1 c1 + s2 c2) ;  out = x' + w D^-1 (b - A x')
__global__ __launch_bounds__(TB) void prolong_smooth(Csr A, const double *__restrict__ dinv,
                                                     const double *__restrict__ b,
                                                     const double *__restrict__ x,
                                                     const int32_t *__restrict__ agg,
                                                     const double *__restrict__ c1,
                                                     const double *__restrict__ c2,
                                                     const double *__restrict__ part, int nparts,
                                                     double *__restrict__ out) {
    const KCoef k = kcycle_coefficients(part, nparts);
    const bool two = nparts != 0;
    auto corrected = [&](int64_t j) {
        const int J = agg[j];
        return x[j] + k.s1 * c1[J] + (two ? k.s2 * c2[J] : 0.0);
    };
    stream::for_rows(
        A.indptr, A.indices, A.data, A.n,
        [&](int32_t, int32_t col, double val) { return val * corrected(col); },
        [&](int64_t i, double sum) { out[i] = fma(OMEGA * dinv[i], b[i] - sum, corrected(i)); });
}

// coarse correction fused with the residual (CSR-stream; small and medium levels, block mode):
//   xp = x + P (s1 c1 + s2 c2) ;  r = b - A xp      (xp_j is recomputed per entry)
__global__ __launch_bounds__(TB) void prolong_residual(Csr A, const double *__restrict__ b,
                                                       const double *__restrict__ x,
                                                       const int32_t *__restrict__ agg,
                                                       const double *__restrict__ c1,
                                                       const double *__restrict__ c2,
                                                       const double *__restrict__ part, int nparts,
                                                       double *__restrict__ xp, double *__restrict__ r) {
    const KCoef k = kcycle_coefficients(part, nparts);
    const bool two = nparts != 0;
    auto corrected = [&](int64_t j) {
        const int J = agg[j];
        return x[j] + k.s1 * c1[J] + (two ? k.s2 * c2[J] : 0.0);
    };
    stream::for_rows(
        A.indptr, A.indices, A.data, A.n,
        [&](int32_t, int32_t col, double val) { return val * corrected(col); },
        [&](int64_t i, double sum) {
            xp[i] = corrected(i);
            r[i] = b[i] - sum;
        });
}

// The same in two launches for large levels, where gathering agg / c1 / c2 once per
// ENTRY costs more HBM traffic than one extra pass over the vectors:
//   (1) xp = x + P (s1 c1 + s2 c2)        (2) out = xp + w D^-1 (b - A xp)
__global__ __launch_bounds__(TB) void prolong_add(int64_t n, const double *__restrict__ x,
                                                  const int32_t *__restrict__ agg,
                                                  const double *__restrict__ c1,
                                                  const double *__restrict__ c2,
                                                  const double *__restrict__ part, int nparts,
                                                  double *__restrict__ xp) {
    const KCoef k = kcycle_coefficients(part, nparts);
    const bool two = nparts != 0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const int I = agg[i];
        xp[i] = x[i] + k.s1 * c1[I] + (two ? k.s2 * c2[I] : 0.0);
    }
}
__global__ __launch_bounds__(TB) void post_smooth(Csr A, const double *__restrict__ dinv,
                                                  const double *__restrict__ b,
                                                  const double *__restrict__ xp,
                                                  double *__restrict__ out) {
    stream::for_rows(
        A.indptr, A.indices, A.data, A.n,
        [&](int32_t, int32_t col, double val) { return val * xp[col]; },
        [&](int64_t i, double sum) { out[i] = fma(OMEGA * dinv[i], b[i] - sum, xp[i]); });
}

// v = A c and the partial dot products c.v, c.u1 (and c.u2 when given)  (CSR-stream)
__global__ __launch_bounds__(TB) void spmv_dots(Csr A, const double *__restrict__ c,
                                                double *__restrict__ v,
                                                const double *__restrict__ u1,
                                                const double *__restrict__ u2,
                                                double *__restrict__ p_cv,
                                                double *__restrict__ p_cu1,
                                                double *__restrict__ p_cu2) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    stream::for_rows(
        A.indptr, A.indices, A.data, A.n,
        [&](int32_t, int32_t col, double val) { return val * c[col]; },
        [&](int64_t i, double sum) {
            v[i] = sum;
            const double ci = c[i];
            a0 = fma(ci, sum, a0);
            a1 = fma(ci, u1[i], a1);
            if (u2) a2 = fma(ci, u2[i], a2);
        });
    a0 = block_sum(a0);
    a1 = block_sum(a1);
    a2 = block_sum(a2);
    if (threadIdx.x == 0) {
        p_cv[blockIdx.x] = a0;
        p_cu1[blockIdx.x] = a1;
        if (p_cu2) p_cu2[blockIdx.x] = a2;
    }
}

// r2 = rc - (alpha1 / rho1) v1
__global__ __launch_bounds__(TB) void second_residual(int64_t n, const double *__restrict__ rc,
                                                      const double *__restrict__ v1,
                                                      const double *__restrict__ part, int nparts,
                                                      double *__restrict__ r2) {
    const double rho1 = reduce_partials(part + 0 * DOT_BLOCKS, nparts);
    const double alpha1 = reduce_partials(part + 1 * DOT_BLOCKS, nparts);
    const double t = rho1 > 0.0 ? alpha1 / rho1 : 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        r2[i] = fma(-t, v1[i], rc[i]);
}

__global__ __launch_bounds__(TB) void dense_apply(int n, const double *__restrict__ inv,
                                                  const double *__restrict__ b,
                                                  double *__restrict__ out) {
    for (int i = threadIdx.x; i < n; i += TB) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s = fma(inv[i * n + j], b[j], s);
        out[i] = s;
    }
}

__global__ __launch_bounds__(TB) void jacobi_apply(int64_t n, const double *__restrict__ dinv,
                                                   const double *__restrict__ b,
                                                   double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        out[i] = dinv[i] * b[i];
}


// ---- tail kernels ----------------------------------------------------------------

__global__ __launch_bounds__(TB) void pack_f64(char *img, int off, const double *src, int64_t n) {
    double *dst = reinterpret_cast<double *>(img + off);
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        dst[i] = src[i];
}
__global__ __launch_bounds__(TB) void pack_i32(char *img, int off, const int32_t *src, int64_t n) {
    int32_t *dst = reinterpret_cast<int32_t *>(img + off);
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        dst[i] = src[i];
}
__global__ __launch_bounds__(TB) void pack_u16(char *img, int off, const int32_t *src, int64_t n) {
    uint16_t *dst = reinterpret_cast<uint16_t *>(img + off);
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        dst[i] = (uint16_t)src[i];
}

struct TailView {  // one level, pointers into LDS
    int n, nnz, nc;
    const double *data, *dinv;
    const int32_t *indptr, *memptr;
    const uint16_t *indices, *agg, *mem;
    double *B, *X, *E, *R;
};

__device__ __forceinline__ TailView tail_view(char *smem, const TailLevel &t) {
    TailView v;
    v.n = t.n; v.nnz = t.nnz; v.nc = t.nc;
    v.data = reinterpret_cast<const double *>(smem + t.o_data);
    v.dinv = reinterpret_cast<const double *>(smem + t.o_dinv);
    v.indptr = reinterpret_cast<const int32_t *>(smem + t.o_indptr);
    v.memptr = reinterpret_cast<const int32_t *>(smem + t.o_memptr);
    v.indices = reinterpret_cast<const uint16_t *>(smem + t.o_indices);
    v.agg = reinterpret_cast<const uint16_t *>(smem + t.o_agg);
    v.mem = reinterpret_cast<const uint16_t *>(smem + t.o_mem);
    v.B = reinterpret_cast<double *>(smem + t.o_B);
    v.X = reinterpret_cast<double *>(smem + t.o_X);
    v.E = reinterpret_cast<double *>(smem + t.o_E);
    v.R = reinterpret_cast<double *>(smem + t.o_R);
    return v;
}

// sums of three per-thread values over the 1024-thread workgroup, valid everywhere
__device__ __forceinline__ void tail_reduce3(double &a, double &b, double &c, double *scratch) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        scratch[wave] = a;
        scratch[16 + wave] = b;
        scratch[32 + wave] = c;
    }
    __syncthreads();
    double sa = 0.0, sb = 0.0, sc = 0.0;
#pragma unroll
    for (int w = 0; w < TAIL_THREADS / 64; ++w) {
        sa += scratch[w];
        sb += scratch[16 + w];
        sc += scratch[32 + w];
    }
    a = sa; b = sb; c = sc;
}

// V-cycle over the tail levels [0, nlev): rhs in lv[0].B, result in lv[0].E
__device__ void tail_vcycle(char *smem, const TailDesc &d) {
    const int last = d.nlev - 1;
    for (int l = 0; l < last; ++l) {
        const TailView v = tail_view(smem, d.lv[l]);
        for (int i = threadIdx.x; i < v.n; i += TAIL_THREADS) {
            double s = v.B[i];
            for (int e = v.indptr[i]; e < v.indptr[i + 1]; ++e) {
                const int j = v.indices[e];
                s = fma(-v.data[e], OMEGA * v.dinv[j] * v.B[j], s);
            }
            v.X[i] = OMEGA * v.dinv[i] * v.B[i];
            v.R[i] = s;
        }
        __syncthreads();
        double *Bc = reinterpret_cast<double *>(smem + d.lv[l + 1].o_B);
        for (int I = threadIdx.x; I < v.nc; I += TAIL_THREADS) {
            double s = 0.0;
            for (int p = v.memptr[I]; p < v.memptr[I + 1]; ++p) s += v.R[v.mem[p]];
            Bc[I] = s;
        }
        __syncthreads();
    }
    {   // dense coarsest solve
        const TailLevel &t = d.lv[last];
        const double *inv = reinterpret_cast<const double *>(smem + d.o_inv);
        const double *B = reinterpret_cast<const double *>(smem + t.o_B);
        double *E = reinterpret_cast<double *>(smem + t.o_E);
        for (int i = threadIdx.x; i < t.n; i += TAIL_THREADS) {
            double s = 0.0;
            for (int j = 0; j < t.n; ++j) s = fma(inv[i * t.n + j], B[j], s);
            E[i] = s;
        }
        __syncthreads();
    }
    for (int l = last - 1; l >= 0; --l) {
        const TailView v = tail_view(smem, d.lv[l]);
        const double *Ec = reinterpret_cast<const double *>(smem + d.lv[l + 1].o_E);
        for (int i = threadIdx.x; i < v.n; i += TAIL_THREADS) {
            double s = v.B[i];
            for (int e = v.indptr[i]; e < v.indptr[i + 1]; ++e) {
                const int j = v.indices[e];
                s = fma(-v.data[e], v.X[j] + Ec[v.agg[j]], s);
            }
            v.E[i] = fma(OMEGA * v.dinv[i], s, v.X[i] + Ec[v.agg[i]]);
        }
        __syncthreads();
    }
}

// Coarse solve at level `tail` for the K-cycle of the level above: two flexible-CG
// steps preconditioned by the tail V-cycle.  Outputs c1, c2 (global) and the five
// dot products (one partial each) that prolong_smooth turns into s1, s2.
__global__ __launch_bounds__(TAIL_THREADS) void amg_tail_kernel(TailDesc d, const char *image,
                                                                const double *__restrict__ rc,
                                                                double *__restrict__ c1_out,
                                                                double *__restrict__ c2_out,
                                                                double *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[48];
    // image -> LDS (16-byte pieces)
    {
        const int4 *src = reinterpret_cast<const int4 *>(image);
        int4 *dst = reinterpret_cast<int4 *>(smem);
        for (int i = threadIdx.x; i < d.image_bytes / 16; i += TAIL_THREADS) dst[i] = src[i];
    }
    const TailView top = tail_view(smem, d.lv[0]);
    double *C1 = reinterpret_cast<double *>(smem + d.o_C1);
    double *V1 = reinterpret_cast<double *>(smem + d.o_V1);
    for (int i = threadIdx.x; i < top.n; i += TAIL_THREADS) top.B[i] = rc[i];
    __syncthreads();

    tail_vcycle(smem, d);  // c1 in top.E
    double rho1 = 0.0, alpha1 = 0.0, zero = 0.0;
    for (int i = threadIdx.x; i < top.n; i += TAIL_THREADS) {
        double s = 0.0;
        for (int e = top.indptr[i]; e < top.indptr[i + 1]; ++e)
            s = fma(top.data[e], top.E[top.indices[e]], s);
        const double ci = top.E[i];
        V1[i] = s;
        C1[i] = ci;
        rho1 = fma(ci, s, rho1);
        alpha1 = fma(ci, top.B[i], alpha1);
    }
    tail_reduce3(rho1, alpha1, zero, red);
    const double t1 = rho1 > 0.0 ? alpha1 / rho1 : 0.0;
    for (int i = threadIdx.x; i < top.n; i += TAIL_THREADS) top.B[i] = fma(-t1, V1[i], top.B[i]);
    __syncthreads();

    tail_vcycle(smem, d);  // c2 in top.E, r2 in top.B
    double gamma = 0.0, beta = 0.0, alpha2 = 0.0;
    for (int i = threadIdx.x; i < top.n; i += TAIL_THREADS) {
        double s = 0.0;
        for (int e = top.indptr[i]; e < top.indptr[i + 1]; ++e)
            s = fma(top.data[e], top.E[top.indices[e]], s);
        const double ci = top.E[i];
        gamma = fma(ci, V1[i], gamma);
        beta = fma(ci, s, beta);
        alpha2 = fma(ci, top.B[i], alpha2);
        c1_out[i] = C1[i];
        c2_out[i] = ci;
    }
    tail_reduce3(gamma, beta, alpha2, red);
    if (threadIdx.x == 0) {
        part[0 * DOT_BLOCKS] = rho1;
        part[1 * DOT_BLOCKS] = alpha1;
        part[2 * DOT_BLOCKS] = gamma;
        part[3 * DOT_BLOCKS] = beta;
        part[4 * DOT_BLOCKS] = alpha2;
    }
}


// ---- structural singularity check -----------------------------------------------------
// A passive network is singular iff some connected component has no resistor to
// ground.  Aggregation never merges across components and the Galerkin product keeps
// an edge between two aggregates iff a fine edge joins them, so the LAST level has the
// same components as the netlist; "touches ground" flags are OR-ed up the hierarchy.

__global__ __launch_bounds__(TB) void flags_restrict(int64_t nc, const int32_t *__restrict__ memptr,
                                                     const int32_t *__restrict__ mem,
                                                     const uint8_t *__restrict__ fine,
                                                     uint8_t *__restrict__ coarse) {
    for (int64_t I = (int64_t)blockIdx.x * TB + threadIdx.x; I < nc; I += (int64_t)gridDim.x * TB) {
        uint8_t f = 0;
        for (int32_t p = memptr[I]; p < memptr[I + 1]; ++p) f |= fine[mem[p]];
        coarse[I] = f;
    }
}

__global__ __launch_bounds__(TB) void cc_init(int64_t n, int32_t *__restrict__ label) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        label[i] = (int32_t)i;
}
// hook: the root of the larger label is attached to the smaller label
__global__ __launch_bounds__(TB) void cc_hook(Csr A, int32_t *__restrict__ label,
                                              int32_t *__restrict__ changed) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        const int li = label[i];
        for (int32_t e = A.indptr[i]; e < A.indptr[i + 1]; ++e) {
            if (A.data[e] == 0.0) continue;
            const int lj = label[A.indices[e]];
            if (lj < li) {
                atomicMin(&label[li], lj);
                *changed = 1;
            }
        }
    }
}
__global__ __launch_bounds__(TB) void cc_jump(int64_t n, int32_t *__restrict__ label) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        int l = label[i];
        while (label[l] != l) l = label[l];
        label[i] = l;
    }
}
// per component root: does it hold a grounded node?
__global__ __launch_bounds__(TB) void cc_mark(int64_t n, const int32_t *__restrict__ label,
                                              const uint8_t *__restrict__ grounded,
                                              int32_t *__restrict__ root_ok) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        if (grounded[i]) root_ok[label[i]] = 1;
}
__global__ __launch_bounds__(TB) void cc_verdict(int64_t n, const int32_t *__restrict__ label,
                                                 const int32_t *__restrict__ root_ok,
                                                 int32_t *__restrict__ floating) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        if (label[i] == (int32_t)i && !root_ok[i]) *floating = 1;
}

unsigned dot_grid(int64_t n) { return stream::grid_for_rows(n, DOT_BLOCKS); }

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------

// one pairwise-matching pass on graph A: agg (i32[n]) and the number of aggregates
int matching_pass(nodal_ctx *h, const Csr &A, double match_theta, int32_t *agg, int64_t *nagg) {
    hipStream_t st = h->stream;
    const int64_t n = A.n;
    const size_t a4 = ((size_t)n * 4 + 255) & ~(size_t)255;
    // work layout: match | prop/join | leader/lead_id (n+1) | scan tmp
    NODAL_HIP_TRY(h, h->work.reserve(3 * a4 + 512 + scan_tmp_bytes(n + 1)));
    char *w = h->work.as<char>();
    int32_t *match = reinterpret_cast<int32_t *>(w);
    int32_t *prop = reinterpret_cast<int32_t *>(w + a4);
    uint32_t *lead = reinterpret_cast<uint32_t *>(w + 2 * a4);
    void *scan_tmp = w + 3 * a4 + 512;
    NODAL_HIP_TRY(h, hipMemsetAsync(match, 0xff, (size_t)n * 4, st));
    for (int r = 0; r < MATCH_ROUNDS; ++r) {
        match_propose<<<grid_for(n), TB, 0, st>>>(A, match, prop, match_theta);
        match_confirm<<<grid_for(n), TB, 0, st>>>(n, prop, match);
    }
    match_join<<<grid_for(n), TB, 0, st>>>(A, match, prop, lead);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipMemsetAsync(lead + n, 0, 4, st));
    NODAL_TRY(scan_exclusive_u32(h, lead, lead, n + 1, nullptr, scan_tmp));
    assign_aggregates<<<grid_for(n), TB, 0, st>>>(n, match, prop, lead, agg);
    NODAL_HIP_TRY(h, hipGetLastError());
    uint32_t total = 0;
    NODAL_HIP_TRY(h, hipMemcpyAsync(&total, lead + n, 4, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    *nagg = total;
    return NODAL_OK;
}

// A_c = P^T A P into the owned buffers of `out`
int galerkin(nodal_ctx *h, const Csr &A, const int32_t *agg, int64_t nc, Level *out) {
    CoarseEntries en{A.rowidx, A.indices, agg, A.nnz};
    int64_t nent = 0, ncon = 0;
    NODAL_TRY(grp::build_lists(h, en, nc, &nent, &ncon, out->indices, out->rowidx, out->cptr,
                               out->contrib, &out->indptr, &out->diag_pos));
    NODAL_HIP_TRY(h, out->data.reserve((size_t)nent * 8 + 8));
    if (nent > 0) {
        sum_groups<<<grid_for(nent), TB, 0, h->stream>>>(out->cptr.as<int32_t>(),
                                                        out->contrib.as<uint32_t>(), A.data,
                                                        out->data.as<double>(), nent);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    out->A.n = nc;
    out->A.nnz = nent;
    out->A.indptr = out->indptr.as<int32_t>();
    out->A.indices = out->indices.as<int32_t>();
    out->A.rowidx = out->rowidx.as<int32_t>();
    out->A.data = out->data.as<double>();
    return NODAL_OK;
}

// inverses of the aggregates' diagonal blocks of level L (block smoother)
int build_blocks(nodal_ctx *h, Level *L, double *flag) {
    hipStream_t st = h->stream;
    const int64_t nc = L->nc;
    NODAL_HIP_TRY(h, L->boff.reserve((size_t)(nc + 1) * 4 + 64));
    NODAL_HIP_TRY(h, h->work.reserve(scan_tmp_bytes(nc + 1) + 256));
    uint32_t *boff = L->boff.as<uint32_t>();
    uint32_t *total_dev = reinterpret_cast<uint32_t *>(h->work.as<char>());
    block_sizes<<<grid_for(nc + 1), TB, 0, st>>>(nc, L->memptr.as<int32_t>(), boff);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_TRY(scan_exclusive_u32(h, boff, boff, nc + 1, total_dev, h->work.as<char>() + 256));
    uint32_t total = 0;
    NODAL_HIP_TRY(h, hipMemcpyAsync(&total, total_dev, 4, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    NODAL_HIP_TRY(h, L->binv.reserve((size_t)total * 8 + 64));
    block_build_reg<16, 0><<<grid_for(nc * 16), TB, 0, st>>>(L->A, nc, L->agg.as<int32_t>(),
                                                            L->memptr.as<int32_t>(), L->mem.as<int32_t>(), boff,
                                                            L->binv.as<double>(), flag);
    block_build_reg<32, 16><<<grid_for(nc * 32), TB, 0, st>>>(L->A, nc, L->agg.as<int32_t>(),
                                                             L->memptr.as<int32_t>(), L->mem.as<int32_t>(), boff,
                                                             L->binv.as<double>(), flag);
    block_build_diag<<<grid_for(L->A.n), TB, 0, st>>>(L->A, L->A.n, L->agg.as<int32_t>(),
                                                     L->memptr.as<int32_t>(), L->mem.as<int32_t>(), boff,
                                                     L->binv.as<double>(), flag);
    NODAL_HIP_TRY(h, hipGetLastError());
    L->block = true;
    if (getenv("NODAL_TRACE")) {
        uint32_t *st3 = reinterpret_cast<uint32_t *>(h->work.as<char>());
        NODAL_HIP_TRY(h, hipMemsetAsync(st3, 0, 12, st));
        block_stats<<<grid_for(nc), TB, 0, st>>>(nc, L->memptr.as<int32_t>(), st3);
        uint32_t hs3[3];
        NODAL_HIP_TRY(h, hipMemcpyAsync(hs3, st3, 12, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        fprintf(stderr, "[amg] blocks: %lld aggregates, %u larger than 16, %u larger than 32, largest %u, %u doubles\n",
                (long long)nc, hs3[0], hs3[1], hs3[2], total);
    }
    return NODAL_OK;
}

int finish_level(nodal_ctx *h, Level *l, const int32_t *diag_pos, double *flag) {
    const int64_t n = l->A.n;
    NODAL_HIP_TRY(h, l->dinv.reserve((size_t)n * 8 + 8));
    diag_inverse<<<grid_for(n), TB, 0, h->stream>>>(l->A, diag_pos, l->dinv.as<double>(), flag);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, l->vec.reserve((size_t)V_COUNT * (((size_t)n + 31) & ~(size_t)31) * 8 + 256));
    NODAL_HIP_TRY(h, l->part.reserve(5 * DOT_BLOCKS * 8));
    return NODAL_OK;
}


__global__ __launch_bounds__(TB) void max_row_length(int64_t n, const int32_t *__restrict__ indptr,
                                                     int32_t *__restrict__ out) {
    int32_t m = 0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const int32_t len = indptr[i + 1] - indptr[i];
        m = len > m ? len : m;
    }
    if (m > 0) atomicMax(out, m);
}

// Choose the first level from which everything fits in LDS and pack its image.
int build_tail(nodal_ctx *h, Hierarchy *H) {
    const int L = (int)H->levels.size() - 1;
    H->tail = -1;
    auto up16 = [](int x) { return (x + 15) & ~15; };
    // The tail kernel gives every row to one thread: a hub row of hundreds of entries makes that
    // thread the whole kernel (406 instead of 25 us with a 5000-spoke hub on level 0).  Levels with
    // such rows stay outside the tail, where the CSR-stream kernels sum long rows cooperatively.
    constexpr int32_t TAIL_MAX_ROW = 128;
    int32_t longest[MAX_LEVELS] = {0};
    {
        NODAL_HIP_TRY(h, h->work.reserve(MAX_LEVELS * 4 + 64));
        int32_t *dev = h->work.as<int32_t>();
        NODAL_HIP_TRY(h, hipMemsetAsync(dev, 0, MAX_LEVELS * 4, h->stream));
        for (int t = 1; t < L; ++t)
            if (H->levels[t]->A.n <= 65535)
                max_row_length<<<grid_for(H->levels[t]->A.n), TB, 0, h->stream>>>(
                    H->levels[t]->A.n, H->levels[t]->A.indptr, dev + t);
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_HIP_TRY(h, hipMemcpyAsync(longest, dev, MAX_LEVELS * 4, hipMemcpyDeviceToHost, h->stream));
        NODAL_WAIT_STREAM(h, h->stream);
    }
    for (int t = 1; t < L; ++t) {  // level 0 and the coarsest alone never form a tail
        if (L - t + 1 > TAIL_MAX_LEVELS) continue;
        if (H->levels[t]->A.n > 65535 || H->levels[t]->A.nnz > (1 << 22)) continue;
        bool hub = false;
        for (int k = t; k < L; ++k) hub = hub || longest[k] > TAIL_MAX_ROW;
        if (hub) continue;
        TailDesc d;
        memset(&d, 0, sizeof d);
        d.nlev = L - t + 1;
        int off = 0;
        for (int k = 0; k < d.nlev; ++k) {
            const Level *lv = H->levels[t + k];
            TailLevel &tl = d.lv[k];
            tl.n = (int)lv->A.n;
            tl.nnz = (int)lv->A.nnz;
            tl.nc = (int)lv->nc;
            if (k == d.nlev - 1) continue;  // the coarsest level only needs vectors + inverse
            tl.o_data = off; off = up16(off + tl.nnz * 8);
            tl.o_dinv = off; off = up16(off + tl.n * 8);
            tl.o_indptr = off; off = up16(off + (tl.n + 1) * 4);
            tl.o_memptr = off; off = up16(off + (tl.nc + 1) * 4);
            tl.o_indices = off; off = up16(off + tl.nnz * 2);
            tl.o_agg = off; off = up16(off + tl.n * 2);
            tl.o_mem = off; off = up16(off + tl.n * 2);
        }
        const int nco = d.lv[d.nlev - 1].n;
        d.o_inv = off; off = up16(off + nco * nco * 8);
        d.image_bytes = off;
        for (int k = 0; k < d.nlev; ++k) {
            TailLevel &tl = d.lv[k];
            tl.o_B = off; off = up16(off + tl.n * 8);
            tl.o_X = off; off = up16(off + tl.n * 8);
            tl.o_E = off; off = up16(off + tl.n * 8);
            tl.o_R = off; off = up16(off + tl.n * 8);
        }
        d.o_C1 = off; off = up16(off + d.lv[0].n * 8);
        d.o_V1 = off; off = up16(off + d.lv[0].n * 8);
        d.total_bytes = off;
        if (off > TAIL_LDS_BUDGET) continue;
        // pack
        NODAL_HIP_TRY(h, H->tail_image.reserve((size_t)d.image_bytes + 64));
        char *img = H->tail_image.as<char>();
        hipStream_t st = h->stream;
        for (int k = 0; k + 1 < d.nlev; ++k) {
            const Level *lv = H->levels[t + k];
            const TailLevel &tl = d.lv[k];
            pack_f64<<<grid_for(tl.nnz), TB, 0, st>>>(img, tl.o_data, lv->A.data, tl.nnz);
            pack_f64<<<grid_for(tl.n), TB, 0, st>>>(img, tl.o_dinv, lv->dinv.as<double>(), tl.n);
            pack_i32<<<grid_for(tl.n + 1), TB, 0, st>>>(img, tl.o_indptr, lv->A.indptr, tl.n + 1);
            pack_i32<<<grid_for(tl.nc + 1), TB, 0, st>>>(img, tl.o_memptr, lv->memptr.as<int32_t>(), tl.nc + 1);
            pack_u16<<<grid_for(tl.nnz), TB, 0, st>>>(img, tl.o_indices, lv->A.indices, tl.nnz);
            pack_u16<<<grid_for(tl.n), TB, 0, st>>>(img, tl.o_agg, lv->agg.as<int32_t>(), tl.n);
            pack_u16<<<grid_for(tl.n), TB, 0, st>>>(img, tl.o_mem, lv->mem.as<int32_t>(), tl.n);
        }
        pack_f64<<<grid_for(nco * nco), TB, 0, st>>>(img, d.o_inv, H->coarse_inv.as<double>(),
                                                   (int64_t)nco * nco);
        NODAL_HIP_TRY(h, hipGetLastError());
        {   // the attribute belongs to the device's code object, not to this handle: raised ONCE per device
            // to the budget (a per-setup value could be lowered by another handle's smaller tail between
            // this setup and this hierarchy's launches -- the same rule as sagg.hip's tail)
            static std::atomic<bool> lds_allowed[64];
            const int dev = h->device >= 0 && h->device < 64 ? h->device : 0;
            if (!lds_allowed[dev].load(std::memory_order_acquire)) {
                NODAL_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(amg_tail_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS_BUDGET));
                lds_allowed[dev].store(true, std::memory_order_release);
            }
        }
        H->tdesc = d;
        H->tail = t;
        break;
    }
    return NODAL_OK;
}

}  // namespace

// ---- interface used by sparse.hip ------------------------------------------------

void amg_destroy(nodal_ctx *h) {
    delete static_cast<Hierarchy *>(h->amg);
    h->amg = nullptr;
}

// Build the hierarchy for the CSR matrix of the context.  `flag` (device double) is
// set non-zero if a diagonal or coarse pivot is not positive (not SPD).
int amg_setup(nodal_ctx *h, double *flag) {
    return amg_setup_csr(h, h->n, h->nnz, h->indptr.as<int32_t>(), h->indices.as<int32_t>(),
                         h->rowidx.as<int32_t>(), h->data.as<double>(),
                         h->diag_pos.as<int32_t>(), flag);
}

// The same for any device CSR matrix (sorted columns, rowidx = row of each entry,
// diag_pos = position of the diagonal entry of each row).  The arrays must stay
// alive as long as the hierarchy is used: level 0 aliases them.
int amg_setup_csr(nodal_ctx *h, int64_t n0, int64_t nnz0, const int32_t *indptr,
                  const int32_t *indices, const int32_t *rowidx, const double *data,
                  const int32_t *diag_pos, double *flag) {
    Hierarchy *H = static_cast<Hierarchy *>(h->amg);
    if (!H) {
        H = new Hierarchy();
        h->amg = H;
    }
    H->begin_setup();
    if (const char *e = getenv("NODAL_AMG_PASSES0")) { H->passes0 = atoi(e); H->passes_forced = true; }
    if (const char *e = getenv("NODAL_AMG_SWEEPS0")) H->sweeps0 = atoi(e);
    if (const char *e = getenv("NODAL_AMG_PASSES1")) { H->passes1 = atoi(e); H->passes_forced = true; }
    hipStream_t st = h->stream;

    Level *l0 = H->take(0);
    H->levels.push_back(l0);
    l0->A.n = n0;
    l0->A.nnz = nnz0;
    l0->A.indptr = indptr;
    l0->A.indices = indices;
    l0->A.rowidx = rowidx;
    l0->A.data = data;
    NODAL_TRY(finish_level(h, l0, diag_pos, flag));
    // Smoother: point Jacobi, or -- where at least 32 nodes (1 % in small networks) have graded links (one link
    // above 0.9 of the diagonal, or the strongest above 8 x the weakest: resistances spread over
    // a decade or more -- 5 % of the nodes at one decade, 75 % at two -- or anisotropy) -- Jacobi
    // over the aggregates' diagonal blocks.
    // NODAL_AMG_BLOCK=0 / 1 forces the choice.
    if (const char *e = getenv("NODAL_AMG_BLOCK")) {
        H->block_smoother = atoi(e) != 0;
    } else {
        NODAL_HIP_TRY(h, h->work.reserve(256));
        uint32_t *cnt = h->work.as<uint32_t>();
        NODAL_HIP_TRY(h, hipMemsetAsync(cnt, 0, 4, st));
        count_dominated<<<grid_for(n0), TB, 0, st>>>(l0->A, 0.9, 8.0, cnt);
        NODAL_HIP_TRY(h, hipGetLastError());
        uint32_t dominated = 0;
        NODAL_HIP_TRY(h, hipMemcpyAsync(&dominated, cnt, 4, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        // (a count, not a share: every such node is a near-null mode the Krylov iteration has to
        // find by itself -- 100 near-shorts in a 90 000-node grid cost point Jacobi 880 iterations)
        const int64_t bar = n0 / 100 < 32 ? (n0 / 100 > 0 ? n0 / 100 : 1) : 32;
        H->block_smoother = (int64_t)dominated >= bar;
        if (getenv("NODAL_TRACE"))
            fprintf(stderr, "[amg] %u of %lld nodes have graded links (one > 0.9 of the diagonal, or > 8 x the weakest): %s smoother\n", dominated,
                    (long long)n0, H->block_smoother ? "aggregate-block" : "point Jacobi");
    }
    // With dominant links around, aggregates must not cut them: a free node whose strong
    // neighbour is already matched waits and joins that pair instead of pairing up over a link
    // below a tenth of its strongest one; two passes per level then keep the aggregates (and
    // the smoother's blocks) at ~8 nodes.  300 x 300 grid over 4 / 6 decades: 76 / 220
    // iterations with plain matching and three passes, 40 / 68 with this.
    H->theta = H->block_smoother ? 0.1 : 0.0;
    if (const char *e = getenv("NODAL_AMG_THETA")) H->theta = atof(e);

    while ((int)H->levels.size() < MAX_LEVELS) {
        Level *fine = H->levels.back();
        const int64_t n = fine->A.n;
        if (n <= COARSEST_MAX) break;
        // PASSES matching passes; the intermediate pair graphs live in scratch levels
        NODAL_HIP_TRY(h, fine->agg.reserve((size_t)n * 4 + 8));
        int32_t *agg = fine->agg.as<int32_t>();
        Level *tmp = H->scratch;
        Csr cur = fine->A;
        int64_t nc = n;
        DevBuf &pass_map = H->pass_map;
        bool stalled = false;
        Level *coarse = H->take(H->levels.size());
        // aggregates of ~8 (three pairwise passes) suit 2-D-connected networks; along a wire
        // (fewer than 4 entries per row) piecewise constants over 8 nodes correct too little:
        // two passes there (ladder of 1e5 sections: 460 instead of 1140 iterations)
        int passes = (int)H->levels.size() == 1 ? H->passes0 : H->passes1;
        if (!H->passes_forced && (fine->A.nnz < 4 * n || H->block_smoother)) passes = passes < 2 ? passes : 2;
        for (int p = 0; p < passes; ++p) {
            NODAL_HIP_TRY(h, pass_map.reserve((size_t)cur.n * 4 + 8));
            int32_t *map = p == 0 ? agg : pass_map.as<int32_t>();
            int64_t na = 0;
            int s = matching_pass(h, cur, H->theta, map, &na);
            if (s != NODAL_OK) return s;
            if (p == 0 && na > (int64_t)(0.8 * (double)n)) { stalled = true; break; }
            if (p > 0) {
                compose_map<<<grid_for(n), TB, 0, st>>>(n, agg, map);
                NODAL_HIP_TRY(h, hipGetLastError());
            }
            const bool last = (p == passes - 1) || na <= COARSEST_MAX || na > (int64_t)(0.9 * (double)cur.n);
            Level *dst = last ? coarse : &tmp[p & 1];
            s = galerkin(h, cur, map, na, dst);
            if (s != NODAL_OK) return s;
            cur = dst->A;
            nc = na;
            if (last) break;
        }
        if (stalled) break;
        fine->nc = nc;
        // aggregate -> members
        {
            Members en{agg, n};
            int64_t nent = 0, ncon = 0;
            int s = grp::build_lists(h, en, nc, &nent, &ncon, H->members_none, H->members_rows, fine->memptr,
                                     fine->cptr, nullptr, nullptr, /*known_nent=*/nc);  // one entry per aggregate
            if (s != NODAL_OK) return s;
            if (nent != nc || ncon != n) return nodal_fail(h, NODAL_E_INVALID, "amg: empty aggregate");
            NODAL_HIP_TRY(h, fine->mem.reserve((size_t)n * 4 + 8));
            unpack_members<<<grid_for(n), TB, 0, st>>>(fine->cptr.as<uint32_t>(),
                                                      fine->mem.as<int32_t>(), n);
            NODAL_HIP_TRY(h, hipGetLastError());
        }
        // Expander-like networks (random long-range connections): the very first Galerkin product
        // fills in -- 9 -> 65 entries per row on a random graph of degree 6, 547 on the level after
        // -- and every level below is close to dense: 112 ms for a solve that takes 20 iterations.
        // Such networks are well conditioned; they keep the one-level "hierarchy" (Jacobi
        // preconditioner).  Grids: 5 -> 6.9 entries per row, 3-D grids 7 -> 13.
        if (H->levels.size() == 1 && coarse->A.n > 0) {
            const double fine_row = (double)fine->A.nnz / (double)n;
            const double coarse_row = (double)coarse->A.nnz / (double)coarse->A.n;
            if (coarse_row > 32.0 && coarse_row > 4.0 * fine_row && !getenv("NODAL_AMG_KEEP_DENSE")) {
                if (getenv("NODAL_TRACE"))
                    fprintf(stderr, "[amg] first coarse level has %.0f entries per row (fine: %.1f): expander-like, "
                                    "Jacobi preconditioner only\n", coarse_row, fine_row);
                fine->nc = 0;
                break;
            }
        }
        H->levels.push_back(coarse);
        NODAL_TRY(finish_level(h, coarse, coarse->diag_pos.as<int32_t>(), flag));
    }

    Level *last = H->levels.back();
    H->coarse_direct = last->A.n <= COARSEST_MAX && H->levels.size() > 1;
    if (H->coarse_direct) {
        NODAL_HIP_TRY(h, H->coarse_inv.reserve((size_t)last->A.n * last->A.n * 8 + 8));
        coarsest_inverse<<<1, 256, 0, st>>>(last->A, H->coarse_inv.as<double>(), flag);
        NODAL_HIP_TRY(h, hipGetLastError());
        if (!getenv("NODAL_AMG_NOTAIL")) NODAL_TRY(build_tail(h, H));
    }
    if (const char *e = getenv("NODAL_AMG_KMAX")) H->kmax = atoi(e);
    else {
        // K-cycle on the coarse levels that are still large (>= K_MIN_ROWS rows), plain V
        // hand-over below, where a V-cycle over a few thousand rows (and the tail's own two
        // inner steps) is accurate enough and every K level would double the launches
        H->kmax = 1;
        for (int l = 2; l < (int)H->levels.size(); ++l)
            if (H->levels[l]->A.n >= K_MIN_ROWS) H->kmax = l;
    }
    if (H->block_smoother) {
        const int nl = (int)H->levels.size();
        const int upto = H->tail >= 0 ? H->tail : nl - 1;  // levels [0, upto) run through cycle()
        for (int l = 0; l < upto; ++l) NODAL_TRY(build_blocks(h, H->levels[l], flag));
    }
    if (getenv("NODAL_TRACE")) {
        fprintf(stderr, "[amg] levels (rows/entries):");
        for (const Level *l : H->levels) fprintf(stderr, " %lld/%lld", (long long)l->A.n, (long long)l->A.nnz);
        fprintf(stderr, "  kmax %d tail %d\n", H->kmax, H->tail);
    }
    return NODAL_OK;
}

int amg_num_levels(nodal_ctx *h) {
    return h->amg ? (int)static_cast<Hierarchy *>(h->amg)->levels.size() : 0;
}

int64_t amg_level_size(nodal_ctx *h, int l) {
    return static_cast<Hierarchy *>(h->amg)->levels[l]->A.n;
}

namespace {

// out ~= A_l^-1 b
int cycle(nodal_ctx *h, Hierarchy *H, int l, const double *b, double *out) {
    hipStream_t st = h->stream;
    Level *L = H->levels[l];
    const int64_t n = L->A.n;
    if (l == (int)H->levels.size() - 1) {
        if (H->coarse_direct)
            dense_apply<<<1, TB, 0, st>>>((int)n, H->coarse_inv.as<double>(), b, out);
        else
            jacobi_apply<<<grid_for(n), TB, 0, st>>>(n, L->dinv.as<double>(), b, out);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    }
    Level *C = H->levels[l + 1];
    const int64_t nc = C->A.n;
    const double *dinv = L->dinv.as<double>();
    double *x = L->v(V_X), *r = L->v(V_R);
    double *rc = C->v(V_RC), *c1 = C->v(V_C1), *c2 = C->v(V_C2);
    const int32_t *agg_ = L->agg.as<int32_t>(), *memptr_ = L->memptr.as<int32_t>(), *mem_ = L->mem.as<int32_t>();
    if (L->block) {
        // x = w Binv b ;  r = b - A x
        block_apply<<<grid_for(n), TB, 0, st>>>(n, agg_, memptr_, mem_, L->boff.as<uint32_t>(),
                                               L->binv.as<double>(), b, nullptr, x, OMEGA);
        residual_vec<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, b, x, r);
    } else {
        smooth_residual<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, dinv, b, x, r);
    }
    const bool two_sweeps = !L->block && l == 0 && H->sweeps0 == 2;  // level 0 only: bandwidth-bound there, latency-bound below
    if (two_sweeps) {
        double *x2 = L->v(V_V1), *r2 = L->v(V_V2);  // (the K-cycle vectors of level 0 are never used)
        smooth_again<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, dinv, b, x, r, x2, r2);
        x = x2;
        r = r2;
    }
    restrict_sum<<<stream::grid_for_rows(nc), TB, 0, st>>>(nc, L->memptr.as<int32_t>(), L->mem.as<int32_t>(), r, rc);
    NODAL_HIP_TRY(h, hipGetLastError());
    nodal_nan_probe(h, b, n, "amg cycle b");
    nodal_nan_probe(h, dinv, n, "amg cycle dinv");
    nodal_nan_probe(h, x, n, "amg cycle x");
    nodal_nan_probe(h, r, n, "amg cycle r");
    nodal_nan_probe(h, rc, nc, "amg cycle rc");
    int nparts = 0;
    const bool coarse_is_last = (l + 1 == (int)H->levels.size() - 1);
    if (l + 1 == H->tail) {
        // the whole coarse solve (two FCG steps + everything below) in one LDS-resident launch
        nparts = 1;
        amg_tail_kernel<<<1, TAIL_THREADS, H->tdesc.total_bytes, st>>>(
            H->tdesc, H->tail_image.as<char>(), rc, c1, c2, C->part.as<double>());
        NODAL_HIP_TRY(h, hipGetLastError());
        nodal_nan_probe(h, c1, nc, "amg tail c1");
        nodal_nan_probe(h, c2, nc, "amg tail c2");
        nodal_nan_probe(h, C->part.as<double>(), 5 * DOT_BLOCKS, "amg tail part (only 0,128,..512 matter)");
        nodal_nan_probe(h, H->coarse_inv.as<double>(), H->levels.back()->A.n * H->levels.back()->A.n, "amg coarse inverse");
    } else if (coarse_is_last || l + 1 > H->kmax) {
        NODAL_TRY(cycle(h, H, l + 1, rc, c1));  // direct coarse solve (exact) or plain V hand-over
    } else {
        // two flexible-CG steps on the coarse problem, preconditioned by its own cycle
        double *v1 = C->v(V_V1), *v2 = C->v(V_V2), *r2 = C->v(V_R2);
        double *part = C->part.as<double>();
        const unsigned g = dot_grid(nc);
        nparts = (int)g;
        NODAL_TRY(cycle(h, H, l + 1, rc, c1));
        spmv_dots<<<g, TB, 0, st>>>(C->A, c1, v1, rc, nullptr, part + 0 * DOT_BLOCKS,
                                    part + 1 * DOT_BLOCKS, nullptr);
        second_residual<<<grid_for(nc), TB, 0, st>>>(nc, rc, v1, part, nparts, r2);
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_TRY(cycle(h, H, l + 1, r2, c2));
        spmv_dots<<<g, TB, 0, st>>>(C->A, c2, v2, v1, r2, part + 3 * DOT_BLOCKS,
                                    part + 2 * DOT_BLOCKS, part + 4 * DOT_BLOCKS);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    if (L->block) {
        // xp = x + P (s1 c1 + s2 c2) ;  out = xp + w Binv (b - A xp)
        double *xp = r, *r2 = x;  // r is dead after the restriction, x after the prolongation
        if (n >= SPLIT_PROLONG_MIN) {
            prolong_add<<<grid_for(n), TB, 0, st>>>(n, x, agg_, c1, c2, C->part.as<double>(), nparts, xp);
            residual_vec<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, b, xp, r2);
        } else {  // one launch: x is read per entry there, so the residual goes to a third vector
            r2 = L->v(V_T);
            prolong_residual<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, b, x, agg_, c1, c2,
                                                                     C->part.as<double>(), nparts, xp, r2);
        }
        block_apply<<<grid_for(n), TB, 0, st>>>(n, agg_, memptr_, mem_, L->boff.as<uint32_t>(),
                                               L->binv.as<double>(), r2, xp, out, OMEGA);
    } else if (n >= SPLIT_PROLONG_MIN) {
        double *xp = r;  // the residual vector is dead after the restriction
        prolong_add<<<grid_for(n), TB, 0, st>>>(n, x, L->agg.as<int32_t>(), c1, c2,
                                               C->part.as<double>(), nparts, xp);
        if (two_sweeps) {
            double *mid = L->v(V_R2);
            post_smooth<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, dinv, b, xp, mid);
            post_smooth<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, dinv, b, mid, out);
        } else {
            post_smooth<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, dinv, b, xp, out);
        }
    } else {
        prolong_smooth<<<stream::grid_for_rows(n), TB, 0, st>>>(L->A, dinv, b, x, L->agg.as<int32_t>(),
                                                               c1, c2, C->part.as<double>(), nparts,
                                                               out);
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

}  // namespace

// z ~= G^-1 r with one K-cycle
int amg_apply(nodal_ctx *h, const double *r, double *z) {
    Hierarchy *H = static_cast<Hierarchy *>(h->amg);
    if (!H || H->levels.empty()) return nodal_fail(h, NODAL_E_INVALID, "amg_setup not called");
    if (H->levels.size() == 1) {
        const int64_t n0 = H->levels[0]->A.n;
        jacobi_apply<<<grid_for(n0), TB, 0, h->stream>>>(n0, H->levels[0]->dinv.as<double>(), r, z);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    }
    return cycle(h, H, 0, r, z);
}

// grounded0: u8[n] at level 0, 1 where a resistor joins the node to ground.
// *floating = 1 if some connected component of the network has no such node.
int amg_has_floating_component(nodal_ctx *h, const uint8_t *grounded0, int32_t *floating) {
    Hierarchy *H = static_cast<Hierarchy *>(h->amg);
    if (!H || H->levels.empty()) return nodal_fail(h, NODAL_E_INVALID, "amg_setup not called");
    hipStream_t st = h->stream;
    // OR the flags up the hierarchy (two ping-pong byte vectors in work2)
    const int64_t n0 = H->levels[0]->A.n;
    const size_t half = ((size_t)n0 + 255) & ~(size_t)255;
    const Level *last = H->levels.back();
    const int64_t nl = last->A.n;
    const size_t a4 = ((size_t)nl * 4 + 255) & ~(size_t)255;
    NODAL_HIP_TRY(h, h->work2.reserve(2 * half + 2 * a4 + 512));
    uint8_t *f[2] = {h->work2.as<uint8_t>(), h->work2.as<uint8_t>() + half};
    const uint8_t *cur = grounded0;
    for (size_t l = 0; l + 1 < H->levels.size(); ++l) {
        const Level *L = H->levels[l];
        uint8_t *dst = f[l & 1];
        flags_restrict<<<grid_for(L->nc), TB, 0, st>>>(L->nc, L->memptr.as<int32_t>(),
                                                      L->mem.as<int32_t>(), cur, dst);
        cur = dst;
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    // connected components of the last level
    int32_t *label = reinterpret_cast<int32_t *>(h->work2.as<char>() + 2 * half);
    int32_t *root_ok = reinterpret_cast<int32_t *>(h->work2.as<char>() + 2 * half + a4);
    int32_t *flags = reinterpret_cast<int32_t *>(h->work2.as<char>() + 2 * half + 2 * a4);  // changed, floating
    if (nl <= 4096) {
        // the usual case (a last level of a few dozen nodes): components and verdict in ONE
        // single-workgroup launch (lowdeg.hip) instead of a host round trip per propagation step
        NODAL_HIP_TRY(h, hipMemsetAsync(flags + 1, 0, 4, st));
        NODAL_TRY(csr_floating_check_small(h, nl, last->A.indptr, last->A.indices, cur,
                                           reinterpret_cast<uint32_t *>(flags + 1)));
        NODAL_HIP_TRY(h, hipMemcpyAsync(floating, flags + 1, 4, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        return NODAL_OK;
    }
    cc_init<<<grid_for(nl), TB, 0, st>>>(nl, label);
    for (int it = 0; it < 4096; ++it) {
        NODAL_HIP_TRY(h, hipMemsetAsync(flags, 0, 4, st));
        cc_hook<<<grid_for(nl), TB, 0, st>>>(last->A, label, flags);
        cc_jump<<<grid_for(nl), TB, 0, st>>>(nl, label);
        int32_t changed = 0;
        NODAL_HIP_TRY(h, hipMemcpyAsync(&changed, flags, 4, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        if (!changed) break;
    }
    NODAL_HIP_TRY(h, hipMemsetAsync(root_ok, 0, (size_t)nl * 4, st));
    NODAL_HIP_TRY(h, hipMemsetAsync(flags + 1, 0, 4, st));
    cc_mark<<<grid_for(nl), TB, 0, st>>>(nl, label, cur, root_ok);
    cc_verdict<<<grid_for(nl), TB, 0, st>>>(nl, label, root_ok, flags + 1);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipMemcpyAsync(floating, flags + 1, 4, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    return NODAL_OK;
}

// The same verdict straight on the context's CSR matrix (no hierarchy): used by the dense
// passive path when a solution does not satisfy the equations, to tell "singular: floating
// sub-network" (reference: LinAlgError -> UnconnectedCircuitError) from anything else.
int csr_has_floating_component(nodal_ctx *h, const uint8_t *grounded, int32_t *floating) {
    hipStream_t st = h->stream;
    const int64_t n = h->n;
    Csr A;
    A.n = n;
    A.nnz = h->nnz;
    A.indptr = h->indptr.as<int32_t>();
    A.indices = h->indices.as<int32_t>();
    A.rowidx = h->rowidx.as<int32_t>();
    A.data = h->data.as<double>();
    const size_t a4 = ((size_t)n * 4 + 255) & ~(size_t)255;
    NODAL_HIP_TRY(h, h->work2.reserve(2 * a4 + 512));
    int32_t *label = h->work2.as<int32_t>();
    int32_t *root_ok = reinterpret_cast<int32_t *>(h->work2.as<char>() + a4);
    int32_t *flags = reinterpret_cast<int32_t *>(h->work2.as<char>() + 2 * a4);  // changed, floating
    cc_init<<<grid_for(n), TB, 0, st>>>(n, label);
    for (int it = 0; it < 65536; ++it) {
        NODAL_HIP_TRY(h, hipMemsetAsync(flags, 0, 4, st));
        cc_hook<<<grid_for(n), TB, 0, st>>>(A, label, flags);
        cc_jump<<<grid_for(n), TB, 0, st>>>(n, label);
        int32_t changed = 0;
        NODAL_HIP_TRY(h, hipMemcpyAsync(&changed, flags, 4, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        if (!changed) break;
    }
    NODAL_HIP_TRY(h, hipMemsetAsync(root_ok, 0, (size_t)n * 4, st));
    NODAL_HIP_TRY(h, hipMemsetAsync(flags + 1, 0, 4, st));
    cc_mark<<<grid_for(n), TB, 0, st>>>(n, label, grounded, root_ok);
    cc_verdict<<<grid_for(n), TB, 0, st>>>(n, label, root_ok, flags + 1);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipMemcpyAsync(floating, flags + 1, 4, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    return NODAL_OK;
}
