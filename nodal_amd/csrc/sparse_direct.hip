// Sparse direct solve: the route that cannot say "unsupported".
//
// The reference hands every sparse system to SuperLU (scipy.sparse.linalg.spsolve, reference
// nodal/nodal.py:325): supernodal LU with partial pivoting behind a COLAMD ordering -- any non-singular G is
// solved, an exactly singular one gives NaNs + MatrixRankWarning.  The iterative routes of sparse.hip /
// sparse_general.hip are faster on the networks they converge on; this file is what stands behind them:
// a multifrontal LU of the CSR matrix of the context, used as the preconditioner of the flexible GMRES of
// sparse_general.hip (= iterative refinement in fp64 with a true-residual test).
//
//   host, per sparsity pattern (kept while the context's struct_epoch stands):
//     1. row matching: a maximum transversal that prefers large entries (diagonal first, then the
//        largest free entry, then augmenting paths) -- the branch rows of voltage-defined components
//        have a zero diagonal (reference nodal/models.py:35-78) and get the +-1 incidence entry of one
//        of their lead nodes instead; no perfect matching = structurally singular;
//     2. nested dissection of the graph of P A + (P A)^T by breadth-first level structures from a
//        pseudo-peripheral vertex (George's automatic nested dissection): pieces of at most LEAF
//        vertices and the separators between them are the supernodes, numbered pieces first;
//        vertices of very high degree (hub nets) are set aside as a last supernode;
//     3. symbolic factorisation over the supernodes: the boundary of every front, the assembly tree
//        (parent = owner of the boundary's first vertex), levels, the index maps of the extend-add
//        and the destination of every CSR entry inside its front.
//   device, per set of values:
//     4. row / column equilibration (max-norm), fronts zeroed, entries scattered;
//     5. level by level up the assembly tree: extend-add of the children's Schur complements (one
//        workgroup per parent, children in a fixed order: deterministic), then a blocked right-looking
//        partial LU of the front's pivot block -- partial pivoting RESTRICTED to the fully summed rows
//        of the front, a pivot below sqrt(eps) |A| replaced by that bound (SuperLU_DIST's static-pivot
//        rule; the refinement absorbs it or, if the matrix is singular, fails to converge);
//     6. apply: forward substitution up the tree (contribution vectors pulled by the parents),
//        backward substitution down the tree.
// All fronts live in one buffer (sum of dim^2 doubles: ~2 GB for the 1e6-node grid; the 288 GB of
// the MI355X are what makes "no stack management" a reasonable design).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <vector>

#include "ctx.h"
#include "slu_analyse.h"

int general_krylov_direct(nodal_ctx *h, const double *b, double *x, int32_t *info, int32_t *iters,
                          double *resid);  // sparse_general.hip

namespace {

using slu::Symbolic;
using slu::analyse;
constexpr int TB = 256;

struct SluState {
    uint64_t epoch = 0;           // struct_epoch of the context the symbolic part belongs to
    int64_t n = 0, nnz = 0;
    bool have_symbolic = false, have_numeric = false;
    int32_t nsn = 0, nlev = 0;
    int64_t front_doubles = 0, vec_doubles = 0;
    int32_t max_dim = 0;
    std::vector<int32_t> lvl_ptr;  // [nlev + 1] into d_lvl_sn
    std::vector<int32_t> lvl_maxdim;
    // fronts wider than BIG_DIM are factored by a sequence of launches per front (big_front below): inside
    // every level's list they come last, lvl_small[l] = how many of the level's fronts take the batched kernel
    std::vector<int32_t> lvl_small, lvl_maxchildren, lvl_maxdim_all;
    // the substitutions' own split (APPLY_BIG): the last lvl_abig[l] fronts of a level's list are wider than that and are
    // substituted together, block step by block step; lvl_amax_s / lvl_amax_dim: their largest pivot count / width, and
    // lvl_arest_dim: the widest of the others (it selects the one-workgroup kernel's size)
    std::vector<int32_t> lvl_abig, lvl_amax_s, lvl_amax_dim, lvl_arest_dim;
    std::vector<int32_t> h_lvl_sn, h_start, h_dim;   // host copies: level lists, first pivot column, front width
    std::vector<int64_t> h_front_off;
    DevBuf bigpiv;                                     // pivot rows of the panels in flight (one set per lane)
    static constexpr int LANES = 6;                    // wide fronts of a level factored side by side
    hipStream_t lane_st[LANES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // [0]: the context's stream (not owned)
    hipEvent_t lane_ev[LANES] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [0]: fork, [k]: lane k done
    int lanes = 1;
    // device copies of the symbolic part
    DevBuf rowof, colof, newrow;   // permuted position -> original row / column; original row -> position
    DevBuf sn_of_row;              // supernode of every pivot position
    DevBuf sn_start, struct_ptr, struct_idx, front_off, vec_off, lvl_sn, child_ptr, child_idx, cmap, dest;
    // numeric part
    DevBuf fronts, vec, lperm, rs, cs, xb, stats;
    DevBuf liperm;                 // inverse of lperm per front: where a front's row ends after the interchanges
    DevBuf blk_sn, blk_b0;         // the DB x DB diagonal blocks of all fronts' pivot parts (invert_diag_blocks)
    int64_t nblocks = 0;
    int vec_nr = 0;                // right-hand sides `vec` and `xb` are sized for
    int64_t perturbed = 0;
    bool analysis_kept = false;  // the last slu_factor reused an analysis (row matching!) made for EARLIER values
};

SluState *state_of(nodal_ctx *h) { return static_cast<SluState *>(h->slu); }

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------

inline unsigned grid_for(int64_t n, unsigned cap = 8192) {
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

// row / column max-norm equilibration: rs[i] = 1 / max_j |a_ij|, cs[j] = 1 / max_i |rs_i a_ij|
__global__ __launch_bounds__(TB) void row_scales(int64_t n, const int32_t *__restrict__ indptr,
                                                 const double *__restrict__ data, double *__restrict__ rs) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        double m = 0.0;
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) m = fmax(m, fabs(data[e]));
        rs[i] = (m > 0.0 && m < 1.0 / 0.0) ? 1.0 / m : 1.0;
    }
}
__global__ __launch_bounds__(TB) void col_maxima(int64_t n, const int32_t *__restrict__ indptr,
                                                 const int32_t *__restrict__ indices, const double *__restrict__ data,
                                                 const double *__restrict__ rs, unsigned long long *__restrict__ cmax) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double r = rs[i];
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const double v = fabs(data[e]) * r;
            if (v > 0.0 && v == v)  // (non-negative doubles order like their bit patterns: exact, order-independent)
                atomicMax(&cmax[indices[e]], (unsigned long long)__double_as_longlong(v));
        }
    }
}
__global__ __launch_bounds__(TB) void col_scales(int64_t n, const unsigned long long *__restrict__ cmax,
                                                 double *__restrict__ cs) {
    for (int64_t j = (int64_t)blockIdx.x * TB + threadIdx.x; j < n; j += (int64_t)gridDim.x * TB) {
        const double m = __longlong_as_double((long long)cmax[j]);
        cs[j] = (m > 0.0 && m < 1.0 / 0.0) ? 1.0 / m : 1.0;
    }
}

__global__ __launch_bounds__(TB) void scatter_entries(int64_t n, const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const double *__restrict__ data, const double *__restrict__ rs,
                                                      const double *__restrict__ cs, const int64_t *__restrict__ dest,
                                                      double *__restrict__ fronts) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double r = rs[i];
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) fronts[dest[e]] = data[e] * r * cs[indices[e]];
    }
}

struct Tree {
    const int32_t *sn_start;
    const int64_t *struct_ptr;
    const int32_t *struct_idx;
    const int64_t *front_off, *vec_off;
    const int32_t *child_ptr, *child_idx, *cmap;
};

// parent front += the Schur complements of its children (children in list order: a fixed summation order)
template <int BS>
__global__ __launch_bounds__(BS) void extend_add(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts) {
    const int32_t p = sns[blockIdx.x];
    const int64_t pdim = (T.sn_start[p + 1] - T.sn_start[p]) + (T.struct_ptr[p + 1] - T.struct_ptr[p]);
    double *P = fronts + T.front_off[p];
    for (int32_t q = T.child_ptr[p]; q < T.child_ptr[p + 1]; ++q) {
        const int32_t c = T.child_idx[q];
        const int32_t s = T.sn_start[c + 1] - T.sn_start[c];
        const int64_t b = T.struct_ptr[c + 1] - T.struct_ptr[c];
        const int64_t cdim = s + b;
        const double *C = fronts + T.front_off[c];
        const int32_t *map = T.cmap + T.struct_ptr[c];
        // columns over the waves, rows over the lanes (column-major: the lanes read contiguous words)
        for (int64_t j = threadIdx.x >> 6; j < b; j += BS >> 6) {
            const int64_t pj = (int64_t)map[j] * pdim;
            const double *col = C + (s + j) * cdim + s;
            for (int64_t i = threadIdx.x & 63; i < b; i += 64) P[pj + map[i]] += col[i];
        }
        __syncthreads();  // (two children may add to the same entry: one after the other)
    }
}

// Blocked right-looking partial LU of the s leading columns of a front (column-major, ld = dim), partial
// pivoting restricted to the s fully summed rows; lperm[k] = local row that ended at position k.
constexpr int NB = 16;
// INLDS: the whole front is copied to LDS, factored there and copied back (levels whose widest front fits: the leaves
// and the small separators, i.e. most fronts; every column step is four barriers around memory operations, at LDS
// latency instead of a round trip to L2 each).  Same operations in the same order: the same factors.
template <int BS, bool INLDS = false>
__global__ __launch_bounds__(BS) void factor_fronts(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts,
                                                    int32_t *__restrict__ lperm, double tiny, double repl,
                                                    unsigned long long *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) double front_lds[];
    const int32_t t = sns[blockIdx.x];
    const int s = T.sn_start[t + 1] - T.sn_start[t];
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    double *Fg = fronts + T.front_off[t];
    double *F = INLDS ? front_lds : Fg;
    if constexpr (INLDS) {
        for (int e = threadIdx.x; e < dim * dim; e += BS) front_lds[e] = Fg[e];
        __syncthreads();
    }
    int32_t *perm = lperm + T.sn_start[t];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = BS / 64;
    __shared__ double red_v[NW];
    __shared__ int red_i[NW];
    __shared__ int piv_row;
    __shared__ double Ls[NB][NB + 1];
    for (int i = tid; i < s; i += BS) perm[i] = i;
    __syncthreads();
    for (int k0 = 0; k0 < s; k0 += NB) {
        const int nb = s - k0 < NB ? s - k0 : NB;
        for (int k = k0; k < k0 + nb; ++k) {
            // pivot search in column k among the fully summed rows k .. s-1 (first row of maximal |a|)
            double best = -1.0;
            int bi = k;
            for (int i = k + tid; i < s; i += BS) {
                const double v = fabs(F[i + (int64_t)k * dim]);
                if (v > best) { best = v; bi = i; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_down(best, off, 64);
                const int oi = __shfl_down(bi, off, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
            __syncthreads();
            if (tid == 0) {
                double bv = red_v[0];
                int br = red_i[0];
                for (int w = 1; w < NW; ++w)
                    if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < br)) { bv = red_v[w]; br = red_i[w]; }
                const double d = F[k + (int64_t)k * dim];
                if (!(bv >= tiny)) {  // nothing usable in the pivot block: SuperLU_DIST's static-pivot rule
                    F[k + (int64_t)k * dim] = d < 0.0 ? -repl : repl;  // (|repl| >= tiny; the second opinion uses another value)
                    br = k;
                    atomicAdd(stats, 1ull);
                } else if (fabs(d) >= 0.25 * bv) {
                    br = k;  // (threshold pivoting: the diagonal stays when it is within a factor 4 of the best)
                }
                piv_row = br;
                if (br != k) {
                    const int32_t tmp = perm[k];
                    perm[k] = perm[br];
                    perm[br] = tmp;
                }
            }
            __syncthreads();
            const int p = piv_row;
            if (p != k) {
                for (int j = tid; j < dim; j += BS) {
                    const double a = F[k + (int64_t)j * dim], b2 = F[p + (int64_t)j * dim];
                    F[k + (int64_t)j * dim] = b2;
                    F[p + (int64_t)j * dim] = a;
                }
                __syncthreads();
            }
            const double rp = 1.0 / F[k + (int64_t)k * dim];
            const int jend = k0 + nb;
            for (int i = k + 1 + tid; i < dim; i += BS) {
                const double l = F[i + (int64_t)k * dim] * rp;
                F[i + (int64_t)k * dim] = l;
                for (int j = k + 1; j < jend; ++j) F[i + (int64_t)j * dim] = fma(-l, F[k + (int64_t)j * dim], F[i + (int64_t)j * dim]);
            }
            __syncthreads();
        }
        const int j0 = k0 + nb;
        if (j0 >= dim) break;
        // U12 = L11^-1 A12: rows k0 .. j0-1, columns j0 .. dim-1
        for (int idx = tid; idx < nb * nb; idx += BS) {
            const int r = idx % nb, q = idx / nb;
            Ls[r][q] = F[(k0 + r) + (int64_t)(k0 + q) * dim];
        }
        __syncthreads();
        for (int j = j0 + tid; j < dim; j += BS) {
            double u[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) u[r] = r < nb ? F[(k0 + r) + (int64_t)j * dim] : 0.0;
#pragma unroll
            for (int r = 1; r < NB; ++r)
#pragma unroll
                for (int q = 0; q < r; ++q)
                    if (r < nb) u[r] = fma(-Ls[r][q], u[q], u[r]);
#pragma unroll
            for (int r = 0; r < NB; ++r)
                if (r < nb) F[(k0 + r) + (int64_t)j * dim] = u[r];
        }
        __syncthreads();
        // trailing update: rows and columns j0 .. dim-1 (columns over the waves, rows over the lanes)
        for (int j = j0 + wave; j < dim; j += NW) {
            double u[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) u[q] = q < nb ? F[(k0 + q) + (int64_t)j * dim] : 0.0;
            for (int i = j0 + lane; i < dim; i += 64) {
                double acc = F[i + (int64_t)j * dim];
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    if (q < nb) acc = fma(-F[i + (int64_t)(k0 + q) * dim], u[q], acc);
                F[i + (int64_t)j * dim] = acc;
            }
        }
        __syncthreads();
    }
    if constexpr (INLDS) {
        __syncthreads();
        for (int e = threadIdx.x; e < dim * dim; e += BS) Fg[e] = front_lds[e];
    }
}

// ---- small fronts, one WAVEFRONT each, the panel in registers (round 5) --------------------------------------------
// factor_fronts<64, true> keeps the front in LDS and walks it with loops of LDS round trips: 80 us for a 60 x 60 leaf,
// 3.4 ms for the 48 888 leaves of config 5's matrix.  Here the lane IS the row: the 16 columns of a panel live in the
// lane's registers for the whole panel step (the pivot row travels by v_readlane -- the pivot's row number is uniform --,
// the arg-max by wave shuffles, no barrier that means anything with one wave), and the lane's rows of L21 stay in
// those registers for the rank-16 update of the trailing block, which reads U12 by broadcast from LDS.  The same
// operations on the same values in the same order as factor_fronts: identical factors (NODAL_DIRECT_WAVE=0 keeps the
// LDS kernel as the cross-check).  RPL: rows per lane (1: fronts of at most 64 rows; 2: of at most 128).
__device__ __forceinline__ double lane_bcast(double v, int src) {  // src uniform
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <int RPL>
__global__ __launch_bounds__(64) void factor_fronts_wave(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts,
                                                         int32_t *__restrict__ lperm, double tiny, double repl,
                                                         unsigned long long *__restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) double Fl[];
    const int32_t t = sns[blockIdx.x];
    const int s = T.sn_start[t + 1] - T.sn_start[t];
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    double *Fg = fronts + T.front_off[t];
    int32_t *perm = lperm + T.sn_start[t];
    const int lane = threadIdx.x;
    for (int e = lane; e < dim * dim; e += 64) Fl[e] = Fg[e];
    for (int i = lane; i < s; i += 64) perm[i] = i;
    __syncthreads();
    for (int k0 = 0; k0 < s; k0 += NB) {
        const int nb = s - k0 < NB ? s - k0 : NB;
        double a[RPL][NB];
        auto load_panel = [&]() {
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int i = lane + 64 * r;
#pragma unroll
                for (int c = 0; c < NB; ++c) a[r][c] = (i < dim && c < nb) ? Fl[i + (k0 + c) * dim] : 0.0;
            }
        };
        auto store_panel = [&]() {
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int i = lane + 64 * r;
#pragma unroll
                for (int c = 0; c < NB; ++c)
                    if (i < dim && c < nb) Fl[i + (k0 + c) * dim] = a[r][c];
            }
        };
        load_panel();
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            if (c < nb) {  // (uniform)
                const int k = k0 + c;
                // pivot search in column k among the fully summed rows k .. s-1 (first row of maximal |a|)
                double best = -1.0;
                int bi = k;
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const int i = lane + 64 * r;
                    const double v = fabs(a[r][c]);
                    if (i >= k && i < s && v > best) { best = v; bi = i; }
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ov = __shfl_xor(best, off, 64);
                    const int oi = __shfl_xor(bi, off, 64);
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                double d = lane_bcast(a[0][c], k & 63);
                if (RPL > 1 && k >= 64) d = lane_bcast(a[RPL - 1][c], k & 63);
                int p = bi;
                double forced = 0.0;
                if (!(best >= tiny)) {  // nothing usable in the pivot block: SuperLU_DIST's static-pivot rule
                    forced = d < 0.0 ? -repl : repl;
                    p = k;
                    if (lane == 0) atomicAdd(stats, 1ull);
                } else if (fabs(d) >= 0.25 * best) {
                    p = k;  // (threshold pivoting: the diagonal stays when it is within a factor 4 of the best)
                }
                if (forced != 0.0) {
#pragma unroll
                    for (int r = 0; r < RPL; ++r)
                        if (lane + 64 * r == k) a[r][c] = forced;
                }
                if (p != k) {  // (uniform; rare) the whole rows k and p change places: through LDS, every column
                    store_panel();
                    __syncthreads();
                    for (int j = lane; j < dim; j += 64) {
                        const double x = Fl[k + j * dim], y = Fl[p + j * dim];
                        Fl[k + j * dim] = y;
                        Fl[p + j * dim] = x;
                    }
                    if (lane == 0) {
                        const int32_t tmp = perm[k];
                        perm[k] = perm[p];
                        perm[p] = tmp;
                    }
                    __syncthreads();
                    load_panel();
                }
                // the pivot row (row k after the interchange), columns c .. nb-1 of the panel
                double prow[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    prow[j] = 0.0;
                    if (j >= c) {
                        prow[j] = lane_bcast(a[0][j], k & 63);
                        if (RPL > 1 && k >= 64) prow[j] = lane_bcast(a[RPL - 1][j], k & 63);
                    }
                }
                const double rp = 1.0 / prow[c];
#pragma unroll
                for (int r = 0; r < RPL; ++r) {
                    const int i = lane + 64 * r;
                    if (i > k && i < dim) {
                        const double l = a[r][c] * rp;
                        a[r][c] = l;
#pragma unroll
                        for (int j = 0; j < NB; ++j)
                            if (j > c && j < nb) a[r][j] = fma(-l, prow[j], a[r][j]);
                    }
                }
            }
        }
        store_panel();
        __syncthreads();
        const int j0 = k0 + nb;
        if (j0 >= dim) break;
        // U12 = L11^-1 A12: rows k0 .. j0-1 of the columns j >= j0, one lane per column
        for (int j = j0 + lane; j < dim; j += 64) {
            double u[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) u[r] = r < nb ? Fl[(k0 + r) + j * dim] : 0.0;
#pragma unroll
            for (int r = 1; r < NB; ++r)
#pragma unroll
                for (int q = 0; q < r; ++q)
                    if (r < nb) u[r] = fma(-Fl[(k0 + r) + (k0 + q) * dim], u[q], u[r]);
#pragma unroll
            for (int r = 0; r < NB; ++r)
                if (r < nb) Fl[(k0 + r) + j * dim] = u[r];
        }
        __syncthreads();
        // trailing update: the lane's rows of L21 are in its registers, U12 by broadcast from LDS
        for (int j = j0; j < dim; ++j) {
            double u[NB];
#pragma unroll
            for (int q = 0; q < NB; ++q) u[q] = q < nb ? Fl[(k0 + q) + j * dim] : 0.0;
#pragma unroll
            for (int r = 0; r < RPL; ++r) {
                const int i = lane + 64 * r;
                if (i >= j0 && i < dim) {
                    double acc = Fl[i + j * dim];
#pragma unroll
                    for (int q = 0; q < NB; ++q)
                        if (q < nb) acc = fma(-a[r][q], u[q], acc);
                    Fl[i + j * dim] = acc;
                }
            }
        }
        __syncthreads();
    }
    for (int e = lane; e < dim * dim; e += 64) Fg[e] = Fl[e];
}

// ---- fronts wider than BIG_DIM: one front, many workgroups ------------------------------------------------
// A 1500-wide front holds 7e8 of the factorisation's flops and ONE workgroup needed 30-90 ms for it (a 2944-wide
// front of a binary tree: 1.5 s).  Such fronts are factored panel by panel from the host: a single-workgroup
// panel kernel (NBB columns, the same pivoting among the fully summed rows), the panel's interchanges applied to
// the rest of the front, the U block row by a triangular solve over the columns, and the trailing update
// A22 -= L21 U12 by the fp64 MFMA GEMM of gemm_f64.hip -- the one place of the sparse path where the matrix cores
// have work (north_star: MFMA where G is dense enough to be a panel factorisation).
constexpr int BIG_DIM = 84;  // (512 until round 5; measured at cfg5(1000): 84 / 128 / 192 / 256 -> 30.3 / 31.6 / 33.9 / 34.0 ms per repeated solve)
constexpr int NBB = 64;  // (64: the panel kernel -- one CU's bandwidth -- 694 us and the triangular solve 375 us per panel, its 64 values per thread spilling)

__global__ __launch_bounds__(256) void iota_i32(int32_t *__restrict__ p, int n) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = i;
}

// columns [k0, k0 + nb) of the front, rows k0 .. dim-1: right-looking within the panel, pivot rows among
// k .. s-1 recorded in piv[0 .. nb) (rows are interchanged inside the panel only: apply_swaps does the rest)
__device__ __forceinline__ void panel_factor_body(double *__restrict__ F, int dim, int s, int k0, int nb,
                                                  int32_t *__restrict__ piv, double tiny, double repl,
                                                  unsigned long long *__restrict__ stats) {
    constexpr int BS = 1024, NW = BS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ double red_v[NW];
    __shared__ int red_i[NW];
    __shared__ int piv_row;
    const int jend = k0 + nb;
    for (int k = k0; k < jend; ++k) {
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < s; i += BS) {
            const double v = fabs(F[i + (int64_t)k * dim]);
            if (v > best) { best = v; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(best, off, 64);
            const int oi = __shfl_down(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            double bv = red_v[0];
            int br = red_i[0];
            for (int w = 1; w < NW; ++w)
                if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < br)) { bv = red_v[w]; br = red_i[w]; }
            const double d = F[k + (int64_t)k * dim];
            if (!(bv >= tiny)) {
                F[k + (int64_t)k * dim] = d < 0.0 ? -repl : repl;
                br = k;
                atomicAdd(stats, 1ull);
            } else if (fabs(d) >= 0.25 * bv) {
                br = k;
            }
            piv_row = br;
            piv[k - k0] = br;
        }
        __syncthreads();
        const int p = piv_row;
        if (p != k) {
            for (int j = k0 + tid; j < jend; j += BS) {
                const double a = F[k + (int64_t)j * dim], b2 = F[p + (int64_t)j * dim];
                F[k + (int64_t)j * dim] = b2;
                F[p + (int64_t)j * dim] = a;
            }
            __syncthreads();
        }
        const double rp = 1.0 / F[k + (int64_t)k * dim];
        for (int i = k + 1 + tid; i < dim; i += BS) {
            const double l = F[i + (int64_t)k * dim] * rp;
            F[i + (int64_t)k * dim] = l;
            for (int j = k + 1; j < jend; ++j) F[i + (int64_t)j * dim] = fma(-l, F[k + (int64_t)j * dim], F[i + (int64_t)j * dim]);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void panel_factor(double *__restrict__ F, int dim, int s, int k0, int nb,
                                                     int32_t *__restrict__ piv, double tiny, double repl,
                                                     unsigned long long *__restrict__ stats) {
    panel_factor_body(F, dim, s, k0, nb, piv, tiny, repl, stats);
}

// The same panel with its rows IN REGISTERS: thread t keeps rows k0 + t, k0 + t + 256, ... (RPT of them) of the
// panel's PNB columns, loaded once and stored once; a column step is a block arg-max (shuffles + one LDS round), the
// pivot row and row k through LDS, and the rank-1 update in registers -- no global memory inside the loop, where the
// version above streams the shrinking panel through ONE compute unit twice per column (79 us per 16 columns of a
// 1489-row front).  Same pivots (largest magnitude among the fully summed rows, the smallest row on ties, the
// diagonal kept within a factor 4, the static replacement), the same operations on every entry in the same order:
// bit-identical factors.  Panels of at most RPT * 256 rows (RPT <= 6: 96 values per thread).
constexpr int PNB = 16;
template <int RPT>
__device__ __forceinline__ void panel_factor_regs_body(double *__restrict__ F, int dim, int s, int k0, int nb,
                                                       int32_t *__restrict__ piv, double tiny, double repl,
                                                       unsigned long long *__restrict__ stats) {
    constexpr int BS = 256, NW = BS / 64;  // (256 threads: a budget of 256 registers each; 512 / 1024 threads leave 128 / 64 and spill)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Two barriers per column (round 5; four before): every thread takes the pivot decision itself from the four waves'
    // candidates -- no thread-0 step with a barrier behind it --, and the words a column hands round (candidates, row k,
    // the pivot row) live in slots of the column's parity, so that column c + 1 writes nothing a slow thread of column c
    // still reads (between a slot's two uses lie the two barriers of the column in between).
    __shared__ double red_v[2][NW];
    __shared__ int red_i[2][NW];
    __shared__ double prow2[2][PNB], krow2[2][PNB];
    double a[RPT][PNB];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = k0 + tid + r * BS;
#pragma unroll
        for (int c = 0; c < PNB; ++c) a[r][c] = (i < dim && c < nb) ? F[i + (int64_t)(k0 + c) * dim] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < PNB; ++c) {  // (fully unrolled: every index into a[][] is a constant -- registers, not scratch)
        if (c < nb) {                // (uniform over the workgroup)
        const int k = k0 + c;
        double *prow = prow2[c & 1], *krow = krow2[c & 1];
        double best = -1.0;
        int bi = k;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = k0 + tid + r * BS;
            const double v = fabs(a[r][c]);
            if (i >= k && i < s && v > best) { best = v; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double ov = __shfl_down(best, off, 64);
            const int oi = __shfl_down(bi, off, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_v[c & 1][wave] = best; red_i[c & 1][wave] = bi; }
        // row k of the panel for whoever needs it (the diagonal for the pivot rule, the row for the interchange)
        if (tid == c) {  // (row k = k0 + c is thread c's first row)
#pragma unroll
            for (int j = 0; j < PNB; ++j) krow[j] = a[0][j];
        }
        __syncthreads();
        int p;
        double forced = 0.0;  // 0: the pivot is whatever row p holds
        {
            double bv = red_v[c & 1][0];
            int br = red_i[c & 1][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                const double wv = red_v[c & 1][w];
                const int wi = red_i[c & 1][w];
                if (wv > bv || (wv == bv && wi < br)) { bv = wv; br = wi; }
            }
            const double d = krow[c];
            if (!(bv >= tiny)) {
                forced = d < 0.0 ? -repl : repl;
                br = k;
                if (tid == 0) atomicAdd(stats, 1ull);
            } else if (fabs(d) >= 0.25 * bv) {
                br = k;
            }
            p = br;
            if (tid == 0) piv[c] = br;
        }
        // the owner of row p publishes it
        {
            const int off = p - k0 - tid;  // row p is mine iff off = r * BS for some r < RPT
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (off == r * BS) {
                    if (forced != 0.0) a[r][c] = forced;  // (then p == k: the replaced diagonal)
#pragma unroll
                    for (int j = 0; j < PNB; ++j) prow[j] = a[r][j];
                }
        }
        __syncthreads();
        if (p != k) {  // interchange rows k and p inside the panel
            if (tid == c) {
#pragma unroll
                for (int j = 0; j < PNB; ++j) a[0][j] = prow[j];
            }
            const int off = p - k0 - tid;
#pragma unroll
            for (int r = 0; r < RPT; ++r)
                if (off == r * BS) {
#pragma unroll
                    for (int j = 0; j < PNB; ++j) a[r][j] = krow[j];
                }
        }
        // (the pivot row -- row k after the interchange -- is read from LDS: broadcast reads, no registers)
        const double rp = 1.0 / prow[c];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = k0 + tid + r * BS;
            if (i > k && i < dim) {
                const double l = a[r][c] * rp;
                a[r][c] = l;
#pragma unroll
                for (int j = 0; j < PNB; ++j)
                    if (j > c) a[r][j] = fma(-l, prow[j], a[r][j]);
            }
        }
        }
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = k0 + tid + r * BS;
#pragma unroll
        for (int c = 0; c < PNB; ++c)
            if (i < dim && c < nb) F[i + (int64_t)(k0 + c) * dim] = a[r][c];
    }
}

template <int RPT>
__global__ __launch_bounds__(256) void panel_factor_regs(double *__restrict__ F, int dim, int s, int k0, int nb,
                                                          int32_t *__restrict__ piv, double tiny, double repl,
                                                          unsigned long long *__restrict__ stats) {
    panel_factor_regs_body<RPT>(F, dim, s, k0, nb, piv, tiny, repl, stats);
}

// ---- the same steps for ALL the wide fronts of a level at once (round 5) ------------------------------------------
// Round 4 walked every wide front's panels from the host, front after front (four side by side on streams of their
// own): a chain of one-workgroup panel kernels, 100 us per 16 columns per front.  The fronts of a level are independent
// and their panels line up: step k0 of the level is ONE panel launch over its fronts (blockIdx = front; a front whose
// pivot columns are exhausted returns at once), one launch for the interchanges, one for the U block rows, one for the
// rank-16 updates of all trailing blocks (64 x 64 tiles, blockIdx.z = front).  A level costs 4 launches per 16 pivot
// columns of its WIDEST front whatever the number of fronts, and the fronts between 192 and 512 rows -- one workgroup
// each until now, their trailing blocks streamed through one compute unit -- take the same path.
struct FrontRef { double *F; int dim, s; int32_t *perm; };
__device__ __forceinline__ FrontRef front_ref(const Tree &T, const int32_t *__restrict__ sns, int which,
                                              double *__restrict__ fronts, int32_t *__restrict__ lperm) {
    const int32_t t = sns[which];
    FrontRef f;
    f.s = T.sn_start[t + 1] - T.sn_start[t];
    f.dim = f.s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    f.F = fronts + T.front_off[t];
    f.perm = lperm + T.sn_start[t];
    return f;
}

__global__ __launch_bounds__(256) void level_iota(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts,
                                                  int32_t *__restrict__ lperm) {
    const FrontRef f = front_ref(T, sns, blockIdx.x, fronts, lperm);
    for (int i = threadIdx.x; i < f.s; i += 256) f.perm[i] = i;
}

template <int RPT>
__global__ __launch_bounds__(256) void level_panel_regs(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts,
                                                        int32_t *__restrict__ lperm, int k0, int32_t *__restrict__ pivs,
                                                        double tiny, double repl, unsigned long long *__restrict__ stats) {
    const FrontRef f = front_ref(T, sns, blockIdx.x, fronts, lperm);
    if (k0 >= f.s) return;
    const int nb = f.s - k0 < PNB ? f.s - k0 : PNB;
    panel_factor_regs_body<RPT>(f.F, f.dim, f.s, k0, nb, pivs + (int64_t)blockIdx.x * NBB, tiny, repl, stats);
}
__global__ __launch_bounds__(1024) void level_panel_stream(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts,
                                                           int32_t *__restrict__ lperm, int k0, int32_t *__restrict__ pivs,
                                                           double tiny, double repl, unsigned long long *__restrict__ stats) {
    const FrontRef f = front_ref(T, sns, blockIdx.x, fronts, lperm);
    if (k0 >= f.s) return;
    const int nb = f.s - k0 < PNB ? f.s - k0 : PNB;
    panel_factor_body(f.F, f.dim, f.s, k0, nb, pivs + (int64_t)blockIdx.x * NBB, tiny, repl, stats);
}

// the panel's interchanges on the columns outside it and on the row permutation, then U12 = L11^-1 A12 for this
// thread's column (one thread per column outside the panel: blockIdx.x tiles of 256 columns, blockIdx.y = front)
__global__ __launch_bounds__(256) void level_swaps_trsm(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts,
                                                        int32_t *__restrict__ lperm, int k0, const int32_t *__restrict__ pivs) {
    const FrontRef f = front_ref(T, sns, blockIdx.y, fronts, lperm);
    if (k0 >= f.s) return;
    const int nb = f.s - k0 < PNB ? f.s - k0 : PNB;
    const int dim = f.dim;
    if ((int)blockIdx.x * 256 >= dim - nb + 1) return;
    const int32_t *piv = pivs + (int64_t)blockIdx.y * NBB;
    __shared__ double Ls[PNB][PNB + 1];
    for (int idx = threadIdx.x; idx < nb * nb; idx += 256) {
        const int r = idx % nb, q = idx / nb;
        Ls[r][q] = f.F[(k0 + r) + (int64_t)(k0 + q) * dim];
    }
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int outside = dim - nb;
    if (t < outside) {
        const int j = t < k0 ? t : t + nb;
        double *col = f.F + (int64_t)j * dim;
        for (int q = 0; q < nb; ++q) {
            const int p = piv[q];
            if (p != k0 + q) {
                const double a = col[k0 + q];
                col[k0 + q] = col[p];
                col[p] = a;
            }
        }
        if (j >= k0 + nb) {  // a column right of the panel: its rows k0 .. k0+nb-1 become U12
            double u[PNB];
#pragma unroll
            for (int r = 0; r < PNB; ++r) u[r] = r < nb ? col[k0 + r] : 0.0;
#pragma unroll
            for (int r = 1; r < PNB; ++r)
#pragma unroll
                for (int q = 0; q < r; ++q)
                    if (r < nb) u[r] = fma(-Ls[r][q], u[q], u[r]);
#pragma unroll
            for (int r = 0; r < PNB; ++r)
                if (r < nb) col[k0 + r] = u[r];
        }
    } else if (t == outside) {
        for (int q = 0; q < nb; ++q) {
            const int p = piv[q];
            if (p != k0 + q) {
                const int32_t a = f.perm[k0 + q];
                f.perm[k0 + q] = f.perm[p];
                f.perm[p] = a;
            }
        }
    }
}

// A22 -= L21 U12 of every front of the list: 64 x 64 tiles (blockIdx.x, blockIdx.y) of the trailing block, blockIdx.z =
// front; 256 threads, a 4 x 4 micro-tile each (rows tx + 16 i: a wavefront reads whole 128-byte segments of a column)
__global__ __launch_bounds__(256) void level_rank_update(Tree T, const int32_t *__restrict__ sns, double *__restrict__ fronts,
                                                         int32_t *__restrict__ lperm, int k0) {
    const FrontRef f = front_ref(T, sns, blockIdx.z, fronts, lperm);
    if (k0 >= f.s) return;
    const int nb = f.s - k0 < PNB ? f.s - k0 : PNB;
    const int dim = f.dim, j0 = k0 + nb;
    const int i0 = j0 + (int)blockIdx.x * 64, c0 = j0 + (int)blockIdx.y * 64;
    if (i0 >= dim || c0 >= dim) return;
    __shared__ double Lt[PNB][64 + 1], Ut[PNB][64 + 1];
    for (int e = threadIdx.x; e < PNB * 64; e += 256) {
        const int q = e / 64, r = e % 64;
        Lt[q][r] = (q < nb && i0 + r < dim) ? f.F[(i0 + r) + (int64_t)(k0 + q) * dim] : 0.0;
        Ut[q][r] = (q < nb && c0 + r < dim) ? f.F[(k0 + q) + (int64_t)(c0 + r) * dim] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    double acc[4][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = i0 + tx + 16 * ii, c = c0 + ty * 4 + jj;
            acc[ii][jj] = (i < dim && c < dim) ? f.F[i + (int64_t)c * dim] : 0.0;
        }
#pragma unroll
    for (int q = 0; q < PNB; ++q) {
        double l[4], u[4];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) l[ii] = Lt[q][tx + 16 * ii];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) u[jj] = Ut[q][ty * 4 + jj];
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[ii][jj] = fma(-l[ii], u[jj], acc[ii][jj]);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const int i = i0 + tx + 16 * ii, c = c0 + ty * 4 + jj;
            if (i < dim && c < dim) f.F[i + (int64_t)c * dim] = acc[ii][jj];
        }
}

// the panel's interchanges on the columns outside it (one thread per column, the nb swaps in order), and on
// the front's row permutation (one extra thread)
__global__ __launch_bounds__(256) void apply_swaps(double *__restrict__ F, int dim, int k0, int nb,
                                                   const int32_t *__restrict__ piv, int32_t *__restrict__ perm) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int outside = dim - nb;
    if (t < outside) {
        const int j = t < k0 ? t : t + nb;
        double *col = F + (int64_t)j * dim;
        for (int q = 0; q < nb; ++q) {
            const int p = piv[q];
            if (p != k0 + q) {
                const double a = col[k0 + q];
                col[k0 + q] = col[p];
                col[p] = a;
            }
        }
    } else if (t == outside) {
        for (int q = 0; q < nb; ++q) {
            const int p = piv[q];
            if (p != k0 + q) {
                const int32_t a = perm[k0 + q];
                perm[k0 + q] = perm[p];
                perm[p] = a;
            }
        }
    }
}

// U12 = L11^-1 A12: rows k0 .. k0+nb-1 of the columns j >= k0 + nb (one thread per column; L11 unit lower, in LDS).
// Sixteen rows at a time in registers, the rows above them re-read from the thread's own column: the same code
// for every panel width.  (A version templated on the width with the whole column of the panel in registers
// gave wrong values for the width 32 -- and only that one -- with this toolchain; measured, not understood.)
__global__ __launch_bounds__(256) void trsm_u12(double *__restrict__ F, int dim, int k0, int nb) {
    __shared__ double Ls[NBB][NBB + 1];
    for (int idx = threadIdx.x; idx < nb * nb; idx += 256) {
        const int r = idx % nb, q = idx / nb;
        Ls[r][q] = F[(k0 + r) + (int64_t)(k0 + q) * dim];
    }
    __syncthreads();
    const int j = k0 + nb + blockIdx.x * 256 + threadIdx.x;
    if (j >= dim) return;
    double *col = F + (int64_t)j * dim + k0;
    for (int rb = 0; rb < nb; rb += 16) {
        double u[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = rb + r < nb ? col[rb + r] : 0.0;
        for (int q = 0; q < rb; ++q) {  // the rows already solved
            const double uq = col[q];
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rb + r < nb) u[r] = fma(-Ls[rb + r][q], uq, u[r]);
        }
#pragma unroll
        for (int r = 1; r < 16; ++r)
#pragma unroll
            for (int q = 0; q < r; ++q)
                if (rb + r < nb) u[r] = fma(-Ls[rb + r][rb + q], u[q], u[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (rb + r < nb) col[rb + r] = u[r];
    }
}

// child number `ordinal` of every parent of the list: its Schur complement into the parent's front, columns
// over blockIdx.y (children of one parent are added by successive launches: a fixed order, no two writers)
__global__ __launch_bounds__(256) void extend_add_child(Tree T, const int32_t *__restrict__ sns, int ordinal,
                                                        double *__restrict__ fronts) {
    const int32_t p = sns[blockIdx.x];
    const int32_t q = T.child_ptr[p] + ordinal;
    if (q >= T.child_ptr[p + 1]) return;
    const int32_t c = T.child_idx[q];
    const int64_t pdim = (T.sn_start[p + 1] - T.sn_start[p]) + (T.struct_ptr[p + 1] - T.struct_ptr[p]);
    const int32_t s = T.sn_start[c + 1] - T.sn_start[c];
    const int64_t b = T.struct_ptr[c + 1] - T.struct_ptr[c];
    const int64_t cdim = s + b;
    double *P = fronts + T.front_off[p];
    const double *C = fronts + T.front_off[c];
    const int32_t *map = T.cmap + T.struct_ptr[c];
    for (int64_t j = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6); j < b; j += (int64_t)gridDim.y * 4) {
        const int64_t pj = (int64_t)map[j] * pdim;
        const double *col = C + (s + j) * cdim + s;
        for (int64_t i = threadIdx.x & 63; i < b; i += 64) P[pj + map[i]] += col[i];
    }
}

// a fixed pseudo-random vector in [-1, 1) (the second opinion of sparse_direct_solve)
__global__ __launch_bounds__(TB) void hashed_rhs(int64_t n, double *__restrict__ b) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
        z ^= z >> 31;
        z *= 0xBF58476D1CE4E5B9ull;
        z ^= z >> 29;
        b[i] = (double)(int64_t)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    }
}

// b' = P (rs .* b) in elimination order
// (nr right-hand sides interleaved by row: element (i, c) at [i * nr + c])
__global__ __launch_bounds__(TB) void permute_rhs(int64_t n, int nr, const int32_t *__restrict__ rowof,
                                                  const double *__restrict__ rs, const double *__restrict__ b,
                                                  double *__restrict__ xb) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < n * nr; e += (int64_t)gridDim.x * TB) {
        const int64_t k = e / nr;
        const int32_t i = rowof[k];
        xb[e] = b[(int64_t)i * nr + e % nr] * rs[i];
    }
}
__global__ __launch_bounds__(TB) void unpermute_solution(int64_t n, int nr, const int32_t *__restrict__ colof,
                                                         const double *__restrict__ cs, const double *__restrict__ xb,
                                                         double *__restrict__ x) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < n * nr; e += (int64_t)gridDim.x * TB) {
        const int64_t k = e / nr;
        const int32_t j = colof[k];
        x[(int64_t)j * nr + e % nr] = xb[e] * cs[j];
    }
}

// ---- apply: blocked substitution with the diagonal blocks' inverses, any number of right-hand sides -----------
// After the factorisation every DB x DB diagonal block of a front's pivot part is replaced IN PLACE by its inverses
// (invert_diag_blocks: the unit lower triangle by inv(L_bb) without its diagonal, the upper triangle by inv(U_bb)), so
// a substitution is a sequence of small dense products instead of s dependent column steps with a barrier each (round
// 4: a front of 1000 pivot columns took a millisecond per sweep, the apply of config 5's factors 8 ms):
//   forward   y_b = inv(L_bb) v_b;  v_below -= L[below, b] y_b         block by block down the pivot columns
//   backward  v_S -= U12 x_B;  x_b = inv(U_bb) v_b;  v_above -= U[above, b] x_b    block by block upwards
// NR right-hand sides travel together, interleaved by row (element (i, c) at [i * NR + c]): every entry of L and U is
// read once for all of them.  One workgroup per front; the rows of a product over the threads (coalesced reads of a
// column of the front), CW columns of the right-hand sides per thread.
constexpr int DB = 32;

// one wavefront per (front, diagonal block): lanes 0..31 the columns of inv(U_bb), lanes 32..63 those of inv(L_bb)
__global__ __launch_bounds__(64) void invert_diag_blocks(Tree T, int64_t nblocks, const int32_t *__restrict__ blk_sn,
                                                         const int32_t *__restrict__ blk_b0, double *__restrict__ fronts) {
    __shared__ double D[DB][DB + 1];
    __shared__ double X[64][DB + 1];
    const int64_t b = blockIdx.x;
    if (b >= nblocks) return;
    const int32_t t = blk_sn[b];
    const int b0 = blk_b0[b];
    const int s = T.sn_start[t + 1] - T.sn_start[t];
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const int nb = s - b0 < DB ? s - b0 : DB;
    double *F = fronts + T.front_off[t] + (int64_t)b0 * dim + b0;
    const int lane = threadIdx.x;
    for (int e = lane; e < nb * nb; e += 64) {
        const int r = e % nb, c = e / nb;
        D[r][c] = F[r + (int64_t)c * dim];
    }
    __syncthreads();
    const int j = lane & 31;
    if (j < nb) {
        double *x = X[lane];
        if (lane < 32) {  // U x = e_j
            x[j] = 1.0 / D[j][j];
            for (int i = j - 1; i >= 0; --i) {
                double acc = 0.0;
                for (int k = i + 1; k <= j; ++k) acc = fma(D[i][k], x[k], acc);
                x[i] = -acc / D[i][i];
            }
        } else {          // L x = e_j, unit diagonal
            x[j] = 1.0;
            for (int i = j + 1; i < nb; ++i) {
                double acc = 0.0;
                for (int k = j; k < i; ++k) acc = fma(D[i][k], x[k], acc);
                x[i] = -acc;
            }
        }
    }
    __syncthreads();
    for (int e = lane; e < nb * nb; e += 64) {
        const int r = e % nb, c = e / nb;
        F[r + (int64_t)c * dim] = r <= c ? X[c][r] : X[32 + c][r];
    }
}

// liperm[start + lperm[start + k]] = k for every front: the position a front's row takes after its interchanges
__global__ __launch_bounds__(TB) void invert_perm(int64_t n, const int32_t *__restrict__ sn_of_row,
                                                  const int32_t *__restrict__ sn_start, const int32_t *__restrict__ lperm,
                                                  int32_t *__restrict__ liperm) {
    for (int64_t k = (int64_t)blockIdx.x * TB + threadIdx.x; k < n; k += (int64_t)gridDim.x * TB) {
        const int32_t start = sn_start[sn_of_row[k]];
        liperm[start + lperm[k]] = (int32_t)(k - start);
    }
}

template <int NR> struct ColGroup { static constexpr int CW = NR >= 4 ? 4 : NR; static constexpr int CG = NR / CW; };

// Staging into LDS without a memory round trip per element: a loop `dst[e] = test ? src[e] : 0` keeps its test as a
// branch around the load and waits for every load before its LDS store (the level steps spent six dependent round
// trips, 5 of their 8.6 us, filling Li and Vb).  Here every load is unconditional (the address of an element that is
// not wanted is clamped to one that is), all of a thread's loads are requested first, and the stores follow.
// inverse factor of the diagonal block at (b0, b0): LOWER: its strictly lower part (unit diagonal implied), else its
// upper triangle with the diagonal; zeros elsewhere and beyond nb
template <int BS, bool LOWER>
__device__ __forceinline__ void stage_diag_block(const double *__restrict__ F, int64_t dim, int b0, int nb,
                                                 double (*D)[DB + 1], int tid) {
    constexpr int PER = (DB * DB + BS - 1) / BS;
    double t[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * BS, r = e % DB, k = (e / DB) % DB;
        const bool in = e < DB * DB && (LOWER ? (r < nb && k < r) : (k < nb && r <= k));
        t[u] = F[(b0 + (in ? r : 0)) + (int64_t)(b0 + (in ? k : 0)) * dim];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * BS, r = e % DB, k = (e / DB) % DB;
        const bool in = e < DB * DB && (LOWER ? (r < nb && k < r) : (k < nb && r <= k));
        if (e < DB * DB) D[r][k] = in ? t[u] : 0.0;
    }
}
// ROWS x NR block of the interleaved vector starting at src (count = wanted rows * NR elements; zeros behind them)
template <int BS, int NR, int ROWS>
__device__ __forceinline__ void stage_rows(const double *__restrict__ src, int count, double (*dst)[NR], int tid) {
    constexpr int PER = (ROWS * NR + BS - 1) / BS;
    double t[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * BS;
        t[u] = src[e < count ? e : 0];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * BS;
        if (e < ROWS * NR) dst[e / NR][e % NR] = e < count ? t[u] : 0.0;
    }
}
// the chunk of x_B behind the boundary indices bidx[0 .. nb): rows of the ancestors' solution xb (two dependent loads
// per element, all of a thread's in flight together)
template <int BS, int NR>
__device__ __forceinline__ void stage_boundary(const double *__restrict__ xb, const int32_t *__restrict__ bidx, int nb,
                                               double (*Y)[NR], int tid) {
    constexpr int PER = (DB * NR + BS - 1) / BS;
    int32_t bi[PER];
    double t[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * BS;
        bi[u] = bidx[e < nb * NR ? e / NR : 0];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * BS;
        t[u] = xb[(int64_t)bi[u] * NR + e % NR];
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int e = tid + u * BS;
        if (e < DB * NR) Y[e / NR][e % NR] = e < nb * NR ? t[u] : 0.0;
    }
}

// acc[0 .. CW) -= sum_k F[i + (c0 + k) dim] * Y[k][col0 ..]: the nb (<= DB) entries of row i in the column chunk c0,
// ALL requested before the first is used -- one round trip to memory per chunk, not one per entry (a loop that
// loads, multiplies and loads again took 1.1 ms per sweep of a 1489-row front).  Y: the chunk's vector rows, in LDS.
// UNCOND (the kernels of the wide fronts): a short last chunk is read with clamped, unconditional loads too (the tests
// of the other form stay branches around the loads, one round trip each; the small fronts' kernels keep that form: most
// of their chunks are short and other workgroups hide the round trips)
template <int NR, int CW, bool UNCOND = false>
__device__ __forceinline__ void chunk_update(const double *__restrict__ Frow, int64_t dim, int nb, const double (*Y)[NR],
                                             int col0, double (&acc)[CW]) {
    double l[DB];
    if (nb == DB) {
#pragma unroll
        for (int k = 0; k < DB; ++k) l[k] = Frow[(int64_t)k * dim];
    } else if (UNCOND) {
#pragma unroll
        for (int k = 0; k < DB; ++k) l[k] = Frow[(int64_t)(k < nb ? k : 0) * dim];  // (nb >= 1)
#pragma unroll
        for (int k = 0; k < DB; ++k) l[k] = k < nb ? l[k] : 0.0;
    } else {
#pragma unroll
        for (int k = 0; k < DB; ++k) l[k] = k < nb ? Frow[(int64_t)k * dim] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < DB; ++k)
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) acc[jj] = fma(-l[k], Y[k][col0 + jj], acc[jj]);
}

// forward substitution of one level: v = [b'_S ; 0] + children's contributions, rows permuted like the
// factorisation's, y_S = L11^-1 v_S, v_B -= L21 y_S.  y_S stays in v[0:s), the contribution in v[s:dim).
// v = [b'_S ; 0] + the children's contributions, rows permuted like the factorisation's (the first part of a front's
// forward substitution; ends with the workgroup in step)
template <int BS, int NR>
__device__ __forceinline__ void forward_gather(const Tree &T, int32_t t, const int32_t *__restrict__ lperm,
                                               const double *__restrict__ xb, double *__restrict__ vec) {
    const int start = T.sn_start[t];
    const int s = T.sn_start[t + 1] - start;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    double *v = vec + T.vec_off[t] * NR;
    const int tid = threadIdx.x;
    // (plain loops: this runs with one workgroup per front on levels of thousands of small fronts, where more
    // workgroups in flight hide the round trips better than more loads per thread -- batching them was measured: 149 ->
    // 197 us per level launch)
    for (int e = tid; e < dim * NR; e += BS) v[e] = e < s * NR ? xb[(int64_t)start * NR + e] : 0.0;
    __syncthreads();
    for (int32_t q = T.child_ptr[t]; q < T.child_ptr[t + 1]; ++q) {
        const int32_t c = T.child_idx[q];
        const int cs_ = T.sn_start[c + 1] - T.sn_start[c];
        const int64_t cb = T.struct_ptr[c + 1] - T.struct_ptr[c];
        const double *cv = vec + (T.vec_off[c] + cs_) * NR;
        const int32_t *map = T.cmap + T.struct_ptr[c];
        for (int64_t e = tid; e < cb * NR; e += BS) v[(int64_t)map[e / NR] * NR + e % NR] += cv[e];
        __syncthreads();
    }
    // the row interchanges of the factorisation (through the s scratch rows behind the front's vector)
    const int32_t *perm = lperm + start;
    double *tmp = v + (int64_t)dim * NR;
    for (int e = tid; e < s * NR; e += BS) tmp[e] = v[(int64_t)perm[e / NR] * NR + e % NR];
    __syncthreads();
    for (int e = tid; e < s * NR; e += BS) v[e] = tmp[e];
    __syncthreads();
}

// forward substitution of one level: v = [b'_S ; 0] + children's contributions, rows permuted like the
// factorisation's, y_S = L11^-1 v_S, v_B -= L21 y_S.  y_S stays in v[0:s), the contribution in v[s:dim).
template <int BS, int NR>
__global__ __launch_bounds__(BS) void forward_level(Tree T, const int32_t *__restrict__ sns,
                                                    const double *__restrict__ fronts,
                                                    const int32_t *__restrict__ lperm, const double *__restrict__ xb,
                                                    double *__restrict__ vec) {
    constexpr int CW = ColGroup<NR>::CW, CG = ColGroup<NR>::CG;
    __shared__ double Y[DB][NR];       // the block's solved rows (zero rows behind a short last block)
    __shared__ double Vb[DB][NR];      // ... before the multiplication by inv(L_bb)
    __shared__ double Li[DB][DB + 1];  // inv(L_bb)
    const int32_t t = sns[blockIdx.x];
    const int start = T.sn_start[t];
    const int s = T.sn_start[t + 1] - start;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const double *F = fronts + T.front_off[t];
    double *v = vec + T.vec_off[t] * NR;
    const int tid = threadIdx.x;
    forward_gather<BS, NR>(T, t, lperm, xb, vec);
    for (int b0 = 0; b0 < s; b0 += DB) {
        const int nb = s - b0 < DB ? s - b0 : DB;
        // the block and its rows of v into LDS (coalesced), the product from there
        // (plain loops here and in backward_level: levels of thousands of small fronts, mostly with a handful of pivot
        // columns -- the tests skip most of the loads, and other workgroups hide the round trips of the rest)
        for (int e = tid; e < DB * DB; e += BS) {
            const int r = e % DB, k = e / DB;
            Li[r][k] = (r < nb && k < r) ? F[(b0 + r) + (int64_t)(b0 + k) * dim] : 0.0;
        }
        for (int e = tid; e < DB * NR; e += BS) Vb[e / NR][e % NR] = e < nb * NR ? v[(int64_t)b0 * NR + e] : 0.0;
        __syncthreads();
        for (int e = tid; e < DB * NR; e += BS) {  // y_b = inv(L_bb) v_b (unit diagonal)
            const int r = e / NR, c = e % NR;
            double acc = Vb[r][c];
#pragma unroll
            for (int k = 0; k < DB; ++k) acc = fma(Li[r][k], Vb[k][c], acc);
            Y[r][c] = acc;
            if (r < nb) v[(int64_t)(b0 + r) * NR + c] = acc;
        }
        __syncthreads();
        const int below = b0 + nb, rows = dim - below;
        for (int e = tid; e < rows * CG; e += BS) {
            const int i = below + e % rows, g = e / rows;
            double acc[CW];
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) acc[jj] = v[(int64_t)i * NR + g * CW + jj];
            chunk_update<NR, CW>(F + i + (int64_t)b0 * dim, dim, nb, Y, g * CW, acc);
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
        }
        __syncthreads();
    }
}

// backward substitution of one level: x_S = U11^-1 (y_S - U12 x_B), x_B gathered from the ancestors' solution
template <int BS, int NR>
__global__ __launch_bounds__(BS) void backward_level(Tree T, const int32_t *__restrict__ sns,
                                                     const double *__restrict__ fronts, double *__restrict__ xb,
                                                     double *__restrict__ vec) {
    constexpr int CW = ColGroup<NR>::CW, CG = ColGroup<NR>::CG;
    __shared__ double Y[DB][NR];
    __shared__ double Vb[DB][NR];
    __shared__ double Ui[DB][DB + 1];
    const int32_t t = sns[blockIdx.x];
    const int start = T.sn_start[t];
    const int s = T.sn_start[t + 1] - start;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const double *F = fronts + T.front_off[t];
    double *v = vec + T.vec_off[t] * NR;
    const int32_t *bidx = T.struct_idx + T.struct_ptr[t];
    const int tid = threadIdx.x;
    // v_S -= U12 x_B, the boundary's columns in chunks of DB: the chunk of x_B goes to LDS straight from the ancestors'
    // solution, every thread then takes its row's entries of the chunk in one round trip
    for (int c0 = s; c0 < dim; c0 += DB) {
        const int nb = dim - c0 < DB ? dim - c0 : DB;
        for (int e = tid; e < DB * NR; e += BS)
            Y[e / NR][e % NR] = e < nb * NR ? xb[(int64_t)bidx[c0 - s + e / NR] * NR + e % NR] : 0.0;
        __syncthreads();
        for (int e = tid; e < s * CG; e += BS) {
            const int i = e % s, g = e / s;
            double acc[CW];
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) acc[jj] = v[(int64_t)i * NR + g * CW + jj];
            chunk_update<NR, CW>(F + i + (int64_t)c0 * dim, dim, nb, Y, g * CW, acc);
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
        }
        __syncthreads();
    }
    for (int b0 = ((s - 1) / DB) * DB; b0 >= 0; b0 -= DB) {
        const int nb = s - b0 < DB ? s - b0 : DB;
        for (int e = tid; e < DB * DB; e += BS) {
            const int r = e % DB, k = e / DB;
            Ui[r][k] = (k < nb && r <= k) ? F[(b0 + r) + (int64_t)(b0 + k) * dim] : 0.0;
        }
        for (int e = tid; e < DB * NR; e += BS) Vb[e / NR][e % NR] = e < nb * NR ? v[(int64_t)b0 * NR + e] : 0.0;
        __syncthreads();
        for (int e = tid; e < DB * NR; e += BS) {  // x_b = inv(U_bb) v_b
            const int r = e / NR, c = e % NR;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < DB; ++k) acc = fma(Ui[r][k], Vb[k][c], acc);
            Y[r][c] = acc;
            if (r < nb) v[(int64_t)(b0 + r) * NR + c] = acc;
        }
        __syncthreads();
        for (int e = tid; e < b0 * CG; e += BS) {  // the rows above
            const int i = e % b0, g = e / b0;
            double acc[CW];
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) acc[jj] = v[(int64_t)i * NR + g * CW + jj];
            chunk_update<NR, CW>(F + i + (int64_t)b0 * dim, dim, nb, Y, g * CW, acc);
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
        }
        __syncthreads();
    }
    for (int e = tid; e < s * NR; e += BS) xb[(int64_t)start * NR + e] = v[e];
}

// ---- the WIDE fronts of a level substituted together, block step by block step (round 5) --------------------------
// One workgroup per front leaves a 1489-row front to one compute unit: 0.65 ms per sweep of each of the top levels with
// sixteen right-hand sides.  Here step b0 of a level is ONE launch over (row tiles, fronts): every workgroup forms the
// block's solved rows itself (a 32 x 32 product from LDS: cheaper than a launch of its own) and updates its tile of
// the rows below (forward) / above (backward).  The solved rows go to the front's scratch rows (forward: read back by
// the backward sweep) or straight into the solution (backward), never into rows another workgroup of the step reads.
constexpr int APPLY_BIG = 256;  // fronts wider than this take the stepped form

constexpr int GATHER_RT = 64;
// v = [b'_S ; 0] + the children's contributions, every row written straight to the place the factorisation's row
// interchanges give it (liperm): row tiles x fronts.  A workgroup owns the destination rows [r0, r0 + RT) and takes, child
// after child (two children may add to one row: a fixed order), the child rows that land there.
template <int NR>
__global__ __launch_bounds__(256) void level_fwd_gather(Tree T, const int32_t *__restrict__ sns,
                                                        const int32_t *__restrict__ liperm, const double *__restrict__ xb,
                                                        double *__restrict__ vec) {
    constexpr int RT = GATHER_RT;  // destination rows per workgroup
    const int32_t t = sns[blockIdx.y];
    const int start = T.sn_start[t];
    const int s = T.sn_start[t + 1] - start;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const int r0 = (int)blockIdx.x * RT;
    if (r0 >= dim) return;
    double *v = vec + T.vec_off[t] * NR;
    const int32_t *ip = liperm + start;
    const int tid = threadIdx.x;
    // (a destination row below s holds the right-hand side of the source row that is interchanged into it)
    for (int e = tid; e < RT * NR; e += 256)
        if (r0 + e / NR < dim) v[(int64_t)r0 * NR + e] = 0.0;
    __syncthreads();
    for (int e = tid; e < s * NR; e += 256) {  // (s source rows: the ones that land in this tile)
        const int pos = ip[e / NR];
        if (pos >= r0 && pos < r0 + RT) v[(int64_t)pos * NR + e % NR] = xb[(int64_t)start * NR + e];
    }
    __syncthreads();
    for (int32_t q = T.child_ptr[t]; q < T.child_ptr[t + 1]; ++q) {
        const int32_t c = T.child_idx[q];
        const int cs_ = T.sn_start[c + 1] - T.sn_start[c];
        const int64_t cb = T.struct_ptr[c + 1] - T.struct_ptr[c];
        const double *cv = vec + (T.vec_off[c] + cs_) * NR;
        const int32_t *map = T.cmap + T.struct_ptr[c];
        for (int64_t e = tid; e < cb * NR; e += 256) {
            const int m = map[e / NR];
            const int pos = m < s ? ip[m] : m;
            if (pos >= r0 && pos < r0 + RT) v[(int64_t)pos * NR + e % NR] += cv[e];
        }
        __syncthreads();
    }
}

template <int NR>
__global__ __launch_bounds__(256) void level_fwd_step(Tree T, const int32_t *__restrict__ sns, const double *__restrict__ fronts,
                                                      double *__restrict__ vec, int b0) {
    constexpr int CW = ColGroup<NR>::CW, CG = ColGroup<NR>::CG, RT = 256 / CG;  // rows per workgroup
    __shared__ double Y[DB][NR];
    __shared__ double Vb[DB][NR];
    __shared__ double Li[DB][DB + 1];
    const int32_t t = sns[blockIdx.y];
    const int s = T.sn_start[t + 1] - T.sn_start[t];
    if (b0 >= s) return;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const int nb = s - b0 < DB ? s - b0 : DB;
    const int below = b0 + nb, rows = dim - below;
    const int r0 = (int)blockIdx.x * RT;
    if (blockIdx.x > 0 && r0 >= rows) return;
    const double *F = fronts + T.front_off[t];
    double *v = vec + T.vec_off[t] * NR;
    const int tid = threadIdx.x;
    stage_diag_block<256, true>(F, dim, b0, nb, Li, tid);
    stage_rows<256, NR, DB>(v + (int64_t)b0 * NR, nb * NR, Vb, tid);
    __syncthreads();
    double *ys = v + (int64_t)dim * NR;  // (the scratch rows: y_S of the whole front ends there)
    for (int e = tid; e < DB * NR; e += 256) {
        const int r = e / NR, c = e % NR;
        double acc = Vb[r][c];
#pragma unroll
        for (int k = 0; k < DB; ++k) acc = fma(Li[r][k], Vb[k][c], acc);
        Y[r][c] = acc;
        if (blockIdx.x == 0 && r < nb) ys[(int64_t)(b0 + r) * NR + c] = acc;
    }
    __syncthreads();
    const int i = below + r0 + tid % RT, g = tid / RT;
    if (i < dim) {
        double acc[CW];
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) acc[jj] = v[(int64_t)i * NR + g * CW + jj];
        chunk_update<NR, CW, true>(F + i + (int64_t)b0 * dim, dim, nb, Y, g * CW, acc);
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
    }
}

// v_S = y_S - U12 x_B (y_S from the scratch rows, x_B from the ancestors' solution): row tiles x fronts
template <int NR>
__global__ __launch_bounds__(256) void level_bwd_u12(Tree T, const int32_t *__restrict__ sns, const double *__restrict__ fronts,
                                                     const double *__restrict__ xb, double *__restrict__ vec) {
    constexpr int CW = ColGroup<NR>::CW, CG = ColGroup<NR>::CG, RT = 256 / CG;
    __shared__ double Y[DB][NR];
    const int32_t t = sns[blockIdx.y];
    const int s = T.sn_start[t + 1] - T.sn_start[t];
    const int r0 = (int)blockIdx.x * RT;
    if (r0 >= s) return;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const double *F = fronts + T.front_off[t];
    double *v = vec + T.vec_off[t] * NR;
    const double *ys = v + (int64_t)dim * NR;
    const int32_t *bidx = T.struct_idx + T.struct_ptr[t];
    const int tid = threadIdx.x;
    const int i = r0 + tid % RT, g = tid / RT;
    double acc[CW];
#pragma unroll
    for (int jj = 0; jj < CW; ++jj) acc[jj] = i < s ? ys[(int64_t)i * NR + g * CW + jj] : 0.0;
    for (int c0 = s; c0 < dim; c0 += DB) {
        const int nb = dim - c0 < DB ? dim - c0 : DB;
        __syncthreads();
        stage_boundary<256, NR>(xb, bidx + (c0 - s), nb, Y, tid);
        __syncthreads();
        if (i < s) chunk_update<NR, CW, true>(F + i + (int64_t)c0 * dim, dim, nb, Y, g * CW, acc);
    }
    if (i < s) {
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
    }
}

template <int NR>
__global__ __launch_bounds__(256) void level_bwd_step(Tree T, const int32_t *__restrict__ sns, const double *__restrict__ fronts,
                                                      double *__restrict__ xb, double *__restrict__ vec, int b0) {
    constexpr int CW = ColGroup<NR>::CW, CG = ColGroup<NR>::CG, RT = 256 / CG;
    __shared__ double Y[DB][NR];
    __shared__ double Vb[DB][NR];
    __shared__ double Ui[DB][DB + 1];
    const int32_t t = sns[blockIdx.y];
    const int start = T.sn_start[t];
    const int s = T.sn_start[t + 1] - start;
    if (b0 >= s) return;
    const int r0 = (int)blockIdx.x * RT;
    if (blockIdx.x > 0 && r0 >= b0) return;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const int nb = s - b0 < DB ? s - b0 : DB;
    const double *F = fronts + T.front_off[t];
    double *v = vec + T.vec_off[t] * NR;
    const int tid = threadIdx.x;
    stage_diag_block<256, false>(F, dim, b0, nb, Ui, tid);
    stage_rows<256, NR, DB>(v + (int64_t)b0 * NR, nb * NR, Vb, tid);
    __syncthreads();
    for (int e = tid; e < DB * NR; e += 256) {
        const int r = e / NR, c = e % NR;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DB; ++k) acc = fma(Ui[r][k], Vb[k][c], acc);
        Y[r][c] = acc;
        if (blockIdx.x == 0 && r < nb) xb[(int64_t)(start + b0 + r) * NR + c] = acc;  // (the solution itself)
    }
    __syncthreads();
    const int i = r0 + tid % RT, g = tid / RT;
    if (i < b0) {
        double acc[CW];
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) acc[jj] = v[(int64_t)i * NR + g * CW + jj];
        chunk_update<NR, CW, true>(F + i + (int64_t)b0 * dim, dim, nb, Y, g * CW, acc);
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
    }
}

// ---- super steps: SUPER_SB diagonal blocks (128 pivot columns) of every wide front of the level per launch -------
// A block step is a launch of ~8 us whatever it moves (117 of them per sweep of config 5's factors).  Here a workgroup
// solves the super block's rows ITSELF, sub-block after sub-block from LDS (the triangle inside the super block is
// 64 KB of the front, read by every workgroup of the step: it stays in the L2), and then updates its tile of the rows
// outside the super block with all of its columns at once: a quarter of the launches, the same sums in the same order
// per row (sub-block after sub-block), hence the same bits as the block steps.
constexpr int SUPER_SB = 4, SUPER_W = SUPER_SB * DB;

template <int NR>
__global__ __launch_bounds__(256) void level_fwd_super(Tree T, const int32_t *__restrict__ sns, const double *__restrict__ fronts,
                                                       double *__restrict__ vec, int B0) {
    constexpr int CW = ColGroup<NR>::CW, CG = ColGroup<NR>::CG, RT = 256 / CG;  // rows per workgroup
    __shared__ double Y[SUPER_W][NR];   // the super block's solved rows
    __shared__ double Vs[SUPER_W][NR];  // its rows of v, updated sub-block by sub-block
    __shared__ double Li[DB][DB + 1];
    const int32_t t = sns[blockIdx.y];
    const int s = T.sn_start[t + 1] - T.sn_start[t];
    if (B0 >= s) return;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const int sw = s - B0 < SUPER_W ? s - B0 : SUPER_W;  // pivot columns of this super block
    const int below = B0 + sw, rows = dim - below;
    const int r0 = (int)blockIdx.x * RT;
    if (blockIdx.x > 0 && r0 >= rows) return;
    const double *F = fronts + T.front_off[t];
    double *v = vec + T.vec_off[t] * NR;
    double *ys = v + (int64_t)dim * NR;  // (the scratch rows: y_S of the whole front ends there)
    const int tid = threadIdx.x;
    stage_rows<256, NR, SUPER_W>(v + (int64_t)B0 * NR, sw * NR, Vs, tid);
    for (int e = tid; e < SUPER_W * NR; e += 256) Y[e / NR][e % NR] = 0.0;
    for (int k = 0; k * DB < sw; ++k) {
        const int b0 = B0 + k * DB, nb = s - b0 < DB ? s - b0 : DB;
        __syncthreads();
        stage_diag_block<256, true>(F, dim, b0, nb, Li, tid);
        __syncthreads();
        for (int e = tid; e < DB * NR; e += 256) {  // y_b = inv(L_bb) v_b (unit diagonal)
            const int r = e / NR, c = e % NR;
            double acc = Vs[k * DB + r][c];
#pragma unroll
            for (int q = 0; q < DB; ++q) acc = fma(Li[r][q], Vs[k * DB + q][c], acc);
            Y[k * DB + r][c] = acc;
            if (blockIdx.x == 0 && r < nb) ys[(int64_t)(b0 + r) * NR + c] = acc;
        }
        __syncthreads();
        // the rows of the super block below this sub-block
        const int ra = sw - (k + 1) * DB;
        for (int e = tid; e < ra * CG; e += 256) {
            const int il = (k + 1) * DB + e % ra, g = e / ra;
            double acc[CW];
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) acc[jj] = Vs[il][g * CW + jj];
            chunk_update<NR, CW, true>(F + (B0 + il) + (int64_t)b0 * dim, dim, nb, (const double (*)[NR]) & Y[k * DB], g * CW, acc);
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) Vs[il][g * CW + jj] = acc[jj];
        }
    }
    __syncthreads();
    const int i = below + r0 + tid % RT, g = tid / RT;
    if (i < dim) {
        double acc[CW];
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) acc[jj] = v[(int64_t)i * NR + g * CW + jj];
        for (int k = 0; k * DB < sw; ++k) {
            const int b0 = B0 + k * DB, nb = s - b0 < DB ? s - b0 : DB;
            chunk_update<NR, CW, true>(F + i + (int64_t)b0 * dim, dim, nb, (const double (*)[NR]) & Y[k * DB], g * CW, acc);
        }
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
    }
}

// backward: the super block [B0, B0 + sw) from its last sub-block up, then the rows above it
template <int NR>
__global__ __launch_bounds__(256) void level_bwd_super(Tree T, const int32_t *__restrict__ sns, const double *__restrict__ fronts,
                                                       double *__restrict__ xb, double *__restrict__ vec, int B0) {
    constexpr int CW = ColGroup<NR>::CW, CG = ColGroup<NR>::CG, RT = 256 / CG;
    __shared__ double Y[SUPER_W][NR];
    __shared__ double Vs[SUPER_W][NR];
    __shared__ double Ui[DB][DB + 1];
    const int32_t t = sns[blockIdx.y];
    const int start = T.sn_start[t];
    const int s = T.sn_start[t + 1] - start;
    if (B0 >= s) return;
    const int r0 = (int)blockIdx.x * RT;
    if (blockIdx.x > 0 && r0 >= B0) return;
    const int dim = s + (int)(T.struct_ptr[t + 1] - T.struct_ptr[t]);
    const int sw = s - B0 < SUPER_W ? s - B0 : SUPER_W;
    const double *F = fronts + T.front_off[t];
    double *v = vec + T.vec_off[t] * NR;
    const int tid = threadIdx.x;
    stage_rows<256, NR, SUPER_W>(v + (int64_t)B0 * NR, sw * NR, Vs, tid);
    for (int e = tid; e < SUPER_W * NR; e += 256) Y[e / NR][e % NR] = 0.0;
    for (int k = (sw - 1) / DB; k >= 0; --k) {
        const int b0 = B0 + k * DB, nb = s - b0 < DB ? s - b0 : DB;
        __syncthreads();
        stage_diag_block<256, false>(F, dim, b0, nb, Ui, tid);
        __syncthreads();
        for (int e = tid; e < DB * NR; e += 256) {  // x_b = inv(U_bb) v_b
            const int r = e / NR, c = e % NR;
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < DB; ++q) acc = fma(Ui[r][q], Vs[k * DB + q][c], acc);
            Y[k * DB + r][c] = acc;
            if (blockIdx.x == 0 && r < nb) xb[(int64_t)(start + b0 + r) * NR + c] = acc;  // (the solution itself)
        }
        __syncthreads();
        const int ra = k * DB;  // the rows of the super block above this sub-block
        for (int e = tid; e < ra * CG; e += 256) {
            const int il = e % ra, g = e / ra;
            double acc[CW];
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) acc[jj] = Vs[il][g * CW + jj];
            chunk_update<NR, CW, true>(F + (B0 + il) + (int64_t)b0 * dim, dim, nb, (const double (*)[NR]) & Y[k * DB], g * CW, acc);
#pragma unroll
            for (int jj = 0; jj < CW; ++jj) Vs[il][g * CW + jj] = acc[jj];
        }
    }
    __syncthreads();
    const int i = r0 + tid % RT, g = tid / RT;
    if (i < B0) {
        double acc[CW];
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) acc[jj] = v[(int64_t)i * NR + g * CW + jj];
        for (int k = (sw - 1) / DB; k >= 0; --k) {
            const int b0 = B0 + k * DB, nb = s - b0 < DB ? s - b0 : DB;
            chunk_update<NR, CW, true>(F + i + (int64_t)b0 * dim, dim, nb, (const double (*)[NR]) & Y[k * DB], g * CW, acc);
        }
#pragma unroll
        for (int jj = 0; jj < CW; ++jj) v[(int64_t)i * NR + g * CW + jj] = acc[jj];
    }
}

template <class T>
int upload_vec(nodal_ctx *h, DevBuf &buf, const std::vector<T> &v) {
    NODAL_HIP_TRY(h, buf.reserve(v.size() * sizeof(T) + 64));
    if (!v.empty())
        NODAL_HIP_TRY(h, hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, h->stream));
    return NODAL_OK;
}

Tree tree_of(const SluState *S) {
    Tree T;
    T.sn_start = S->sn_start.as<int32_t>();
    T.struct_ptr = S->struct_ptr.as<int64_t>();
    T.struct_idx = S->struct_idx.as<int32_t>();
    T.front_off = S->front_off.as<int64_t>();
    T.vec_off = S->vec_off.as<int64_t>();
    T.child_ptr = S->child_ptr.as<int32_t>();
    T.child_idx = S->child_idx.as<int32_t>();
    T.cmap = S->cmap.as<int32_t>();
    return T;
}

}  // namespace

// NODAL_POISON=2: the numeric part (fronts, vectors, interchanges, scales) holds nothing between solves
void slu_poison(nodal_ctx *h) {
    SluState *S = state_of(h);
    if (!S) return;
    DevBuf *bufs[] = {&S->fronts, &S->vec, &S->lperm, &S->rs, &S->cs, &S->xb, &S->stats};
    for (DevBuf *b : bufs) b->poison(h->stream);
    S->have_numeric = false;
}

void slu_destroy(nodal_ctx *h) {
    SluState *S = state_of(h);
    if (!S) return;
    DevBuf *bufs[] = {&S->rowof, &S->colof, &S->newrow, &S->sn_start, &S->struct_ptr, &S->struct_idx, &S->front_off,
                      &S->vec_off, &S->lvl_sn, &S->child_ptr, &S->child_idx, &S->cmap, &S->dest, &S->fronts,
                      &S->vec, &S->lperm, &S->rs, &S->cs, &S->xb, &S->stats, &S->bigpiv, &S->blk_sn, &S->blk_b0, &S->liperm, &S->sn_of_row};
    for (DevBuf *b : bufs) b->release();
    for (int k = 1; k < SluState::LANES; ++k)
        if (S->lane_st[k]) (void)hipStreamDestroy(S->lane_st[k]);
    for (int k = 0; k < SluState::LANES; ++k)
        if (S->lane_ev[k]) (void)hipEventDestroy(S->lane_ev[k]);
    delete S;
    h->slu = nullptr;
}

// Analysis (kept per struct_epoch) + numeric factorisation of the context's CSR matrix.
// *info = 1: structurally singular (no perfect matching).
int slu_factor(nodal_ctx *h, int32_t *info, double tiny_factor, double tiny_threshold) {
    *info = 0;
    const bool trace = getenv("NODAL_TRACE") != nullptr;
    const int64_t n = h->n, nnz = h->nnz;
    hipStream_t st = h->stream;
    if (n >= (1ll << 31) - 2 || nnz >= (1ll << 31) - 2) return nodal_fail(h, NODAL_E_UNSUPPORTED, "direct solve: more than 2^31 rows or entries");
    SluState *S = state_of(h);
    if (!S) {
        S = new SluState();
        h->slu = S;
    }
    S->perturbed = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto ms_since = [&](auto t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
    S->analysis_kept = S->have_symbolic && S->epoch == h->struct_epoch && S->n == n && S->nnz == nnz;
    if (!S->analysis_kept) {
        S->have_symbolic = S->have_numeric = false;
        std::vector<int32_t> indptr((size_t)n + 1), indices((size_t)nnz);
        std::vector<double> data((size_t)nnz);
        NODAL_HIP_TRY(h, hipMemcpyAsync(indptr.data(), h->indptr.p, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost, st));
        if (nnz) {
            NODAL_HIP_TRY(h, hipMemcpyAsync(indices.data(), h->indices.p, (size_t)nnz * 4, hipMemcpyDeviceToHost, st));
            NODAL_HIP_TRY(h, hipMemcpyAsync(data.data(), h->data.p, (size_t)nnz * 8, hipMemcpyDeviceToHost, st));
        }
        NODAL_WAIT_STREAM(h, st);
        Symbolic sym;
        if (!analyse(n, indptr.data(), indices.data(), data.data(), sym, trace)) {
            if (trace) fprintf(stderr, "[direct] no perfect matching: structurally singular\n");
            *info = 1;
            return NODAL_OK;
        }
        S->nsn = (int32_t)sym.sn_start.size() - 1;
        S->nlev = (int32_t)sym.lvl_ptr.size() - 1;
        S->front_doubles = sym.front_off.back();
        S->vec_doubles = sym.vec_off.back();
        S->max_dim = sym.max_dim;
        S->lvl_ptr = sym.lvl_ptr;
        S->lvl_maxdim = sym.lvl_maxdim;
        S->lvl_maxdim_all = sym.lvl_maxdim;  // (substitution kernels walk ALL fronts of a level)
        {   // inside every level: the fronts of the batched kernel first, the wide ones last
            const int32_t nsn = S->nsn;
            S->h_start.assign(sym.sn_start.begin(), sym.sn_start.end());
            S->h_front_off = sym.front_off;
            S->h_dim.resize((size_t)nsn);
            for (int32_t t = 0; t < nsn; ++t)
                S->h_dim[(size_t)t] = (int32_t)((sym.sn_start[(size_t)t + 1] - sym.sn_start[(size_t)t]) +
                                                (sym.struct_ptr[(size_t)t + 1] - sym.struct_ptr[(size_t)t]));
            S->lvl_small.assign((size_t)S->nlev, 0);
            S->lvl_abig.clear(); S->lvl_amax_s.clear(); S->lvl_amax_dim.clear(); S->lvl_arest_dim.clear();
            S->lvl_maxchildren.assign((size_t)S->nlev, 0);
            for (int32_t l = 0; l < S->nlev; ++l) {
                auto b = sym.lvl_sn.begin() + sym.lvl_ptr[(size_t)l], e = sym.lvl_sn.begin() + sym.lvl_ptr[(size_t)l + 1];
                // (NODAL_DIRECT_BIG_DIM: the front width above which a front is stepped with its level's wide ones)
                static const int big_dim = getenv("NODAL_DIRECT_BIG_DIM") ? atoi(getenv("NODAL_DIRECT_BIG_DIM")) : BIG_DIM;
                auto mid = std::stable_partition(b, e, [&](int32_t t) { return S->h_dim[(size_t)t] <= big_dim; });
                S->lvl_small[(size_t)l] = (int32_t)(mid - b);
                auto amid = std::stable_partition(mid, e, [&](int32_t t) { return S->h_dim[(size_t)t] <= APPLY_BIG; });
                int32_t am_s = 0, am_d = 0, rest_d = 0;
                for (auto it = b; it != e; ++it) {
                    if (it < amid) rest_d = std::max(rest_d, S->h_dim[(size_t)*it]);
                    else {
                        am_s = std::max(am_s, S->h_start[(size_t)*it + 1] - S->h_start[(size_t)*it]);
                        am_d = std::max(am_d, S->h_dim[(size_t)*it]);
                    }
                }
                S->lvl_abig.push_back((int32_t)(e - amid));
                S->lvl_amax_s.push_back(am_s);
                S->lvl_amax_dim.push_back(am_d);
                S->lvl_arest_dim.push_back(rest_d);
                int32_t small_max = 0;
                for (auto it = b; it != e; ++it) {
                    if (it < mid) small_max = std::max(small_max, S->h_dim[(size_t)*it]);
                    S->lvl_maxchildren[(size_t)l] = std::max(S->lvl_maxchildren[(size_t)l],
                                                            sym.child_ptr[(size_t)*it + 1] - sym.child_ptr[(size_t)*it]);
                }
                S->lvl_maxdim[(size_t)l] = small_max;  // (of the batched kernel's fronts: selects its block size)
            }
            S->h_lvl_sn = sym.lvl_sn;
        }
        std::vector<int32_t> blk_sn, blk_b0;
        for (int32_t t = 0; t < S->nsn; ++t)
            for (int32_t b0 = 0; b0 < S->h_start[(size_t)t + 1] - S->h_start[(size_t)t]; b0 += DB) {
                blk_sn.push_back(t);
                blk_b0.push_back(b0);
            }
        S->nblocks = (int64_t)blk_sn.size();
        // memory the fronts may take: half of what is free on the device (NODAL_DIRECT_MAX_GB overrides)
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        double cap = 0.5 * (double)free_b;
        if (const char *e = getenv("NODAL_DIRECT_MAX_GB")) cap = atof(e) * 1e9;
        if ((double)S->front_doubles * 8.0 > cap) {
            char msg[256];
            snprintf(msg, sizeof msg, "direct solve: the fronts of this matrix need %.1f GB (limit %.1f GB)",
                     (double)S->front_doubles * 8e-9, cap * 1e-9);
            return nodal_fail(h, NODAL_E_NOMEM, msg);
        }
        NODAL_TRY(upload_vec(h, S->rowof, sym.rowof));
        NODAL_TRY(upload_vec(h, S->colof, sym.colof));
        NODAL_TRY(upload_vec(h, S->newrow, sym.newrow));
        NODAL_TRY(upload_vec(h, S->sn_start, sym.sn_start));
        NODAL_TRY(upload_vec(h, S->struct_ptr, sym.struct_ptr));
        NODAL_TRY(upload_vec(h, S->struct_idx, sym.struct_idx));
        NODAL_TRY(upload_vec(h, S->front_off, sym.front_off));
        NODAL_TRY(upload_vec(h, S->vec_off, sym.vec_off));
        NODAL_TRY(upload_vec(h, S->lvl_sn, S->h_lvl_sn));
        NODAL_TRY(upload_vec(h, S->child_ptr, sym.child_ptr));
        NODAL_TRY(upload_vec(h, S->child_idx, sym.child_idx));
        NODAL_TRY(upload_vec(h, S->cmap, sym.cmap));
        NODAL_TRY(upload_vec(h, S->dest, sym.dest));
        {
            std::vector<int32_t> sn_of_row((size_t)n);
            for (int32_t t = 0; t < S->nsn; ++t)
                for (int32_t k = S->h_start[(size_t)t]; k < S->h_start[(size_t)t + 1]; ++k) sn_of_row[(size_t)k] = t;
            NODAL_TRY(upload_vec(h, S->sn_of_row, sn_of_row));
        }
        NODAL_TRY(upload_vec(h, S->blk_sn, blk_sn));
        NODAL_TRY(upload_vec(h, S->blk_b0, blk_b0));
        NODAL_WAIT_STREAM(h, st);  // (the host vectors go out of scope)
        S->epoch = h->struct_epoch;
        S->n = n;
        S->nnz = nnz;
        S->have_symbolic = true;
    }
    const double t_sym = ms_since(t0);
    // ---- numeric ----
    NODAL_HIP_TRY(h, S->fronts.reserve((size_t)S->front_doubles * 8 + 64));
    if (S->vec_nr < 1) S->vec_nr = 1;
    NODAL_HIP_TRY(h, S->vec.reserve((size_t)S->vec_doubles * 8 * S->vec_nr + 64));
    NODAL_HIP_TRY(h, S->lperm.reserve((size_t)n * 4 + 64));
    NODAL_HIP_TRY(h, S->liperm.reserve((size_t)n * 4 + 64));
    NODAL_HIP_TRY(h, S->rs.reserve((size_t)n * 8 + 64));
    NODAL_HIP_TRY(h, S->cs.reserve((size_t)n * 8 + 64));
    NODAL_HIP_TRY(h, S->xb.reserve((size_t)n * 8 * S->vec_nr + 64));
    NODAL_HIP_TRY(h, S->stats.reserve(64));
    const int32_t *indptr = h->indptr.as<int32_t>();
    const int32_t *indices = h->indices.as<int32_t>();
    const double *data = h->data.as<double>();
    NODAL_HIP_TRY(h, hipMemsetAsync(S->stats.p, 0, 16, st));
    NODAL_HIP_TRY(h, hipMemsetAsync(S->fronts.p, 0, (size_t)S->front_doubles * 8, st));
    // (the column maxima use cs as their integer scratch first)
    NODAL_HIP_TRY(h, hipMemsetAsync(S->xb.p, 0, (size_t)n * 8, st));
    row_scales<<<grid_for(n), TB, 0, st>>>(n, indptr, data, S->rs.as<double>());
    col_maxima<<<grid_for(n), TB, 0, st>>>(n, indptr, indices, data, S->rs.as<double>(), S->xb.as<unsigned long long>());
    col_scales<<<grid_for(n), TB, 0, st>>>(n, S->xb.as<unsigned long long>(), S->cs.as<double>());
    scatter_entries<<<grid_for(n), TB, 0, st>>>(n, indptr, indices, data, S->rs.as<double>(), S->cs.as<double>(),
                                                S->dest.as<int64_t>(), S->fronts.as<double>());
    NODAL_HIP_TRY(h, hipGetLastError());
    const Tree T = tree_of(S);
    // after the equilibration every row and column has max-norm <= 1: the static-pivot bound is sqrt(eps)
    // (tiny_threshold: the pivot magnitude below which a pivot is replaced -- sqrt(eps) by default, the static-pivot
    // rule; sparse_direct_solve's last resort lowers it so that small TRUE pivots of a badly scaled regular matrix are used)
    const double sqrt_eps = 1.4901161193847656e-08;
    const double tiny = tiny_threshold > 0.0 ? tiny_threshold : sqrt_eps, repl = sqrt_eps * tiny_factor;
    NODAL_HIP_TRY(h, S->bigpiv.reserve(SluState::LANES * NBB * 4 + 64));  // (one set of panel pivots per lane of wide fronts)
    // columns per panel of the wide fronts: 16 measured best (config 5 at 1e6 unknowns, analysis kept: 210 / 230 /
    // 270 / 260 ms for 16 / 32 / 48 / 64 -- the single-workgroup panel kernel is what a wider panel makes longer);
    // NODAL_DIRECT_NB = 16 / 32 / 48 / 64
    int panel_nb = 16;
    if (const char *e = getenv("NODAL_DIRECT_NB")) {
        const int v = atoi(e);
        if (v == 16 || v == 32 || v == 48 || v == 64) panel_nb = v;
    }
    // (NODAL_DIRECT_LANES=1: the wide fronts of a level one after the other on the main stream)
    int want_lanes = 4;  // (config 5's factorisation with 1 / 2 / 3 / 4 / 6 lanes: 93 / 73 / 67 / 65 / 83 ms)
    if (const char *e = getenv("NODAL_DIRECT_LANES")) want_lanes = atoi(e) < 1 ? 1 : (atoi(e) > SluState::LANES ? SluState::LANES : atoi(e));
    S->lane_st[0] = st;
    if (S->lanes < want_lanes && nodal_extra_streams_ok(h)) {  // streams and events of the lanes: once per context (a failure leaves fewer lanes)
        if (!S->lane_ev[0] && hipEventCreateWithFlags(&S->lane_ev[0], hipEventDisableTiming) != hipSuccess) S->lane_ev[0] = nullptr;
        for (int k = S->lanes; k < want_lanes && S->lane_ev[0]; ++k) {
            if (hipStreamCreateWithFlags(&S->lane_st[k], hipStreamNonBlocking) != hipSuccess) { S->lane_st[k] = nullptr; break; }
            if (hipEventCreateWithFlags(&S->lane_ev[k], hipEventDisableTiming) != hipSuccess) {
                (void)hipStreamDestroy(S->lane_st[k]);
                S->lane_st[k] = nullptr;
                S->lane_ev[k] = nullptr;
                break;
            }
            S->lanes = k + 1;
        }
        (void)hipGetLastError();
    }
    const int max_lanes = !nodal_extra_streams_ok(h) ? 1 : (S->lanes < want_lanes ? S->lanes : want_lanes);  // (see api.hip)
    const bool fronts_in_lds = !(getenv("NODAL_DIRECT_FRONT_LDS") && atoi(getenv("NODAL_DIRECT_FRONT_LDS")) == 0);
    const bool panel_regs = !(getenv("NODAL_DIRECT_PANEL_REGS") && atoi(getenv("NODAL_DIRECT_PANEL_REGS")) == 0);
    const bool batched = !(getenv("NODAL_DIRECT_BATCHED") && atoi(getenv("NODAL_DIRECT_BATCHED")) == 0);
    // NODAL_DIRECT_LEVELS=1: an event behind every level, and a table of times, flops and bytes per level on stderr
    const bool level_table = getenv("NODAL_DIRECT_LEVELS") != nullptr;
    std::vector<hipEvent_t> lev_ev;
    if (level_table) {
        lev_ev.resize((size_t)S->nlev + 1);
        for (auto &e : lev_ev) NODAL_HIP_TRY(h, hipEventCreate(&e));
        NODAL_HIP_TRY(h, hipEventRecord(lev_ev[0], st));
    }
    struct LevelEvents {  // (records the event of a level whatever way the loop body is left)
        std::vector<hipEvent_t> &ev; hipStream_t st; size_t at; bool on;
        ~LevelEvents() { if (on) (void)hipEventRecord(ev[at], st); }
    };
    int64_t big_fronts = 0;
    for (int32_t l = 0; l < S->nlev; ++l) {
        LevelEvents level_done{lev_ev, st, (size_t)l + 1, level_table};
        const int32_t cnt = S->lvl_ptr[(size_t)l + 1] - S->lvl_ptr[(size_t)l];
        const int32_t nsmall = S->lvl_small[(size_t)l], nbig = cnt - nsmall;
        const int32_t *sns = S->lvl_sn.as<int32_t>() + S->lvl_ptr[(size_t)l];
        const bool wide = S->lvl_maxdim[(size_t)l] > 192;
        if (l > 0 && nsmall > 0) {
            if (wide) extend_add<1024><<<nsmall, 1024, 0, st>>>(T, sns, S->fronts.as<double>());
            else extend_add<256><<<nsmall, 256, 0, st>>>(T, sns, S->fronts.as<double>());
        }
        if (l > 0 && nbig > 0)  // wide parents: one launch per child ordinal, columns over the grid's second dimension
            for (int32_t o = 0; o < S->lvl_maxchildren[(size_t)l]; ++o)
                extend_add_child<<<dim3((unsigned)nbig, 64), 256, 0, st>>>(T, sns + nsmall, o, S->fronts.as<double>());
        if (nsmall > 0) {
            if (wide) factor_fronts<1024><<<nsmall, 1024, 0, st>>>(T, sns, S->fronts.as<double>(), S->lperm.as<int32_t>(), tiny, repl,
                                                                  S->stats.as<unsigned long long>());
            else if (fronts_in_lds && (size_t)S->lvl_maxdim[(size_t)l] * S->lvl_maxdim[(size_t)l] * 8 <= 56 * 1024) {
                // (NODAL_DIRECT_LDS_BS: threads per front of the in-LDS kernel -- one wavefront per front makes its four
                // barriers per column step next to free; round 5)
                static const int lds_bs = getenv("NODAL_DIRECT_LDS_BS") ? atoi(getenv("NODAL_DIRECT_LDS_BS")) : 64;
                const size_t lds = (size_t)S->lvl_maxdim[(size_t)l] * S->lvl_maxdim[(size_t)l] * 8;
                const bool wave_fronts = !(getenv("NODAL_DIRECT_WAVE") && atoi(getenv("NODAL_DIRECT_WAVE")) == 0);
                if (lds_bs == 64 && wave_fronts && S->lvl_maxdim[(size_t)l] <= 64)
                    factor_fronts_wave<1><<<nsmall, 64, lds, st>>>(T, sns, S->fronts.as<double>(), S->lperm.as<int32_t>(), tiny, repl,
                                                                   S->stats.as<unsigned long long>());
                else if (lds_bs == 64 && wave_fronts)
                    factor_fronts_wave<2><<<nsmall, 64, lds, st>>>(T, sns, S->fronts.as<double>(), S->lperm.as<int32_t>(), tiny, repl,
                                                                   S->stats.as<unsigned long long>());
                else if (lds_bs == 64)
                    factor_fronts<64, true><<<nsmall, 64, lds, st>>>(T, sns, S->fronts.as<double>(), S->lperm.as<int32_t>(), tiny, repl,
                                                                     S->stats.as<unsigned long long>());
                else if (lds_bs == 128)
                    factor_fronts<128, true><<<nsmall, 128, lds, st>>>(T, sns, S->fronts.as<double>(), S->lperm.as<int32_t>(), tiny, repl,
                                                                       S->stats.as<unsigned long long>());
                else
                    factor_fronts<256, true><<<nsmall, 256, lds, st>>>(T, sns, S->fronts.as<double>(), S->lperm.as<int32_t>(), tiny, repl,
                                                                       S->stats.as<unsigned long long>());
            }
            else factor_fronts<256><<<nsmall, 256, 0, st>>>(T, sns, S->fronts.as<double>(), S->lperm.as<int32_t>(), tiny, repl,
                                                            S->stats.as<unsigned long long>());
        }
        NODAL_HIP_TRY(h, hipGetLastError());
        // The wide fronts of the level, all at once, panel step by panel step (see level_panel_regs): four launches per
        // 16 pivot columns of the level's widest front.  (NODAL_DIRECT_BATCHED=0: round 4's chains, front after front.)
        if (nbig > 0 && batched) {
            const int32_t *bsns = sns + nsmall;
            int max_s = 0, max_dim = 0;
            for (int32_t q = 0; q < nbig; ++q) {
                const int32_t t = S->h_lvl_sn[(size_t)S->lvl_ptr[(size_t)l] + nsmall + q];
                max_s = std::max(max_s, S->h_start[(size_t)t + 1] - S->h_start[(size_t)t]);
                max_dim = std::max(max_dim, S->h_dim[(size_t)t]);
            }
            NODAL_HIP_TRY(h, S->bigpiv.reserve((size_t)nbig * NBB * 4 + 64));
            int32_t *pivs = S->bigpiv.as<int32_t>();
            double *Fd = S->fronts.as<double>();
            int32_t *pd = S->lperm.as<int32_t>();
            unsigned long long *pst = S->stats.as<unsigned long long>();
            level_iota<<<nbig, 256, 0, st>>>(T, bsns, Fd, pd);
            for (int k0 = 0; k0 < max_s; k0 += PNB) {
                const int prows = max_dim - k0;  // (the tallest panel of the step)
                if (panel_regs && prows <= 512) level_panel_regs<2><<<nbig, 256, 0, st>>>(T, bsns, Fd, pd, k0, pivs, tiny, repl, pst);
                else if (panel_regs && prows <= 1024) level_panel_regs<4><<<nbig, 256, 0, st>>>(T, bsns, Fd, pd, k0, pivs, tiny, repl, pst);
                else if (panel_regs && prows <= 1536) level_panel_regs<6><<<nbig, 256, 0, st>>>(T, bsns, Fd, pd, k0, pivs, tiny, repl, pst);
                else level_panel_stream<<<nbig, 1024, 0, st>>>(T, bsns, Fd, pd, k0, pivs, tiny, repl, pst);
                level_swaps_trsm<<<dim3((unsigned)((max_dim + 255) / 256), (unsigned)nbig), 256, 0, st>>>(T, bsns, Fd, pd, k0, pivs);
                const int rest = max_dim - k0 - 1;  // (at least: a front's last panel may be a single column)
                if (rest > 0) {
                    const unsigned tiles = (unsigned)((rest + 63) / 64);
                    level_rank_update<<<dim3(tiles, tiles, (unsigned)nbig), 256, 0, st>>>(T, bsns, Fd, pd, k0);
                }
                NODAL_HIP_TRY(h, hipGetLastError());
            }
            big_fronts += nbig;
            continue;
        }
        // (round 4) The wide fronts of a level are independent and each one's chain is a sequence of small launches (a
        // one-workgroup panel, its interchanges, a triangular solve, a thin GEMM): up to three of them run side by
        // side on streams of the factorisation's own (NODAL_DIRECT_LANES, four), forked and joined by events around the level.
        hipStream_t main_st = st;
        const int nl = nbig < max_lanes ? (nbig > 0 ? nbig : 1) : max_lanes;
        StreamJoinGuard join(main_st);  // (a failed call inside the level must not leave its lanes running unjoined)
        if (nl > 1) {
            NODAL_HIP_TRY(h, hipEventRecord(S->lane_ev[0], main_st));
            for (int k = 1; k < nl; ++k) {
                NODAL_HIP_TRY(h, hipStreamWaitEvent(S->lane_st[k], S->lane_ev[0], 0));
                join.add(S->lane_st[k], S->lane_ev[k]);
            }
        }
        for (int32_t q = 0; q < nbig; ++q) {  // wide fronts: panel by panel (see big-front kernels above)
            const int lane_no = q % nl;
            hipStream_t st = lane_no == 0 ? main_st : S->lane_st[lane_no];  // (shadows the main stream)
            const int32_t t = S->h_lvl_sn[(size_t)S->lvl_ptr[(size_t)l] + nsmall + q];
            const int dim = S->h_dim[(size_t)t], start = S->h_start[(size_t)t], sz = S->h_start[(size_t)t + 1] - start;
            double *F = S->fronts.as<double>() + S->h_front_off[(size_t)t];
            int32_t *perm = S->lperm.as<int32_t>() + start;
            int32_t *piv = S->bigpiv.as<int32_t>() + lane_no * NBB;
            iota_i32<<<(unsigned)((sz + 255) / 256), 256, 0, st>>>(perm, sz);
            for (int k0 = 0; k0 < sz; k0 += panel_nb) {
                const int nb = sz - k0 < panel_nb ? sz - k0 : panel_nb;
                // (the panel in registers when its rows fit: NODAL_DIRECT_PANEL_REGS=0 keeps the streaming kernel)
                const int prows = dim - k0;
                unsigned long long *pst = S->stats.as<unsigned long long>();
                if (panel_regs && nb <= PNB && prows <= 512) panel_factor_regs<2><<<1, 256, 0, st>>>(F, dim, sz, k0, nb, piv, tiny, repl, pst);
                else if (panel_regs && nb <= PNB && prows <= 1024) panel_factor_regs<4><<<1, 256, 0, st>>>(F, dim, sz, k0, nb, piv, tiny, repl, pst);
                else if (panel_regs && nb <= PNB && prows <= 1536) panel_factor_regs<6><<<1, 256, 0, st>>>(F, dim, sz, k0, nb, piv, tiny, repl, pst);
                else panel_factor<<<1, 1024, 0, st>>>(F, dim, sz, k0, nb, piv, tiny, repl, pst);
                apply_swaps<<<(unsigned)((dim - nb + 1 + 255) / 256), 256, 0, st>>>(F, dim, k0, nb, piv, perm);
                const int rest = dim - k0 - nb;
                if (rest > 0) {
                    const unsigned gt = (unsigned)((rest + 255) / 256);
                    trsm_u12<<<gt, 256, 0, st>>>(F, dim, k0, nb);
                    NODAL_HIP_TRY(h, hipGetLastError());
                    NODAL_TRY(gemm_sub_f64(h, st, F + (k0 + nb) + (int64_t)(k0 + nb) * dim, dim, F + (k0 + nb) + (int64_t)k0 * dim,
                                           dim, F + k0 + (int64_t)(k0 + nb) * dim, dim, rest, rest, nb));
                }
            }
            NODAL_HIP_TRY(h, hipGetLastError());
            ++big_fronts;
        }
        for (int k = 1; k < nl; ++k) {
            NODAL_HIP_TRY(h, hipEventRecord(S->lane_ev[k], S->lane_st[k]));
            NODAL_HIP_TRY(h, hipStreamWaitEvent(main_st, S->lane_ev[k], 0));
        }
        join.disarm();
    }
    if (level_table) {
        NODAL_TRY(nodal_wait_stream(h, st, NODAL_SITE));
        fprintf(stderr, "[direct] numeric factorisation by level of the assembly tree (leaves first): fronts, widest front / most pivot "
                        "columns, time, flops of the partial LUs (2/3 (d^3 - (d - s)^3)) and their rate, bytes of the fronts "
                        "(8 d^2, read and written once = the floor) and their rate\n");
        double tot_ms = 0.0, tot_fl = 0.0, tot_by = 0.0;
        for (int32_t l = 0; l < S->nlev; ++l) {
            double fl = 0.0, by = 0.0;
            int md = 0, ms_ = 0;
            for (int32_t q = S->lvl_ptr[(size_t)l]; q < S->lvl_ptr[(size_t)l + 1]; ++q) {
                const int32_t t = S->h_lvl_sn[(size_t)q];
                const double d = S->h_dim[(size_t)t], sz = S->h_start[(size_t)t + 1] - S->h_start[(size_t)t];
                fl += 2.0 / 3.0 * (d * d * d - (d - sz) * (d - sz) * (d - sz));
                by += 16.0 * d * d;
                md = std::max(md, (int)d);
                ms_ = std::max(ms_, (int)sz);
            }
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, lev_ev[(size_t)l], lev_ev[(size_t)l + 1]);
            fprintf(stderr, "[direct]   level %2d: %6d fronts, widest %4d / %4d pivots  %8.3f ms  %9.3f GFLOP %8.2f TFLOP/s  %8.1f MB %7.1f GB/s\n",
                    l, S->lvl_ptr[(size_t)l + 1] - S->lvl_ptr[(size_t)l], md, ms_, ms, fl * 1e-9, ms > 0 ? fl / ms * 1e-9 : 0.0,
                    by * 1e-6, ms > 0 ? by / ms * 1e-6 : 0.0);
            tot_ms += ms; tot_fl += fl; tot_by += by;
        }
        fprintf(stderr, "[direct]   all levels: %.3f ms, %.2f GFLOP (%.2f TFLOP/s), %.1f MB of fronts r+w (%.1f GB/s)\n", tot_ms,
                tot_fl * 1e-9, tot_ms > 0 ? tot_fl / tot_ms * 1e-9 : 0.0, tot_by * 1e-6, tot_ms > 0 ? tot_by / tot_ms * 1e-6 : 0.0);
        for (auto &e : lev_ev) (void)hipEventDestroy(e);
    }
    // the substitutions multiply by the diagonal blocks' inverses (see forward_level): in place, once per factorisation
    if (S->nblocks > 0) {
        invert_diag_blocks<<<(unsigned)S->nblocks, 64, 0, st>>>(T, S->nblocks, S->blk_sn.as<int32_t>(), S->blk_b0.as<int32_t>(),
                                                                S->fronts.as<double>());
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    invert_perm<<<grid_for(n), TB, 0, st>>>(n, S->sn_of_row.as<int32_t>(), S->sn_start.as<int32_t>(), S->lperm.as<int32_t>(),
                                           S->liperm.as<int32_t>());
    NODAL_HIP_TRY(h, hipGetLastError());
    unsigned long long pert = 0;
    NODAL_TRY(nodal_read_words(h, &pert, S->stats.p, 8));
    S->perturbed = (int64_t)pert;
    S->have_numeric = true;
    if (trace)
        fprintf(stderr, "[direct] analysis %.1f ms (%s), numeric factorisation %.1f ms (%lld fronts wider than %d panel by "
                        "panel), %lld perturbed pivots\n", t_sym, t_sym < 0.5 ? "kept" : "new", ms_since(t0) - t_sym,
                (long long)big_fronts, BIG_DIM, (long long)pert);
    return NODAL_OK;
}

namespace {
template <int NR>
int slu_apply_nr(nodal_ctx *h, SluState *S, const double *r, double *z) {
    hipStream_t st = h->stream;
    const int64_t n = S->n;
    const Tree T = tree_of(S);
    if (S->vec_nr < NR) {  // (grown on the handle's stream: zero-filled in order)
        NODAL_HIP_TRY(h, S->vec.reserve((size_t)S->vec_doubles * 8 * NR + 64));
        NODAL_HIP_TRY(h, S->xb.reserve((size_t)n * 8 * NR + 64));
        S->vec_nr = NR;
    }
    permute_rhs<<<grid_for(n * NR), TB, 0, st>>>(n, NR, S->rowof.as<int32_t>(), S->rs.as<double>(), r, S->xb.as<double>());
    static const bool stepped = !(getenv("NODAL_DIRECT_APPLY_STEPPED") && atoi(getenv("NODAL_DIRECT_APPLY_STEPPED")) == 0);
    static const bool super_steps = !(getenv("NODAL_DIRECT_SUPER") && atoi(getenv("NODAL_DIRECT_SUPER")) == 0);
    constexpr int RT = 256 / ColGroup<NR>::CG;  // rows per workgroup of the stepped kernels
    const double *Fd = S->fronts.as<double>();
    double *xbd = S->xb.as<double>(), *vd = S->vec.as<double>();
    const int32_t *pd = S->lperm.as<int32_t>();
    for (int32_t l = 0; l < S->nlev; ++l) {
        const int32_t cnt = S->lvl_ptr[(size_t)l + 1] - S->lvl_ptr[(size_t)l];
        const int32_t *sns = S->lvl_sn.as<int32_t>() + S->lvl_ptr[(size_t)l];
        const int32_t nab = stepped ? S->lvl_abig[(size_t)l] : 0, rest = cnt - nab;
        const int md = stepped ? S->lvl_arest_dim[(size_t)l] : S->lvl_maxdim_all[(size_t)l];
        if (rest > 0) {
            if (md > 192) forward_level<1024, NR><<<rest, 1024, 0, st>>>(T, sns, Fd, pd, xbd, vd);
            else if (md > 64 / (NR > 4 ? 4 : 1)) forward_level<256, NR><<<rest, 256, 0, st>>>(T, sns, Fd, pd, xbd, vd);
            else forward_level<64, NR><<<rest, 64, 0, st>>>(T, sns, Fd, pd, xbd, vd);
        }
        if (nab > 0) {  // the wide fronts of the level together (level_fwd_step)
            const int32_t *bs = sns + rest;
            const int ms = S->lvl_amax_s[(size_t)l], mdim = S->lvl_amax_dim[(size_t)l];
            {
                constexpr int GRT = GATHER_RT;
                level_fwd_gather<NR><<<dim3((unsigned)((mdim + GRT - 1) / GRT), (unsigned)nab), 256, 0, st>>>(
                    T, bs, S->liperm.as<int32_t>(), xbd, vd);
            }
            if (super_steps) {
                for (int b0 = 0; b0 < ms; b0 += SUPER_W) {
                    const int tiles = std::max(1, (mdim - b0 - 1 + RT - 1) / RT);
                    level_fwd_super<NR><<<dim3((unsigned)tiles, (unsigned)nab), 256, 0, st>>>(T, bs, Fd, vd, b0);
                }
            } else {
                for (int b0 = 0; b0 < ms; b0 += DB) {
                    const int tiles = std::max(1, (mdim - b0 - 1 + RT - 1) / RT);
                    level_fwd_step<NR><<<dim3((unsigned)tiles, (unsigned)nab), 256, 0, st>>>(T, bs, Fd, vd, b0);
                }
            }
        }
    }
    for (int32_t l = S->nlev - 1; l >= 0; --l) {
        const int32_t cnt = S->lvl_ptr[(size_t)l + 1] - S->lvl_ptr[(size_t)l];
        const int32_t *sns = S->lvl_sn.as<int32_t>() + S->lvl_ptr[(size_t)l];
        const int32_t nab = stepped ? S->lvl_abig[(size_t)l] : 0, rest = cnt - nab;
        const int md = stepped ? S->lvl_arest_dim[(size_t)l] : S->lvl_maxdim_all[(size_t)l];
        if (rest > 0) {
            if (md > 192) backward_level<1024, NR><<<rest, 1024, 0, st>>>(T, sns, Fd, xbd, vd);
            else if (md > 64 / (NR > 4 ? 4 : 1)) backward_level<256, NR><<<rest, 256, 0, st>>>(T, sns, Fd, xbd, vd);
            else backward_level<64, NR><<<rest, 64, 0, st>>>(T, sns, Fd, xbd, vd);
        }
        if (nab > 0) {
            const int32_t *bs = sns + rest;
            const int ms = S->lvl_amax_s[(size_t)l];
            level_bwd_u12<NR><<<dim3((unsigned)((ms + RT - 1) / RT), (unsigned)nab), 256, 0, st>>>(T, bs, Fd, xbd, vd);
            if (super_steps) {
                for (int b0 = ((ms - 1) / SUPER_W) * SUPER_W; b0 >= 0; b0 -= SUPER_W) {
                    const int tiles = std::max(1, (b0 + RT - 1) / RT);
                    level_bwd_super<NR><<<dim3((unsigned)tiles, (unsigned)nab), 256, 0, st>>>(T, bs, Fd, xbd, vd, b0);
                }
            } else {
                for (int b0 = ((ms - 1) / DB) * DB; b0 >= 0; b0 -= DB) {
                    const int tiles = std::max(1, (b0 + RT - 1) / RT);
                    level_bwd_step<NR><<<dim3((unsigned)tiles, (unsigned)nab), 256, 0, st>>>(T, bs, Fd, xbd, vd, b0);
                }
            }
        }
    }
    unpermute_solution<<<grid_for(n * NR), TB, 0, st>>>(n, NR, S->colof.as<int32_t>(), S->cs.as<double>(), S->xb.as<double>(), z);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}
}  // namespace

// z ~= A^-1 r with the factorisation of the last slu_factor
int slu_apply(nodal_ctx *h, const double *r, double *z) {
    SluState *S = state_of(h);
    if (!S || !S->have_numeric) return nodal_fail(h, NODAL_E_INVALID, "direct solve: no factorisation");
    return slu_apply_nr<1>(h, S, r, z);
}

// the same for SLU_MULTI right-hand sides at once, interleaved by row: r, z hold n x SLU_MULTI doubles, element
// (i, c) at [i * SLU_MULTI + c] -- every entry of the factors is read once for all of them (pair sweeps, sparse.hip)
int slu_apply_multi(nodal_ctx *h, const double *r, double *z) {
    SluState *S = state_of(h);
    if (!S || !S->have_numeric) return nodal_fail(h, NODAL_E_INVALID, "direct solve: no factorisation");
    return slu_apply_nr<SLU_MULTI>(h, S, r, z);
}

bool slu_analysis_kept(nodal_ctx *h) {
    SluState *S = state_of(h);
    return S && S->have_symbolic && S->epoch == h->struct_epoch && S->n == h->n && S->nnz == h->nnz;
}

int64_t slu_perturbed(nodal_ctx *h) {
    SluState *S = state_of(h);
    return S ? S->perturbed : 0;
}

static int sparse_direct_solve_once(nodal_ctx *h, const double *b, double *x, int32_t *info, int32_t *iters, double *resid,
                                    double threshold) {
    *info = 0;
    *iters = 0;
    *resid = 0.0;
    NODAL_TRY(slu_factor(h, info, 1.0, threshold));
    if (*info > 0) return NODAL_OK;
    NODAL_TRY(general_krylov_direct(h, b, x, info, iters, resid));
    SluState *S = state_of(h);
    if (*info > 0 || S->perturbed == 0) return NODAL_OK;
    // Pivots were replaced and the refinement converged all the same.  Either G is regular and the pivot it
    // needed sat outside the fully summed rows of its front (the refinement has made up for it), or G is
    // singular and the equations happen to be CONSISTENT -- then the answer is one of infinitely many
    // solutions, where the reference's SuperLU meets the zero pivot (NaNs + MatrixRankWarning, reference
    // nodal/nodal.py:323-336).  (Another replacement value does not tell them apart: for a consistent right-hand
    // side the solution of the perturbed system does not depend on it to first order.)  A pseudo-random
    // right-hand side does: a singular matrix makes it inconsistent and the refinement on the same factors
    // stalls; a regular one solves it like any other.
    const int64_t n = h->n;
    NODAL_HIP_TRY(h, h->work2.reserve((size_t)n * 8 + 64));
    NODAL_HIP_TRY(h, h->work3.reserve((size_t)n * 8 + 64));
    hashed_rhs<<<grid_for(n), TB, 0, h->stream>>>(n, h->work2.as<double>());
    NODAL_HIP_TRY(h, hipGetLastError());
    int32_t info2 = 0, it2 = 0;
    double rs2 = 0.0;
    h->slu_strict = true;
    const int s2 = general_krylov_direct(h, h->work2.as<double>(), h->work3.as<double>(), &info2, &it2, &rs2);
    h->slu_strict = false;
    NODAL_TRY(s2);
    if (getenv("NODAL_TRACE"))
        fprintf(stderr, "[direct] %lld replaced pivots: a pseudo-random right-hand side %s\n", (long long)S->perturbed,
                info2 > 0 ? "does not refine: singular" : "is solved too: regular");
    if (info2 > 0) *info = 1;
    *iters += it2;
    return NODAL_OK;
}

// G x = b by the multifrontal LU + flexible GMRES refinement.  *info > 0: singular (structurally, or replaced pivots and
// a refinement that stalls): the caller fills NaNs, as the reference's spsolve does (reference nodal/nodal.py:323-336:
// NaNs + MatrixRankWarning, no exception).
// The analysis is kept per sparsity pattern, but its row matching looked at VALUES (a diagonal is kept when it carries
// weight, the largest entry of a free column otherwise): a value sweep, nodal_run(reuse_symbolic) or a pair sweep can
// hand the kept matching entries that are tiny or zero now.  So a verdict that rests on a kept analysis -- singular, or
// pivots were replaced -- is not final: the analysis is redone with the current values, once (advisor, round 4).
int sparse_direct_solve(nodal_ctx *h, const double *b, double *x, int32_t *info, int32_t *iters, double *resid) {
    bool redone = false, relaxed = false;
    double threshold = 0.0;  // (the default: sqrt(eps))
    for (;;) {
        NODAL_TRY(sparse_direct_solve_once(h, b, x, info, iters, resid, threshold));
        SluState *S = state_of(h);
        if (!S) return NODAL_OK;
        const bool trace = getenv("NODAL_TRACE") != nullptr;
        const bool doubtful = *info > 0 || S->perturbed > 0;
        if (doubtful && S->analysis_kept && !redone) {
            if (trace)
                fprintf(stderr, "[direct] %s on an analysis kept from earlier values: the analysis is redone with the current ones\n",
                        *info > 0 ? "singular verdict" : "replaced pivots");
            redone = true;
            S->have_symbolic = S->have_numeric = false;
            continue;
        }
        // Last resort before "singular": pivots were replaced under the sqrt(eps) rule and the refinement could not make
        // up for them.  In a regular matrix whose entries span many decades (resistances over 14 decades: the scaled
        // pivots of the weak links sit below sqrt(eps) by rights) those were TRUE pivots -- the reference's SuperLU
        // divides by them and returns a solution.  Factor once more with the bar at 1e-13: what is still replaced then is
        // a zero to working precision, and the verdicts stand as before.
        if (*info > 0 && S->perturbed > 0 && !relaxed) {
            if (trace) fprintf(stderr, "[direct] %lld pivots below sqrt(eps) and no refinement: once more with the pivot bar at 1e-13\n",
                               (long long)S->perturbed);
            relaxed = true;
            threshold = 1e-13;
            continue;
        }
        return NODAL_OK;
    }
}
