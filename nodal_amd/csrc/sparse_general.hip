// Large general MNA systems on the sparse path: branch equations present, so G is
// non-symmetric with zero diagonals in the branch rows (E, VCVS, CCVS) -- the case
// the reference hands to SuperLU (scipy spsolve, reference nodal/nodal.py:325).
//
//     [ Gn  Bc ] [ e ]   [ a ]      Gn (K x K): only resistor stamps land here, so it
//     [ Cr  D  ] [ i ] = [ v ]      is a symmetric M-matrix whatever the sources are.
//
// Right-preconditioned flexible GMRES(m) on the full system with the block
// preconditioner
//     z_e = AMG(Gn)^-1 r_e                        (K-cycle of amg.hip on the node block)
//     z_i = S~^-1 (r_i - Cr z_e),  S~ = diag(D - Cr diag(Gn)^-1 Bc)
// i.e. block forward substitution with the multigrid cycle standing in for Gn^-1 and
// a diagonal approximation of the Schur complement of the branch block.  Nodes
// without any resistor get a unit diagonal in the preconditioner's copy of Gn.
// Orthogonalisation: classical Gram-Schmidt, twice, with all inner products of a
// step fused in one kernel; the small Hessenberg least-squares problem stays on the
// host.  The answer is accepted only if the TRUE residual meets the tolerance;
// otherwise the call fails loudly (no silent wrong answer, no CPU fallback).
#include <cmath>

#include "ctx.h"

namespace {

constexpr int TB = 256;
constexpr int RESTART = 40;
constexpr int MAXV = RESTART + 1;
constexpr int DOT_GRID = 1024;
// the small least-squares problem's state on the device (see gs_finish)
enum { G_H = 0, G_CS = G_H + MAXV * RESTART, G_SN = G_CS + RESTART, G_G = G_SN + RESTART,
       G_INV_H = G_G + MAXV, G_EST, G_DONE, G_COUNT, G_INFO, G_TOLB,
       G_LOG /* per column: |w after the projections|^2 / |w|^2 (NODAL_TRACE) */, G_WORDS = G_LOG + RESTART };

inline unsigned grid_for(int64_t n, unsigned cap = 4096) {
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

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

// ---- node block extraction ----------------------------------------------------------

__global__ __launch_bounds__(TB) void gn_count(const int32_t *__restrict__ indptr,
                                               const int32_t *__restrict__ indices, int K,
                                               uint32_t *__restrict__ cnt) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < K; i += (int64_t)gridDim.x * TB) {
        uint32_t c = 0;
        bool diag = false;
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int j = indices[e];
            if (j < K) { ++c; diag |= (j == (int)i); }
        }
        cnt[i] = c + (diag ? 0u : 1u);
    }
}

__global__ __launch_bounds__(TB) void gn_fill(const int32_t *__restrict__ indptr,
                                              const int32_t *__restrict__ indices,
                                              const double *__restrict__ data, int K,
                                              const uint32_t *__restrict__ start,
                                              int32_t *__restrict__ o_indptr,
                                              int32_t *__restrict__ o_indices,
                                              int32_t *__restrict__ o_rowidx,
                                              double *__restrict__ o_data,
                                              int32_t *__restrict__ o_diag) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i <= K; i += (int64_t)gridDim.x * TB) {
        o_indptr[i] = (int32_t)start[i];
        if (i == K) continue;
        uint32_t p = start[i];
        bool placed = false;
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) {
            const int j = indices[e];
            if (j >= K) break;  // sorted columns
            if (!placed && j > (int)i) {  // no diagonal stamp: insert a unit one
                o_indices[p] = (int32_t)i; o_rowidx[p] = (int32_t)i; o_data[p] = 1.0; o_diag[i] = (int32_t)p;
                ++p;
                placed = true;
            }
            double v = data[e];
            if (j == (int)i) {
                placed = true;
                o_diag[i] = (int32_t)p;
                if (!(v > 0.0)) v = 1.0;  // preconditioner only: keep it an M-matrix
            }
            o_indices[p] = j; o_rowidx[p] = (int32_t)i; o_data[p] = v;
            ++p;
        }
        if (!placed) {
            o_indices[p] = (int32_t)i; o_rowidx[p] = (int32_t)i; o_data[p] = 1.0; o_diag[i] = (int32_t)p;
        }
    }
}

// S~_m = D_mm - sum_j Cr_mj Bc_jm / Gn_jj ; stored as 1 / S~_m (1 if degenerate)
__global__ __launch_bounds__(TB) void schur_diag(const int32_t *__restrict__ indptr,
                                                 const int32_t *__restrict__ indices,
                                                 const double *__restrict__ data, int K, int n,
                                                 const double *__restrict__ gn_data,
                                                 const int32_t *__restrict__ gn_diag,
                                                 double *__restrict__ sinv) {
    for (int64_t m = K + (int64_t)blockIdx.x * TB + threadIdx.x; m < n; m += (int64_t)gridDim.x * TB) {
        double s = 0.0;
        for (int32_t e = indptr[m]; e < indptr[m + 1]; ++e) {
            const int j = indices[e];
            if (j == (int)m) { s += data[e]; continue; }
            if (j >= K) continue;
            // Bc_jm: entry (j, m) of G, by binary search in the sorted row j
            int lo = indptr[j], hi = indptr[j + 1] - 1;
            double bjm = 0.0;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const int c = indices[mid];
                if (c == (int)m) { bjm = data[mid]; break; }
                if (c < (int)m) lo = mid + 1; else hi = mid - 1;
            }
            s -= data[e] * bjm / gn_data[gn_diag[j]];
        }
        sinv[m - K] = (s != 0.0 && s == s) ? 1.0 / s : 1.0;
    }
}

// z_i = S~^-1 (r_i - Cr z_e)
__global__ __launch_bounds__(TB) void branch_solve(const int32_t *__restrict__ indptr,
                                                   const int32_t *__restrict__ indices,
                                                   const double *__restrict__ data, int K, int n,
                                                   const double *__restrict__ sinv,
                                                   const double *__restrict__ r,
                                                   double *__restrict__ z) {
    for (int64_t m = K + (int64_t)blockIdx.x * TB + threadIdx.x; m < n; m += (int64_t)gridDim.x * TB) {
        double s = r[m];
        for (int32_t e = indptr[m]; e < indptr[m + 1]; ++e) {
            const int j = indices[e];
            if (j < K) s = fma(-data[e], z[j], s);
        }
        z[m] = s * sinv[m - K];
    }
}

// ---- Gram-Schmidt kernels -----------------------------------------------------------

// All three kernels are instantiated for NV = nv rounded up to a multiple of 4, so the
// loop over the basis is fully unrolled with unconditional loads (a run-time bound turns
// every load into its own branch + wait).  The padding lanes re-read vector nv-1 (L2
// hits) with a zero coefficient.

// partial[block][i] = sum over the block's rows of V_i . w, i < nv
template <int NV>
__global__ __launch_bounds__(TB) void gs_dots(const double *__restrict__ V, int64_t ld, int nv,
                                              const double *__restrict__ w, int64_t n,
                                              double *__restrict__ partial) {
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TB) {
        const double wr = w[r];
#pragma unroll
        for (int i = 0; i < NV; ++i)
            acc[i] = fma(V[(int64_t)(i < nv ? i : nv - 1) * ld + r], wr, acc[i]);
    }
    __shared__ double ws[TB / 64][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double v = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6][i] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < TB / 64; ++wv) s += ws[wv][threadIdx.x];
        partial[(int64_t)blockIdx.x * MAXV + threadIdx.x] = s;
    }
}

// w -= sum_i h_i V_i, and in the same pass over V the inner products of the UPDATED w with
// the basis (the second Gram-Schmidt sweep): partial[block][i] = sum V_i . w'
template <int NV>
__global__ __launch_bounds__(TB) void gs_update_dots(const double *__restrict__ V, int64_t ld, int nv,
                                                     const double *__restrict__ hh,
                                                     double *__restrict__ w, int64_t n,
                                                     double *__restrict__ partial) {
    double hv[NV], acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        hv[i] = i < nv ? hh[i] : 0.0;
        acc[i] = 0.0;
    }
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TB) {
        double v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = V[(int64_t)(i < nv ? i : nv - 1) * ld + r];
        double wr = w[r];
#pragma unroll
        for (int i = 0; i < NV; ++i) wr = fma(-hv[i], v[i], wr);
        w[r] = wr;
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = fma(v[i], wr, acc[i]);
    }
    __shared__ double ws[TB / 64][NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6][i] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double s = 0.0;
#pragma unroll
        for (int wv = 0; wv < TB / 64; ++wv) s += ws[wv][threadIdx.x];
        partial[(int64_t)blockIdx.x * MAXV + threadIdx.x] = s;
    }
}

// w -= sum_i h_i V_i ; partial2[block] = |w|^2 of the block's rows afterwards
template <int NV>
__global__ __launch_bounds__(TB) void gs_update(const double *__restrict__ V, int64_t ld, int nv,
                                                const double *__restrict__ hh,
                                                double *__restrict__ w, int64_t n,
                                                double *__restrict__ partial2) {
    double hv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) hv[i] = i < nv ? hh[i] : 0.0;
    double nrm = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TB) {
        double v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = V[(int64_t)(i < nv ? i : nv - 1) * ld + r];
        double wr = w[r];
#pragma unroll
        for (int i = 0; i < NV; ++i) wr = fma(-hv[i], v[i], wr);
        w[r] = wr;
        nrm = fma(wr, wr, nrm);
    }
    nrm = block_sum(nrm);
    if (threadIdx.x == 0) partial2[blockIdx.x] = nrm;
}

// Block i < nv: out[i] = sum over blocks of partial[.][i]; block nv (when partial2 is
// given): out[MAXV] = sum of partial2.  Fixed order, so the result is reproducible.
__global__ __launch_bounds__(TB) void gs_reduce(const double *__restrict__ partial, int nblocks,
                                                int nv, double *__restrict__ out,
                                                const double *__restrict__ partial2,
                                                int nblocks2) {
    const int i = blockIdx.x;
    double s = 0.0;
    if (i < nv) {
        for (int b = threadIdx.x; b < nblocks; b += TB) s += partial[(int64_t)b * MAXV + i];
    } else if (partial2) {
        for (int b = threadIdx.x; b < nblocks2; b += TB) s += partial2[b];
    }
    s = block_sum(s);
    if (threadIdx.x == 0) {
        if (i < nv) out[i] = s;
        else if (partial2) out[MAXV] = s;
    }
}

#define DISPATCH_NV(nv, CALL)                          \
    switch (((nv) + 3) / 4) {                          \
    case 1: { constexpr int NV = 4; CALL; } break;     \
    case 2: { constexpr int NV = 8; CALL; } break;     \
    case 3: { constexpr int NV = 12; CALL; } break;    \
    case 4: { constexpr int NV = 16; CALL; } break;    \
    case 5: { constexpr int NV = 20; CALL; } break;    \
    case 6: { constexpr int NV = 24; CALL; } break;    \
    case 7: { constexpr int NV = 28; CALL; } break;    \
    case 8: { constexpr int NV = 32; CALL; } break;    \
    case 9: { constexpr int NV = 36; CALL; } break;    \
    case 10: { constexpr int NV = 40; CALL; } break;   \
    default: { constexpr int NV = 44; CALL; } break;   \
    }

// dst = src * scale
__global__ __launch_bounds__(TB) void scale_to(const double *__restrict__ src, double scale,
                                               double *__restrict__ dst, int64_t n) {
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TB)
        dst[r] = src[r] * scale;
}

// x += sum_i y_i Z_i
__global__ __launch_bounds__(TB) void add_combination(const double *__restrict__ Z, int64_t ld,
                                                      int nv_max, const double *__restrict__ y,
                                                      double *__restrict__ x, const double *__restrict__ gst,
                                                      int64_t n) {
    __shared__ double ys[MAXV];
    const int count = (int)gst[G_COUNT], nv = count < nv_max ? count : nv_max;  // (columns of this cycle)
    if (threadIdx.x < MAXV) ys[threadIdx.x] = (int)threadIdx.x < nv ? y[threadIdx.x] : 0.0;
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TB) {
        double xr = x[r];
        for (int i = 0; i < nv; ++i) xr = fma(ys[i], Z[(int64_t)i * ld + r], xr);
        x[r] = xr;
    }
}

// ---- the small least-squares problem, on the device -----------------------------------
// The Hessenberg column, the Givens rotations and the residual estimate of every iteration
// live in one block of doubles, so the host never waits inside a restart cycle: it enqueues
// iterations in batches and only polls G_DONE / G_EST (as the flexible CG of sagg.hip does).

__global__ void gst_begin(double *__restrict__ gst, double rnorm, double tolb) {
    for (int i = threadIdx.x; i < MAXV; i += blockDim.x) gst[G_G + i] = i == 0 ? rnorm : 0.0;
    if (threadIdx.x == 0) {
        gst[G_INV_H] = 0.0;
        gst[G_EST] = rnorm;
        gst[G_DONE] = 0.0;
        gst[G_COUNT] = 0.0;
        gst[G_INFO] = 0.0;
        gst[G_TOLB] = tolb;
    }
}

// Iteration j's last step: |w|^2 from the blocks' partial sums (fixed order), the new column
// h = h1 + h2 of H, the rotations so far applied to it, the new rotation, g and the estimate.
// Once G_DONE is raised (converged, happy breakdown or singular operator) later iterations'
// calls leave the state alone.
__global__ __launch_bounds__(TB) void gs_finish(const double *__restrict__ h1, const double *__restrict__ h2,
                                                const double *__restrict__ partial2, int nblocks2, int j, int s0,
                                                double *__restrict__ gst) {
    if (gst[G_DONE] != 0.0) return;
    // (the column, the rotations and g are staged in LDS by all threads: thread 0 walking them through global
    // memory was a chain of ~150 dependent loads, 11.7 us per iteration)
    __shared__ double col[MAXV + 1], cs_[RESTART], sn_[RESTART], gj;
    const int nv = j + 1;
    for (int i = threadIdx.x; i < nv; i += TB) col[i] = i < s0 ? 0.0 : h1[i] + h2[i];  // (s0 > 0: a window only)
    for (int i = threadIdx.x; i < j; i += TB) {
        cs_[i] = gst[G_CS + i];
        sn_[i] = gst[G_SN + i];
    }
    if (threadIdx.x == 0) gj = gst[G_G + j];
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks2; b += TB) s += partial2[b];
    s = block_sum(s);  // (its barriers also publish the staged values)
    __syncthreads();
    double *H = gst + G_H;
    if (threadIdx.x == 0) {
        const double hnext = sqrt(s);
        double proj = 0.0;
        for (int i = 0; i < nv; ++i) proj = fma(col[i], col[i], proj);
        gst[G_LOG + j] = s / (proj + s);
        col[nv] = hnext;
        for (int i = 0; i < j; ++i) {
            const double a = col[i], b = col[i + 1];
            col[i] = cs_[i] * a + sn_[i] * b;
            col[i + 1] = -sn_[i] * a + cs_[i] * b;
        }
        const double d = hypot(col[j], col[j + 1]);
        if (!(d > 0.0) || d != d) {  // breakdown: singular operator
            gst[G_INFO] = 1.0;
            gst[G_DONE] = 1.0;
        } else {
            const double c = col[j] / d, sn = col[j + 1] / d;
            gst[G_CS + j] = c;
            gst[G_SN + j] = sn;
            col[j] = d;
            col[j + 1] = 0.0;
            gst[G_G + j + 1] = -sn * gj;
            gst[G_G + j] = c * gj;
            const double est = fabs(sn * gj);
            gst[G_EST] = est;
            gst[G_COUNT] = (double)(j + 1);
            gst[G_INV_H] = hnext > 0.0 ? 1.0 / hnext : 0.0;
            if (est <= gst[G_TOLB] || hnext == 0.0) gst[G_DONE] = 1.0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= nv; i += TB) H[i * RESTART + j] = col[i];
}

// v_{j+1} = w / h_{j+1,j}
// (x0 != nullptr: also the start iterate w D^-1 v of the multigrid cycle that preconditions v_{j+1} next, rows < n0:
// the launch sagg_apply would spend on it)
__global__ __launch_bounds__(TB) void scale_by_device(const double *__restrict__ src, const double *__restrict__ gst,
                                                      double *__restrict__ dst, int64_t n,
                                                      const double *__restrict__ dinv, nodal_cyc_t *__restrict__ x0,
                                                      int64_t n0, double omega) {
    if (gst[G_DONE] != 0.0) return;
    const double scale = gst[G_INV_H];
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < n; r += (int64_t)gridDim.x * TB) {
        const double v = src[r] * scale;
        dst[r] = v;
        if (x0 && r < n0) x0[r] = (nodal_cyc_t)(omega * dinv[r] * v);
    }
}

// y = H^-1 g over the G_COUNT columns of the cycle (zero beyond)
__global__ void gst_solve(const double *__restrict__ gst, double *__restrict__ y) {
    if (threadIdx.x != 0) return;
    const int m = (int)gst[G_COUNT];
    const double *H = gst + G_H, *g = gst + G_G;
    for (int i = m; i < MAXV; ++i) y[i] = 0.0;
    for (int i = m - 1; i >= 0; --i) {
        double s = g[i];
        for (int k = i + 1; k < m; ++k) s -= H[i * RESTART + k] * y[k];
        y[i] = s / H[i * RESTART + i];
    }
}

// r = b - t ; partial2 = |r|^2
__global__ __launch_bounds__(TB) void residual_of(const double *__restrict__ b,
                                                  const double *__restrict__ t,
                                                  double *__restrict__ r, int64_t n,
                                                  double *__restrict__ partial2) {
    double nrm = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double v = b[i] - (t ? t[i] : 0.0);
        r[i] = v;
        nrm = fma(v, v, nrm);
    }
    nrm = block_sum(nrm);
    if (threadIdx.x == 0) partial2[blockIdx.x] = nrm;
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

int csr_spmv(nodal_ctx *h, const double *x, double *y);  // sparse.hip
int csr_scaled_residual(nodal_ctx *h, const double *x, const double *b, double *scaled);  // sparse.hip
bool general_source_loop(const nodal_ctx *h);                // sparse.hip
bool general_floating_island(const nodal_ctx *h);            // sparse.hip
int grounded_flags(nodal_ctx *h, uint8_t *flags_dev);         // lowdeg.hip

// A row whose stored entries are all exactly zero (a VCVS from a node to ground sensing that very node
// with gain 1: its branch row reads (1 - 1) e = 0, reference nodal/models.py:53-78): G is exactly
// singular, the reference's SuperLU meets the zero pivot whatever the rounding (NaNs +
// MatrixRankWarning) -- but the system is consistent, so a Krylov iteration "converges" to one of its
// solutions.  Looked for before anything is solved.
__global__ __launch_bounds__(TB) void find_zero_row(const int32_t *__restrict__ indptr, const double *__restrict__ data,
                                                    int64_t n, uint32_t *__restrict__ flag) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        bool any = false;
        for (int32_t e = indptr[i]; e < indptr[i + 1]; ++e) any = any || data[e] != 0.0;
        if (!any) *flag = 1u;  // benign race
    }
}

namespace {

// `direct`: the preconditioner is the multifrontal LU of sparse_direct.hip (factorised by the caller) --
// the iteration is then iterative refinement with a true-residual test: no presolve, no structural
// verdicts (the caller has been through them), no multigrid; any right-hand side / solution vectors.
int general_impl(nodal_ctx *h, const double *b, double *x, int32_t *info, int32_t *iters, double *resid,
                 bool direct) {
    const int64_t n = h->n;
    const int K = direct ? (int)n : h->K;
    hipStream_t st = h->stream;
    const int32_t *indptr = h->indptr.as<int32_t>();
    const int32_t *indices = h->indices.as<int32_t>();
    const double *data = h->data.as<double>();
    *info = 0;

    if (h->B > 0 && !direct) {  // (branch rows are the ones that can vanish; a node row of an all-resistor network cannot)
        NODAL_HIP_TRY(h, h->status.reserve(64));
        uint32_t *zr = reinterpret_cast<uint32_t *>(h->status.as<char>() + 48);
        NODAL_HIP_TRY(h, hipMemsetAsync(zr, 0, 4, st));
        find_zero_row<<<grid_for(n), TB, 0, st>>>(indptr, data, n, zr);
        NODAL_HIP_TRY(h, hipGetLastError());
        uint32_t zero_row = 0;
        NODAL_TRY(nodal_read_words(h, &zero_row, zr, 4));
        if (zero_row) {
            *info = 1;
            return NODAL_OK;
        }
    }
    // branch equations present: first try to eliminate them exactly (presolve.hip)
    if (h->B > 0 && !direct) {
        const bool attempted = h->use_presolve && !h->host.type.empty();
        if (attempted) {
            bool done = false;
            NODAL_TRY(presolve_solve(h, &done, info, iters, resid));
            if (done) return NODAL_OK;
        }
        // A loop of voltage-defined branches (one of the patterns the presolve declines) makes the matrix
        // exactly singular whatever the values; with consistent values a Krylov iteration would still hand
        // back one of the infinitely many solutions, where the reference's spsolve reports the singular matrix
        // (NaNs + MatrixRankWarning).  The same for an island that nothing ties to the ground node: its
        // equations are consistent (no net current can enter it), so the iteration would "converge" with
        // arbitrary island potentials.  (The presolved system asks the multigrid hierarchy instead, below.)
        if (general_source_loop(h) || general_floating_island(h)) {
            *info = 1;
            return NODAL_OK;
        }
        // What the presolve declined or could not get accepted -- cyclic definitions among dependent sources,
        // duplicated branches, a reduced system that did not converge -- is where the reference's PIVOTING
        // decides (spsolve, reference nodal/nodal.py:325): e_p = 2 e_q together with e_q = 0.5 e_p is singular
        // for its values only, the equations are consistent, and the iteration below converges to one of the
        // solutions.  Such systems go to the pivoting solvers (sparse.hip: the dense LU when small, the sparse
        // direct route of sparse_direct.hip otherwise) instead of the full-system iteration (round 3: 240-360
        // iterations at 1e6 unknowns, and the wrong kind of answer on a singular matrix).
        if (attempted)
            return nodal_fail(h, NODAL_E_UNSUPPORTED, "the presolve declined this pattern of sources");
    }

    bool use_sa = false;
    bool ell_spmv = false;
    if (!direct) {
    // ---- preconditioner setup: node block + multigrid + Schur diagonal ----
    NODAL_HIP_TRY(h, h->work.reserve(align_up((size_t)(K + 1) * 4) + scan_tmp_bytes(K + 1) + 512));
    uint32_t *cnt = h->work.as<uint32_t>();
    void *scan_tmp = h->work.as<char>() + align_up((size_t)(K + 1) * 4);
    NODAL_HIP_TRY(h, hipMemsetAsync(cnt, 0, (size_t)(K + 1) * 4, st));
    gn_count<<<grid_for(K), TB, 0, st>>>(indptr, indices, K, cnt);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_TRY(scan_exclusive_u32(h, cnt, cnt, (int64_t)K + 1, nullptr, scan_tmp));
    uint32_t gn_nnz = 0;
    NODAL_TRY(nodal_read_words(h, &gn_nnz, cnt + K, 4));
    NODAL_HIP_TRY(h, h->gn_indptr.reserve((size_t)(K + 1) * 4));
    NODAL_HIP_TRY(h, h->gn_indices.reserve((size_t)gn_nnz * 4 + 4));
    NODAL_HIP_TRY(h, h->gn_rowidx.reserve((size_t)gn_nnz * 4 + 4));
    NODAL_HIP_TRY(h, h->gn_data.reserve((size_t)gn_nnz * 8 + 8));
    NODAL_HIP_TRY(h, h->gn_diag.reserve((size_t)K * 4 + 4));
    gn_fill<<<grid_for(K + 1), TB, 0, st>>>(indptr, indices, data, K, cnt, h->gn_indptr.as<int32_t>(),
                                           h->gn_indices.as<int32_t>(), h->gn_rowidx.as<int32_t>(),
                                           h->gn_data.as<double>(), h->gn_diag.as<int32_t>());
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, h->schur.reserve((size_t)(n - K) * 8 + 64));
    double *flag = h->schur.as<double>() + (n - K);  // spare word: multigrid's SPD flag (unused here)
    NODAL_HIP_TRY(h, hipMemsetAsync(flag, 0, 8, st));
    // node block: smoothed aggregation where it takes the matrix (sagg.hip), else plain aggregation
    {
        // a presolved system (no branch unknowns left, resistors / current sources / transconductances):
        // its table says which nodes touch ground, so the hierarchy can tell a floating island
        const bool check_floating = h->B == 0 && h->have_table && !h->csr_only && K == (int)n;
        int32_t floating = 0;
        NODAL_TRY(sagg_setup_csr(h, K, gn_nnz, h->gn_indptr.as<int32_t>(), h->gn_indices.as<int32_t>(),
                                 h->gn_data.as<double>(), true, check_floating, &use_sa, &floating));
        if (!use_sa && check_floating) {  // the hierarchy declined the matrix: the same verdict on the CSR pattern
            NODAL_HIP_TRY(h, h->work3.reserve((size_t)n + 256));
            NODAL_TRY(grounded_flags(h, h->work3.as<uint8_t>()));
            NODAL_TRY(csr_has_floating_component(h, h->work3.as<uint8_t>(), &floating));
        }
        if (floating) {  // singular: NaNs + warning, as the reference's spsolve (quirk 3)
            *info = 1;
            return NODAL_OK;
        }
    }
    // the hierarchy's level-0 matrix is the system itself (a presolved network: no branch rows, no
    // diagonal had to be added): its ELL copy serves the Krylov SpMV too (13 instead of 22 us at 1e6 rows)
    ell_spmv = use_sa && K == (int)n && gn_nnz == (uint32_t)h->nnz;
    if (use_sa) {
        h->amg_levels = sagg_levels(h);
    } else {
        NODAL_TRY(amg_setup_csr(h, K, gn_nnz, h->gn_indptr.as<int32_t>(), h->gn_indices.as<int32_t>(),
                                h->gn_rowidx.as<int32_t>(), h->gn_data.as<double>(),
                                h->gn_diag.as<int32_t>(), flag));
        h->amg_levels = amg_num_levels(h);
    }
    if (n > K) {
        schur_diag<<<grid_for(n - K), TB, 0, st>>>(indptr, indices, data, K, (int)n,
                                                  h->gn_data.as<double>(), h->gn_diag.as<int32_t>(),
                                                  h->schur.as<double>());
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    }  // (!direct)
    auto system_spmv = [&](const double *in, double *out) -> int {
        return ell_spmv ? sagg_spmv(h, in, out) : csr_spmv(h, in, out);
    };

    // ---- Krylov storage: V (m+1), Z (m), w, r ----
    const int64_t ld = (int64_t)(align_up((size_t)n * 8) / 8);
    const size_t vecs = (size_t)(2 * RESTART + 3) * ld * 8;
    const size_t scal = ((size_t)DOT_GRID * MAXV + DOT_GRID + 2 * (MAXV + 1) + MAXV + G_WORDS) * 8;
    NODAL_HIP_TRY(h, h->krylov.reserve(vecs + scal + 1024));
    double *V = h->krylov.as<double>();
    double *Z = V + (int64_t)(RESTART + 1) * ld;
    double *w = Z + (int64_t)RESTART * ld;
    double *r = w + ld;
    double *partial = r + ld;
    double *partial2 = partial + (int64_t)DOT_GRID * MAXV;
    double *hdev = partial2 + DOT_GRID;        // MAXV + 1
    double *hdev2 = hdev + (MAXV + 1);         // MAXV + 1
    double *ydev = hdev2 + (MAXV + 1);         // MAXV
    double *gst = ydev + MAXV;                 // G_WORDS: the small least-squares problem

    const unsigned gd = grid_for(n, DOT_GRID), gv = grid_for(n);
    NODAL_HIP_TRY(h, hipMemsetAsync(x, 0, (size_t)n * 8, st));

    auto device_norm = [&](double *out) -> int {  // sqrt(sum partial2) -> host
        gs_reduce<<<1, TB, 0, st>>>(partial, 0, 0, hdev, partial2, (int)gd);
        double v = 0.0;
        NODAL_TRY(nodal_read_words(h, &v, hdev + MAXV, 8));
        *out = std::sqrt(v);
        return NODAL_OK;
    };

    residual_of<<<gd, TB, 0, st>>>(b, nullptr, r, n, partial2);
    double bnorm = 0.0;
    NODAL_TRY(device_norm(&bnorm));
    *iters = 0;
    *resid = 0.0;
    if (bnorm == 0.0) return NODAL_OK;  // x = 0

    const double tol = 1e-13;
    const int max_cycles = direct ? 3 : 10;  // (refinement on good factors takes 1-3 ITERATIONS; 120 say "singular")
    // Orthogonalisation window of the FIRST restart cycle.  A system without branch rows (a presolved network:
    // a conductance matrix plus a few transconductance terms, preconditioned by its own multigrid cycle) is
    // nearly symmetric, its Hessenberg matrix nearly tridiagonal: orthogonalising against the last 8 basis
    // vectors costs no iteration (config 5: 23 either way, window 4 too) and saves 43 % of the Gram-Schmidt
    // passes -- 11.4 -> 10.6 ms.  A saddle-point system (branch rows present) needs all of them: with a window
    // of 12 or 16 the full-system fallback of config 5 + cascaded stages stalls at 2e-9 after 400 iterations
    // where full orthogonalisation converges in 240.  So: the window only without branch rows, only in the
    // first cycle (a system that needs a restart is a hard one: all vectors from then on), and the true
    // residual at the end of every cycle decides.  NODAL_FGMRES_WINDOW=k forces k everywhere (experiments).
    static const int window_env = getenv("NODAL_FGMRES_WINDOW") ? std::max(2, atoi(getenv("NODAL_FGMRES_WINDOW"))) : 0;
    const int window_first = window_env ? window_env : ((n == K && !direct) ? 8 : RESTART + 1);
    double rnorm = bnorm;
    int total = 0;
    bool converged = false;
    h->kern_ms = 0;
    h->kern_launches = 0;
    hipEvent_t e0 = h->ev[2], e1 = h->ev[3];
    double hst[6];  // G_INV_H .. G_TOLB, as polled

    for (int cyc = 0; cyc < max_cycles && !converged; ++cyc) {
        const int window = (cyc == 0 || window_env) ? window_first : RESTART + 1;
        scale_to<<<gv, TB, 0, st>>>(r, 1.0 / rnorm, V, n);
        gst_begin<<<1, 64, 0, st>>>(gst, rnorm, tol * bnorm);
        // the kernel that normalises v_{j+1} also leaves the start iterate of the cycle that preconditions it
        const double *x0_dinv = nullptr;
        nodal_cyc_t *x0_slot = nullptr;
        int64_t x0_n = 0;
        double x0_omega = 0.0;
        if (use_sa && !direct && !sagg_x0_slot(h, &x0_dinv, &x0_slot, &x0_n, &x0_omega)) x0_slot = nullptr;
        // Iterations are enqueued in batches; between batches the host reads the estimate and
        // sizes the next batch from the convergence rate seen so far (three quarters of what is
        // still missing, at least two).  A batch that overshoots costs idle launches only: every
        // kernel that touches the small problem returns once G_DONE is up.
        // (behind the direct factorisation an iteration is a 10-ms pair of triangular sweeps and one or two of them
        // converge: the host looks after every one -- a batch of six ran four sweeps for nothing)
        int enq = 0, batch = direct ? 1 : 6;
        double est_prev = rnorm;
        int at_prev = 0;
        bool done = false;
        while (!done && enq < RESTART) {
            if (batch > RESTART - enq) batch = RESTART - enq;
            for (int c = 0; c < batch; ++c, ++enq) {
                const int j = enq;
                double *vj = V + (int64_t)j * ld, *zj = Z + (int64_t)j * ld;
                // z_j = M^-1 v_j
                if (j == 0) nodal_nan_probe(h, vj, n, "fgmres v0");
                if (direct) NODAL_TRY(slu_apply(h, vj, zj));
                else if (use_sa) NODAL_TRY(sagg_apply(h, vj, zj, j > 0 && x0_slot != nullptr, cyc == 0 ? j : -1));
                else NODAL_TRY(amg_apply(h, vj, zj));
                if (j == 0) nodal_nan_probe(h, zj, K, "fgmres z0 (node block)");
                if (n > K) {
                    branch_solve<<<grid_for(n - K), TB, 0, st>>>(indptr, indices, data, K, (int)n,
                                                                h->schur.as<double>(), vj, zj);
                    NODAL_HIP_TRY(h, hipGetLastError());
                }
                // w = A z_j   (one launch per batch is timed)
                if (ell_spmv) {
                    NODAL_TRY(sagg_spmv(h, zj, w, c == 0 ? e0 : nullptr, c == 0 ? e1 : nullptr));
                } else {
                    if (c == 0) NODAL_HIP_TRY(h, hipEventRecord(e0, st));
                    NODAL_TRY(csr_spmv(h, zj, w));
                    if (c == 0) NODAL_HIP_TRY(h, hipEventRecord(e1, st));
                }
                if (j == 0) {
                    nodal_nan_probe(h, zj, n, "fgmres z0");
                    nodal_nan_probe(h, w, n, "fgmres w0 = A z0");
                }
                // classical Gram-Schmidt, twice, against the last `window` basis vectors (see window_first above)
                const int nv_all = j + 1;
                const int s0 = nv_all > window ? nv_all - window : 0, nv = nv_all - s0;
                const double *Vw = V + (int64_t)s0 * ld;
                DISPATCH_NV(nv, (gs_dots<NV><<<gd, TB, 0, st>>>(Vw, ld, nv, w, n, partial)));
                gs_reduce<<<nv, TB, 0, st>>>(partial, (int)gd, nv, hdev + s0, nullptr, 0);
                DISPATCH_NV(nv, (gs_update_dots<NV><<<gd, TB, 0, st>>>(Vw, ld, nv, hdev + s0, w, n, partial)));
                gs_reduce<<<nv, TB, 0, st>>>(partial, (int)gd, nv, hdev2 + s0, nullptr, 0);
                DISPATCH_NV(nv, (gs_update<NV><<<gd, TB, 0, st>>>(Vw, ld, nv, hdev2 + s0, w, n, partial2)));
                gs_finish<<<1, TB, 0, st>>>(hdev, hdev2, partial2, (int)gd, j, s0, gst);
                scale_by_device<<<gv, TB, 0, st>>>(w, gst, V + (int64_t)(j + 1) * ld, n, x0_dinv, x0_slot, x0_n, x0_omega);
                NODAL_HIP_TRY(h, hipGetLastError());
            }
            NODAL_TRY(nodal_read_words(h, hst, gst + G_INV_H, sizeof hst));
            float ms = 0;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { h->kern_ms += ms; h->kern_launches += 1; }
            const double est = hst[G_EST - G_INV_H];
            done = hst[G_DONE - G_INV_H] != 0.0;
            if (hst[G_INFO - G_INV_H] != 0.0) *info = 1;
            if (done) break;
            // contraction per iteration since the last poll -> iterations still missing
            double need = RESTART;
            if (est > 0.0 && est < est_prev && enq > at_prev) {
                const double rate = std::log(est / est_prev) / (double)(enq - at_prev);
                need = std::log(tol * bnorm / est) / rate;
            }
            batch = (int)std::floor(0.75 * need);
            if (batch < 2) batch = 2;
            if (batch > enq) batch = enq;  // (at most doubling)
            if (direct) batch = 1;
            est_prev = est;
            at_prev = enq;
        }
        total += (int)hst[G_COUNT - G_INV_H];
        if (getenv("NODAL_TRACE")) {
            double lg[RESTART];
            NODAL_TRY(nodal_read_words(h, lg, gst + G_LOG, sizeof lg));
            fprintf(stderr, "[fgmres] |w after Gram-Schmidt|^2 / |w|^2 per column:");
            for (int i = 0; i < (int)hst[G_COUNT - G_INV_H] && i < RESTART; ++i) fprintf(stderr, " %.2g", lg[i]);
            fprintf(stderr, "\n");
        }
        if (*info) break;
        // y = H^-1 g ; x += Z y
        gst_solve<<<1, 64, 0, st>>>(gst, ydev);
        add_combination<<<gv, TB, 0, st>>>(Z, ld, RESTART, ydev, x, gst, n);
        // true residual
        NODAL_TRY(system_spmv(x, w));
        residual_of<<<gd, TB, 0, st>>>(b, w, r, n, partial2);
        NODAL_TRY(device_norm(&rnorm));
        if (!(rnorm == rnorm)) { *info = 1; break; }
        if (rnorm <= 10.0 * tol * bnorm) converged = true;
        if (!converged && direct && h->slu_strict) {
            // (second opinion of sparse_direct_solve: a pseudo-random right-hand side.  A singular G leaves a
            // residual of ~|b| / sqrt(n) however large x grows -- and a huge x makes the backward error small)
            if (rnorm <= 1e-8 * bnorm) converged = true;
        } else if (!converged && direct) {
            // Refinement on LU factors is judged by the backward error, like every direct solve here (tests:
            // scaled residual <= 1e-14): |r|_2 / |b|_2 has a floor of eps |A| |x| / |b| -- 1.4e-12 for the
            // 1e6-node grid driven at one corner -- that no solver gets under in fp64.
            double scaled = 1.0;
            NODAL_TRY(csr_scaled_residual(h, x, b, &scaled));
            if (scaled <= 2e-15) converged = true;
            else if (cyc > 0 && scaled <= 1e-14) converged = true;  // (a second cycle did not improve it further)
        }
    }
    *iters = total;
    *resid = rnorm / bnorm;
    h->kern_alg = 12.0 * (double)h->nnz + 4.0 * (double)(n + 1) + 16.0 * (double)n;
    if (*info) return NODAL_OK;  // numerically singular: caller fills NaNs (reference quirk 3)
    if (!converged && direct && !h->slu_strict && slu_perturbed(h) == 0) {
        // No pivot was replaced: the factors are those of G itself (pivoting restricted to each front's fully summed
        // rows), and a refinement that misses the 1e-14 backward-error bar says "ill-conditioned or high growth", not
        // "singular" -- the reference's SuperLU (nodal/nodal.py:325) returns its solution for such a system, and so
        // does this route: the best iterate, provided it is a solution at all (backward error <= 1e-9).  (Advisor,
        // round 4: singular only on positive evidence.)
        double scaled = 1.0;
        NODAL_TRY(csr_scaled_residual(h, x, b, &scaled));
        if (getenv("NODAL_TRACE"))
            fprintf(stderr, "[direct] refinement stopped at backward error %.3e with no replaced pivot: %s\n", scaled,
                    scaled <= 1e-9 ? "accepted (ill-conditioned, not singular)" : "not a solution");
        if (scaled <= 1e-9) return NODAL_OK;
    }
    if (!converged && direct) {
        // Refinement on the LU factors did not reach the residual bar: the statically perturbed pivots stood in
        // for zero ones -- G is singular to working precision.  The reference's spsolve meets the zero pivot
        // and returns NaNs + MatrixRankWarning (reference nodal/nodal.py:323-336); so does the caller.
        if (getenv("NODAL_TRACE"))
            fprintf(stderr, "[direct] refinement stalled at relative residual %.3e after %d iterations: singular\n",
                    rnorm / bnorm, total);
        *info = 1;
        return NODAL_OK;
    }
    if (!converged) {
        char msg[256];
        snprintf(msg, sizeof msg,
                 "sparse general solve: FGMRES did not converge (relative residual %.3e after %d "
                 "iterations); the matrix may be singular or too ill-conditioned for the "
                 "iterative path", rnorm / bnorm, total);
        return nodal_fail(h, NODAL_E_UNSUPPORTED, msg);
    }
    return NODAL_OK;
}

}  // namespace

int sparse_general_solve(nodal_ctx *h, int32_t *info, int32_t *iters, double *resid) {
    return general_impl(h, h->rhs.as<double>(), h->x.as<double>(), info, iters, resid, false);
}

int general_krylov_direct(nodal_ctx *h, const double *b, double *x, int32_t *info, int32_t *iters, double *resid) {
    return general_impl(h, b, x, info, iters, resid, true);
}
