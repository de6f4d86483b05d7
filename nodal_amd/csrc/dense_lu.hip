// Dense path of Circuit.solve: LU with partial (row) pivoting + triangular
// solves, fp64 -- the arithmetic LAPACK dgesv performs behind
// np.linalg.solve(G, A) (reference nodal/nodal.py:327).
//
// Layout: column-major n x (n+1) in HBM; column n is the right-hand side, so the
// row interchanges and the forward substitution L y = P b happen as part of the
// blocked factorisation (the rhs is just one more trailing column).  Pivot rule
// as LAPACK idamax: first row of maximal |a|; an exactly zero pivot sets
// info = column + 1 and the factorisation continues without scaling.
#include "ctx.h"

namespace {

constexpr int NB = 32;  // panel width

struct MaxLoc {
    double v;
    int i;
};

__device__ __forceinline__ MaxLoc better(MaxLoc a, MaxLoc b) {
    // larger magnitude wins; on ties the smaller row index (idamax semantics)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// One workgroup: find the pivot of column c among rows c..n-1, record it, then
// swap rows c and pivot inside the panel columns [j0, j1).
__global__ __launch_bounds__(1024) void lu_pivot_swap(double *__restrict__ A, int64_t n,
                                                      int64_t lda, int c, int j0, int j1,
                                                      int32_t *__restrict__ piv,
                                                      int32_t *__restrict__ info) {
    __shared__ MaxLoc part[16];
    __shared__ int prow;
    const double *col = A + (int64_t)c * lda;
    MaxLoc best{-1.0, 0x7fffffff};
    for (int i = c + threadIdx.x; i < n; i += 1024) {
        const double v = fabs(col[i]);
        // NaN never compares greater: it is skipped exactly as idamax skips it
        if (v > best.v) best = MaxLoc{v, i};
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        MaxLoc o{__shfl_down(best.v, off, 64), __shfl_down(best.i, off, 64)};
        best = better(best, o);
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        MaxLoc b = part[0];
        for (int w = 1; w < 16; ++w) b = better(b, part[w]);
        if (b.i == 0x7fffffff) b.i = c;  // column of NaNs: keep the diagonal
        prow = b.i;
        piv[c] = b.i;
        if (!(b.v > 0.0) && !(b.v != b.v)) atomicCAS(info, 0, c + 1);  // exact zero pivot
    }
    __syncthreads();
    const int p = prow;
    if (p != c) {
        for (int q = j0 + threadIdx.x; q < j1; q += 1024) {
            double *cq = A + (int64_t)q * lda;
            const double t = cq[c];
            cq[c] = cq[p];
            cq[p] = t;
        }
    }
}

// rows below the diagonal of column c: l = a / pivot, then rank-1 update of the
// remaining panel columns (c, j1)
__global__ __launch_bounds__(256) void lu_scale_update(double *__restrict__ A, int64_t n,
                                                       int64_t lda, int c, int j1) {
    const double pivot = A[(int64_t)c * lda + c];
    if (pivot == 0.0) return;
    const double rcp = 1.0 / pivot;
    for (int64_t i = c + 1 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const double l = A[(int64_t)c * lda + i] * rcp;
        A[(int64_t)c * lda + i] = l;
        for (int q = c + 1; q < j1; ++q) {
            double *cq = A + (int64_t)q * lda;
            cq[i] = fma(-l, cq[c], cq[i]);
        }
    }
}

// apply the panel's interchanges to every column outside the panel
__global__ __launch_bounds__(256) void lu_swap_outside(double *__restrict__ A, int64_t lda,
                                                       int64_t ncols, int j0, int j1,
                                                       const int32_t *__restrict__ piv) {
    for (int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x; q < ncols;
         q += (int64_t)gridDim.x * 256) {
        if (q >= j0 && q < j1) continue;
        double *cq = A + q * lda;
        for (int c = j0; c < j1; ++c) {
            const int p = piv[c];
            if (p != c) {
                const double t = cq[c];
                cq[c] = cq[p];
                cq[p] = t;
            }
        }
    }
}

// U12 = L11^-1 A12 for the columns right of the panel (unit lower triangular)
__global__ __launch_bounds__(256) void lu_trsm(double *__restrict__ A, int64_t lda, int64_t ncols,
                                               int j0, int j1) {
    __shared__ double L[NB][NB + 1];
    const int nb = j1 - j0;
    for (int t = threadIdx.x; t < NB * NB; t += 256) {
        const int r = t % NB, s = t / NB;
        L[r][s] = (r < nb && s < nb) ? A[(int64_t)(j0 + s) * lda + j0 + r] : 0.0;
    }
    __syncthreads();
    for (int64_t q = j1 + (int64_t)blockIdx.x * 256 + threadIdx.x; q < ncols;
         q += (int64_t)gridDim.x * 256) {
        double *cq = A + q * lda + j0;
        double u[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) u[r] = r < nb ? cq[r] : 0.0;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
#pragma unroll
            for (int s = r + 1; s < NB; ++s) u[s] = fma(-L[s][r], u[r], u[s]);
        }
#pragma unroll
        for (int r = 0; r < NB; ++r)
            if (r < nb) cq[r] = u[r];
    }
}

// trailing update C -= L21 * U12 : C rows [j1, n), cols [j1, ncols), K = [j0, j1)
// 64 x 64 tile per workgroup, 4 x 4 per lane, operands staged through LDS.
__global__ __launch_bounds__(256) void lu_gemm(double *__restrict__ A, int64_t n, int64_t lda,
                                               int64_t ncols, int j0, int j1) {
    __shared__ double Ls[NB][64 + 1];  // [k][row]
    __shared__ double Us[NB][64 + 1];  // [k][col]
    const int nb = j1 - j0;
    const int64_t row0 = j1 + (int64_t)blockIdx.x * 64;
    const int64_t col0 = j1 + (int64_t)blockIdx.y * 64;
    for (int t = threadIdx.x; t < NB * 64; t += 256) {
        const int r = t % 64, kk = t / 64;
        const int64_t gr = row0 + r;
        Ls[kk][r] = (kk < nb && gr < n) ? A[(int64_t)(j0 + kk) * lda + gr] : 0.0;
    }
    for (int t = threadIdx.x; t < NB * 64; t += 256) {
        const int kk = t % NB, cc = t / NB;
        const int64_t gc = col0 + cc;
        Us[kk][cc] = (kk < nb && gc < ncols) ? A[gc * lda + j0 + kk] : 0.0;
    }
    __syncthreads();
    const int tr = (threadIdx.x % 16) * 4, tc = (threadIdx.x / 16) * 4;
    double acc[4][4] = {};
#pragma unroll 8
    for (int kk = 0; kk < NB; ++kk) {
        double a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = Ls[kk][tr + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[j] = Us[kk][tc + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = fma(a[i], b[j], acc[i][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t gc = col0 + tc + j;
        if (gc >= ncols) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gr = row0 + tr + i;
            if (gr < n) A[gc * lda + gr] -= acc[i][j];
        }
    }
}

// back substitution, one diagonal block: solve U[j0:j1, j0:j1] x = y in place
__global__ __launch_bounds__(64) void bs_diag(const double *__restrict__ A, int64_t lda,
                                              double *__restrict__ y, int j0, int j1) {
    __shared__ double U[NB][NB + 1];
    __shared__ double x[NB];
    const int nb = j1 - j0;
    for (int t = threadIdx.x; t < NB * NB; t += 64) {
        const int r = t % NB, s = t / NB;
        U[r][s] = (r < nb && s < nb) ? A[(int64_t)(j0 + s) * lda + j0 + r] : 0.0;
    }
    if (threadIdx.x < NB) x[threadIdx.x] = threadIdx.x < nb ? y[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    for (int r = nb - 1; r >= 0; --r) {
        if (threadIdx.x == 0) x[r] = x[r] / U[r][r];
        __syncthreads();
        if ((int)threadIdx.x < r) x[threadIdx.x] = fma(-U[threadIdx.x][r], x[r], x[threadIdx.x]);
        __syncthreads();
    }
    if ((int)threadIdx.x < nb) y[j0 + threadIdx.x] = x[threadIdx.x];
}

// y[0:j0] -= U[0:j0, j0:j1] * x[j0:j1]
__global__ __launch_bounds__(256) void bs_update(const double *__restrict__ A, int64_t lda,
                                                 double *__restrict__ y, int j0, int j1) {
    __shared__ double x[NB];
    if (threadIdx.x < NB) x[threadIdx.x] = (j0 + (int)threadIdx.x < j1) ? y[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    const int nb = j1 - j0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j0;
         i += (int64_t)gridDim.x * 256) {
        double acc = y[i];
        for (int s = 0; s < nb; ++s) acc = fma(-A[(int64_t)(j0 + s) * lda + i], x[s], acc);
        y[i] = acc;
    }
}

__global__ __launch_bounds__(256) void fill_nan(double *x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        x[i] = __builtin_nan("");
}

unsigned blocks_for(int64_t work, int per_block) {
    int64_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (unsigned)b;
}

}  // namespace

int dense_fill_nan(nodal_ctx *h, double *x, int64_t n) {
    fill_nan<<<blocks_for(n, 256), 256, 0, h->stream>>>(x, n);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// Factor the column-major n x (n+1) augmented matrix in h->dense and leave the
// solution in h->x.  *info as LAPACK dgesv.
int dense_factor_solve(nodal_ctx *h, int32_t *info) {
    const int64_t n = h->n, lda = n, ncols = n + 1;
    hipStream_t st = h->stream;
    double *A = h->dense.as<double>();
    NODAL_HIP_TRY(h, h->piv.reserve((size_t)n * 4 + 64));
    int32_t *piv = h->piv.as<int32_t>();
    int32_t *dinfo = piv + n;  // one spare word after the pivots
    NODAL_HIP_TRY(h, hipMemsetAsync(dinfo, 0, 4, st));

    // HIP-event pairs around every trailing-update launch (the dominant kernel)
    const size_t npanels = (size_t)((n + NB - 1) / NB);
    while (h->evpool.size() < 2 * npanels) {
        hipEvent_t e;
        NODAL_HIP_TRY(h, hipEventCreate(&e));
        h->evpool.push_back(e);
    }
    size_t timed = 0;
    double flops = 0;

    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        const int64_t j1 = j0 + NB < n ? j0 + NB : n;
        for (int64_t c = j0; c < j1; ++c) {
            lu_pivot_swap<<<1, 1024, 0, st>>>(A, n, lda, (int)c, (int)j0, (int)j1, piv, dinfo);
            if (c + 1 < n)
                lu_scale_update<<<blocks_for(n - c - 1, 256), 256, 0, st>>>(A, n, lda, (int)c,
                                                                           (int)j1);
        }
        lu_swap_outside<<<blocks_for(ncols, 256), 256, 0, st>>>(A, lda, ncols, (int)j0, (int)j1,
                                                               piv);
        lu_trsm<<<blocks_for(ncols - j1, 256), 256, 0, st>>>(A, lda, ncols, (int)j0, (int)j1);
        if (j1 < n) {
            dim3 grid((unsigned)((n - j1 + 63) / 64), (unsigned)((ncols - j1 + 63) / 64));
            NODAL_HIP_TRY(h, hipEventRecord(h->evpool[2 * timed], st));
            lu_gemm<<<grid, 256, 0, st>>>(A, n, lda, ncols, (int)j0, (int)j1);
            NODAL_HIP_TRY(h, hipEventRecord(h->evpool[2 * timed + 1], st));
            ++timed;
            flops += 2.0 * (double)(j1 - j0) * (double)(n - j1) * (double)(ncols - j1);
        }
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    // back substitution on the transformed rhs (column n)
    double *y = A + n * lda;
    for (int64_t j1 = n; j1 > 0;) {
        int64_t j0 = ((j1 - 1) / NB) * NB;
        bs_diag<<<1, 64, 0, st>>>(A, lda, y, (int)j0, (int)j1);
        if (j0 > 0) bs_update<<<blocks_for(j0, 256), 256, 0, st>>>(A, lda, y, (int)j0, (int)j1);
        j1 = j0;
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipMemcpyAsync(h->x.p, y, (size_t)n * 8, hipMemcpyDeviceToDevice, st));
    int32_t hinfo = 0;
    NODAL_HIP_TRY(h, hipMemcpyAsync(&hinfo, dinfo, 4, hipMemcpyDeviceToHost, st));
    NODAL_HIP_TRY(h, hipStreamSynchronize(st));
    *info = hinfo;
    h->kern_ms = 0;
    for (size_t i = 0; i < timed; ++i) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, h->evpool[2 * i], h->evpool[2 * i + 1]) == hipSuccess)
            h->kern_ms += ms;
    }
    h->kern_launches = (int64_t)timed;
    h->kern_alg = timed ? flops / (double)timed : 0.0;  // average flops per launch
    return NODAL_OK;
}
