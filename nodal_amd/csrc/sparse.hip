// Sparse path of Circuit.solve (replaces scipy.sparse.linalg.spsolve,
// reference nodal/nodal.py:325) on the CSR matrix built by stamp.hip.
//
//   * B == 0 (resistors and current sources only), more than 4096 unknowns: G is a
//     symmetric weighted graph Laplacian plus ground conductances -> flexible conjugate
//     gradients in fp64 preconditioned by the aggregation multigrid of amg.hip (Jacobi-CG
//     as the fallback).  No atomics, no host round trip per iteration: dot products are
//     block partials that the NEXT kernel's prologue reduces (every workgroup redundantly,
//     in a fixed order, so the result is deterministic).  Row kernels are CSR-stream
//     (spmv_stream.h).  Smaller passive systems are solved directly (block_elim.hip).
//   * otherwise (branch equations present: zero diagonals, non-symmetric): small
//     systems are scattered into a dense column-major panel and factorised by
//     the dense LU (dense_lu.hip); large ones go to the block-preconditioned flexible
//     GMRES of sparse_general.hip.
//
// Singular systems: x is filled with NaN and info > 0, without an error status,
// because the reference's sparse path warns and returns NaNs (SURVEY.md section 0
// quirk 3).
#include <chrono>
#include <utility>
#include <cstdio>
#include <cstdlib>

#include "ctx.h"
#include "spmv_stream.h"

int dense_fill_nan(nodal_ctx *h, double *x, int64_t n);

namespace {

constexpr int TB = 256;
constexpr int MAX_PARTIALS = 1024;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// sum of one value per thread over the workgroup; valid in every thread
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

// every workgroup reduces the same partial array in the same order
__device__ __forceinline__ double reduce_partials(const double *__restrict__ part, int count) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += TB) s += part[i];
    return block_sum(s);
}

template <int LPR>
__device__ __forceinline__ double row_dot(const int32_t *__restrict__ indptr,
                                          const int32_t *__restrict__ indices,
                                          const double *__restrict__ data,
                                          const double *__restrict__ x, int64_t row, int sub) {
    double s = 0.0;
    const int32_t e1 = indptr[row + 1];
    for (int32_t e = indptr[row] + sub; e < e1; e += LPR) s = fma(data[e], x[indices[e]], s);
#pragma unroll
    for (int off = LPR >> 1; off > 0; off >>= 1) s += __shfl_down(s, off, LPR);
    return s;  // valid in sub == 0
}

// y = A x  (CSR-stream)
__global__ __launch_bounds__(TB) void spmv_kernel(const int32_t *__restrict__ indptr,
                                                  const int32_t *__restrict__ indices,
                                                  const double *__restrict__ data,
                                                  const double *__restrict__ x,
                                                  double *__restrict__ y, int64_t n) {
    stream::for_rows(
        indptr, indices, data, n, [&](int32_t, int32_t col, double val) { return val * x[col]; },
        [&](int64_t r, double sum) { y[r] = sum; });
}

// scalars kept on the device (doubles)
enum { S_RZ0 = 0, S_RZ1 = 1, S_RR = 2, S_BB = 3, S_PAP = 4, S_FLAG = 5, S_COUNT = 8 };

// z = r * dinv (r = b, x = 0), partials of r.z and r.r
__global__ __launch_bounds__(TB) void pcg_init(const double *__restrict__ b,
                                               const double *__restrict__ dinv,
                                               double *__restrict__ x, double *__restrict__ r,
                                               double *__restrict__ z, double *__restrict__ part_rz,
                                               double *__restrict__ part_rr, int64_t n) {
    double srz = 0.0, srr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double ri = b[i];
        const double zi = ri * dinv[i];
        x[i] = 0.0;
        r[i] = ri;
        z[i] = zi;
        srz = fma(ri, zi, srz);
        srr = fma(ri, ri, srr);
    }
    srz = block_sum(srz);
    srr = block_sum(srr);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = srz;
        part_rr[blockIdx.x] = srr;
    }
}

// K1: rz_new from partials; beta = rz_new / rz_old; p = z + beta p
__global__ __launch_bounds__(TB) void pcg_direction(const double *__restrict__ z,
                                                    double *__restrict__ p,
                                                    const double *__restrict__ part_rz,
                                                    const double *__restrict__ part_rr,
                                                    int nparts, double *__restrict__ sc, int iter,
                                                    int64_t n) {
    const double rz_new = reduce_partials(part_rz, nparts);
    const double rr = reduce_partials(part_rr, nparts);
    const double rz_old = iter > 0 ? sc[(iter - 1) & 1] : 1.0;
    const double beta = (iter > 0 && rz_old != 0.0) ? rz_new / rz_old : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[iter & 1] = rz_new;
        sc[S_RR] = rr;
        if (iter == 0) sc[S_BB] = rr;
    }
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        p[i] = iter > 0 ? fma(beta, p[i], z[i]) : z[i];
}

// K2: Ap = A p, partials of p.Ap  (CSR-stream: spmv_stream.h)
__global__ __launch_bounds__(TB) void pcg_spmv(const int32_t *__restrict__ indptr,
                                               const int32_t *__restrict__ indices,
                                               const double *__restrict__ data,
                                               const double *__restrict__ p,
                                               double *__restrict__ Ap,
                                               double *__restrict__ part_pap, int64_t n) {
    double acc = 0.0;
    stream::for_rows(
        indptr, indices, data, n,
        [&](int32_t, int32_t col, double val) { return val * p[col]; },
        [&](int64_t r, double sum) {
            Ap[r] = sum;
            acc = fma(p[r], sum, acc);
        });
    acc = block_sum(acc);
    if (threadIdx.x == 0) part_pap[blockIdx.x] = acc;
}

// K3: alpha = rz / pAp; x += alpha p; r -= alpha Ap; z = r dinv; partials r.z, r.r
__global__ __launch_bounds__(TB) void pcg_update(double *__restrict__ x, double *__restrict__ r,
                                                 double *__restrict__ z,
                                                 const double *__restrict__ p,
                                                 const double *__restrict__ Ap,
                                                 const double *__restrict__ dinv,
                                                 const double *__restrict__ part_pap,
                                                 double *__restrict__ part_rz,
                                                 double *__restrict__ part_rr, int nparts,
                                                 double *__restrict__ sc, int iter, int64_t n) {
    const double pap = reduce_partials(part_pap, nparts);
    const double rz = sc[iter & 1];
    // breakdown (indefinite or singular G): freeze the iteration, raise the flag
    const bool bad = !(pap > 0.0) && rz != 0.0;
    const double alpha = (pap > 0.0) ? rz / pap : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[S_PAP] = pap;
        if (bad) sc[S_FLAG] = 1.0;
    }
    double srz = 0.0, srr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        x[i] = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, Ap[i], r[i]);
        const double zi = ri * dinv[i];
        r[i] = ri;
        z[i] = zi;
        srz = fma(ri, zi, srz);
        srr = fma(ri, ri, srr);
    }
    srz = block_sum(srz);
    srr = block_sum(srr);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = srz;
        part_rr[blockIdx.x] = srr;
    }
}

// dinv = 1 / diag(G); flag rows whose diagonal is missing or not positive
__global__ __launch_bounds__(TB) void jacobi_setup(const int32_t *__restrict__ diag_pos,
                                                   const double *__restrict__ data,
                                                   double *__restrict__ dinv,
                                                   double *__restrict__ sc, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const int32_t e = diag_pos[i];
        const double d = e >= 0 ? data[e] : 0.0;
        if (!(d > 0.0)) sc[S_FLAG] = 2.0;  // benign race: every writer stores the same value
        dinv[i] = d > 0.0 ? 1.0 / d : 1.0;
    }
}

__device__ __forceinline__ void atomic_max_nonneg(double *addr, double v) {
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

// out[0] = max_i |(A x - b)_i|, out[1] = max row sum |A|, out[2] = max|x|, out[3] = max|b|
// NaNs anywhere in x poison out[0] via out[4].
template <int LPR>
__global__ __launch_bounds__(TB) void residual_kernel(const int32_t *__restrict__ indptr,
                                                      const int32_t *__restrict__ indices,
                                                      const double *__restrict__ data,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ b,
                                                      double *__restrict__ out, int64_t n) {
    const int sub = threadIdx.x % LPR;
    const int64_t rows_per_pass = (int64_t)gridDim.x * (TB / LPR);
    const int64_t passes = (n + rows_per_pass - 1) / rows_per_pass;
    // per-thread maxima, reduced per block: one atomic per block and quantity (a million
    // same-address atomics serialise in L2 and cost milliseconds)
    double m_res = 0.0, m_an = 0.0, m_x = 0.0, m_b = 0.0;
    bool poisoned = false;
    for (int64_t it = 0; it < passes; ++it) {
        const int64_t row = it * rows_per_pass + (int64_t)blockIdx.x * (TB / LPR) + threadIdx.x / LPR;
        const bool live = row < n;
        double s = 0.0, an = 0.0;
        if (live) {
            const int32_t e1 = indptr[row + 1];
            for (int32_t e = indptr[row] + sub; e < e1; e += LPR) {
                s = fma(data[e], x[indices[e]], s);
                an += fabs(data[e]);
            }
        }
#pragma unroll
        for (int off = LPR >> 1; off > 0; off >>= 1) {
            s += __shfl_down(s, off, LPR);
            an += __shfl_down(an, off, LPR);
        }
        if (live && sub == 0) {
            const double res = fabs(s - b[row]);
            if (res != res || x[row] != x[row]) poisoned = true;
            else {
                m_res = fmax(m_res, res);
                m_x = fmax(m_x, fabs(x[row]));
            }
            m_an = fmax(m_an, an);
            m_b = fmax(m_b, fabs(b[row]));
        }
    }
    __shared__ double red[4][TB / 64];
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    if (poisoned) bad = 1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        m_res = fmax(m_res, __shfl_down(m_res, off));
        m_an = fmax(m_an, __shfl_down(m_an, off));
        m_x = fmax(m_x, __shfl_down(m_x, off));
        m_b = fmax(m_b, __shfl_down(m_b, off));
    }
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        red[0][w] = m_res; red[1][w] = m_an; red[2][w] = m_x; red[3][w] = m_b;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double m = 0.0;
        for (int w = 0; w < TB / 64; ++w) m = fmax(m, red[threadIdx.x][w]);
        // out[0] res, out[1] |A| row sum, out[2] |x|, out[3] |b|
        if (m > 0.0) atomic_max_nonneg(&out[threadIdx.x], m);
    }
    if (threadIdx.x == 0 && bad) out[4] = 1.0;
}

int lanes_per_row(nodal_ctx *h) {
    const double avg = h->n > 0 ? (double)h->nnz / (double)h->n : 1.0;
    int lpr = 2;
    while (lpr < 64 && lpr < avg) lpr <<= 1;
    return lpr;
}

unsigned grid_rows(int64_t n, int lpr) {
    int64_t g = (n * lpr + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > MAX_PARTIALS ? MAX_PARTIALS : g);
}

#define DISPATCH_LPR(lpr, CALL)          \
    switch (lpr) {                       \
    case 2: { constexpr int L = 2; CALL; } break;   \
    case 4: { constexpr int L = 4; CALL; } break;   \
    case 8: { constexpr int L = 8; CALL; } break;   \
    case 16: { constexpr int L = 16; CALL; } break; \
    case 32: { constexpr int L = 32; CALL; } break; \
    default: { constexpr int L = 64; CALL; } break; \
    }

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

int pcg_solve(nodal_ctx *h, int32_t *info, int32_t *iters, double *resid) {
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    const size_t vec = align_up((size_t)n * 8);
    // solver layout: r | z | p | Ap | dinv | partials 3 x MAX | scalars
    NODAL_HIP_TRY(h, h->solver.reserve(5 * vec + 3 * MAX_PARTIALS * 8 + 256));
    char *base = h->solver.as<char>();
    double *r = reinterpret_cast<double *>(base);
    double *z = reinterpret_cast<double *>(base + vec);
    double *p = reinterpret_cast<double *>(base + 2 * vec);
    double *Ap = reinterpret_cast<double *>(base + 3 * vec);
    double *dinv = reinterpret_cast<double *>(base + 4 * vec);
    double *part_rz = reinterpret_cast<double *>(base + 5 * vec);
    double *part_rr = part_rz + MAX_PARTIALS;
    double *part_pap = part_rr + MAX_PARTIALS;
    double *sc = part_pap + MAX_PARTIALS;
    double *x = h->x.as<double>();
    const int32_t *indptr = h->indptr.as<int32_t>();
    const int32_t *indices = h->indices.as<int32_t>();
    const double *data = h->data.as<double>();
    const double *b = h->rhs.as<double>();

    const unsigned gv = grid_rows(n, 1);     // vector kernels
    const unsigned gs = stream::grid_for_rows(n, MAX_PARTIALS);   // spmv kernels
    const int nparts_v = (int)gv, nparts_s = (int)gs;

    NODAL_HIP_TRY(h, hipMemsetAsync(sc, 0, S_COUNT * 8, st));
    jacobi_setup<<<gv, TB, 0, st>>>(h->diag_pos.as<int32_t>(), data, dinv, sc, n);
    pcg_init<<<gv, TB, 0, st>>>(b, dinv, x, r, z, part_rz, part_rr, n);
    NODAL_HIP_TRY(h, hipGetLastError());

    const double tol = 1e-13;
    // Jacobi-CG needs ~5.5 sqrt(n) iterations on a grid; ten times that and it is not going to
    // converge (the old cap of 20 n iterations meant minutes of silence at 1e5 unknowns)
    const int64_t maxit = 1000 + (int64_t)(50.0 * sqrt((double)n));
    const int check = 50;
    double hs[S_COUNT];
    int64_t it = 0;
    int status = 0;  // 0 running, 1 converged, 2 breakdown, 3 maxit
    hipEvent_t e0 = h->ev[2], e1 = h->ev[3];
    h->kern_ms = 0;
    h->kern_launches = 0;
    while (status == 0) {
        for (int c = 0; c < check; ++c, ++it) {
            pcg_direction<<<gv, TB, 0, st>>>(z, p, part_rz, part_rr, nparts_v, sc, (int)it, n);
            const bool timed = (c == check - 1);
            if (timed) NODAL_HIP_TRY(h, hipEventRecord(e0, st));
            pcg_spmv<<<gs, TB, 0, st>>>(indptr, indices, data, p, Ap, part_pap, n);
            if (timed) NODAL_HIP_TRY(h, hipEventRecord(e1, st));
            pcg_update<<<gv, TB, 0, st>>>(x, r, z, p, Ap, dinv, part_pap, part_rz, part_rr,
                                          nparts_s, sc, (int)it, n);
        }
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_TRY(nodal_read_words(h, hs, sc, S_COUNT * 8));
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
            h->kern_ms += ms;
            h->kern_launches += 1;
        }
        // sc[S_RR] is |r|^2 as of the direction kernel of the last iteration
        if (hs[S_FLAG] != 0.0) status = 2;
        else if (!(hs[S_RR] == hs[S_RR])) status = 2;
        else if (hs[S_BB] == 0.0 || hs[S_RR] <= tol * tol * hs[S_BB]) status = 1;
        else if (it >= maxit) status = 3;
    }
    *iters = (int32_t)it;
    *resid = hs[S_BB] > 0 ? sqrt(hs[S_RR] / hs[S_BB]) : 0.0;
    // algorithmic bytes of one SpMV launch: matrix once, x and y once
    h->kern_alg = 12.0 * (double)h->nnz + 4.0 * (double)(n + 1) + 16.0 * (double)n;
    if (status == 2) return -1;  // not SPD / singular: caller falls back
    if (status == 3) return -1;  // not converged: never hand back an unconverged vector as a solution
    *info = 0;
    return NODAL_OK;
}

// ---- flexible CG with the multigrid preconditioner (amg.hip) ----------------------
// The K-cycle is a (mildly) non-linear operator, so beta uses the flexible
// (Polak-Ribiere) form  beta = z_new.(r_new - r_old) / (z_old.r_old)
//                             = -alpha (z_new . A p) / (z_old . r_old).

enum { F_RZ0 = 0, F_RZ1 = 1, F_RR = 2, F_BB = 3, F_ALPHA = 4, F_FLAG = 5, F_ITER = 6, F_COUNT = 8 };

// The iteration number lives on the device so that one iteration can be captured into a
// hipGraph and replayed (kernel arguments are frozen in a graph): sc[F_ITER] holds the number
// of the iteration minus one until fcg_dots -- the first CG kernel of an iteration and the
// only one that does not read it -- bumps it.

__global__ __launch_bounds__(TB) void fcg_init(const double *__restrict__ b, double *__restrict__ x,
                                               double *__restrict__ r, double *__restrict__ part_rr,
                                               double *__restrict__ sc, int64_t n) {
    double srr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double ri = b[i];
        x[i] = 0.0;
        r[i] = ri;
        srr = fma(ri, ri, srr);
    }
    srr = block_sum(srr);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = srr;
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[F_ITER] = -1.0;
}

// partials of z.r and z.Ap
__global__ __launch_bounds__(TB) void fcg_dots(const double *__restrict__ z,
                                               const double *__restrict__ r,
                                               const double *__restrict__ Ap,
                                               double *__restrict__ part_rz,
                                               double *__restrict__ part_zap, double *__restrict__ sc,
                                               int64_t n) {
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[F_ITER] += 1.0;  // read by the kernels AFTER this one
    double a = 0.0, c = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double zi = z[i];
        a = fma(zi, r[i], a);
        c = fma(zi, Ap[i], c);
    }
    a = block_sum(a);
    c = block_sum(c);
    if (threadIdx.x == 0) {
        part_rz[blockIdx.x] = a;
        part_zap[blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(TB) void fcg_direction(const double *__restrict__ z,
                                                    double *__restrict__ p,
                                                    const double *__restrict__ part_rz,
                                                    const double *__restrict__ part_zap,
                                                    const double *__restrict__ part_rr, int nparts,
                                                    double *__restrict__ sc, int64_t n) {
    const int iter = (int)sc[F_ITER];
    const double rz_new = reduce_partials(part_rz, nparts);
    const double zap = reduce_partials(part_zap, nparts);
    const double rr = reduce_partials(part_rr, nparts);
    const double rz_old = iter > 0 ? sc[(iter - 1) & 1] : 1.0;
    const double beta = (iter > 0 && rz_old != 0.0) ? -sc[F_ALPHA] * zap / rz_old : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[iter & 1] = rz_new;
        sc[F_RR] = rr;
        if (iter == 0) sc[F_BB] = rr;
        if (!(rz_new >= 0.0)) sc[F_FLAG] = 1.0;  // preconditioner not positive: not SPD
    }
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        p[i] = iter > 0 ? fma(beta, p[i], z[i]) : z[i];
}

__global__ __launch_bounds__(TB) void fcg_update(double *__restrict__ x, double *__restrict__ r,
                                                 const double *__restrict__ p,
                                                 const double *__restrict__ Ap,
                                                 const double *__restrict__ part_pap, int nparts_s,
                                                 double *__restrict__ part_rr,
                                                 double *__restrict__ sc, int64_t n) {
    const int iter = (int)sc[F_ITER];
    const double pap = reduce_partials(part_pap, nparts_s);
    const double rz = sc[iter & 1];
    const bool bad = !(pap > 0.0) && rz != 0.0;
    const double alpha = (pap > 0.0) ? rz / pap : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[F_ALPHA] = alpha;
        if (bad) sc[F_FLAG] = 1.0;
    }
    double srr = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        x[i] = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, Ap[i], r[i]);
        r[i] = ri;
        srr = fma(ri, ri, srr);
    }
    srr = block_sum(srr);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = srr;
}

// b: right-hand side (device).  do_setup: build the hierarchy and run the structural
// singularity check first; false re-uses the hierarchy of the previous call (sweeps
// over right-hand sides, nodal_solve_pairs).
int amg_fcg_solve_ex(nodal_ctx *h, const double *b, bool do_setup, int32_t *info, int32_t *iters,
                     double *resid) {
    {   // networks without graded links / hubs: smoothed aggregation (sagg.hip); it declines the rest
        const int s = sagg_fcg_solve(h, b, do_setup, info, iters, resid);
        if (s == -1) {
            // the smoothed hierarchy took the matrix but the iteration broke down or hit its cap
            // (slowly converging chain-like / graded networks): the plain-aggregation hierarchy below,
            // with its contrast mode and its higher cap, gets its turn before Jacobi-CG does
            sagg_invalidate(h);
            do_setup = true;
        } else if (s != -3) {
            return s;
        }
    }
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    const size_t vec = align_up((size_t)n * 8);
    NODAL_HIP_TRY(h, h->solver.reserve(4 * vec + 4 * MAX_PARTIALS * 8 + 256));
    char *base = h->solver.as<char>();
    double *r = reinterpret_cast<double *>(base);
    double *z = reinterpret_cast<double *>(base + vec);
    double *p = reinterpret_cast<double *>(base + 2 * vec);
    double *Ap = reinterpret_cast<double *>(base + 3 * vec);
    double *part_rz = reinterpret_cast<double *>(base + 4 * vec);
    double *part_rr = part_rz + MAX_PARTIALS;
    double *part_pap = part_rr + MAX_PARTIALS;
    double *part_zap = part_pap + MAX_PARTIALS;
    double *sc = part_zap + MAX_PARTIALS;
    double *x = h->x.as<double>();
    const int32_t *indptr = h->indptr.as<int32_t>();
    const int32_t *indices = h->indices.as<int32_t>();
    const double *data = h->data.as<double>();

    const unsigned gv = grid_rows(n, 1), gs = stream::grid_for_rows(n, MAX_PARTIALS);
    NODAL_HIP_TRY(h, hipMemsetAsync(sc, 0, F_COUNT * 8, st));
    NODAL_HIP_TRY(h, hipMemsetAsync(part_zap, 0, MAX_PARTIALS * 8, st));
    static const bool trace = getenv("NODAL_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    if (do_setup) {
        NODAL_TRY(amg_setup(h, sc + F_FLAG));
        h->amg_levels = amg_num_levels(h);
        if (trace) {
            NODAL_WAIT_STREAM(h, st);
            fprintf(stderr, "[amg] hierarchy %.2f ms\n",
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
        }
        // structurally singular (a floating island): the reference's spsolve returns
        // NaNs; CG would happily return one of the infinitely many solutions
        NODAL_HIP_TRY(h, h->work3.reserve((size_t)n + 256));
        NODAL_TRY(grounded_flags(h, h->work3.as<uint8_t>()));
        int32_t floating = 0;
        NODAL_TRY(amg_has_floating_component(h, h->work3.as<uint8_t>(), &floating));
        if (floating) {
            *info = 1;
            *iters = 0;
            *resid = 0.0;
            return -2;  // singular: caller fills NaNs
        }
    }
    if (trace) {
        NODAL_WAIT_STREAM(h, st);
        fprintf(stderr, "[amg] setup + structural check %.2f ms (%d levels)\n",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(),
                h->amg_levels);
    }
    fcg_init<<<gv, TB, 0, st>>>(b, x, r, part_rr, sc, n);
    NODAL_HIP_TRY(h, hipGetLastError());

    const double tol = 1e-13;
    // chain-like networks need ~1000 iterations; beyond the cap: Jacobi-CG fallback
    const int64_t maxit = getenv("NODAL_FCG_MAXIT") ? atoll(getenv("NODAL_FCG_MAXIT")) : 5000;
    int check = 6;  // iterations before the next look at the residual (first: six; then from the rate seen so far)
    double rr_prev = -1.0;
    int64_t it_prev = 0;
    double hs[F_COUNT];
    int64_t it = 0;
    int status = 0;  // 0 running, 1 converged, 2 breakdown, 3 maxit
    hipEvent_t e0 = h->ev[2], e1 = h->ev[3];
    h->kern_ms = 0;
    h->kern_launches = 0;
    // one iteration = the multigrid cycle + four CG kernels: ~45 small dependent
    // launches.  Capture it once into a hipGraph and replay it; the last iteration of
    // every check interval runs eagerly with HIP events around the SpMV (roofline).
    auto iteration = [&](bool timed) -> int {
        NODAL_TRY(amg_apply(h, r, z));
        fcg_dots<<<gv, TB, 0, st>>>(z, r, Ap, part_rz, part_zap, sc, n);
        fcg_direction<<<gv, TB, 0, st>>>(z, p, part_rz, part_zap, part_rr, (int)gv, sc, n);
        if (timed) NODAL_HIP_TRY(h, hipEventRecord(e0, st));
        pcg_spmv<<<gs, TB, 0, st>>>(indptr, indices, data, p, Ap, part_pap, n);
        if (timed) NODAL_HIP_TRY(h, hipEventRecord(e1, st));
        fcg_update<<<gv, TB, 0, st>>>(x, r, p, Ap, part_pap, (int)gs, part_rr, sc, n);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    };
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    if (h->use_graphs) {
        if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            const int cs = iteration(false);
            const hipError_t ce = hipStreamEndCapture(st, &graph);
            if (cs != NODAL_OK || ce != hipSuccess || !graph ||
                hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
                exec = nullptr;
                (void)hipGetLastError();
            }
        } else {
            (void)hipGetLastError();
        }
    }
    while (status == 0) {
        for (int c = 0; c < check; ++c, ++it) {
            const bool timed = (c == check - 1);
            if (exec && !timed) NODAL_HIP_TRY(h, hipGraphLaunch(exec, st));
            else NODAL_TRY(iteration(timed));
        }
        // The look brings the scalars and the partial sums of |r|^2 the last fcg_update left (one copy: they lie in
        // front of the scalars); the host adds them up in a fixed order -- the residual AFTER the last iteration,
        // where sc[F_RR] is the one before it -- and sizes the next batch from the reduction per iteration seen so
        // far (three quarters of what is still missing), as sagg_fcg_solve does.  (Until round 4: a look every
        // four iterations -- eight of them and two iterations too many on a 32-iteration solve.)
        {
            const size_t bytes = (size_t)(3 * MAX_PARTIALS + F_COUNT) * 8;  // part_rr | part_pap | part_zap | sc
            double *stage = static_cast<double *>(nodal_pinned_arena(h, bytes));
            if (stage) {
                NODAL_HIP_TRY(h, hipMemcpyAsync(stage, part_rr, bytes, hipMemcpyDeviceToHost, st));
                NODAL_WAIT_STREAM(h, st);
                memcpy(hs, stage + 3 * MAX_PARTIALS, F_COUNT * 8);
                double acc = 0.0;
                for (unsigned k = 0; k < gv; ++k) acc += stage[k];
                if (acc == acc && hs[F_FLAG] == 0.0) hs[F_RR] = acc;
            } else {
                NODAL_TRY(nodal_read_words(h, hs, sc, F_COUNT * 8));
            }
        }
        float ms = 0;
        if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
            h->kern_ms += ms;
            h->kern_launches += 1;
        }
        if (hs[F_FLAG] != 0.0 || !(hs[F_RR] == hs[F_RR])) status = 2;
        else if (hs[F_BB] == 0.0 || hs[F_RR] <= tol * tol * hs[F_BB]) status = 1;
        else if (it >= maxit) status = 3;
        else {
            int next = 2;
            const double target = tol * tol * hs[F_BB];
            const double rr_ref = rr_prev > 0.0 ? rr_prev : hs[F_BB];
            const int64_t it_ref = rr_prev > 0.0 ? it_prev : 0;
            if (hs[F_RR] > 0.0 && hs[F_RR] < rr_ref && it > it_ref) {
                const double rate = log(hs[F_RR] / rr_ref) / (double)(it - it_ref);  // < 0
                next = (int)floor(0.75 * log(target / hs[F_RR]) / rate);
            }
            if (next > it) next = (int)it;  // (at most doubling: early rates are pessimistic)
            check = next < 1 ? 1 : (next > 32 ? 32 : next);
            rr_prev = hs[F_RR];
            it_prev = it;
        }
    }
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    *iters = (int32_t)it;
    *resid = hs[F_BB] > 0 ? sqrt(hs[F_RR] / hs[F_BB]) : 0.0;
    h->kern_alg = 12.0 * (double)h->nnz + 4.0 * (double)(n + 1) + 16.0 * (double)n;
    if (status == 2) return -1;  // not SPD / singular: caller falls back
    if (status == 3) return -1;  // stagnation: let the plain path decide
    *info = 0;
    return NODAL_OK;
}

__global__ void pair_rhs(double *__restrict__ b, int32_t ia, int32_t ib) {
    if (ia >= 0) b[ia] += 1.0;   // 1 A enters the first node ...
    if (ib >= 0) b[ib] -= 1.0;   // ... and leaves the second (reference nodal/models.py:27-32)
}
__global__ void pair_read(const double *__restrict__ x, int32_t ia, int32_t ib,
                          double *__restrict__ out) {
    *out = (ia >= 0 ? x[ia] : 0.0) - (ib >= 0 ? x[ib] : 0.0);
}

// ---- sixteen probe pairs at a time through the sparse LU (sparse_solve_pairs_direct below) ----
// vectors interleaved by row: element (i, c) at [i * SLU_MULTI + c]; pair (-1, -1) = unused column
struct PairBlock { int32_t ia[SLU_MULTI], ib[SLU_MULTI]; };
__global__ __launch_bounds__(64) void pair_rhs_multi(double *__restrict__ b, PairBlock pb) {
    const int c = threadIdx.x;
    if (c >= SLU_MULTI) return;
    if (pb.ia[c] >= 0) b[(int64_t)pb.ia[c] * SLU_MULTI + c] += 1.0;
    if (pb.ib[c] >= 0) b[(int64_t)pb.ib[c] * SLU_MULTI + c] -= 1.0;
}
__global__ __launch_bounds__(64) void pair_read_multi(const double *__restrict__ x, PairBlock pb, int count,
                                                      double *__restrict__ out) {
    const int c = threadIdx.x;
    if (c >= count) return;
    out[c] = (pb.ia[c] >= 0 ? x[(int64_t)pb.ia[c] * SLU_MULTI + c] : 0.0) -
             (pb.ib[c] >= 0 ? x[(int64_t)pb.ib[c] * SLU_MULTI + c] : 0.0);
}
// r = b - A x for the sixteen columns: the lanes of a row share every matrix entry and gather 128 contiguous bytes of x
__global__ __launch_bounds__(TB) void csr_residual_multi(int64_t n, const int32_t *__restrict__ indptr,
                                                         const int32_t *__restrict__ indices, const double *__restrict__ data,
                                                         const double *__restrict__ x, const double *__restrict__ b,
                                                         double *__restrict__ r) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < n * SLU_MULTI; e += (int64_t)gridDim.x * TB) {
        const int64_t i = e / SLU_MULTI;
        const int c = (int)(e % SLU_MULTI);
        double acc = b[e];
        for (int32_t q = indptr[i]; q < indptr[i + 1]; ++q) acc = fma(-data[q], x[(int64_t)indices[q] * SLU_MULTI + c], acc);
        r[e] = acc;
    }
}
__global__ __launch_bounds__(TB) void add_into(int64_t count, const double *__restrict__ d, double *__restrict__ x) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < count; e += (int64_t)gridDim.x * TB) x[e] += d[e];
}

__global__ __launch_bounds__(TB) void copy_rhs_column(const double *__restrict__ rhs,
                                                      double *__restrict__ col, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        col[i] = rhs[i];
}

}  // namespace

int dense_prepare(nodal_ctx *h) {
    const int64_t n = h->n, lda = dense_lda(n);
    NODAL_HIP_TRY(h, h->dense.reserve((size_t)lda * (size_t)(n + 1) * 8 + 64));
    if (h->csr_only) NODAL_TRY(csr_to_dense(h, h->dense.as<double>(), lda));
    else NODAL_TRY(stamp_to_dense(h, h->dense.as<double>(), lda, true));
    if (n > 0) {
        copy_rhs_column<<<grid_rows(n, 1), TB, 0, h->stream>>>(h->rhs.as<double>(),
                                                              h->dense.as<double>() + n * lda, n);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    return NODAL_OK;
}

// dense panel with `nrhs` probe-pair columns (e_ia - e_ib) instead of the circuit's rhs
int dense_prepare_pairs(nodal_ctx *h, int32_t nrhs, const int32_t *ia, const int32_t *ib) {
    const int64_t n = h->n, lda = dense_lda(n);
    NODAL_HIP_TRY(h, h->dense.reserve((size_t)lda * (size_t)(n + nrhs) * 8 + 64));
    NODAL_TRY(stamp_to_dense(h, h->dense.as<double>(), lda, true));
    double *cols = h->dense.as<double>() + n * lda;
    NODAL_HIP_TRY(h, hipMemsetAsync(cols, 0, (size_t)lda * nrhs * 8, h->stream));
    for (int32_t q = 0; q < nrhs; ++q) pair_rhs<<<1, 1, 0, h->stream>>>(cols + (int64_t)q * lda, ia[q], ib[q]);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

int sparse_general_solve(nodal_ctx *h, int32_t *info, int32_t *iters, double *resid);  // sparse_general.hip
int general_krylov_direct(nodal_ctx *h, const double *b, double *x, int32_t *info, int32_t *iters, double *resid);  // sparse_general.hip

int amg_fcg_solve(nodal_ctx *h, int32_t *info, int32_t *iters, double *resid) {
    return amg_fcg_solve_ex(h, h->rhs.as<double>(), true, info, iters, resid);
}

// ---- equivalent-resistance sweeps --------------------------
// The reference rebuilds and re-solves the whole circuit per node pair
// (reference nodal/equiv.py:31-61: deepcopy + Circuit + solve).  G does not depend on the
// pair -- only the probe current source does -- so one multigrid setup (sparse) or one
// LU factorisation (dense, the pairs ride along as extra right-hand-side columns) serves
// every pair.


int pair_read_host(nodal_ctx *h, const double *x, int32_t ia, int32_t ib, double *out) {
    pair_read<<<1, 1, 0, h->stream>>>(x, ia, ib, out);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// Factor once, substitute for sixteen pairs at a time (SURVEY 8f N1 in its own words: "one factorisation + batched
// triangular solves"; reference nodal/equiv.py:31-61 rebuilds and re-solves per pair): X = A^-1 B by slu_apply_multi
// and ONE step of refinement on the block (R = B - A X, X += A^-1 R: the factors are those of a statically pivoted LU).
// *taken = false: pivots had to be replaced -- the caller's pair-by-pair route, which judges every answer, takes over.
static int sparse_solve_pairs_direct(nodal_ctx *h, int32_t npairs, const int32_t *ia, const int32_t *ib, double *res_dev,
                                     int32_t *info, bool *taken) {
    *taken = false;
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    int32_t inf = 0;
    NODAL_TRY(slu_factor(h, &inf));
    if (inf > 0) {
        *info = 1;
        *taken = true;
        return NODAL_OK;
    }
    if (slu_perturbed(h) > 0) return NODAL_OK;
    const size_t vb = (size_t)n * SLU_MULTI * 8;
    NODAL_HIP_TRY(h, h->krylov.reserve(3 * vb + 256));
    double *B = h->krylov.as<double>(), *X = B + (size_t)n * SLU_MULTI, *R = X + (size_t)n * SLU_MULTI;
    const unsigned gv = (unsigned)std::min<int64_t>((n * SLU_MULTI + TB - 1) / TB, 65536);
    for (int32_t q = 0; q < npairs; q += SLU_MULTI) {
        const int cnt = npairs - q < SLU_MULTI ? npairs - q : SLU_MULTI;
        PairBlock pb;
        for (int c = 0; c < SLU_MULTI; ++c) {
            pb.ia[c] = c < cnt ? ia[q + c] : -1;
            pb.ib[c] = c < cnt ? ib[q + c] : -1;
        }
        NODAL_HIP_TRY(h, hipMemsetAsync(B, 0, vb, st));
        pair_rhs_multi<<<1, 64, 0, st>>>(B, pb);
        NODAL_TRY(slu_apply_multi(h, B, X));
        csr_residual_multi<<<gv, TB, 0, st>>>(n, h->indptr.as<int32_t>(), h->indices.as<int32_t>(), h->data.as<double>(), X, B, R);
        NODAL_TRY(slu_apply_multi(h, R, B));  // (B is free: the correction lands there)
        add_into<<<gv, TB, 0, st>>>(n * SLU_MULTI, B, X);
        pair_read_multi<<<1, 64, 0, st>>>(X, pb, cnt, res_dev + q);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    h->last_iterations = 1;
    *taken = true;
    return NODAL_OK;
}

int sparse_solve_pairs(nodal_ctx *h, int32_t npairs, const int32_t *ia, const int32_t *ib,
                       double *res_dev, int32_t *info) {
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    NODAL_HIP_TRY(h, h->ps_buf.reserve((size_t)n * 8 + 64));
    double *b = h->ps_buf.as<double>();
    *info = 0;
    bool direct = false;  // the multigrid CG broke down on this network: one sparse LU serves every pair from then on
    // A network with a non-positive resistance (the only non-passive kind a resistance sweep meets: reference
    // nodal/equiv.py:31-37 admits resistors only) is no M-matrix: one sparse LU serves every pair, as SuperLU
    // serves the reference.
    const bool indefinite = !(h->B == 0 && h->passive_network);
    // The factor-once route.  Measured on the 1e6-node grid (round 5, DESIGN 3.5, tools/pairs_probe.py): a factorisation
    // 27 ms (its analysis -- host work, kept per sparsity pattern -- 0.22 s more), two substitutions and a residual for
    // sixteen pairs 11.3 ms, i.e. 0.71 ms per pair against 0.88-0.93 ms per pair of the block iteration: 1024 pairs 0.75 s
    // against 0.90 s, 192 pairs 0.163 against 0.176 s.  So: sweeps of at least PAIRS_DIRECT_MIN (256) pairs when the
    // analysis is at hand, eight times as many when it has to be made; always where the multigrid has no business
    // (indefinite: a non-positive resistance).  NODAL_PAIRS_DIRECT=1 forces it, =0 forbids it.
    {
        const int forced = getenv("NODAL_PAIRS_DIRECT") ? atoi(getenv("NODAL_PAIRS_DIRECT")) : -1;  // (per call: tests switch it)
        const int64_t min_pairs = getenv("NODAL_PAIRS_DIRECT_MIN") ? atoll(getenv("NODAL_PAIRS_DIRECT_MIN")) : 256;
        const bool worth = npairs >= (slu_analysis_kept(h) ? min_pairs : 8 * min_pairs);
        if (forced != 0 && (forced == 1 || indefinite || worth)) {
            bool taken = false;
            NODAL_TRY(sparse_solve_pairs_direct(h, npairs, ia, ib, res_dev, info, &taken));
            if (taken) return NODAL_OK;
        }
    }
    if (indefinite || (getenv("NODAL_PAIRS_DIRECT") && atoi(getenv("NODAL_PAIRS_DIRECT")) == 1)) {  // (pair by pair on the factors)
        int32_t inf = 0;
        NODAL_TRY(slu_factor(h, &inf));
        if (inf > 0) {
            *info = 1;
            return NODAL_OK;
        }
        direct = true;
    }
    // (latched off by the first block that breaks down or does not converge: the pairs of that block and every later
    // one go singly -- a second failing block would cost its iteration cap again for nothing)
    bool block_allowed = !(getenv("NODAL_PAIRS_BLOCK") && atoi(getenv("NODAL_PAIRS_BLOCK")) == 0);
    for (int32_t q = 0; q < npairs;) {
        // Once the first pair has set the smoothed-aggregation hierarchy up, the others go sixteen at a time
        // through the block iteration (sagg_multi.h: one launch sequence and one pass over every matrix per block
        // instead of per pair).  A block that breaks down is redone pair by pair below.
        if (!direct && block_allowed && q > 0 && sagg_ready(h, n) && npairs - q >= 2) {
            const int32_t bw = sagg_pairs_block_width();
            const int32_t cnt = npairs - q < bw ? npairs - q : bw;
            int32_t it = 0;
            const int s = sagg_fcg_solve_pairs_block(h, cnt, ia + q, ib + q, res_dev + q, &it);
            if (s == NODAL_OK) {
                h->last_iterations = it;
                q += cnt;
                continue;
            }
            if (s > 0) return s;
            block_allowed = false;
        }
        NODAL_HIP_TRY(h, hipMemsetAsync(b, 0, (size_t)n * 8, st));
        pair_rhs<<<1, 1, 0, st>>>(b, ia[q], ib[q]);
        int32_t it = 0, inf = 0;
        double rs = 0;
        if (!direct) {
            const int s = amg_fcg_solve_ex(h, b, q == 0, &inf, &it, &rs);
            if (s == -2) {  // floating island: every pair is singular, as in the reference
                *info = 1;
                return NODAL_OK;
            }
            if (s < 0) {
                direct = true;
                NODAL_TRY(slu_factor(h, &inf));
                if (inf > 0) {
                    *info = 1;
                    return NODAL_OK;
                }
            } else if (s != NODAL_OK) {
                return s;
            }
        }
        if (direct) {
            NODAL_TRY(general_krylov_direct(h, b, h->x.as<double>(), &inf, &it, &rs));
            if (inf > 0) {
                *info = 1;
                return NODAL_OK;
            }
        }
        h->last_iterations = it;
        pair_read<<<1, 1, 0, st>>>(h->x.as<double>(), ia[q], ib[q], res_dev + q);
        ++q;
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// Pair sweep on a passive network made of chains / ladders / trees: every pair goes through the
// exact elimination of lowdeg.hip (the probe pair only changes the right-hand side, but a
// reduction costs a few ms where the multigrid needs hundreds of iterations on such networks).
// *done = false: the network has too few low-degree nodes, nothing was computed.
int lowdeg_solve_pairs(nodal_ctx *h, int32_t npairs, const int32_t *ia, const int32_t *ib, double *res_dev,
                       bool *done, int32_t *info) {
    *done = false;
    *info = 0;
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    NODAL_HIP_TRY(h, h->ps_buf.reserve((size_t)n * 8 + 64));
    for (int32_t q = 0; q < npairs; ++q) {
        NODAL_HIP_TRY(h, hipMemsetAsync(h->ps_buf.p, 0, (size_t)n * 8, st));
        pair_rhs<<<1, 1, 0, st>>>(h->ps_buf.as<double>(), ia[q], ib[q]);
        NODAL_HIP_TRY(h, hipGetLastError());
        bool d = false;
        int32_t inf = 0, it = 0;
        double rs = 0.0;
        std::swap(h->rhs, h->ps_buf);  // the elimination reads the context's right-hand side
        const int s = lowdeg_solve(h, n > 4096 ? 512 : 8, &d, &inf, &it, &rs);
        std::swap(h->rhs, h->ps_buf);
        if (s != NODAL_OK) return s;
        if (!d) return q == 0 ? NODAL_OK : nodal_fail(h, NODAL_E_INVALID, "pair sweep: elimination not repeatable");
        if (inf > 0) {  // floating sub-network: every pair is singular, as in the reference
            *done = true;
            *info = 1;
            return NODAL_OK;
        }
        h->last_iterations = it;
        pair_read<<<1, 1, 0, st>>>(h->x.as<double>(), ia[q], ib[q], res_dev + q);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    *done = true;
    return NODAL_OK;
}

// Structural verdict for a general system (branch equations present) from the host copy of the
// component table, for the cases the iterative path cannot answer: the reference's spsolve
// reports an exactly singular matrix with NaNs + MatrixRankWarning (reference nodal/nodal.py:
// 323-336), it does not raise.
//   (a) a node that no resistor or voltage-defined branch (E, VCVS, CCVS) ties -- directly or
//       through other nodes -- to the ground node has an undetermined potential: the rows of
//       its island sum to a row with no conductance to anything fixed;
//   (b) a loop of independent voltage sources makes their branch equations linearly dependent
//       whatever the values.
// O(components) on the host; only called after the device solve has given up (or, (b) alone,
// when the presolve has declined the pattern).
namespace {
struct UnionFind {
    std::vector<int32_t> p;
    explicit UnionFind(int64_t n) : p((size_t)n) {
        for (int64_t i = 0; i < n; ++i) p[(size_t)i] = (int32_t)i;
    }
    int32_t find(int32_t x) {
        while (p[(size_t)x] != x) {
            p[(size_t)x] = p[(size_t)p[(size_t)x]];
            x = p[(size_t)x];
        }
        return x;
    }
    bool unite(int32_t a, int32_t b) {  // false: already joined
        a = find(a);
        b = find(b);
        if (a == b) return false;
        p[(size_t)(a < b ? b : a)] = a < b ? a : b;
        return true;
    }
};
const double *host_values(const nodal_ctx *h) {
    if (h->batch > 0 && !h->host.values_batch.empty())
        return h->host.values_batch.data() + (size_t)h->member * h->ncomp;
    return h->host.value.data();
}
}  // namespace

// A loop of voltage-defined branches (E, VCVS, CCVS -- independent or dependent alike): the current that
// circulates in it enters and leaves every node of the loop and appears in no other equation, a null vector of
// G whatever the values and gains.  (Round 3 looked at independent sources only; a ring of three VCVS branches
// next to a 1e5-node grid "converged" to a circulating current of 2e16 A.)
bool general_source_loop(const nodal_ctx *h) {
    const HostTable &t = h->host;
    if (t.type.empty()) return false;
    const int32_t K = h->K;
    UnionFind uf((int64_t)K + 1);
    for (const int64_t i : t.branch_rows) {
        const uint8_t ty = t.type[(size_t)i];
        if (ty != NODAL_T_E && ty != NODAL_T_VCVS && ty != NODAL_T_CCVS) continue;
        const int32_t a = t.a[(size_t)i] < 0 ? K : t.a[(size_t)i], b = t.b[(size_t)i] < 0 ? K : t.b[(size_t)i];
        if (!uf.unite(a, b)) return true;  // (a == b included: the reference asserts on it at stamping time)
    }
    return false;
}

bool general_floating_island(const nodal_ctx *h) {
    const HostTable &t = h->host;
    if (t.type.empty()) return false;
    const double *value = host_values(h);
    const int32_t K = h->K;
    UnionFind uf((int64_t)K + 1);
    for (int64_t i = 0; i < h->ncomp; ++i) {
        const uint8_t ty = t.type[(size_t)i];
        const bool ties = (ty == NODAL_T_R && value[i] != 0.0 && value[i] == value[i]) || ty == NODAL_T_E ||
                          ty == NODAL_T_VCVS || ty == NODAL_T_CCVS;
        // (Control terminals never tie an island: the KCL rows of a set of nodes that no resistor or
        // voltage-defined branch joins to the ground node sum to zero whatever the control terms say --
        // what leaves one of its nodes enters another -- so such a system is singular even when a
        // branch equation like e_a - e_b = gain (e_c - 0) seems to fix the island's level.  With the
        // reference's stamps (VCCS rows are stamped as VCVS, reference nodal/nodal.py:377-378) there is
        // no transconductance to ground; tests/test_gpu_parity.py::test_island_with_a_grounded_control_
        // terminal_is_still_singular and ::test_island_grounded_through_a_vccs_row.)
        if (!ties) continue;
        const int32_t a = t.a[(size_t)i] < 0 ? K : t.a[(size_t)i], b = t.b[(size_t)i] < 0 ? K : t.b[(size_t)i];
        uf.unite(a, b);
    }
    const int32_t g = uf.find(K);
    for (int32_t i = 0; i < K; ++i)
        if (uf.find(i) != g) return true;
    return false;
}

int sparse_solve(nodal_ctx *h, int32_t method, int32_t *info, int32_t *iters, double *resid) {
    if (!h->have_numeric) return nodal_fail(h, NODAL_E_INVALID, "assemble_numeric not called");
    const int64_t n = h->n;
    *info = 0;
    *iters = 0;
    *resid = 0.0;
    h->have_x = false;
    if (n == 0) {
        h->have_x = true;
        return NODAL_OK;
    }
    const int64_t densify_max = 4096;
    const int64_t dense_rescue_max = 8192;  // a general system the iteration gives up on is decided by the dense LU up to here, by the sparse direct solve beyond
    // below this a direct solve costs less than an elimination round.  (1024 until round 4: true of a round that is
    // being BUILT, 0.2-0.4 ms; repeated on kept lists it is two 5-us launches, and the dense solve of the 780 unknowns
    // seven rounds leave of a 1e5-section ladder was 0.75 of that solve's 0.91 ms.  NODAL_LOWDEG_MIN to compare.)
    static const int64_t lowdeg_min = getenv("NODAL_LOWDEG_MIN") ? atoll(getenv("NODAL_LOWDEG_MIN")) : 32;
    bool auto_passive = false;
    // NODAL_SPARSE_FORCE_DIRECT=1 (testing): every automatic sparse solve through the direct route
    if (method == NODAL_SPARSE_AUTO && getenv("NODAL_SPARSE_FORCE_DIRECT")) method = NODAL_SPARSE_DIRECT;
    // NODAL_SPARSE_CHILD_DIRECT=1 (testing): the same for matrix-only contexts alone (what lowdeg.hip's rounds leave)
    if (method == NODAL_SPARSE_AUTO && h->csr_only && getenv("NODAL_SPARSE_CHILD_DIRECT")) method = NODAL_SPARSE_DIRECT;
    if (method == NODAL_SPARSE_AUTO) {
        // passive network (B == 0, every R > 0, no transconductance): symmetric M-matrix.
        // Up to densify_max unknowns the direct dense solve is faster than the multigrid
        // (n = 2024: 2.0 vs 3.6 ms, n = 3599: 3.7 vs 4.4 ms; n = 6399: 8.2 vs 4.8 ms) and
        // indifferent to the topology (chain-like networks converge slowly).
        auto_passive = h->B == 0 && h->passive_network;
        if (auto_passive && n > densify_max) method = NODAL_SPARSE_PCG;
        else method = n <= densify_max ? NODAL_SPARSE_DENSIFY : NODAL_SPARSE_LU;
    }
    if (auto_passive && n > lowdeg_min) {
        // chains, ladders, trees: eliminate the nodes with <= 2 neighbours exactly first.  A round
        // costs 0.2-0.4 ms (about 1 ms at 5e6 entries).  Before the multigrid it pays as long as
        // wires are left: a 300 x 300 grid with 300 wires of 200 nodes takes 46 ms untouched,
        // 37 / 20 / 16 / 14 ms with rounds down to 1/32 / 1/128 / 1/256 / 1/512 of the nodes
        // (208 -> 52 iterations, the count of the bare grid).  Before a direct solve only
        // a round that removes 1/8 of the unknowns is cheaper than solving them.
        bool done = false;
        NODAL_TRY(lowdeg_solve(h, n > densify_max ? 512 : 8, &done, info, iters, resid));
        if (done) {
            if (*info > 0) NODAL_TRY(dense_fill_nan(h, h->x.as<double>(), n));
            h->have_x = true;
            return NODAL_OK;
        }
    }
    if (method == NODAL_SPARSE_PCG) {
        // multigrid-preconditioned flexible CG for large networks; Jacobi-CG below
        // that (the hierarchy would not pay for itself) or if the cycle breaks down
        int s = -1;
        if (n >= h->amg_min_n) s = amg_fcg_solve(h, info, iters, resid);
        if (s == -2) {  // structurally singular network
            NODAL_TRY(dense_fill_nan(h, h->x.as<double>(), n));
            h->have_x = true;
            return NODAL_OK;
        }
        if (s < 0) s = pcg_solve(h, info, iters, resid);
        if (s == NODAL_OK) {
            h->have_x = true;
            return NODAL_OK;
        }
        if (s > 0) return s;
        // breakdown: negative resistances or a singular / disconnected network
        method = n <= densify_max ? NODAL_SPARSE_DENSIFY : NODAL_SPARSE_LU;
    }
    if (method == NODAL_SPARSE_DENSIFY) {
        int32_t floating = 0;
        if (h->csr_only) NODAL_TRY(csr_small_floating_check(h, &floating));  // what an elimination round left
        if (floating) {
            *info = 1;
        } else {
            NODAL_TRY(dense_prepare(h));
            NODAL_TRY(dense_factor_solve(h, info));
        }
    } else if (method == NODAL_SPARSE_DIRECT) {
        NODAL_TRY(sparse_direct_solve(h, h->rhs.as<double>(), h->x.as<double>(), info, iters, resid));
    } else if (method == NODAL_SPARSE_LU) {
        static const bool trace = getenv("NODAL_TRACE") != nullptr;
        // (a matrix-only context -- what an elimination round of lowdeg.hip left -- has no component table
        // for the presolve and the verdicts below: the direct route takes it)
        const int s = h->csr_only ? NODAL_E_UNSUPPORTED : sparse_general_solve(h, info, iters, resid);
        if (s == NODAL_E_UNSUPPORTED && !h->csr_only && (general_floating_island(h) || general_source_loop(h))) {
            // the iteration gave up on a matrix that is singular by construction: the reference's
            // answer is NaNs + a warning (quirk 3), not an exception
            *info = 1;
        } else if (s == NODAL_E_UNSUPPORTED && !h->csr_only && n <= dense_rescue_max) {
            // The iteration did not converge and nothing in the STRUCTURE says why: singular for its
            // particular values (a gain of exactly 1 around a loop), or too ill-conditioned for the
            // iterative path.  The reference's spsolve decides by pivoting (reference nodal/nodal.py:325:
            // an exact zero pivot -> NaNs + MatrixRankWarning, else the LU's answer); up to
            // dense_rescue_max unknowns the pivoted dense LU here does the same with LAPACK's own rule.
            const std::string why = h->err;
            const bool forced = h->force_pivoting;
            h->force_pivoting = true;  // (the tournament LU's exact-zero-pivot test, not the pivot-free paths)
            int d = dense_prepare(h);
            if (d == NODAL_OK) d = dense_factor_solve(h, info);
            h->force_pivoting = forced;
            if (d != NODAL_OK) return d;
            *iters = 0;
            *resid = 0.0;
            if (trace) fprintf(stderr, "[general] %s -- decided by the pivoted dense LU: info %d\n", why.c_str(), *info);
        } else if (s == NODAL_E_UNSUPPORTED) {
            // Larger than that (round 4): the multifrontal sparse LU of sparse_direct.hip with fp64 refinement --
            // any non-singular G is solved, a singular one (the refinement stalls on perturbed pivots, or no
            // perfect matching exists) gives NaNs + MatrixRankWarning like the reference's SuperLU, at every size.
            if (trace && !h->csr_only) fprintf(stderr, "[general] %s -- handed to the sparse direct solve\n", h->err.c_str());
            NODAL_TRY(sparse_direct_solve(h, h->rhs.as<double>(), h->x.as<double>(), info, iters, resid));
        } else if (s != NODAL_OK) {
            return s;
        }
    } else {
        return nodal_fail(h, NODAL_E_INVALID, "unknown sparse method");
    }
    if (*info > 0) NODAL_TRY(dense_fill_nan(h, h->x.as<double>(), n));
    h->have_x = true;
    return NODAL_OK;
}

// y = G x with the context's CSR matrix (used by the general Krylov path)
int csr_spmv(nodal_ctx *h, const double *x, double *y) {
    const int64_t n = h->n;
    spmv_kernel<<<stream::grid_for_rows(n), TB, 0, h->stream>>>(
        h->indptr.as<int32_t>(), h->indices.as<int32_t>(), h->data.as<double>(), x, y, n);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// |G x - b|_inf / (|G|_inf |x|_inf + |b|_inf) for any device vectors (NaN if x holds one)
int csr_scaled_residual(nodal_ctx *h, const double *x, const double *b, double *scaled) {
    const int64_t n = h->n;
    if (n == 0) {
        *scaled = 0.0;
        return NODAL_OK;
    }
    NODAL_HIP_TRY(h, h->status.reserve(64));
    double *out = h->status.as<double>();
    NODAL_HIP_TRY(h, hipMemsetAsync(out, 0, 40, h->stream));
    const int lpr = lanes_per_row(h);
    DISPATCH_LPR(lpr, (residual_kernel<L><<<grid_rows(n, lpr), TB, 0, h->stream>>>(
                          h->indptr.as<int32_t>(), h->indices.as<int32_t>(), h->data.as<double>(), x, b, out, n)));
    NODAL_HIP_TRY(h, hipGetLastError());
    double o[5];
    NODAL_TRY(nodal_read_words(h, o, out, 40));
    if (o[4] != 0.0) {
        *scaled = __builtin_nan("");
        return NODAL_OK;
    }
    const double den = o[1] * o[2] + o[3];
    *scaled = den > 0 ? o[0] / den : 0.0;
    return NODAL_OK;
}

int sparse_residual(nodal_ctx *h, double *scaled) {
    if (!h->have_numeric || !h->have_x)
        return nodal_fail(h, NODAL_E_INVALID, "no assembled system / solution on the device");
    return csr_scaled_residual(h, h->x.as<double>(), h->rhs.as<double>(), scaled);
}
