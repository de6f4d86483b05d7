// Block (multi-vector) flexible CG on the smoothed-aggregation hierarchy: MK right-hand sides per iteration.
// (Third part of sagg.hip; included there, same translation unit.)
//
// Why: an equivalent-resistance sweep (reference nodal/equiv.py:31-61: one deepcopy + rebuild + solve per
// node pair) changes only the right-hand side.  One hierarchy already served all pairs, but every pair still
// paid for its own iterations -- and half of an iteration is the launch-latency floor of the coarse levels
// (33 dependent launches of 4-7 us), the other half bandwidth-bound passes over level-0 matrices that are the
// same for every pair.  Here MK = 16 pairs share every launch and every matrix read:
//   * vectors are interleaved by row, element (i, y) at v[i * MK + y]: thread t of a row kernel owns row
//     t / MK of column t % MK, so the 16 lanes of a row read each matrix entry ONCE (one address, one
//     transaction) and gather 128 contiguous bytes of the vector block per neighbour;
//   * every column has its own scalars -- alpha, beta, the K-cycle's coefficients, the convergence flag -- in
//     the same parity-slot protocol as the single-vector iteration (sagg_cycle.h): the columns are independent
//     Krylov processes that merely travel together; a converged column's updates are predicated off;
//   * dot products leave per-workgroup partials laid out [quantity][block][column]; a one-workgroup-per-
//     quantity kernel folds them in a fixed order (deterministic), the consumers read 16 totals;
//   * the tail levels run in one launch of MK workgroups (k_tail, one column each).
// The single-vector kernels of sagg_cycle.h are untouched: the headline path does not go through here.
#pragma once

namespace {

#ifndef NODAL_MK_SHIFT
#define NODAL_MK_SHIFT 4
#endif
constexpr int MK = 1 << NODAL_MK_SHIFT;  // columns of a block (16; 32 / 64 measured: tools/experiments/README.md)
// (cyc_t, the type of the vectors inside the cycle: sagg.hip)
constexpr int MK_SHIFT = NODAL_MK_SHIFT;
constexpr int MSC = 32;       // scalars per column: the single-vector block's F_COUNT words + the functional's (below)
enum { M_SUM = 16, M_INC = 17 /* .. 20: the last increments alpha_j r_j.z_j, a ring */, M_RING = 4, M_FDONE = 21 };
constexpr int MPARTS = 1024;  // workgroups that leave dot partials (grid cap of the level-0 kernels that do)

// per-column sum over the workgroup of NQ quantities; result written to part[(q * nblocks_cap + block) * MK + y]
template <int NQ>
__device__ __forceinline__ void column_partials(double (&a)[NQ], double *__restrict__ part, int cap) {
    __shared__ double ws[NQ][TB / 64][MK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double v = a[q];
#pragma unroll
        for (int off = MK; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);  // (lanes MK apart hold the same column)
        if (lane < MK) ws[q][wave][lane] = v;
    }
    __syncthreads();
    if (threadIdx.x < MK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < TB / 64; ++w) s += ws[q][w][threadIdx.x];
            part[((int64_t)q * cap + blockIdx.x) * MK + threadIdx.x] = s;
        }
    }
}

// totals[q * MK + y] = sum over the blocks of part[(q * cap + block) * MK + y]; one workgroup of MR threads per
// quantity, four loads in flight per thread (256 threads walking 64 partials each one load at a time took 16-28 us,
// the latency of 64 dependent-looking round trips; the order of the additions is fixed either way)
constexpr int MR = 1024;
__global__ __launch_bounds__(MR) void m_reduce_kernel(const double *__restrict__ part, int cap, int count,
                                                      double *__restrict__ totals) {
    __shared__ double ws[MR / 64][MK];
    const int q = blockIdx.x, y = threadIdx.x & (MK - 1), r = threadIdx.x >> MK_SHIFT;
    const double *p = part + (int64_t)q * cap * MK;
    constexpr int RS = MR / MK;  // partials a pass of the workgroup covers
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int k = r; k < count; k += 4 * RS) {
        const double v0 = p[(int64_t)k * MK + y];
        const double v1 = k + RS < count ? p[(int64_t)(k + RS) * MK + y] : 0.0;
        const double v2 = k + 2 * RS < count ? p[(int64_t)(k + 2 * RS) * MK + y] : 0.0;
        const double v3 = k + 3 * RS < count ? p[(int64_t)(k + 3 * RS) * MK + y] : 0.0;
        s0 += v0, s1 += v1, s2 += v2, s3 += v3;
    }
    double s = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int off = MK; off < 64; off <<= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) < MK) ws[threadIdx.x >> 6][y] = s;
    __syncthreads();
    if (threadIdx.x < MK) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < MR / 64; ++w) t += ws[w][threadIdx.x];
        totals[q * MK + threadIdx.x] = t;
    }
}

inline unsigned mgrid(int64_t rows, unsigned cap = 65536) {  // one thread per (row, column)
    int64_t g = (rows * MK + TB - 1) / TB;
    if (g < 1) g = 1;
    if (g >= 64) g = (g + 7) & ~(int64_t)7;
    return (unsigned)(g > cap ? cap : g);
}

template <int W, typename TBV>
__global__ __launch_bounds__(TB) void m_smooth_residual(Ell A, const TBV *__restrict__ b,
                                                        const cyc_t *__restrict__ x0, cyc_t *__restrict__ r, bool f32) {
    const int64_t total = A.n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t i = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        const double s = f32 ? ell_row_w<W>(A, A.valf, i, 0, [&](int32_t j) { return (double)x0[(int64_t)j * MK + y]; })
                             : ell_row_w<W>(A, A.val, i, 0, [&](int32_t j) { return (double)x0[(int64_t)j * MK + y]; });
        r[t] = (cyc_t)((double)b[t] - s);
    }
}

// rc = R r (one thread per coarse row and column; the 16 lanes of a row read every block of R once)
template <typename TO>
__global__ __launch_bounds__(TB) void m_restrict(int64_t nc, int64_t rld, const int32_t *__restrict__ rcol,
                                                 const float *__restrict__ rvalf, const double *__restrict__ rval,
                                                 const int32_t *__restrict__ rlen, const cyc_t *__restrict__ r,
                                                 TO *__restrict__ rc, const double *__restrict__ cdinv,
                                                 cyc_t *__restrict__ x0c) {
    const int64_t total = nc * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t I = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        const int32_t len = rlen[I];
        double s = 0.0;
        for (int32_t q = 0; q * RL < len; ++q) {
            const int64_t at = ((int64_t)q * rld + I) * RL;
            int32_t c[RL];
            double v[RL];
#pragma unroll
            for (int u = 0; u < RL; ++u) {
                c[u] = rcol[at + u];
                v[u] = rvalf ? (double)rvalf[at + u] : rval[at + u];
            }
#pragma unroll
            for (int u = 0; u < RL; ++u) s = fma(v[u], (double)r[(int64_t)c[u] * MK + y], s);
        }
        rc[t] = (TO)s;
        if (x0c) x0c[t] = (cyc_t)(OMEGA * cdinv[I] * s);
    }
}

// xp = x + P (s1 c1 + s2 c2) per column (coef: [MK][2]; nullptr: plain V hand-over)
template <typename TC>
__global__ __launch_bounds__(TB) void m_prolong(int64_t n, int64_t ld, const int32_t *__restrict__ pcol,
                                                const float *__restrict__ pvalf, const double *__restrict__ pval,
                                                const cyc_t *__restrict__ x, const TC *__restrict__ c1,
                                                const TC *__restrict__ c2, const double *__restrict__ coef,
                                                cyc_t *__restrict__ xp) {
    const int64_t total = n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t i = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        const bool two = coef != nullptr;
        const double s1 = two ? coef[y * 2] : 1.0, s2 = two ? coef[y * 2 + 1] : 0.0;
        int32_t J[PW];
        double w[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            J[q] = pcol[(int64_t)q * ld + i];
            w[q] = pvalf ? (double)pvalf[(int64_t)q * ld + i] : pval[(int64_t)q * ld + i];
        }
        double s = (double)x[t];
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int64_t j = (int64_t)(J[q] < 0 ? 0 : J[q]) * MK + y;
            const double e = two ? s1 * (double)c1[j] + s2 * (double)c2[j] : (double)c1[j];
            s = fma(J[q] < 0 ? 0.0 : w[q], e, s);
        }
        xp[t] = (cyc_t)s;
    }
}

// out = xp + w D^-1 (b - A xp); DOTS: per-column partials of out.b and out.u
template <int W, bool DOTS, typename TBV>
__global__ __launch_bounds__(TB) void m_post(Ell A, const double *__restrict__ dinv, const TBV *__restrict__ b,
                                             const cyc_t *__restrict__ xp, cyc_t *__restrict__ out,
                                             const double *__restrict__ u, double *__restrict__ part, int cap,
                                             bool f32) {
    double a[2] = {0.0, 0.0};
    const int64_t total = A.n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t i = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        const double s = f32 ? ell_row_w<W>(A, A.valf, i, 0, [&](int32_t j) { return (double)xp[(int64_t)j * MK + y]; })
                             : ell_row_w<W>(A, A.val, i, 0, [&](int32_t j) { return (double)xp[(int64_t)j * MK + y]; });
        const double bi = (double)b[t];
        // (the dots take the value as stored: z.r and z.Ap are those of the z the direction is built from)
        const cyc_t os = (cyc_t)fma(OMEGA * dinv[i], bi - s, (double)xp[t]);
        const double o = (double)os;
        out[t] = os;
        if (DOTS) {
            a[0] = fma(o, bi, a[0]);
            a[1] = fma(o, u[t], a[1]);
        }
    }
    if (DOTS) column_partials<2>(a, part, cap);
}

// v = A c and the per-column partials c.v, c.u1, c.u2 (u2 may be null)
template <int W>
__global__ __launch_bounds__(TB) void m_spmv_dots(Ell A, const cyc_t *__restrict__ c, cyc_t *__restrict__ v,
                                                  const cyc_t *__restrict__ u1, const cyc_t *__restrict__ u2,
                                                  double *__restrict__ part, int cap, bool f32) {
    double a[3] = {0.0, 0.0, 0.0};
    const int64_t total = A.n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t i = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        const double sd = f32 ? ell_row_w<W>(A, A.valf, i, 0, [&](int32_t j) { return (double)c[(int64_t)j * MK + y]; })
                              : ell_row_w<W>(A, A.val, i, 0, [&](int32_t j) { return (double)c[(int64_t)j * MK + y]; });
        const cyc_t vs = (cyc_t)sd;
        v[t] = vs;
        const double s = (double)vs, ci = (double)c[t];
        a[0] = fma(ci, s, a[0]);
        a[1] = fma(ci, (double)u1[t], a[1]);
        if (u2) a[2] = fma(ci, (double)u2[t], a[2]);
    }
    column_partials<3>(a, part, cap);
}

// r2 = rc - (alpha1 / rho1) v1 per column; tot: [0] rho1 = c1.v1, [1] alpha1 = c1.rc
__global__ __launch_bounds__(TB) void m_second_residual(int64_t n, const cyc_t *__restrict__ rc,
                                                        const cyc_t *__restrict__ v1, const double *__restrict__ tot,
                                                        cyc_t *__restrict__ r2, const double *__restrict__ dinv,
                                                        cyc_t *__restrict__ x0) {
    const int64_t total = n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int y = (int)(t & (MK - 1));
        const double rho1 = tot[y], alpha1 = tot[MK + y];
        const double tt = rho1 > 0.0 ? alpha1 / rho1 : 0.0;
        const double v = fma(-tt, (double)v1[t], (double)rc[t]);
        r2[t] = (cyc_t)v;
        x0[t] = (cyc_t)(OMEGA * dinv[t >> MK_SHIFT] * v);
    }
}

// K-cycle coefficients per column from the totals: first [rho1, alpha1, -] then second [beta, gamma, alpha2]
// slot >= 0: the column's sample (s1, s2, t) also goes to slot `slot` of a ring of three, and the mean of the ring's
// first `count` slots to the FROZEN coefficients (fs: [MK][2], ft: [MK]) the cycles between two calibrations use
// (sagg.hip: kcycle_schedule; m_spmv_resid below)
__global__ void m_kcoef(const double *__restrict__ tot1, const double *__restrict__ tot2, double *__restrict__ coef,
                        int slot, int count) {
    const int y = threadIdx.x;
    if (y >= MK) return;
    const double rho1 = tot1[y], alpha1 = tot1[MK + y];
    const double beta = tot2[y], gamma = tot2[MK + y], alpha2 = tot2[2 * MK + y];
    double s1 = 0.0, s2 = 0.0;
    if (rho1 > 0.0) {
        const double rho2 = beta - gamma * gamma / rho1;
        if (rho2 > 0.0) {
            s1 = alpha1 / rho1 - gamma * alpha2 / (rho1 * rho2);
            s2 = alpha2 / rho2;
        } else {
            s1 = alpha1 / rho1;
        }
    }
    coef[y * 2] = s1;
    coef[y * 2 + 1] = s2;
    if (slot >= 0) {
        double *fs = coef + 2 * MK, *ft = fs + 2 * MK, *ring = ft + MK;  // (behind the coefficients: MLevel::tot)
        double *mine = ring + ((int64_t)slot * MK + y) * 3;
        mine[0] = s1;
        mine[1] = s2;
        mine[2] = rho1 > 0.0 ? alpha1 / rho1 : 0.0;
        double m0 = 0.0, m1 = 0.0, m2 = 0.0;
        for (int q = 0; q < count; ++q) {
            const double *e = ring + ((int64_t)q * MK + y) * 3;
            m0 += e[0];
            m1 += e[1];
            m2 += e[2];
        }
        fs[y * 2] = m0 / count;
        fs[y * 2 + 1] = m1 / count;
        ft[y] = m2 / count;
    }
}

// r2 = rc - t A c1 with the column's FROZEN t and the start iterate of the second visit: m_spmv_dots, m_reduce_kernel
// and m_second_residual of an adaptive cycle in one launch, no dot products
template <int W>
__global__ __launch_bounds__(TB) void m_spmv_resid(Ell A, const cyc_t *__restrict__ c, const cyc_t *__restrict__ rc,
                                                   const double *__restrict__ ft, cyc_t *__restrict__ r2,
                                                   const double *__restrict__ dinv, cyc_t *__restrict__ x0, bool f32) {
    const int64_t total = A.n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t i = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        const double sd = f32 ? ell_row_w<W>(A, A.valf, i, 0, [&](int32_t j) { return (double)c[(int64_t)j * MK + y]; })
                              : ell_row_w<W>(A, A.val, i, 0, [&](int32_t j) { return (double)c[(int64_t)j * MK + y]; });
        const double v = fma(-ft[y], (double)(cyc_t)sd, (double)rc[t]);
        r2[t] = (cyc_t)v;
        x0[t] = (cyc_t)(OMEGA * dinv[i] * v);
    }
}

__global__ __launch_bounds__(TB) void m_coarsest(int64_t n, const double *__restrict__ inv,
                                                 const double *__restrict__ dinv, const double *__restrict__ b,
                                                 double *__restrict__ out) {
    const int64_t total = n * MK;
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t i = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        double s = 0.0;
        if (inv)
            for (int64_t j = 0; j < n; ++j) s = fma(inv[i * n + j], b[j * MK + y], s);
        else
            s = dinv[i] * b[t];
        out[t] = s;
    }
}

// ---- the outer iteration, per column ---------------------------------------------------------------
// sc: [MK][MSC] (the single-vector iteration's scalar block per column, same parity protocol, plus the functional's)
//
// The stopping rule of a PAIR sweep.  The quantity wanted is the functional R = b.x with b the probe itself
// (b = e_ia - e_ib, x0 = 0) and A symmetric.  With e_k = x - x_k and r_k = A e_k:
//     |e_k|_A^2 = (x - x_k).A e_k = b.e_k - x_k.r_k     =>     R = b.x_k + x_k.r_k + |e_k|_A^2          (*)
// so b.x_k + x_k.r_k is R up to the SQUARE of the energy-norm error, whatever the preconditioner (in exact CG the
// term x_k.r_k vanishes; the flexible iteration keeps one old direction only, so it does not: leaving it out costs
// first order, 1e-9 measured at the point where the rule below stops).  Each step is an exact line minimisation, so
// d_k = |e_k|_A^2 - |e_{k+1}|_A^2 = alpha_k r_k.z_k (Hestenes-Stiefel; Strakos & Tichy for the preconditioned form):
// the running sum of the decrements is R, and what is left after step k is the tail d_{k+1} + d_{k+2} + ...,
// estimated as a geometric one with the worse of the last two ratios q = max(d_k / d_{k-1}, d_{k-1} / d_{k-2}).  A
// column is done when d_k q / (1 - q) <= 1e-11 of the sum (q < 0.9; a slower iteration is left to the residual
// rule): the 1e-9 bar of SURVEY 8f N1 with two orders to spare, in under half the iterations the residual rule
// |r| <= 1e-13 |b| needs (that rule stays: NODAL_PAIRS_FUNCTIONAL=0).

__global__ void m_set_scalars(double *__restrict__ sc, double tol2) {
    const int y = threadIdx.x;
    if (y >= MK) return;
    for (int k = 0; k < MSC; ++k) sc[y * MSC + k] = 0.0;
    sc[y * MSC + F_TOL2] = tol2;
}

// b_y = e(ia[y]) - e(ib[y]) (a 1 A probe enters ia and leaves ib, reference nodal/models.py:27-32; -1 = ground)
__global__ __launch_bounds__(TB) void m_init(int64_t n, const int32_t *__restrict__ ia, const int32_t *__restrict__ ib,
                                             double *__restrict__ x, double *__restrict__ r, double *__restrict__ Ap,
                                             const double *__restrict__ dinv, cyc_t *__restrict__ x0,
                                             double *__restrict__ part, int cap) {
    double a[2] = {0.0, 0.0};  // r.r, x.r (x = 0)
    const int64_t total = n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        const int64_t i = t >> MK_SHIFT;
        const int y = (int)(t & (MK - 1));
        const double ri = ((int64_t)ia[y] == i ? 1.0 : 0.0) - ((int64_t)ib[y] == i ? 1.0 : 0.0);
        x[t] = 0.0;
        r[t] = ri;
        x0[t] = (cyc_t)(OMEGA * dinv[i] * ri);
        Ap[t] = 0.0;
        a[0] = fma(ri, ri, a[0]);
    }
    column_partials<2>(a, part, cap);
}

// tot: [0] z.r, [1] z.Ap, [2] r.r, [3] x.r per column ([4] p.Ap)
__global__ __launch_bounds__(TB) void m_direction(const cyc_t *__restrict__ z, double *__restrict__ p,
                                                  const double *__restrict__ tot, double *__restrict__ scs, int parity,
                                                  int64_t n, bool functional) {
    const int cur = parity & 1, prev = cur ^ 1;
    const int y = threadIdx.x & (MK - 1);
    double *sc = scs + y * MSC;
    const int iter = (int)sc[F_ITNO + prev];
    const bool writer = blockIdx.x == 0 && threadIdx.x < MK;
    const bool was_done = iter > 0 && sc[F_CONV + prev] != 0.0;
    const double rz_new = tot[y], zap = tot[MK + y], rr = tot[2 * MK + y];
    const double rz_old = iter > 0 ? sc[F_RZ + prev] : 1.0;
    const double beta = (iter > 0 && rz_old != 0.0) ? -sc[F_ALPHA + prev] * zap / rz_old : 0.0;
    const double bb = iter == 0 ? rr : sc[F_BB];
    const bool bad = !(rz_new >= 0.0) || !(rr == rr);
    // (the functional's words are written by m_update, the kernel before this one: no writer in this launch)
    const bool fdone = functional && sc[M_FDONE] != 0.0;  // (decided by m_update, where the host can see it a cycle earlier)
    const bool converged = was_done || bb == 0.0 || rr <= sc[F_TOL2] * bb || bad || fdone;
    __syncthreads();  // (block 0: every lane has read the previous parity's words before the writers go on)
    if (writer) {
        sc[F_ITNO + cur] = (double)(iter + 1);
        if (was_done) {
            sc[F_CONV + cur] = 1.0;
        } else {
            sc[F_RZ + cur] = rz_new;
            sc[F_RR] = rr;
            if (iter == 0) sc[F_BB] = rr;
            if (bad) sc[F_FLAG] = 1.0;
            sc[F_CONV + cur] = converged ? 1.0 : 0.0;
            if (converged) sc[F_ITERS] = (double)iter;
        }
    }
    if (converged) return;  // (per column: the lanes of the other columns go on)
    const int64_t total = n * MK;
    // (a thread's column is fixed: the stride of the loop is a multiple of MK)
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB)
        p[t] = iter > 0 ? fma(beta, p[t], (double)z[t]) : (double)z[t];
}

template <int W>
__global__ __launch_bounds__(TB) void m_spmv(Ell A, const double *__restrict__ p, double *__restrict__ Ap,
                                             double *__restrict__ part, int cap, const double *__restrict__ scs,
                                             int iter) {
    double a[1] = {0.0};
    const int yy = threadIdx.x & (MK - 1);
    const bool done = scs[yy * MSC + F_CONV + (iter & 1)] != 0.0;
    const int64_t total = A.n * MK;
    if (!done)
        for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
            const int64_t i = t >> MK_SHIFT;
            const double s = ell_row_w<W>(A, A.val, i, 0, [&](int32_t j) { return p[(int64_t)j * MK + yy]; });
            Ap[t] = s;
            a[0] = fma(p[t], s, a[0]);
        }
    column_partials<1>(a, part, cap);
}

// tot_pap: p.Ap per column
__global__ __launch_bounds__(TB) void m_update(double *__restrict__ x, double *__restrict__ r,
                                               const double *__restrict__ p, const double *__restrict__ Ap,
                                               const double *__restrict__ tot_pap, const double *__restrict__ dinv,
                                               cyc_t *__restrict__ x0, double *__restrict__ part, int cap,
                                               double *__restrict__ scs, int iter, int64_t n, bool functional) {
    const int cur = iter & 1;
    const int y = threadIdx.x & (MK - 1);
    double *sc = scs + y * MSC;
    const bool done = sc[F_CONV + cur] != 0.0;
    const double pap = tot_pap[y];
    const double rz = sc[F_RZ + cur];
    const bool bad = !(pap > 0.0) && rz != 0.0;
    const double alpha = (pap > 0.0) ? rz / pap : 0.0;
    const int itno = (int)sc[F_ITNO + cur];  // iterations started so far (written by m_direction of this iteration)
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < MK && !done) {
        sc[F_ALPHA + cur] = alpha;
        if (bad) sc[F_FLAG] = 1.0;
        const double inc = alpha * rz;  // d_k = |x - x_k|_A^2 - |x - x_{k+1}|_A^2
        const double sum = sc[M_SUM] + inc;
        const double d2 = sc[M_INC + ((itno - 1) & (M_RING - 1))], d3 = sc[M_INC + ((itno - 2) & (M_RING - 1))];
        sc[M_SUM] = sum;
        sc[M_INC + (itno & (M_RING - 1))] = inc;
        // what is left, |x - x_{k+1}|_A^2 = d_{k+1} + d_{k+2} + ..., as a geometric tail with the worse of the last
        // two ratios (CG converges superlinearly: the ratios fall, the estimate errs on the safe side); an iteration
        // that contracts by less than 0.9 per step is left to the residual rule
        bool fdone = false;
        if (functional && itno >= 4 && d2 > 0.0 && d3 > 0.0) {
            const double q = fmax(inc / d2, d2 / d3);
            fdone = q < 0.9 && inc * q <= 1e-11 * sum * (1.0 - q);
        }
        if (fdone) {
            sc[M_FDONE] = 1.0;  // x and r of this column are final after this launch: m_direction stops it next time
            sc[F_ITERS] = (double)itno;
        }
    }
    double a[2] = {0.0, 0.0};  // r.r, x.r (the latter closes the functional, see (*) above)
    const int64_t total = n * MK;
    for (int64_t t = (int64_t)xcd_block() * TB + threadIdx.x; t < total; t += (int64_t)gridDim.x * TB) {
        double ri = r[t], xi = x[t];
        if (!done) {
            xi = fma(alpha, p[t], xi);
            x[t] = xi;
            ri = fma(-alpha, Ap[t], ri);
            r[t] = ri;
            x0[t] = (cyc_t)(OMEGA * dinv[t >> MK_SHIFT] * ri);
        }
        a[0] = fma(ri, ri, a[0]);
        a[1] = fma(xi, ri, a[1]);
    }
    column_partials<2>(a, part, cap);
}

// res[q0 + y] = x_y[ia] - x_y[ib] (+ x_y.r_y: the functional rule stops early and closes with that term)
__global__ void m_read_pairs(int count, const int32_t *__restrict__ ia, const int32_t *__restrict__ ib,
                             const double *__restrict__ x, const double *__restrict__ xr, double *__restrict__ res) {
    const int y = threadIdx.x;
    if (y >= count) return;
    const double ea = ia[y] >= 0 ? x[(int64_t)ia[y] * MK + y] : 0.0;
    const double eb = ib[y] >= 0 ? x[(int64_t)ib[y] * MK + y] : 0.0;
    res[y] = (ea - eb) + (xr ? xr[y] : 0.0);
}

// ---- buffers and the cycle -----------------------------------------------------------------------------
struct MLevel {
    double *vec = nullptr;   // V_COUNT slots of ld * MK doubles (cyc_t vectors use the front of their slot)
    double *part = nullptr;  // K-cycle partials: 2 x [3][DOT cap][MK]
    double *tot = nullptr;   // 2 x [3][MK] totals + [MK][2] coefficients
    int64_t ld = 0;
    template <typename T> T *v(int which) const { return reinterpret_cast<T *>(vec + (int64_t)which * ld * MK); }
};
struct MBufs {
    MLevel lv[MAX_LEVELS];
    double *x, *r, *p, *Ap;            // n * MK each
    cyc_t *z, *x0;
    double *part;                      // [5][MPARTS][MK]: z.r, z.Ap | r.r, x.r | p.Ap
    double *tot;                       // [5][MK]
    double *sc;                        // [MK][MSC]
    int32_t *ia, *ib;
    int g0;
};
constexpr int MDOT = 1024;  // partial blocks of the K-cycle level's dot kernels

// the levels that end the recursion: the tail (one workgroup per column, sagg_cycle.h) or the coarsest level; f64
int m_last_level(nodal_ctx *h, SHierarchy *H, int l, const double *b, double *out) {
    hipStream_t st = h->stream;
    SLevel *L = H->pool[l];
    const int64_t n = L->n;
    if (l == H->tail) {
        // vectors strided by MK
        if (H->td.slots <= 8) k_tail<8><<<MK, 1024, (size_t)H->td.lds_bytes, st>>>(H->td, H->tail_image.as<char>(), b, out, MK);
        else if (H->td.slots <= 16) k_tail<16><<<MK, 1024, (size_t)H->td.lds_bytes, st>>>(H->td, H->tail_image.as<char>(), b, out, MK);
        else k_tail<32><<<MK, 1024, (size_t)H->td.lds_bytes, st>>>(H->td, H->tail_image.as<char>(), b, out, MK);
    } else {
        m_coarsest<<<mgrid(n), TB, 0, st>>>(n, H->dense_coarsest ? H->coarse_inv.as<double>() : nullptr,
                                            L->dinv.as<double>(), b, out);
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// one cycle of level l (not a last level): out ~ A_l^-1 b from the start iterate x0 = w D^-1 b
template <typename TBV>
int m_cycle(nodal_ctx *h, SHierarchy *H, const MBufs &M, int l, const TBV *b, const cyc_t *x0, cyc_t *out, bool outer) {
    hipStream_t st = h->stream;
    SLevel *L = H->pool[l];
    const int64_t n = L->n;
    SLevel *C = H->pool[l + 1];
    const int64_t nc = C->n;
    const Ell A = L->A();
    const bool f32 = L->avalf.p != nullptr;  // (levels outside the tail carry f32 copies of A, P, R)
    const double *dinv = L->dinv.as<double>();
    const MLevel &ML = M.lv[l], &MC = M.lv[l + 1];
    const cyc_t *x = x0;
    cyc_t *r = ML.v<cyc_t>(V_R), *xp = ML.v<cyc_t>(V_XP);
    const unsigned g = mgrid(n);                                  // kernels without dot partials: as wide as the rows
    const unsigned gp = outer ? (unsigned)M.g0 : mgrid(n);        // the one that leaves them
    const int nu = H->nu[l < 2 ? l : 2];
    // (levels whose rows are ragged -- wfix == 0 -- take the W = 0 instantiation like the single-vector kernels)
    if (nu >= 2) {
        cyc_t *x1 = ML.v<cyc_t>(V_X1);
        SAGG_DISPATCH_W(L->wfix, (m_post<W, false, TBV><<<g, TB, 0, st>>>(A, dinv, b, x0, x1, nullptr, nullptr, 0, f32)));
        x = x1;
    }
    SAGG_DISPATCH_W(L->wfix, (m_smooth_residual<W, TBV><<<g, TB, 0, st>>>(A, b, x, r, f32)));
    const float *rvf = f32 ? L->rvalf.as<float>() : nullptr, *pvf = f32 ? L->pvalf.as<float>() : nullptr;
    const bool last = l + 1 == H->tail || l + 1 == H->nlev - 1;
    if (last) {
        double *rc = MC.v<double>(V_RC), *c1 = MC.v<double>(V_C1);
        m_restrict<double><<<mgrid(nc), TB, 0, st>>>(nc, L->rld, L->rcol.as<int32_t>(), rvf, L->rval.as<double>(),
                                                     L->rlen.as<int32_t>(), r, rc, C->dinv.as<double>(), nullptr);
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_TRY(m_last_level(h, H, l + 1, rc, c1));
        m_prolong<double><<<mgrid(n), TB, 0, st>>>(n, L->ld, L->pcol.as<int32_t>(), pvf, L->pval.as<double>(), x, c1,
                                                   nullptr, nullptr, xp);
    } else {
        cyc_t *rc = MC.v<cyc_t>(V_RC), *c1 = MC.v<cyc_t>(V_C1), *c2 = MC.v<cyc_t>(V_C2), *x0c = MC.v<cyc_t>(V_X);
        m_restrict<cyc_t><<<mgrid(nc), TB, 0, st>>>(nc, L->rld, L->rcol.as<int32_t>(), rvf, L->rval.as<double>(),
                                                    L->rlen.as<int32_t>(), r, rc, C->dinv.as<double>(), x0c);
        NODAL_HIP_TRY(h, hipGetLastError());
        const bool kcycle = l < H->klevels && H->kcycle;
        const double *coef = nullptr;
        if (kcycle) {
            cyc_t *v1 = MC.v<cyc_t>(V_V1), *v2 = MC.v<cyc_t>(V_V2), *r2 = MC.v<cyc_t>(V_R2);
            const Ell Ac = C->A();
            const bool cf32 = C->avalf.p != nullptr;
            const unsigned gd = mgrid(nc, MDOT);
            double *part1 = MC.part, *part2 = MC.part + (int64_t)3 * MDOT * MK;
            double *tot1 = MC.tot, *tot2 = MC.tot + 3 * MK, *cf = MC.tot + 6 * MK;
            double *fs = cf + 2 * MK, *ft = fs + 2 * MK;  // the frozen coefficients (m_kcoef)
            if (H->kfrozen && l == 0) {
                // between two calibrations (sagg.hip, kcycle_schedule): six launches fewer
                NODAL_TRY(m_cycle<cyc_t>(h, H, M, l + 1, rc, x0c, c1, false));
                SAGG_DISPATCH_W(C->wfix, (m_spmv_resid<W><<<mgrid(nc), TB, 0, st>>>(Ac, c1, rc, ft, r2, C->dinv.as<double>(), x0c,
                                                                                   cf32)));
                NODAL_HIP_TRY(h, hipGetLastError());
                NODAL_TRY(m_cycle<cyc_t>(h, H, M, l + 1, r2, x0c, c2, false));
                coef = fs;
            } else {
                const bool sample = l == 0 && H->kslot >= 0;
                NODAL_TRY(m_cycle<cyc_t>(h, H, M, l + 1, rc, x0c, c1, false));
                // [0] c1.v1 (rho1), [1] c1.rc (alpha1)
                SAGG_DISPATCH_W(C->wfix, (m_spmv_dots<W><<<gd, TB, 0, st>>>(Ac, c1, v1, rc, nullptr, part1, MDOT, cf32)));
                m_reduce_kernel<<<2, MR, 0, st>>>(part1, MDOT, (int)gd, tot1);
                m_second_residual<<<mgrid(nc), TB, 0, st>>>(nc, rc, v1, tot1, r2, C->dinv.as<double>(), x0c);
                NODAL_HIP_TRY(h, hipGetLastError());
                NODAL_TRY(m_cycle<cyc_t>(h, H, M, l + 1, r2, x0c, c2, false));
                // [0] c2.v2 (beta), [1] c2.v1 (gamma), [2] c2.r2 (alpha2)
                SAGG_DISPATCH_W(C->wfix, (m_spmv_dots<W><<<gd, TB, 0, st>>>(Ac, c2, v2, v1, r2, part2, MDOT, cf32)));
                m_reduce_kernel<<<3, MR, 0, st>>>(part2, MDOT, (int)gd, tot2);
                m_kcoef<<<1, 64, 0, st>>>(tot1, tot2, cf, sample ? H->kslot : -1, sample ? H->kcount : 0);
                NODAL_HIP_TRY(h, hipGetLastError());
                coef = cf;
            }
        } else {
            NODAL_TRY(m_cycle<cyc_t>(h, H, M, l + 1, rc, x0c, c1, false));
        }
        m_prolong<cyc_t><<<mgrid(n), TB, 0, st>>>(n, L->ld, L->pcol.as<int32_t>(), pvf, L->pval.as<double>(), x, c1, c2,
                                                  coef, xp);
    }
    const cyc_t *cur = xp;
    if (nu >= 2) {
        cyc_t *mid = ML.v<cyc_t>(V_T);
        SAGG_DISPATCH_W(L->wfix, (m_post<W, false, TBV><<<g, TB, 0, st>>>(A, dinv, b, cur, mid, nullptr, nullptr, 0, f32)));
        cur = mid;
    }
    if (outer) {
        SAGG_DISPATCH_W(L->wfix, (m_post<W, true, TBV><<<gp, TB, 0, st>>>(A, dinv, b, cur, out, M.Ap, M.part, MPARTS, f32)));
    } else {
        SAGG_DISPATCH_W(L->wfix, (m_post<W, false, TBV><<<g, TB, 0, st>>>(A, dinv, b, cur, out, nullptr, nullptr, 0, f32)));
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

}  // namespace

int sagg_pairs_block_width() { return MK; }

// Pairs [0, count), count <= MK, on the hierarchy of the last setup: res_dev[q] = e(ia[q]) - e(ib[q]) for a
// 1 A probe.  Return NODAL_OK (all columns converged), -1 breakdown / no convergence (the caller falls back
// to the single-vector iteration and what stands behind it), > 0 a status.
int sagg_fcg_solve_pairs_block(nodal_ctx *h, int32_t count, const int32_t *ia_host, const int32_t *ib_host,
                               double *res_dev, int32_t *iters) {
    const bool trace = getenv("NODAL_TRACE") != nullptr;  // (per call: tests switch it on for one sweep)
    SHierarchy *H = hierarchy_of(h);
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    if (!H->ready || H->pool[0]->n != n || count < 1 || count > MK) return -1;
    if (H->nlev < 2 || H->tail == 0) return -1;  // (no level outside the tail: the single-vector iteration's case)
    // ---- buffers: one allocation in the hierarchy, grown on demand ----
    size_t bytes = 0;
    auto take = [&](size_t b) { const size_t at = bytes; bytes += (b + 255) & ~(size_t)255; return at; };
    size_t o_vec[MAX_LEVELS], o_part[MAX_LEVELS], o_tot[MAX_LEVELS];
    const int last_outside = H->tail >= 0 ? H->tail : H->nlev - 1;  // levels [0, last_outside] hold vectors
    for (int l = 0; l <= last_outside && l < H->nlev; ++l) {
        o_vec[l] = take((size_t)V_COUNT * H->pool[l]->ld * MK * 8);
        o_part[l] = take((size_t)6 * MDOT * MK * 8);
        o_tot[l] = take((size_t)(8 + 3 + 9) * MK * 8);  // totals, coefficients | frozen (s1, s2), frozen t | ring of samples
    }
    const size_t o_outer = take((size_t)6 * n * MK * 8 + 6 * 256);
    const size_t o_opart = take((size_t)5 * MPARTS * MK * 8);
    const size_t o_otot = take((size_t)5 * MK * 8);
    const size_t o_sc = take((size_t)MK * MSC * 8);
    const size_t o_pairs = take((size_t)2 * MK * 4);
    NODAL_HIP_TRY(h, H->mvec.reserve(bytes + 256));
    char *base = H->mvec.as<char>();
    MBufs M;
    for (int l = 0; l <= last_outside && l < H->nlev; ++l) {
        M.lv[l].vec = reinterpret_cast<double *>(base + o_vec[l]);
        M.lv[l].part = reinterpret_cast<double *>(base + o_part[l]);
        M.lv[l].tot = reinterpret_cast<double *>(base + o_tot[l]);
        M.lv[l].ld = H->pool[l]->ld;
    }
    const int64_t nv = (n * MK + 31) & ~(int64_t)31;
    double *ov = reinterpret_cast<double *>(base + o_outer);
    M.x = ov; M.r = ov + nv; M.p = ov + 2 * nv; M.Ap = ov + 3 * nv;
    M.z = reinterpret_cast<cyc_t *>(ov + 4 * nv); M.x0 = reinterpret_cast<cyc_t *>(ov + 5 * nv);
    M.part = reinterpret_cast<double *>(base + o_opart);
    M.tot = reinterpret_cast<double *>(base + o_otot);
    M.sc = reinterpret_cast<double *>(base + o_sc);
    M.ia = reinterpret_cast<int32_t *>(base + o_pairs);
    M.ib = M.ia + MK;
    M.g0 = (int)mgrid(n, MPARTS);
    // pairs: unused columns get the pair (-1, -1): a zero right-hand side, converged at once
    int32_t pairs[2 * MK];
    for (int y = 0; y < MK; ++y) {
        pairs[y] = y < count ? ia_host[y] : -1;
        pairs[MK + y] = y < count ? ib_host[y] : -1;
    }
    int32_t *pin = static_cast<int32_t *>(nodal_pinned(h));
    if (pin) memcpy(pin, pairs, sizeof pairs);
    NODAL_HIP_TRY(h, hipMemcpyAsync(M.ia, pin ? pin : pairs, sizeof pairs, hipMemcpyHostToDevice, st));
    if (!pin) NODAL_WAIT_STREAM(h, st);

    const SLevel *L0 = H->pool[0];
    const Ell A0 = L0->A();
    const double *dinv0 = L0->dinv.as<double>();
    const double tol = 1e-13;
    const bool functional = !(getenv("NODAL_PAIRS_FUNCTIONAL") && atoi(getenv("NODAL_PAIRS_FUNCTIONAL")) == 0);
    double *part_rr = M.part + (int64_t)2 * MPARTS * MK, *part_pap = M.part + (int64_t)4 * MPARTS * MK;
    m_set_scalars<<<1, 64, 0, st>>>(M.sc, tol * tol);
    m_init<<<M.g0, TB, 0, st>>>(n, M.ia, M.ib, M.x, M.r, M.Ap, dinv0, M.x0, part_rr, MPARTS);
    NODAL_HIP_TRY(h, hipGetLastError());

    auto iteration = [&](int it) -> int {
        kcycle_schedule(H, it);  // (the K-cycle's coefficients, per column: calibrated, then frozen -- sagg.hip)
        const int mrc = m_cycle<double>(h, H, M, 0, M.r, M.x0, M.z, true);  // leaves z.r, z.Ap partials in part[0..1]
        H->kfrozen = false;
        H->kslot = -1;
        NODAL_TRY(mrc);
        m_reduce_kernel<<<4, MR, 0, st>>>(M.part, MPARTS, M.g0, M.tot);  // [0] z.r [1] z.Ap [2] r.r [3] x.r
        m_direction<<<M.g0, TB, 0, st>>>(M.z, M.p, M.tot, M.sc, it & 1, n, functional);
        SAGG_DISPATCH_W(L0->wfix, (m_spmv<W><<<M.g0, TB, 0, st>>>(A0, M.p, M.Ap, part_pap, MPARTS, M.sc, it & 1)));
        m_reduce_kernel<<<1, MR, 0, st>>>(part_pap, MPARTS, M.g0, M.tot + 4 * MK);
        m_update<<<M.g0, TB, 0, st>>>(M.x, M.r, M.p, M.Ap, M.tot + 4 * MK, dinv0, M.x0, part_rr, MPARTS, M.sc, it & 1, n, functional);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    };
    // (a block iteration costs 1.2 ms at 1e6 rows: the cap follows what the first pair's own solve took on this
    // hierarchy -- four times that, at least 64 -- so that a slowly converging network costs a failing block tenths of
    // a second, not 2000 iterations, before the sweep goes on pair by pair)
    int64_t maxit = 2000;
    if (H->last_iters > 0) maxit = std::max<int64_t>(64, 4 * (int64_t)H->last_iters);
    if (getenv("NODAL_FCG_MAXIT")) maxit = atoll(getenv("NODAL_FCG_MAXIT"));
    double hs[MK * MSC];
    int64_t enqueued = 0;
    // A block iteration takes ~1.2 ms at 1e6 rows and a look at the scalars ~50 us: the host looks after one
    // iteration fewer than the block before took, then after every one.
    int batch = (H->mblock_n == n && H->mblock_iters > 7) ? H->mblock_iters - 1 : 10;
    int status = 0, polls = 0;
    while (status == 0) {
        for (int c = 0; c < batch; ++c, ++enqueued) NODAL_TRY(iteration((int)enqueued));
        NODAL_TRY(nodal_read_words(h, hs, M.sc, sizeof hs));
        ++polls;
        bool all = true, flag = false;
        for (int y = 0; y < MK; ++y) {
            const double *s = hs + y * MSC;
            const bool conv = s[F_CONV + ((enqueued - 1) & 1)] != 0.0 || (functional && s[M_FDONE] != 0.0);
            flag = flag || s[F_FLAG] != 0.0 || !(s[F_RR] == s[F_RR]);
            all = all && conv;
        }
        if (flag) status = 2;
        else if (all) status = 1;
        else if (enqueued >= maxit) status = 3;
        batch = 1;
    }
    int its = 0;
    for (int y = 0; y < count; ++y) its = std::max(its, (int)hs[y * MSC + F_ITERS]);
    if (iters) *iters = its;
    if (status == 1) H->mblock_iters = (int)enqueued, H->mblock_n = n;
    if (trace)
        fprintf(stderr, "[sagg] block of %d pairs: %d iterations (%lld enqueued, %d polls), status %d\n", count, its,
                (long long)enqueued, polls, status);
    if (status != 1) return -1;
    // (x.r of the final iterates: the partials of the last m_update, every column's vectors at rest by then)
    if (functional) m_reduce_kernel<<<1, MR, 0, st>>>(part_rr + (int64_t)MPARTS * MK, MPARTS, M.g0, M.tot + 3 * MK);
    m_read_pairs<<<1, 64, 0, st>>>(count, M.ia, M.ib, M.x, functional ? M.tot + 3 * MK : nullptr, res_dev);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}
