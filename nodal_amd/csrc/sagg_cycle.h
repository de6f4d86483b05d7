// Cycle kernels, setup driver and flexible-CG driver of the smoothed-aggregation path
// (second half of sagg.hip; included there, same translation unit).
#pragma once

namespace {

// ---------------------------------------------------------------------------------
// row kernels on ELL
// ---------------------------------------------------------------------------------
// Rows of uneven length (W == 0): LPR_RAGGED adjacent lanes share a row, lane `sub` takes the slots
// sub, sub + LPR, ...; up to eight of them in flight, so a row of 32 entries costs three dependent
// round trips (length, entries, gathers) where one thread walking it alone needed 2 x len / 4 + 1.
// Every lane of the group returns the sum.
constexpr int LPR_RAGGED = 1;
template <int W> struct RowLanes { static constexpr int value = W == 0 ? LPR_RAGGED : 1; };

template <class VT, class XF>
__device__ __forceinline__ double ell_row_lanes(const Ell &A, const VT *__restrict__ val, int64_t i, int sub, XF xf) {
    constexpr int Q = 8;
    const int32_t len = A.len[i];
    double acc = 0.0;
    for (int32_t s0 = sub; s0 < len; s0 += LPR_RAGGED * Q) {
        int32_t c[Q];
        double v[Q];
        if (A.dcol) {  // (uniform) columns as 16-bit offsets from the row
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool ok = s0 + q * LPR_RAGGED < len;
                const int64_t at = (int64_t)(ok ? s0 + q * LPR_RAGGED : s0) * A.ld + i;
                c[q] = (int32_t)i + (int32_t)A.dcol[at];
                const VT loaded = val[at];
                v[q] = ok ? (double)loaded : 0.0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const bool ok = s0 + q * LPR_RAGGED < len;  // (a slot past the end re-reads the lane's first one)
                const int64_t at = (int64_t)(ok ? s0 + q * LPR_RAGGED : s0) * A.ld + i;
                c[q] = A.col[at];
                const VT loaded = val[at];  // (unconditional: `at` is always a slot of the row)
                v[q] = ok ? (double)loaded : 0.0;
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) acc = fma(v[q], xf(c[q]), acc);
    }
#pragma unroll
    for (int off = LPR_RAGGED >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    return acc;
}

// W > 0: every row holds exactly W slots (zero-padded): all loads issued at once, no length load
// in front of them, one thread per row; W == 0: per-row length, LPR_RAGGED lanes per row.
// `t` is the global thread number: row t / lanes, lane t % lanes of the row.
// `val`: A.valf inside the cycle, A.val in the outer iteration's SpMV.
template <int W, class VT, class XF>
__device__ __forceinline__ double ell_row_w(const Ell &A, const VT *__restrict__ val, int64_t i, int sub, XF xf) {
    if constexpr (W == 0) {
        return ell_row_lanes(A, val, i, sub, xf);
    } else {
        int32_t c[W];
        double v[W];
        if (A.dcol) {  // (uniform) columns as 16-bit offsets from the row
#pragma unroll
            for (int q = 0; q < W; ++q) {
                c[q] = (int32_t)i + (int32_t)A.dcol[(int64_t)q * A.ld + i];
                v[q] = (double)val[(int64_t)q * A.ld + i];
            }
        } else {
#pragma unroll
            for (int q = 0; q < W; ++q) {
                c[q] = A.col[(int64_t)q * A.ld + i];
                v[q] = (double)val[(int64_t)q * A.ld + i];
            }
        }
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < W; ++q) acc = fma(v[q], xf(c[q]), acc);
        return acc;
    }
}

// the two halves of ell_row_w for W > 0, for kernels that request a row's slots before they know what to gather
template <int W, class VT>
__device__ __forceinline__ void ell_row_load(const Ell &A, const VT *__restrict__ val, int64_t i, int32_t (&c)[W > 0 ? W : 1],
                                             VT (&v)[W > 0 ? W : 1]) {
    if (A.dcol) {
#pragma unroll
        for (int q = 0; q < W; ++q) {
            c[q] = (int32_t)A.dcol[(int64_t)q * A.ld + i];  // (the offset: the row is added by ell_row_sum)
            v[q] = val[(int64_t)q * A.ld + i];
        }
    } else {
#pragma unroll
        for (int q = 0; q < W; ++q) {
            c[q] = A.col[(int64_t)q * A.ld + i] - (int32_t)i;
            v[q] = val[(int64_t)q * A.ld + i];
        }
    }
}
template <int W, class VT, class XF>
__device__ __forceinline__ double ell_row_sum(int64_t i, const int32_t (&c)[W > 0 ? W : 1], const VT (&v)[W > 0 ? W : 1], XF xf) {
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < W; ++q) acc = fma((double)v[q], xf((int32_t)i + c[q]), acc);
    return acc;
}

#define SAGG_DISPATCH_W(w, CALL)                         \
    switch (w) {                                         \
    case 4: { constexpr int W = 4; CALL; } break;        \
    case 5: { constexpr int W = 5; CALL; } break;        \
    case 6: { constexpr int W = 6; CALL; } break;        \
    case 8: { constexpr int W = 8; CALL; } break;        \
    case 12: { constexpr int W = 12; CALL; } break;      \
    case 16: { constexpr int W = 16; CALL; } break;      \
    case 20: { constexpr int W = 20; CALL; } break;      \
    case 24: { constexpr int W = 24; CALL; } break;      \
    case 32: { constexpr int W = 32; CALL; } break;      \
    default: { constexpr int W = 0; CALL; } break;       \
    }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
template <int NT = TB>
__device__ __forceinline__ double block_sum(double v) {  // NT threads; valid in every thread
    __shared__ double ws[NT / 64];
    __syncthreads();
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) s += ws[w];
    return s;
}
// every workgroup reduces the same partial array in the same order: deterministic, no atomics
__device__ __forceinline__ double reduce_partials(const double *__restrict__ part, int count) {
    double s = 0.0;
    for (int i = threadIdx.x; i < count; i += TB) s += part[i];
    return block_sum(s);
}
// two / three arrays at once: their loads in flight together, one pair of barriers
__device__ __forceinline__ void reduce_partials3(const double *__restrict__ pa, const double *__restrict__ pb,
                                                 const double *__restrict__ pc, int count, double &ra, double &rb,
                                                 double &rc) {
    __shared__ double ws[3][TB / 64];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < count; i += TB) {
        a += pa[i];
        b += pb[i];
        if (pc) c += pc[i];
    }
    __syncthreads();
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) {
        ws[0][threadIdx.x >> 6] = a;
        ws[1][threadIdx.x >> 6] = b;
        ws[2][threadIdx.x >> 6] = c;
    }
    __syncthreads();
    ra = rb = rc = 0.0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w) {
        ra += ws[0][w];
        rb += ws[1][w];
        rc += ws[2][w];
    }
}

// r = b - A x0, where x0 = w D^-1 b is the pre-smoothed iterate from a zero guess.  x0 comes from
// whoever produced b (f_init / f_update, k_restrict, k_second_residual): the row sum then gathers
// one vector, not D^-1 and b.
template <int W, typename TBV>
__global__ __launch_bounds__(TB * RowLanes<W>::value) void k_smooth_residual(Ell A, const TBV *__restrict__ b,
                                                        const cyc_t *__restrict__ x0, cyc_t *__restrict__ r) {
    constexpr int LPR = RowLanes<W>::value, NT = TB * LPR;  // (the workgroup still covers TB rows)
    const int sub = threadIdx.x & (LPR - 1);
    for (int64_t t = (int64_t)xcd_block() * NT + threadIdx.x; t / LPR < A.n; t += (int64_t)gridDim.x * NT) {
        const int64_t i = t / LPR;
        const double s = ell_row_w<W>(A, A.valf, i, sub, [&](int32_t j) { return (double)x0[j]; });
        if (sub == 0) r[i] = (cyc_t)((double)b[i] - s);
    }
}

// rc = R r.  R is stored in blocks of RL = 8 entries per coarse row, block q of row I at
// [(q * rld + I) * 8 .. +8): the eight lanes of a row read one contiguous block per step, a
// wavefront eight consecutive rows = 64 contiguous entries.  (One thread per row walked its
// 17-40 entries alone: 10-20 dependent round trips, 15-20 us whatever the level's size.)
constexpr int RL = 8;
template <typename TO>
__global__ __launch_bounds__(TB) void k_restrict(int64_t nc, int64_t rld, const int32_t *__restrict__ rcol,
                                                 const float *__restrict__ rval, const int32_t *__restrict__ rlen,
                                                 const cyc_t *__restrict__ r, TO *__restrict__ rc,
                                                 const double *__restrict__ cdinv, cyc_t *__restrict__ x0c) {
    const int sub = threadIdx.x & (RL - 1);
    const int64_t rows_per_pass = (int64_t)gridDim.x * (TB / RL);
    for (int64_t I0 = (int64_t)xcd_block() * (TB / RL); I0 < nc; I0 += rows_per_pass) {
        const int64_t I = I0 + threadIdx.x / RL;
        double s = 0.0;
        if (I < nc) {
            const int32_t len = rlen[I];
            // (rows are zero-padded to whole blocks: no per-entry test) RU blocks in flight
            constexpr int RU = 6;
            for (int32_t q0 = 0; q0 * RL < len; q0 += RU) {
                int32_t c[RU];
                double v[RU];
#pragma unroll
                for (int u = 0; u < RU; ++u) {
                    const bool ok = (q0 + u) * RL < len;
                    const int64_t at = ((int64_t)(ok ? q0 + u : q0) * rld + I) * RL + sub;
                    c[u] = rcol[at];
                    const float loaded = rval[at];  // (unconditional: a block of the row either way)
                    v[u] = ok ? (double)loaded : 0.0;
                }
#pragma unroll
                for (int u = 0; u < RU; ++u) s = fma(v[u], (double)r[c[u]], s);
            }
        }
#pragma unroll
        for (int off = RL >> 1; off > 0; off >>= 1) s += __shfl_down(s, off, RL);
        if (I < nc && sub == 0) {
            rc[I] = (TO)s;
            if (x0c) x0c[I] = (cyc_t)(OMEGA * cdinv[I] * s);  // pre-smoothed iterate of the coarse visit
        }
    }
}

// K-cycle coefficients from the five dot products of the two inner FCG steps
struct KCoef { double s1, s2; };
__device__ __forceinline__ KCoef kcycle_coefficients(const double *__restrict__ part, int nparts) {
    if (nparts == 0) return KCoef{1.0, 0.0};  // plain V hand-over
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

// s1, s2 of the K-cycle, once per cycle: one workgroup of five wavefronts, one per dot product
// coef[0..1] = s1, s2 of this cycle.  slot >= 0: the sample (s1, s2, t = alpha1 / rho1) also goes to slot `slot` of a ring
// of three behind them, and coef[3..5] = the mean of the ring's first `count` slots -- the FROZEN coefficients the
// cycles between two calibrations use (sagg.hip, cycle(): k_spmv_resid + k_prolong with coef + 3).
__global__ __launch_bounds__(320) void k_kcoef(const double *__restrict__ part, int nparts, double *__restrict__ coef,
                                               int slot, int count) {
    __shared__ double d[5];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s = 0.0;
    for (int i = lane; i < nparts; i += 64) s += part[w * DOT_BLOCKS + i];
    s = wave_sum(s);
    if (lane == 0) d[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double rho1 = d[0], alpha1 = d[1], gamma = d[2], beta = d[3], alpha2 = d[4];
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
        coef[0] = s1;
        coef[1] = s2;
        if (slot >= 0) {
            double *ring = coef + 6;
            ring[3 * slot + 0] = s1;
            ring[3 * slot + 1] = s2;
            ring[3 * slot + 2] = rho1 > 0.0 ? alpha1 / rho1 : 0.0;
            double m0 = 0.0, m1 = 0.0, m2 = 0.0;
            for (int q = 0; q < count; ++q) {  // (slot `slot` is among them: count = min(samples so far, 3))
                m0 += ring[3 * q + 0];
                m1 += ring[3 * q + 1];
                m2 += ring[3 * q + 2];
            }
            coef[3] = m0 / count;
            coef[4] = m1 / count;
            coef[5] = m2 / count;
        }
    }
}

// r2 = rc - t A c1 with a FROZEN t (frz[2]) and the start iterate of the second visit: the first k_spmv_dots and
// k_second_residual of an adaptive cycle in one launch, no dot products
template <int W>
__global__ __launch_bounds__(TB * RowLanes<W>::value) void k_spmv_resid(Ell A, const cyc_t *__restrict__ c,
                                                   const cyc_t *__restrict__ rc, const double *__restrict__ frz,
                                                   cyc_t *__restrict__ r2, const double *__restrict__ dinv,
                                                   cyc_t *__restrict__ x0) {
    constexpr int LPR = RowLanes<W>::value, NT = TB * LPR;
    const int sub = threadIdx.x & (LPR - 1);
    const double t = frz[2];
    for (int64_t tt = (int64_t)xcd_block() * NT + threadIdx.x; tt / LPR < A.n; tt += (int64_t)gridDim.x * NT) {
        const int64_t i = tt / LPR;
        const double sd = ell_row_w<W>(A, A.valf, i, sub, [&](int32_t j) { return (double)c[j]; });
        if (sub != 0) continue;
        const double v = fma(-t, (double)(cyc_t)sd, (double)rc[i]);  // (A c1 rounded like the stored v1 of the adaptive cycle)
        r2[i] = (cyc_t)v;
        x0[i] = (cyc_t)(OMEGA * dinv[i] * v);
    }
}

// xp = x + P (s1 c1 + s2 c2)   (coef == nullptr: plain V hand-over, xp = x + P c1)
template <typename TC>
__global__ __launch_bounds__(TB) void k_prolong(int64_t n, int64_t ld, const int32_t *__restrict__ pcol,
                                                const float *__restrict__ pval, const cyc_t *__restrict__ x,
                                                const TC *__restrict__ c1, const TC *__restrict__ c2,
                                                const double *__restrict__ coef, cyc_t *__restrict__ xp) {
    const bool two = coef != nullptr;
    const double s1 = two ? coef[0] : 1.0, s2 = two ? coef[1] : 0.0;
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        // all eight loads of the row at once (empty slots hold value 0: their column is clamped), then
        // the gathers: two dependent stages instead of three
        int32_t J[PW];
        double w[PW];
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            J[q] = pcol[(int64_t)q * ld + i];
            w[q] = (double)pval[(int64_t)q * ld + i];
        }
        double s = (double)x[i];
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int32_t j = J[q] < 0 ? 0 : J[q];
            const double e = two ? s1 * (double)c1[j] + s2 * (double)c2[j] : (double)c1[j];
            s = fma(J[q] < 0 ? 0.0 : w[q], e, s);
        }
        xp[i] = (cyc_t)s;
    }
}

// out = xp + w D^-1 (b - A xp); DOTS: partial sums of out.b and out.u (the outer iteration's z.r, z.Ap)
template <int W, bool DOTS, typename TBV, typename TOUT>
__global__ __launch_bounds__(TB * RowLanes<W>::value) void k_post(Ell A, const double *__restrict__ dinv, const TBV *__restrict__ b,
                                             const cyc_t *__restrict__ xp, TOUT *__restrict__ out,
                                             const double *__restrict__ u, double *__restrict__ p_ob,
                                             double *__restrict__ p_ou) {
    double a0 = 0.0, a1 = 0.0;
    constexpr int LPR = RowLanes<W>::value, NT = TB * LPR;  // (the workgroup still covers TB rows)
    const int sub = threadIdx.x & (LPR - 1);
    for (int64_t t = (int64_t)xcd_block() * NT + threadIdx.x; t / LPR < A.n; t += (int64_t)gridDim.x * NT) {
        const int64_t i = t / LPR;
        const double s = ell_row_w<W>(A, A.valf, i, sub, [&](int32_t j) { return (double)xp[j]; });
        if (sub != 0) continue;
        const double bi = (double)b[i];
        // (the dots take the value as stored: z.r and z.Ap are those of the z the direction is built from)
        const TOUT os = (TOUT)fma(OMEGA * dinv[i], bi - s, (double)xp[i]);
        const double o = (double)os;
        out[i] = os;
        if (DOTS) {
            a0 = fma(o, bi, a0);
            a1 = fma(o, u[i], a1);
        }
    }
    if (DOTS) {
        a0 = block_sum<NT>(a0);
        a1 = block_sum<NT>(a1);
        if (threadIdx.x == 0) {
            p_ob[blockIdx.x] = a0;
            p_ou[blockIdx.x] = a1;
        }
    }
}

// v = A c and the partial dot products c.v, c.u1 (and c.u2 when given)
template <int W>
__global__ __launch_bounds__(TB * RowLanes<W>::value) void k_spmv_dots(Ell A, const cyc_t *__restrict__ c, cyc_t *__restrict__ v,
                                                  const cyc_t *__restrict__ u1, const cyc_t *__restrict__ u2,
                                                  double *__restrict__ p_cv, double *__restrict__ p_cu1,
                                                  double *__restrict__ p_cu2) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    constexpr int LPR = RowLanes<W>::value, NT = TB * LPR;  // (the workgroup still covers TB rows)
    const int sub = threadIdx.x & (LPR - 1);
    for (int64_t t = (int64_t)xcd_block() * NT + threadIdx.x; t / LPR < A.n; t += (int64_t)gridDim.x * NT) {
        const int64_t i = t / LPR;
        const double sd = ell_row_w<W>(A, A.valf, i, sub, [&](int32_t j) { return (double)c[j]; });
        if (sub != 0) continue;
        const cyc_t vs = (cyc_t)sd;
        v[i] = vs;
        const double s = (double)vs, ci = (double)c[i];
        a0 = fma(ci, s, a0);
        a1 = fma(ci, (double)u1[i], a1);
        if (u2) a2 = fma(ci, (double)u2[i], a2);
    }
    a0 = block_sum<NT>(a0);
    a1 = block_sum<NT>(a1);
    a2 = block_sum<NT>(a2);
    if (threadIdx.x == 0) {
        p_cv[blockIdx.x] = a0;
        p_cu1[blockIdx.x] = a1;
        if (p_cu2) p_cu2[blockIdx.x] = a2;
    }
}

// r2 = rc - (alpha1 / rho1) v1
__global__ __launch_bounds__(TB) void k_second_residual(int64_t n, const cyc_t *__restrict__ rc,
                                                        const cyc_t *__restrict__ v1,
                                                        const double *__restrict__ part, int nparts,
                                                        cyc_t *__restrict__ r2, const double *__restrict__ dinv,
                                                        cyc_t *__restrict__ x0) {
    double rho1, alpha1, unused;
    reduce_partials3(part + 0 * DOT_BLOCKS, part + 1 * DOT_BLOCKS, nullptr, nparts, rho1, alpha1, unused);
    const double t = rho1 > 0.0 ? alpha1 / rho1 : 0.0;
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
    {
        const double v = fma(-t, (double)v1[i], (double)rc[i]);
        r2[i] = (cyc_t)v;
        x0[i] = (cyc_t)(OMEGA * dinv[i] * v);
    }
}

// coarsest level outside the tail kernel: dense inverse (n <= COARSEST) or a diagonal matrix
__global__ __launch_bounds__(TB) void k_coarsest(int64_t n, const double *__restrict__ inv,
                                                 const double *__restrict__ dinv, const double *__restrict__ b,
                                                 double *__restrict__ out) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        double s = 0.0;
        if (inv)
            for (int64_t j = 0; j < n; ++j) s = fma(inv[i * n + j], b[j], s);
        else
            s = dinv[i] * b[i];
        out[i] = s;
    }
}

// ---- tail: the smallest levels in ONE 1024-thread workgroup, matrices and vectors in LDS -------
// A level visit of separate launches costs 4-7 us per kernel whatever its size (two to three
// dependent memory round trips + the launch); here the levels' matrices (f64 values, u16 columns,
// rows padded to the level's longest), transfer operators and vectors are copied into LDS once
// per launch and the V-cycle -- d.nu Jacobi sweeps before and after each coarse correction,
// cheap at LDS latency -- runs between workgroup barriers.  The last level is solved with its
// dense inverse (or its diagonal: nothing but isolated nodes).


// row i of an LDS-resident ELL matrix times x, by the `lpr` (power of two) adjacent lanes that
// share the row: lane `sub` takes the slots sub, sub + lpr, ...; four slots in flight; every lane
// of the group returns the sum
__device__ __forceinline__ double lds_row(const double *aval, const uint16_t *acol, int n, int width, int i,
                                          int sub, int lpr, const double *x) {
    double s0 = 0.0, s1 = 0.0;
    int t = sub;
    for (; t + 3 * lpr < width; t += 4 * lpr) {
        const double v0 = aval[t * n + i], v1 = aval[(t + lpr) * n + i], v2 = aval[(t + 2 * lpr) * n + i],
                     v3 = aval[(t + 3 * lpr) * n + i];
        const int c0 = acol[t * n + i], c1 = acol[(t + lpr) * n + i], c2 = acol[(t + 2 * lpr) * n + i],
                  c3 = acol[(t + 3 * lpr) * n + i];
        s0 = fma(v0, x[c0], s0);
        s1 = fma(v1, x[c1], s1);
        s0 = fma(v2, x[c2], s0);
        s1 = fma(v3, x[c3], s1);
    }
    for (; t < width; t += lpr) s0 = fma(aval[t * n + i], x[acol[t * n + i]], s0);
    double s = s0 + s1;
    for (int off = lpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// the tail levels' matrices and transfer operators, packed once per setup into a global image
// laid out exactly like the LDS copy
__global__ __launch_bounds__(1024) void k_tail_pack(TailDesc d, char *__restrict__ image) {
    const int tid = threadIdx.x;
    const int last = d.nlev - 1;
    auto f64 = [&](int off) { return reinterpret_cast<double *>(image + off); };
    auto u16 = [&](int off) { return reinterpret_cast<uint16_t *>(image + off); };
    auto copy_f64 = [&](double *dst, const double *src, int rows, int64_t ld, int slots) {
        for (int e = tid; e < rows * slots; e += 1024) {  // dst[t * rows + i] = src[t * ld + i]
            const int t = e / rows, i = e - t * rows;
            dst[e] = src[(int64_t)t * ld + i];
        }
    };
    auto copy_u16 = [&](uint16_t *dst, const int32_t *src, int rows, int64_t ld, int slots) {
        for (int e = tid; e < rows * slots; e += 1024) {
            const int t = e / rows, i = e - t * rows;
            const int32_t v = src[(int64_t)t * ld + i];
            dst[e] = (uint16_t)(v < 0 ? 0 : v);
        }
    };
    for (int k = 0; k < last; ++k) {
        const TailLevelDesc &L = d.lv[k];
        // A: slot t of row i, zero / column 0 past the row's end
        for (int e = tid; e < L.n * L.width; e += 1024) {
            const int t = e / L.n, i = e - t * L.n;
            const bool in = t < L.alen[i];
            f64(L.o_aval)[e] = in ? L.aval[(int64_t)t * L.ld + i] : 0.0;
            const int32_t c = in ? L.acol[(int64_t)t * L.ld + i] : 0;
            u16(L.o_acol)[e] = (uint16_t)(c < 0 ? 0 : c);
        }
        copy_f64(f64(L.o_dinv), L.dinv, L.n, L.ld, 1);
        copy_f64(f64(L.o_pval), L.pval, L.n, L.ld, PW);       // (empty slots: value 0, column -1 -> 0)
        copy_u16(u16(L.o_pcol), L.pcol, L.n, L.ld, PW);
        // R: block q of row I at [(q * rld + I) * 8 ..): nq "slots" of nc * 8 contiguous entries
        copy_f64(f64(L.o_rval), L.rval, L.nc * RL, (int64_t)L.rld * RL, L.nq);
        copy_u16(u16(L.o_rcol), L.rcol, L.nc * RL, (int64_t)L.rld * RL, L.nq);
    }
    const TailLevelDesc &L = d.lv[last];
    if (d.inv) copy_f64(f64(L.o_aval), d.inv, L.n * L.n, L.n * L.n, 1);
    else copy_f64(f64(L.o_aval), L.dinv, L.n, L.n, 1);
}

// row i of the tail's FIRST level, whose slots this lane keeps in registers for the whole launch
// (slot j of the lane is slot sub + j * lpr of the row; padding slots hold value 0, column 0)
template <int SLOTS>
__device__ __forceinline__ double reg_row(const double (&v)[SLOTS], const uint32_t (&c)[SLOTS / 2], int lpr,
                                          const double *x) {
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int j = 0; j < SLOTS; j += 2) {
        s0 = fma(v[j], x[c[j / 2] & 0xffffu], s0);
        s1 = fma(v[j + 1], x[c[j / 2] >> 16], s1);
    }
    double s = s0 + s1;
    for (int off = lpr >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    return s;
}

// SLOTS >= ceil(width / lpr) of the first tail level (8, 16 or 32)
template <int SLOTS>
__global__ __launch_bounds__(1024) void k_tail(TailDesc d, const char *__restrict__ image,
                                               const double *__restrict__ rc, double *__restrict__ out, int vs) {
    // vs: stride of the vectors' elements; vs > 1 = a block of interleaved right-hand sides (sagg_multi.h), one
    // workgroup per column
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int voff = vs > 1 ? (int)blockIdx.x : 0;
    const int last = d.nlev - 1;
    const int TAIL_NU = d.nu;
    // LDS holds the image from d.skip on (the first level's matrix, which leads the image, goes to
    // registers straight from global memory): offsets in the descriptor are image offsets
    auto f64 = [&](int off) { return reinterpret_cast<double *>(smem + (off - d.skip)); };
    auto u16 = [&](int off) { return reinterpret_cast<uint16_t *>(smem + (off - d.skip)); };
    int stamp_no = 0;
    auto stamp = [&]() {
        if (d.stamps && tid == 0) d.stamps[stamp_no] = wall_clock64();
        ++stamp_no;
    };
    stamp();
    // the first level's matrix: image -> registers (no LDS matrix traffic in the sweeps, no LDS space)
    double v0[SLOTS];
    uint32_t c0[SLOTS / 2];
    {
        const TailLevelDesc &L = d.lv[0];
        const double *aval = reinterpret_cast<const double *>(image + L.o_aval);
        const uint16_t *acol = reinterpret_cast<const uint16_t *>(image + L.o_acol);
        const int lpr = L.lpr, sub = tid & (lpr - 1), i = tid / lpr, ii = i < L.n ? i : L.n - 1;
        const bool row = last > 0 && i < L.n;
#pragma unroll
        for (int j = 0; j < SLOTS; j += 2) {
            const int ta = sub + j * lpr, tb = ta + lpr;
            const bool ha = row && ta < L.width, hb = row && tb < L.width;
            const int ea = ha ? ta * L.n + ii : 0, eb = hb ? tb * L.n + ii : 0;  // (unconditional loads)
            const double va = aval[ea], vb = aval[eb];
            const uint32_t ca = acol[ea], cb = acol[eb];
            v0[j] = ha ? va : 0.0;
            v0[j + 1] = hb ? vb : 0.0;
            c0[j / 2] = (ha ? ca : 0u) | ((hb ? cb : 0u) << 16);
        }
    }
    {   // the rest of the image -> LDS: 16-byte pieces, all of a lane's in flight at once (one round
        // trip; the image never survives in L2 between two launches, the level-0 kernels stream past it)
        constexpr int FLY = (TAIL_LDS_BUDGET / 16 + 1023) / 1024;
        const int4 *src = reinterpret_cast<const int4 *>(image + d.skip);
        int4 *dst = reinterpret_cast<int4 *>(smem);
        const int pieces = (d.image_bytes - d.skip) / 16;
        int4 v[FLY];
#pragma unroll
        for (int u = 0; u < FLY; ++u) v[u] = src[tid + u * 1024 < pieces ? tid + u * 1024 : 0];
#pragma unroll
        for (int u = 0; u < FLY; ++u)
            if (tid + u * 1024 < pieces) dst[tid + u * 1024] = v[u];
    }
    for (int i = tid; i < d.lv[0].n; i += 1024) f64(d.lv[0].o_B)[i] = rc[(int64_t)i * vs + voff];
    __syncthreads();
    stamp();
    // ---- down ----
    for (int k = 0; k < last; ++k) {
        const TailLevelDesc &L = d.lv[k];
        const double *aval = f64(L.o_aval), *dinv = f64(L.o_dinv), *B = f64(L.o_B);
        const uint16_t *acol = u16(L.o_acol);
        double *X = f64(L.o_X), *Y = f64(L.o_Y), *R = f64(L.o_R);
        const int lpr = L.lpr, sub = tid & (lpr - 1), rows = 1024 / lpr;  // lpr lanes share a row
        for (int i = tid; i < L.n; i += 1024) X[i] = OMEGA * dinv[i] * B[i];
        __syncthreads();
        for (int sweep = 1; sweep < TAIL_NU; ++sweep) {
            for (int i0 = 0; i0 < L.n; i0 += rows) {  // (one pass at the first level: n * lpr <= 1024)
                const int i = i0 + tid / lpr, ii = i < L.n ? i : L.n - 1;
                const double ax = k == 0 ? reg_row<SLOTS>(v0, c0, lpr, X)
                                         : lds_row(aval, acol, L.n, L.width, ii, sub, lpr, X);
                if (i < L.n && sub == 0) Y[i] = fma(OMEGA * dinv[i], B[i] - ax, X[i]);
            }
            __syncthreads();
            double *t = X; X = Y; Y = t;
        }
        for (int i0 = 0; i0 < L.n; i0 += rows) {
            const int i = i0 + tid / lpr, ii = i < L.n ? i : L.n - 1;
            const double ax = k == 0 ? reg_row<SLOTS>(v0, c0, lpr, X)
                                     : lds_row(aval, acol, L.n, L.width, ii, sub, lpr, X);
            if (i < L.n && sub == 0) R[i] = B[i] - ax;
        }
        __syncthreads();
        stamp();
        // (X holds the smoothed iterate: remember which buffer via the parity of TAIL_NU below)
        const double *rval = f64(L.o_rval);
        const uint16_t *rcol = u16(L.o_rcol);
        double *Bc = f64(d.lv[k + 1].o_B);
        for (int I0 = 0; I0 < L.nc; I0 += 1024 / RL) {  // eight lanes per coarse row
            const int I = I0 + tid / RL, r8 = tid & (RL - 1);
            double s = 0.0;
            if (I < L.nc) {
                double s1 = 0.0;
                const int e0 = I * RL + r8, step = L.nc * RL;
                int q = 0;
                for (; q + 3 < L.nq; q += 4) {  // (four blocks in flight)
                    const int ea = e0 + q * step, eb = ea + step, ec = eb + step, ed = ec + step;
                    const double va = rval[ea], vb = rval[eb], vc = rval[ec], vd = rval[ed];
                    const int ca = rcol[ea], cb = rcol[eb], cc = rcol[ec], cd = rcol[ed];
                    s = fma(va, R[ca], s);
                    s1 = fma(vb, R[cb], s1);
                    s = fma(vc, R[cc], s);
                    s1 = fma(vd, R[cd], s1);
                }
                for (; q < L.nq; ++q) s = fma(rval[e0 + q * step], R[rcol[e0 + q * step]], s);
                s += s1;
            }
#pragma unroll
            for (int off = RL >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            if (I < L.nc && r8 == 0) Bc[I] = s;
        }
        __syncthreads();
        stamp();
    }
    {
        const TailLevelDesc &L = d.lv[last];
        const double *inv = f64(L.o_aval), *B = f64(L.o_B);
        double *E = f64(L.o_X);
        for (int i = tid; i < L.n; i += 1024) {
            double s = 0.0;
            if (d.inv) {
                double s1 = 0.0, s2 = 0.0, s3 = 0.0;
                const double *row = inv + i * L.n;
                int j = 0;
                for (; j + 3 < L.n; j += 4) {
                    s = fma(row[j], B[j], s);
                    s1 = fma(row[j + 1], B[j + 1], s1);
                    s2 = fma(row[j + 2], B[j + 2], s2);
                    s3 = fma(row[j + 3], B[j + 3], s3);
                }
                for (; j < L.n; ++j) s = fma(row[j], B[j], s);
                s = (s + s1) + (s2 + s3);
            } else {
                s = inv[i] * B[i];
            }
            E[i] = s;
        }
        __syncthreads();
        stamp();
    }
    // ---- up ----
    const double *Ec = f64(d.lv[last].o_X);
    for (int k = last - 1; k >= 0; --k) {
        const TailLevelDesc &L = d.lv[k];
        const double *aval = f64(L.o_aval), *dinv = f64(L.o_dinv), *B = f64(L.o_B), *pval = f64(L.o_pval);
        const uint16_t *acol = u16(L.o_acol), *pcol = u16(L.o_pcol);
        double *X = f64((TAIL_NU & 1) ? L.o_X : L.o_Y), *Y = f64((TAIL_NU & 1) ? L.o_Y : L.o_X);  // X: pre-smoothed iterate
        for (int i = tid; i < L.n; i += 1024) {
            double s = X[i];
#pragma unroll
            for (int q = 0; q < PW; ++q) s = fma(pval[q * L.n + i], Ec[pcol[q * L.n + i]], s);
            Y[i] = s;
        }
        __syncthreads();
        stamp();
        const int lpr = L.lpr, sub = tid & (lpr - 1), rows = 1024 / lpr;
        for (int sweep = 0; sweep < TAIL_NU; ++sweep) {
            for (int i0 = 0; i0 < L.n; i0 += rows) {
                const int i = i0 + tid / lpr, ii = i < L.n ? i : L.n - 1;
                const double ax = k == 0 ? reg_row<SLOTS>(v0, c0, lpr, Y)
                                         : lds_row(aval, acol, L.n, L.width, ii, sub, lpr, Y);
                if (i < L.n && sub == 0) X[i] = fma(OMEGA * dinv[i], B[i] - ax, Y[i]);
            }
            __syncthreads();
            double *t = X; X = Y; Y = t;
        }
        Ec = Y;  // the last sweep's result
        stamp();
    }
    for (int i = tid; i < d.lv[0].n; i += 1024) out[(int64_t)i * vs + voff] = Ec[i];
    stamp();
    if (d.stamps && tid == 0) d.stamps[63] = stamp_no;
}

// ---------------------------------------------------------------------------------
// flexible CG (Polak-Ribiere beta: the K-cycle is a mildly non-linear operator)
// ---------------------------------------------------------------------------------
// Scalars on the device.  The host passes the iteration number to every kernel, and the words an
// iteration hands to the next one live in slots indexed by its parity, so a kernel never writes
// a word that other workgroups of the SAME kernel read (they may start later):
//   RZ[it&1], CONV[it&1]  written by f_direction(it);  ALPHA[it&1] by f_update(it).
// Once an iteration has converged, CONV stays raised (f_direction hands it on) and the CG
// kernels of every later iteration return at once.
enum { F_RZ = 0, F_ALPHA = 2, F_CONV = 4, F_RR = 6, F_BB = 7, F_FLAG = 8, F_ITERS = 9, F_TOL2 = 10,
       F_ITNO = 12,  // [parity]: iterations started before the NEXT one, written by f_direction (see there)
       F_COUNT = 16 };
constexpr int MAX_PARTIALS = 1024;

__global__ __launch_bounds__(TB) void f_init(const double *__restrict__ b, double *__restrict__ x,
                                             double *__restrict__ r, double *__restrict__ Ap,
                                             const double *__restrict__ dinv, cyc_t *__restrict__ x0,
                                             double *__restrict__ part_rr, int64_t n) {
    double srr = 0.0;
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const double ri = b[i];
        x[i] = 0.0;
        r[i] = ri;
        x0[i] = (cyc_t)(OMEGA * dinv[i] * ri);
        Ap[i] = 0.0;
        srr = fma(ri, ri, srr);
    }
    srr = block_sum(srr);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = srr;
}

// beta from the dots of the cycle's last kernel; p = z + beta p; convergence test on |r|^2
// The kernels of an iteration take its PARITY only; the iteration number itself lives on the device:
// f_direction(parity cur) reads ITNO[prev] -- written by the previous iteration's f_direction, so no
// workgroup of this launch can see it change -- and leaves ITNO[cur] = that + 1.  A pair of iterations
// (parities 0, 1) therefore has fixed kernel arguments and can be replayed as a hipGraph.
__global__ __launch_bounds__(TB) void f_direction(const cyc_t *__restrict__ z, double *__restrict__ p,
                                                  const double *__restrict__ part_rz,
                                                  const double *__restrict__ part_zap,
                                                  const double *__restrict__ part_rr, int nparts,
                                                  double *__restrict__ sc, int parity, int64_t n) {
    const int cur = parity & 1, prev = cur ^ 1;
    // The vector operands of the thread's first FD_PRE rows are requested BEFORE the scalars and the partial sums are
    // looked at: the reduction in front of the update is a dependent round trip and a barrier (~2 us of an 8-us
    // kernel) during which nothing streamed.
    constexpr int FD_PRE = 4;
    const int64_t i0 = (int64_t)xcd_block() * TB + threadIdx.x, stride = (int64_t)gridDim.x * TB;
    cyc_t zq[FD_PRE];  // (as loaded: a conversion here would wait for the load)
    double pq[FD_PRE];
#pragma unroll
    for (int u = 0; u < FD_PRE; ++u) {
        const int64_t i = i0 + u * stride, ii = i < n ? i : n - 1;  // (unconditional loads from a row of the vector)
        zq[u] = z[ii];
        pq[u] = p[ii];
    }
    const int iter = (int)sc[F_ITNO + prev];
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[F_ITNO + cur] = (double)(iter + 1);
    if (iter > 0 && sc[F_CONV + prev] != 0.0) {  // converged earlier: hand the flag on
        if (blockIdx.x == 0 && threadIdx.x == 0) sc[F_CONV + cur] = 1.0;
        return;
    }
    double rz_new, zap, rr;
    reduce_partials3(part_rz, part_zap, part_rr, nparts, rz_new, zap, rr);
    const double rz_old = iter > 0 ? sc[F_RZ + prev] : 1.0;
    const double beta = (iter > 0 && rz_old != 0.0) ? -sc[F_ALPHA + prev] * zap / rz_old : 0.0;
    const double bb = iter == 0 ? rr : sc[F_BB];
    const bool bad = !(rz_new >= 0.0) || !(rr == rr);  // preconditioner not positive / NaN
    const bool converged = bb == 0.0 || rr <= sc[F_TOL2] * bb || bad;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[F_RZ + cur] = rz_new;
        sc[F_RR] = rr;
        if (iter == 0) sc[F_BB] = rr;
        if (bad) sc[F_FLAG] = 1.0;
        sc[F_CONV + cur] = converged ? 1.0 : 0.0;
        if (converged) sc[F_ITERS] = (double)iter;
    }
    if (converged) return;  // uniform over the grid: every workgroup reduces the same partials
#pragma unroll
    for (int u = 0; u < FD_PRE; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < n) p[i] = iter > 0 ? fma(beta, pq[u], (double)zq[u]) : (double)zq[u];
    }
    for (int64_t i = i0 + FD_PRE * stride; i < n; i += stride)
        p[i] = iter > 0 ? fma(beta, p[i], (double)z[i]) : (double)z[i];
}

// y = A x on the level-0 matrix (fp64 values): the Krylov SpMV of the general path when the hierarchy's
// matrix IS the system's (sagg_spmv)
template <int W>
__global__ __launch_bounds__(TB * RowLanes<W>::value) void k_ell_spmv(Ell A, const double *__restrict__ x,
                                                                      double *__restrict__ y) {
    constexpr int LPR = RowLanes<W>::value, NT = TB * LPR;
    const int sub = threadIdx.x & (LPR - 1);
    for (int64_t t = (int64_t)xcd_block() * NT + threadIdx.x; t / LPR < A.n; t += (int64_t)gridDim.x * NT) {
        const int64_t i = t / LPR;
        const double s = ell_row_w<W>(A, A.val, i, sub, [&](int32_t j) { return x[j]; });
        if (sub == 0) y[i] = s;
    }
}

// Ap = A p, partials of p.Ap
template <int W>
__global__ __launch_bounds__(TB * RowLanes<W>::value) void f_spmv(Ell A, const double *__restrict__ p, double *__restrict__ Ap,
                                             double *__restrict__ part_pap, const double *__restrict__ sc,
                                             int iter) {
    if (sc[F_CONV + (iter & 1)] != 0.0) return;
    double acc = 0.0;
    constexpr int LPR = RowLanes<W>::value, NT = TB * LPR;  // (the workgroup still covers TB rows)
    const int sub = threadIdx.x & (LPR - 1);
    for (int64_t t = (int64_t)xcd_block() * NT + threadIdx.x; t / LPR < A.n; t += (int64_t)gridDim.x * NT) {
        const int64_t i = t / LPR;
        const double s = ell_row_w<W>(A, A.val, i, sub, [&](int32_t j) { return p[j]; });
        if (sub != 0) continue;
        Ap[i] = s;
        acc = fma(p[i], s, acc);
    }
    acc = block_sum<NT>(acc);
    if (threadIdx.x == 0) part_pap[blockIdx.x] = acc;
}

// f_direction and f_spmv in ONE launch (NODAL_SA_FUSE_DIR=0: the two kernels): every workgroup forms beta from the
// partials, the row sum gathers the NEW direction's entries  z[j] + beta p_old[j]  (the same fma as f_direction's: the
// same bits) and the row's own entry is written to the OTHER direction buffer (p_old is still being gathered by the
// neighbours' rows: the two buffers swap roles with the iteration's parity).  One launch floor (~4.5 us) and one pass
// over p less per iteration; the scalars are written by workgroup 0 exactly as f_direction writes them.
template <int W>
__global__ __launch_bounds__(TB * RowLanes<W>::value) void f_dir_spmv(Ell A, const cyc_t *__restrict__ z,
                                                 const double *__restrict__ p_old, double *__restrict__ p_new,
                                                 double *__restrict__ Ap, const double *__restrict__ part_rz,
                                                 const double *__restrict__ part_zap, const double *__restrict__ part_rr,
                                                 int nparts, double *__restrict__ part_pap, double *__restrict__ sc,
                                                 int parity) {
    static_assert(RowLanes<W>::value == 1, "the reduction of the partials assumes TB threads");
    const int cur = parity & 1, prev = cur ^ 1;
    const int64_t i0 = (int64_t)xcd_block() * TB + threadIdx.x, stride = (int64_t)gridDim.x * TB;
    // the first row's slots and the row's own entries are requested before the scalars and partials are looked at
    int32_t c0[W > 0 ? W : 1];
    double v0[W > 0 ? W : 1];
    const int64_t ifirst = i0 < A.n ? i0 : A.n - 1;
    if constexpr (W > 0) ell_row_load<W>(A, A.val, ifirst, c0, v0);
    const cyc_t zfirst = z[ifirst];
    const double pfirst = p_old[ifirst];
    const int iter = (int)sc[F_ITNO + prev];
    if (blockIdx.x == 0 && threadIdx.x == 0) sc[F_ITNO + cur] = (double)(iter + 1);
    if (iter > 0 && sc[F_CONV + prev] != 0.0) {  // converged earlier: hand the flag on
        if (blockIdx.x == 0 && threadIdx.x == 0) sc[F_CONV + cur] = 1.0;
        return;
    }
    double rz_new, zap, rr;
    reduce_partials3(part_rz, part_zap, part_rr, nparts, rz_new, zap, rr);
    const double rz_old = iter > 0 ? sc[F_RZ + prev] : 1.0;
    const double beta = (iter > 0 && rz_old != 0.0) ? -sc[F_ALPHA + prev] * zap / rz_old : 0.0;
    const double bb = iter == 0 ? rr : sc[F_BB];
    const bool bad = !(rz_new >= 0.0) || !(rr == rr);  // preconditioner not positive / NaN
    const bool converged = bb == 0.0 || rr <= sc[F_TOL2] * bb || bad;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[F_RZ + cur] = rz_new;
        sc[F_RR] = rr;
        if (iter == 0) sc[F_BB] = rr;
        if (bad) sc[F_FLAG] = 1.0;
        sc[F_CONV + cur] = converged ? 1.0 : 0.0;
        if (converged) sc[F_ITERS] = (double)iter;
    }
    if (converged) return;  // uniform over the grid: every workgroup reduces the same partials
    const bool first = iter == 0;
    auto pn = [&](int32_t j) { return first ? (double)z[j] : fma(beta, p_old[j], (double)z[j]); };
    double acc = 0.0;
    for (int64_t i = i0; i < A.n; i += stride) {
        double s, pi;
        if (W > 0 && i == i0) {
            s = ell_row_sum<W>(i, c0, v0, pn);
            pi = first ? (double)zfirst : fma(beta, pfirst, (double)zfirst);
        } else {
            s = ell_row_w<W>(A, A.val, i, 0, pn);
            pi = pn((int32_t)i);
        }
        p_new[i] = pi;
        Ap[i] = s;
        acc = fma(pi, s, acc);
    }
    acc = block_sum<TB>(acc);
    if (threadIdx.x == 0) part_pap[blockIdx.x] = acc;
}

// alpha = rz / pAp; x += alpha p; r -= alpha Ap; partials of |r|^2
__global__ __launch_bounds__(TB) void f_update(double *__restrict__ x, double *__restrict__ r,
                                               const double *__restrict__ p, const double *__restrict__ Ap,
                                               const double *__restrict__ part_pap, int nparts,
                                               const double *__restrict__ dinv, cyc_t *__restrict__ x0,
                                               double *__restrict__ part_rr, double *__restrict__ sc, int iter,
                                               int64_t n) {
    const int cur = iter & 1;
    // (operands of the thread's first rows requested before the scalars: see f_direction)
    constexpr int FU_PRE = 4;
    const int64_t i0 = (int64_t)xcd_block() * TB + threadIdx.x, stride = (int64_t)gridDim.x * TB;
    double xq[FU_PRE], rq[FU_PRE], pq[FU_PRE], aq[FU_PRE], dq[FU_PRE];
#pragma unroll
    for (int u = 0; u < FU_PRE; ++u) {
        const int64_t i = i0 + u * stride, ii = i < n ? i : n - 1;  // (unconditional loads from a row of the vectors)
        xq[u] = x[ii];
        rq[u] = r[ii];
        pq[u] = p[ii];
        aq[u] = Ap[ii];
        dq[u] = dinv[ii];
    }
    if (sc[F_CONV + cur] != 0.0) return;
    const double pap = reduce_partials(part_pap, nparts);
    const double rz = sc[F_RZ + cur];
    const bool bad = !(pap > 0.0) && rz != 0.0;  // breakdown (indefinite or singular G)
    const double alpha = (pap > 0.0) ? rz / pap : 0.0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[F_ALPHA + cur] = alpha;
        if (bad) sc[F_FLAG] = 1.0;
    }
    double srr = 0.0;
#pragma unroll
    for (int u = 0; u < FU_PRE; ++u) {
        const int64_t i = i0 + u * stride;
        if (i < n) {
            x[i] = fma(alpha, pq[u], xq[u]);
            const double ri = fma(-alpha, aq[u], rq[u]);
            r[i] = ri;
            x0[i] = (cyc_t)(OMEGA * dq[u] * ri);  // the next cycle's pre-smoothed iterate
            srr = fma(ri, ri, srr);
        }
    }
    for (int64_t i = i0 + FU_PRE * stride; i < n; i += stride) {
        x[i] = fma(alpha, p[i], x[i]);
        const double ri = fma(-alpha, Ap[i], r[i]);
        r[i] = ri;
        x0[i] = (cyc_t)(OMEGA * dinv[i] * ri);
        srr = fma(ri, ri, srr);
    }
    srr = block_sum(srr);
    if (threadIdx.x == 0) part_rr[blockIdx.x] = srr;
}

}  // namespace
