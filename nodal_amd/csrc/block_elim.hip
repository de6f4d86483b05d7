// Dense block elimination with explicitly inverted diagonal blocks: the passive (SPD) dense
// path and the presolved systems of dense_lu.hip's dispatcher.
#include <atomic>
#include <algorithm>
#include <cstdlib>
#include <vector>

#include "dense_common.h"

namespace {

constexpr int W = 256;             // default block width (K of the bulk update)
// ---------------------------------------------------------------------------------
// Passive networks (B == 0, every R > 0): G is symmetric positive definite, so block
// Gaussian elimination with EXPLICITLY INVERTED diagonal blocks needs no pivoting and
// is stable (the diagonal blocks of an SPD matrix are at least as well conditioned as
// the matrix).  Per 256-column block k:
//     Q    = A11^-1                      (small chain on the high-priority stream)
//     A12 <- Q A12                       (GEMM; includes the right-hand sides)
//     A22 <- A22 - A21 A12               (GEMM, the bulk of the flops)
// A21 is never touched: there is no tall-skinny panel factorisation, the critical
// path per block is one 256 x 256 inverse, and that runs while the previous block's
// big GEMM is still busy (its diagonal block is updated first).  Back substitution is
// x1 = A12[:, rhs] - A12[:, rest] x2, block by block: no triangular solves.
//
// Q for w = 256 comes from the 2 x 2 Schur-complement formula on 128 x 128 quadrants:
//     [A B]^-1   [A^-1 + T1 S^-1 T2   -T1 S^-1]     T1 = A^-1 B, T2 = C A^-1,
//     [C D]    = [-S^-1 T2             S^-1   ]     S  = D - C T1
// with the two 128 x 128 inverses by in-register Gauss-Jordan (one workgroup, the
// block distributed 4 x 4 per thread, pivot row / column broadcast through LDS).

constexpr int GJ = 128;
constexpr int BI_MAX = 512;        // largest block width of the block elimination

// dst (m x m, ldd) = inverse of src (m x m, lds_), m <= 128.  No pivoting.  A zero or
// NaN pivot records its 1-based global index in *dinfo (first one wins).
__global__ __launch_bounds__(1024) void gj128(const double *__restrict__ src, int64_t lds_, int m,
                                               double *__restrict__ dst, int64_t ldd,
                                               int32_t *__restrict__ dinfo, int base) {
    __shared__ double rowb[2][GJ], colb[2][GJ];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    double a[4][4];
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int r = ty + 32 * ii, c = tx + 32 * jj;
            a[ii][jj] = (r < m && c < m) ? src[(int64_t)c * lds_ + r] : (r == c ? 1.0 : 0.0);
        }
    // Step k: the owners of column k publish it (with a zero in row k) and clear their
    // copy; the owners of row k publish the scaled row (1/p in column k) and keep it as
    // the new row k.  After the barrier every element takes the SAME update
    // a -= col[i] * row[j]: rows other than k get a_ij - a_ik a_kj / p, column k gets
    // 0 - a_ik / p, and row k is left alone by its zero multiplier.
    // (The loop over k is split as k = 32 kb + kr with kb unrolled, so that the register
    // index of row / column k is a compile-time constant.)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        for (int kr = 0; kr < 32; ++kr) {
            const int k = 32 * kb + kr, buf = kr & 1;
            if (k >= m) break;
            const bool rowowner = ty == kr, colowner = tx == kr;
            double rv[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) rv[jj] = a[kb][jj];
            if (colowner) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    colb[buf][ty + 32 * ii] = (ii == kb && rowowner) ? 0.0 : a[ii][kb];
                    a[ii][kb] = 0.0;
                }
            }
            if (rowowner) {
                // the pivot sits in lane tx == kr of this half-wave, register rv[kb]
                const double p = __shfl(rv[kb], (int)(threadIdx.x & 32u) + kr, 64);
                if (tx == 0 && !(p != 0.0 && p == p) && *dinfo == 0) *dinfo = base + k + 1;
                double ip = __builtin_amdgcn_rcp(p);
                ip = fma(fma(-p, ip, 1.0), ip, ip);  // one Newton step: full double accuracy
                ip = fma(fma(-p, ip, 1.0), ip, ip);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    rv[jj] = (tx + 32 * jj) == k ? ip : rv[jj] * ip;
                    rowb[buf][tx + 32 * jj] = rv[jj];
                    a[kb][jj] = rv[jj];
                }
            }
            __syncthreads();
            double rr[4], ff[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) rr[jj] = rowb[buf][tx + 32 * jj];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) ff[ii] = colb[buf][ty + 32 * ii];
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) a[ii][jj] = fma(-ff[ii], rr[jj], a[ii][jj]);
        }
    }
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int r = ty + 32 * ii, c = tx + 32 * jj;
            if (r < m && c < m) dst[(int64_t)c * ldd + r] = a[ii][jj];
        }
}

// The same inverse with RANK-4 steps on the matrix cores (v_mfma_f64_4x4x4_4b_f64), 32
// block steps instead of 128 scalar ones.  The 128 x 128 block lives in accumulator
// registers in the MFMA result layout: wave w owns the 32 x 32 tile (tr = w & 3, tc = w >> 2),
// acc[mi][c8] lane l = element (32 tr + 16 mi + (l & 15), 32 tc + 4 c8 + (l >> 4)).
// Block step p, pivots k0 = 4p .. 4p+3 (block Gauss-Jordan, no pivoting):
//   1. the owners publish the four pivot rows and the four pivot columns (raw) through LDS
//      and clear their copy of the pivot columns;
//   2. one wave inverts the 4 x 4 pivot block P;
//   3. row panel  Rp = P^-1 [pivot rows]  with P^-1 itself in the pivot columns,
//      column panel Lp = -[pivot columns] with zeros in the pivot rows;
//   4. every wave: acc += Lp Rp (16 MFMAs; the cleared pivot columns become -L P^-1, the
//      pivot rows are untouched by their zero multipliers) and the row owners take Rp as
//      their new pivot rows.
constexpr int RP_S = 132;  // row panel stride (doubles): the 4 k of a fragment on disjoint banks
constexpr int CP_S = 144;  // column panel stride

__device__ __forceinline__ double rcp_f64(double p) {
    double ip = __builtin_amdgcn_rcp(p);
    ip = fma(fma(-p, ip, 1.0), ip, ip);
    return fma(fma(-p, ip, 1.0), ip, ip);
}
#include "gj16_wave.h"

__global__ __launch_bounds__(1024) void gj128_mfma(const double *__restrict__ src, int64_t lds_, int m,
                                                    double *__restrict__ dst, int64_t ldd,
                                                    int32_t *__restrict__ dinfo, int base) {
    __shared__ double rowraw[4][128], colraw[4][128], pinv[16];
    __shared__ double rowpan[4][RP_S], colpan[4][CP_S];
    // on the critical chain, usually sharing its CU with a workgroup of the bulk update: its
    // waves go first at instruction issue
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tr = wave & 3, tc = wave >> 2;
    const int lr = lane & 15, lc = lane >> 4, lq = lane & 3;
    double acc0[8], acc1[8];  // two separate arrays: a select between them cannot be turned
                              // into a dynamically indexed (scratch-resident) array access
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int r0 = 32 * tr + lr, r1 = r0 + 16, c = 32 * tc + 4 * c8 + lc;
        acc0[c8] = (r0 < m && c < m) ? src[(int64_t)c * lds_ + r0] : (r0 == c ? 1.0 : 0.0);
        acc1[c8] = (r1 < m && c < m) ? src[(int64_t)c * lds_ + r1] : (r1 == c ? 1.0 : 0.0);
    }
    const int nsteps = (m + 3) / 4;
    for (int p = 0; p < nsteps; ++p) {
        const int k0 = 4 * p, tp = p >> 3, mip = (p >> 2) & 1, ro = 4 * (p & 3), c8p = p & 7;
        // ---- 1. publish ----
        if (tr == tp && lr >= ro && lr < ro + 4) {
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8)
                rowraw[lr - ro][32 * tc + 4 * c8 + lc] = mip == 0 ? acc0[c8] : acc1[c8];
        }
        if (tc == tp) {  // uniform per wave; static register indices, uniform selects
            double v0 = acc0[0], v1 = acc1[0];
#pragma unroll
            for (int c8 = 1; c8 < 8; ++c8) {
                v0 = c8 == c8p ? acc0[c8] : v0;
                v1 = c8 == c8p ? acc1[c8] : v1;
            }
            colraw[lc][32 * tr + lr] = v0;
            colraw[lc][32 * tr + 16 + lr] = v1;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                acc0[c8] = c8 == c8p ? 0.0 : acc0[c8];
                acc1[c8] = c8 == c8p ? 0.0 : acc1[c8];
            }
        }
        __syncthreads();
        // ---- 2. invert the pivot block (wave 0, every lane redundantly) ----
        if (wave == 0) {
            double a[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) a[i][j] = rowraw[i][k0 + j];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double pv = a[k][k];
                if (lane == 0 && !(pv != 0.0 && pv == pv) && *dinfo == 0) *dinfo = base + k0 + k + 1;
                const double ip = rcp_f64(pv);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[k][j] = j == k ? ip : a[k][j] * ip;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i == k) continue;
                    const double f = a[i][k];
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[i][j] = j == k ? -f * ip : fma(-f, a[k][j], a[i][j]);
                }
            }
            if (lane < 16) {
                double v = 0.0;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (lane == 4 * i + j) v = a[i][j];
                pinv[lane] = v;
            }
        }
        __syncthreads();
        // ---- 3. panels ----
        {
            const int t = threadIdx.x & 511, k = t >> 7, x = t & 127;
            if (threadIdx.x < 512) {
                double v;
                if (x >= k0 && x < k0 + 4) v = pinv[4 * k + (x - k0)];
                else {
                    v = 0.0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v = fma(pinv[4 * k + j], rowraw[j][x], v);
                }
                rowpan[k][x] = v;
            } else {
                colpan[k][x] = (x >= k0 && x < k0 + 4) ? 0.0 : -colraw[k][x];
            }
        }
        __syncthreads();
        // ---- 4. rank-4 update ----
        {
            double bf[2], af[8];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) bf[mi] = colpan[lc][32 * tr + 16 * mi + lr];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) af[c8] = rowpan[lc][32 * tc + 4 * c8 + lq];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                acc0[c8] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[c8], bf[0], acc0[c8], 0, 0, 0);
                acc1[c8] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[c8], bf[1], acc1[c8], 0, 0, 0);
            }
            if (tr == tp && lr >= ro && lr < ro + 4) {
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const double v = rowpan[lr - ro][32 * tc + 4 * c8 + lc];
                    acc0[c8] = mip == 0 ? v : acc0[c8];
                    acc1[c8] = mip == 1 ? v : acc1[c8];
                }
            }
        }
        // (the next step's publish writes rowraw / colraw only; the panels are rewritten
        // after two more barriers)
    }
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int r0 = 32 * tr + lr, r1 = r0 + 16, c = 32 * tc + 4 * c8 + lc;
        if (r0 < m && c < m) dst[(int64_t)c * ldd + r0] = acc0[c8];
        if (r1 < m && c < m) dst[(int64_t)c * ldd + r1] = acc1[c8];
    }
}

// Two-level variant: RANK-16 outer steps (8 instead of 32 block steps, 2 barriers each instead of
// 4 x 3).  Outer step q, pivots k0 = 16 q .. 16 q + 15 = the 16-row half `mip` of tile `tp`:
//   1. the owners publish the 16 pivot rows (raw) and the 16 pivot columns, already negated and with
//      zeros in the pivot rows (= the column panel Lp), and clear their copy of the pivot columns;
//   2. row panel Rp = P^-1 [pivot rows] on the matrix cores (every wave eight columns), with P^-1
//      itself in the pivot columns;
//   3. every wave: acc += Lp Rp as four rank-4 MFMA sweeps; the row owners take Rp as their rows;
//   4. the wave that owns the NEXT pivot block inverts it right after its own update, in registers:
//      lane (r, g) holds P[r][4 j + g], sixteen Gauss-Jordan steps whose pivot row / column travel
//      by wave shuffles -- no LDS traffic, no barrier, and hidden behind the other waves' updates
//      (the sixteen dependent reciprocals are 2 us per outer step on the critical path otherwise).
// The column panel is double-buffered (step q+1 publishes it while slower waves still read step q's).
constexpr int GJ16_LDS = (16 * 128 + 16 * RP_S + 2 * 16 * CP_S + 16 * 17) * 8;

// In-wave inverse of the 16 x 16 pivot block: gj16_wave.h (two forms with the same arithmetic: lane exchanges by
// ds_bpermute, or -- DPP -- by v_readlane and DPP row broadcasts: 4 instead of 28 trips through the LDS crossbar per
// 2 x 2 step on the critical path of an outer step; NODAL_GJ_DPP=0 selects the former).
template <bool DPP>
__global__ __launch_bounds__(1024) void gj128_mfma16(const double *__restrict__ src, int64_t lds_, int m,
                                                      double *__restrict__ dst, int64_t ldd,
                                                      int32_t *__restrict__ dinfo, int base) {
    extern __shared__ __attribute__((aligned(16))) double gj16_smem[];
    double (*rowraw)[128] = reinterpret_cast<double (*)[128]>(gj16_smem);
    double (*rowpan)[RP_S] = reinterpret_cast<double (*)[RP_S]>(gj16_smem + 16 * 128);
    double (*colpan2)[CP_S] = reinterpret_cast<double (*)[CP_S]>(gj16_smem + 16 * 128 + 16 * RP_S);
    double (*pinv)[17] = reinterpret_cast<double (*)[17]>(gj16_smem + 16 * 128 + 16 * RP_S + 2 * 16 * CP_S);
    __builtin_amdgcn_s_setprio(3);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tr = wave & 3, tc = wave >> 2;
    const int lr = lane & 15, lc = lane >> 4, lq = lane & 3;
    double acc0[8], acc1[8];
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int r0 = 32 * tr + lr, r1 = r0 + 16, c = 32 * tc + 4 * c8 + lc;
        acc0[c8] = (r0 < m && c < m) ? src[(int64_t)c * lds_ + r0] : (r0 == c ? 1.0 : 0.0);
        acc1[c8] = (r1 < m && c < m) ? src[(int64_t)c * lds_ + r1] : (r1 == c ? 1.0 : 0.0);
    }
    if (wave == 0) {  // the first pivot block
        double a[4] = {acc0[0], acc0[1], acc0[2], acc0[3]};
        if (DPP) gj16_in_wave(a, lane, dinfo, base);
        else gj16_in_wave_bperm(a, lane, dinfo, base);
#pragma unroll
        for (int j = 0; j < 4; ++j) pinv[lr][4 * j + lc] = a[j];
    }
    const int nouter = (m + 15) / 16;
    for (int q = 0; q < nouter; ++q) {
        const int k0 = 16 * q, tp = q >> 1, mip = q & 1;
        double (*colpan)[CP_S] = colpan2 + 16 * (q & 1);
        // ---- 1. publish ----
        if (tr == tp) {
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) rowraw[lr][32 * tc + 4 * c8 + lc] = mip == 0 ? acc0[c8] : acc1[c8];
        }
        if (tc == tp) {  // uniform per wave; the pivot columns are c8 = 4 mip + j
            // (two branches with static register indices: a select `mip ? acc0[4 + j] : acc0[j]` is
            // turned into a dynamically indexed, scratch-resident array)
            const bool p0 = tr == tp && mip == 0, p1 = tr == tp && mip == 1;  // my rows ARE the pivot rows
            if (mip == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    colpan[4 * j + lc][32 * tr + lr] = p0 ? 0.0 : -acc0[j];
                    colpan[4 * j + lc][32 * tr + 16 + lr] = p1 ? 0.0 : -acc1[j];
                    acc0[j] = 0.0;
                    acc1[j] = 0.0;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    colpan[4 * j + lc][32 * tr + lr] = p0 ? 0.0 : -acc0[4 + j];
                    colpan[4 * j + lc][32 * tr + 16 + lr] = p1 ? 0.0 : -acc1[4 + j];
                    acc0[4 + j] = 0.0;
                    acc1[4 + j] = 0.0;
                }
            }
        }
        __syncthreads();
        // ---- 2. row panel Rp = P^-1 [pivot rows]: wave w computes columns 8 w .. 8 w + 7 ----
        {
            double d0 = 0.0, d1 = 0.0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const double bf = pinv[lr][4 * ks + lc];
                const double a0 = rowraw[4 * ks + lc][8 * wave + lq], a1 = rowraw[4 * ks + lc][8 * wave + 4 + lq];
                d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, bf, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, bf, d1, 0, 0, 0);
            }
            const int x0 = 8 * wave + lc, x1 = x0 + 4;
            rowpan[lr][x0] = (x0 >= k0 && x0 < k0 + 16) ? pinv[lr][x0 - k0] : d0;
            rowpan[lr][x1] = (x1 >= k0 && x1 < k0 + 16) ? pinv[lr][x1 - k0] : d1;
        }
        __syncthreads();
        // ---- 3. rank-16 update ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            double bf[2], af[8];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) bf[mi] = colpan[4 * ks + lc][32 * tr + 16 * mi + lr];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) af[c8] = rowpan[4 * ks + lc][32 * tc + 4 * c8 + lq];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                acc0[c8] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[c8], bf[0], acc0[c8], 0, 0, 0);
                acc1[c8] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[c8], bf[1], acc1[c8], 0, 0, 0);
            }
        }
        if (tr == tp) {
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const double v = rowpan[lr][32 * tc + 4 * c8 + lc];
                acc0[c8] = mip == 0 ? v : acc0[c8];
                acc1[c8] = mip == 1 ? v : acc1[c8];
            }
        }
        // ---- 4. the next pivot block's inverse, by its owner (its tile is up to date now) ----
        if (q + 1 < nouter && tr == ((q + 1) >> 1) && tc == tr) {
            double a[4];
            if (((q + 1) & 1) == 0) { a[0] = acc0[0]; a[1] = acc0[1]; a[2] = acc0[2]; a[3] = acc0[3]; }
            else { a[0] = acc1[4]; a[1] = acc1[5]; a[2] = acc1[6]; a[3] = acc1[7]; }
            if (DPP) gj16_in_wave(a, lane, dinfo, base + k0 + 16);
            else gj16_in_wave_bperm(a, lane, dinfo, base + k0 + 16);
            // (pinv of this step was last read before the barrier above)
#pragma unroll
            for (int j = 0; j < 4; ++j) pinv[lr][4 * j + lc] = a[j];
        }
        // (the next publish writes rowraw and the OTHER column panel; rowpan is rewritten after a barrier
        // that every wave reaches only after this step's reads of it)
    }
#pragma unroll
    for (int c8 = 0; c8 < 8; ++c8) {
        const int r0 = 32 * tr + lr, r1 = r0 + 16, c = 32 * tc + 4 * c8 + lc;
        if (r0 < m && c < m) dst[(int64_t)c * ldd + r0] = acc0[c8];
        if (r1 < m && c < m) dst[(int64_t)c * ldd + r1] = acc1[c8];
    }
}

// dst (rows x cols, ldd) = src (rows x cols, lds_): four columns per workgroup pass,
// whole lines moved
__global__ __launch_bounds__(256) void copy_block(const double *__restrict__ src, int64_t lds_,
                                                  double *__restrict__ dst, int64_t ldd, int rows,
                                                  int64_t cols) {
    for (int64_t c = (int64_t)blockIdx.x * 4; c < cols; c += (int64_t)gridDim.x * 4)
        for (int r = threadIdx.x; r < rows; r += 256) {
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = c + u < cols ? src[(c + u) * lds_ + r] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c + u < cols) dst[(c + u) * ldd + r] = v[u];
        }
}

// dst (cols x rows, ldd) = src (rows x cols, lds_)^T, both at most BI_MAX wide: 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_block(const double *__restrict__ src, int64_t lds_,
                                                       double *__restrict__ dst, int64_t ldd, int rows, int cols) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + tx, c = c0 + ty + 8 * k;
        tile[ty + 8 * k][tx] = (r < rows && c < cols) ? src[(int64_t)c * lds_ + r] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + tx, r = r0 + ty + 8 * k;  // dst row index = src column
        if (c < cols && r < rows) dst[(int64_t)r * ldd + c] = tile[tx][ty + 8 * k];
    }
}

// max |A_ij - A_ji| and max |A_ij| over the n x n leading block (NODAL_TRACE: is the matrix the symmetric
// elimination is about to take really symmetric?)
__global__ __launch_bounds__(256) void asymmetry(const double *__restrict__ A, int64_t lda, int64_t n,
                                                 unsigned long long *__restrict__ out) {
    double d = 0.0, m = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n * n; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e % n, j = e / n;
        const double a = A[j * lda + i], b = A[i * lda + j];
        d = fmax(d, fabs(a - b));
        m = fmax(m, fabs(a));
    }
    atomicMax(&out[0], (unsigned long long)__double_as_longlong(d));
    atomicMax(&out[1], (unsigned long long)__double_as_longlong(m));
}

// Back substitution for the block-inverse form: x[j0:j1] = y[j0:j1] is final; the rows
// above lose A[0:j0, j0:j1] x[j0:j1].  A workgroup owns 64 rows (one per lane); its four
// waves split the block's columns, 16 independent loads in flight each, and meet in
// LDS.  blockIdx.y = right-hand side.
__global__ __launch_bounds__(256) void bs_block(const double *__restrict__ A, int64_t lda,
                                                double *__restrict__ y, double *__restrict__ xout,
                                                int64_t ldx, int j0, int j1) {
    __shared__ double xs[BI_MAX];
    __shared__ double part[3][64];
    y += (int64_t)blockIdx.y * lda;
    xout += (int64_t)blockIdx.y * ldx;
    const int w = j1 - j0, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < w; t += 256) xs[t] = y[j0 + t];
    __syncthreads();
    if (blockIdx.x == 0)
        for (int t = threadIdx.x; t < w; t += 256) xout[j0 + t] = xs[t];
    const int per = (w + 3) / 4, s0 = wave * per, s1 = s0 + per < w ? s0 + per : w;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < j0; base += (int64_t)gridDim.x * 64) {
        const int64_t i = base + lane;
        double acc = 0.0;
        if (i < j0) {
            const double *col = A + (int64_t)j0 * lda + i;
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int s = s0;
            for (; s + 16 <= s1; s += 16) {
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = col[(int64_t)(s + u) * lda];
#pragma unroll
                for (int u = 0; u < 16; u += 4) {
                    a0 = fma(v[u + 0], xs[s + u + 0], a0);
                    a1 = fma(v[u + 1], xs[s + u + 1], a1);
                    a2 = fma(v[u + 2], xs[s + u + 2], a2);
                    a3 = fma(v[u + 3], xs[s + u + 3], a3);
                }
            }
            for (; s < s1; ++s) a0 = fma(col[(int64_t)s * lda], xs[s], a0);
            acc = (a0 + a1) + (a2 + a3);
        }
        if (wave > 0) part[wave - 1][lane] = acc;
        __syncthreads();
        if (wave == 0 && i < j0) y[i] -= (acc + part[0][lane]) + (part[1][lane] + part[2][lane]);
        __syncthreads();
    }
}

// Q (ld ldq) = inverse of the w x w block at D (ld lda), w <= BI_MAX; D itself is overwritten.
// w <= 128: one Gauss-Jordan workgroup; otherwise the 2 x 2 Schur-complement formula with the
// leading m1 = 128 (w <= 256) or 256 columns, recursively.  `scratch` holds 2 * m1 * m2 doubles
// per recursion level (T1, T2).
int invert_diag(nodal_ctx *h, hipStream_t sp, double *D, int64_t lda, int w, double *Q, int64_t ldq,
                double *scratch, int32_t *dinfo, int base) {
    if (w <= GJ) {
        if (h->gj_scalar == 1) {
            gj128<<<1, 1024, 0, sp>>>(D, lda, w, Q, ldq, dinfo, base);
        } else if (h->gj_scalar == 2) {
            gj128_mfma<<<1, 1024, 0, sp>>>(D, lda, w, Q, ldq, dinfo, base);
        } else {
            static std::atomic<bool> lds_allowed[64];  // per device: the attribute belongs to the device's code object
            const int dev = h->device >= 0 && h->device < 64 ? h->device : 0;
            if (!lds_allowed[dev].load(std::memory_order_acquire)) {
                NODAL_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(gj128_mfma16<true>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, GJ16_LDS));
                NODAL_HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void *>(gj128_mfma16<false>),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, GJ16_LDS));
                lds_allowed[dev].store(true, std::memory_order_release);
            }
            static const bool gj_dpp = !(getenv("NODAL_GJ_DPP") && atoi(getenv("NODAL_GJ_DPP")) == 0);
            if (gj_dpp) gj128_mfma16<true><<<1, 1024, GJ16_LDS, sp>>>(D, lda, w, Q, ldq, dinfo, base);
            else gj128_mfma16<false><<<1, 1024, GJ16_LDS, sp>>>(D, lda, w, Q, ldq, dinfo, base);
        }
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    }
    const int m1 = w <= 2 * GJ ? GJ : 2 * GJ, m2 = w - m1;
    double *T1 = scratch, *T2 = scratch + (size_t)m1 * m2, *deeper = T2 + (size_t)m1 * m2;
    double *Bq = D + (int64_t)m1 * lda, *Cq = D + m1, *Dq = D + (int64_t)m1 * lda + m1;
    double *Q11 = Q, *Q12 = Q + (int64_t)m1 * ldq, *Q21 = Q + m1, *Q22 = Q + (int64_t)m1 * ldq + m1;
    NODAL_TRY(invert_diag(h, sp, D, lda, m1, Q11, ldq, deeper, dinfo, base));
    NODAL_TRY(gemm_pair_f64(h, sp, GEMM_SET, GemmProblem{T1, m1, Q11, ldq, Bq, lda, m1, m2, m1},    // T1 = A^-1 B
                            GemmProblem{T2, m2, Cq, lda, Q11, ldq, m2, m1, m1}));                  // T2 = C A^-1
    NODAL_TRY(gemm_f64(h, sp, GEMM_SUB, Dq, lda, Cq, lda, T1, m1, m2, m2, m1));                    // S = D - C T1
    NODAL_TRY(invert_diag(h, sp, Dq, lda, m2, Q22, ldq, deeper, dinfo, base + m1));                // S^-1
    NODAL_TRY(gemm_pair_f64(h, sp, GEMM_SETNEG, GemmProblem{Q12, ldq, T1, m1, Q22, ldq, m1, m2, m2},  // -T1 S^-1
                            GemmProblem{Q21, ldq, Q22, ldq, T2, m2, m2, m1, m2}));                    // -S^-1 T2
    NODAL_TRY(gemm_f64(h, sp, GEMM_SUB, Q11, ldq, Q12, ldq, T2, m2, m1, m1, m2));                  // + T1 S^-1 T2
    return NODAL_OK;
}

// bnd: block boundaries 0 = bnd[0] < bnd[1] < ... < bnd[nb] = n (widths <= BI_MAX, possibly mixed)
// sym: the matrix is symmetric (a passive network) -- only its upper block triangle is updated.  The
// block column below the diagonal block k is the transpose of the block row V(k) = A12(k) right of it
// (as it stood before A12 <- Q A12), which is kept in a scratch panel: the bulk update
// A22 -= V^T W touches the upper tiles only (half the flops, both operands read along k), the small
// products of the chain take the 256 x 256 head of V^T from an explicit transpose.
int factor_blockinv(nodal_ctx *h, double *A, int64_t n, int64_t lda, int64_t ncols, int32_t *dinfo,
                    GemmTimer &tm, const std::vector<int64_t> &bnd, bool sym) {
    // Three streams.  Per block k = [J0, J1), next block [J1, J2), once A12(k) <- Q(k) A12(k)
    // (called W(k) below) is done and the previous bulk update has retired:
    //   sp (high priority): diag   A[J1:J2, J1:J2] -= A[J1:J2, J0:J1] W(k)       (small)
    //                       Q(k+1) = inv(A[J1:J2, J1:J2])                         (chain)
    //                       W(k+1), first FIRST columns  (after the strip)
    //   s3:                 strip  A[J1:J2, J2:]   -= A[J1:J2, J0:J1] W(k)       (block row k+1)
    //                       W(k+1), remaining columns    (after Q(k+1))
    //   sg:                 rest   A[J2:,   J1:]   -= A[J2:,   J0:J1] W(k)       (the bulk)
    // The critical path per block is diag -> inverse -> first columns of W -> next diag; the
    // strip, the wide part of W and the bulk update run beside it.
    // Bulk updates on a CU-masked stream that leaves 32 CUs (NODAL_PANEL_CUS) to the chain's small kernels:
    // since the symmetric form halved the bulk work the chain bounds 30 of config 2's 39 blocks, and its
    // workgroups no longer queue behind 58-us GEMM workgroups (14.37 -> 13.92 ms; 8 / 16 / 64 CUs: 14.15 /
    // 14.12 / 14.69).  NODAL_BI_MASKED=0: bulk updates on all CUs (the default until the symmetric form).
    static const bool full_mask = getenv("NODAL_BI_MASKED") != nullptr && atoi(getenv("NODAL_BI_MASKED")) == 0;
    // NODAL_BI_UNMASK_ROWS=r (round 5, measured, off): the bulk update on ALL compute units while more than r rows are
    // left, on the masked stream afterwards (the two side streams swap roles at that block).  The idea -- the bulk update
    // bounds the first two thirds of the solve and the mask costs it an eighth of the chip -- did not survive the
    // measurement: config 2 13.65 ms without, 14.28 / 13.98 / 13.70 / 13.85 ms for r = 4000 / 5632 / 7000 / 8500: the chain's
    // small kernels queueing behind 58-us GEMM workgroups cost more than the 32 CUs give, in every phase.
    static const int64_t unmask_rows = getenv("NODAL_BI_UNMASK_ROWS") ? atoll(getenv("NODAL_BI_UNMASK_ROWS")) : 0;
    const bool adaptive = !full_mask && unmask_rows > 0 && unmask_rows < n;
    hipStream_t sp = h->stream, sg = (full_mask || adaptive) ? h->stream3 : h->stream2;
    hipStream_t s3 = (full_mask || adaptive) ? h->stream2 : h->stream3;
    bool swapped = false;
    hipEvent_t ev_wfirst = h->ev_bi[0], ev_strip = h->ev_bi[1], ev_wrest = h->ev_bi[2],
               ev_start = h->ev_bi[3], ev_done = h->ev_bi[4], ev_q = h->ev_bi[5], ev_rest = nullptr;
    const int nb = (int)bnd.size() - 1;
    int64_t wmax = 0;
    for (int k = 0; k < nb; ++k) wmax = std::max(wmax, bnd[k + 1] - bnd[k]);
    const int64_t FIRST = 2 * wmax;  // (capacity) columns of W(k) that the next two diagonal blocks need
    // scratch: Q[2] (wmax x wmax), the inverse's T1 / T2 per recursion level, S1 (wmax x FIRST), S (wmax x ncols);
    // sym: two panels V[2] (wmax x ncols: block k's bulk update reads V(k) while the chain fills V(k + 1))
    // and two transposed heads Lt[2] (wmax x wmax) instead of S1 / S
    const size_t qb = (size_t)wmax * wmax, tb = 2 * (size_t)(2 * GJ) * (2 * GJ) + 2 * (size_t)GJ * GJ;
    const size_t panel = (size_t)wmax * (size_t)ncols;
    NODAL_HIP_TRY(h, h->work.reserve((2 * qb + tb + (sym ? 2 * panel + 2 * qb : (size_t)wmax * FIRST + panel)) * 8 + 256));
    double *Q[2] = {h->work.as<double>(), h->work.as<double>() + qb};
    double *T = Q[1] + qb, *S1 = T + tb, *S = S1 + (size_t)wmax * FIRST;
    double *V[2] = {T + tb, T + tb + panel};
    double *Lt[2] = {V[1] + panel, V[1] + panel + qb};

    // A12 <- Q A12 for the block [J0, J1): columns [c0, c1) on stream st through scratch buf
    // (sym: buf is the block's panel V, column j of it <-> global column J1 + j, and it is kept)
    // copied: the columns are in buf already (round 5: the first columns of the next block are copied, and their head
    // transposed, on the strip's stream as soon as the strip is done -- off the chain)
    auto scale_cols = [&](hipStream_t st, const double *Qk, double *buf, int64_t J0, int64_t J1, int64_t c0,
                          int64_t c1, bool copied = false) -> int {
        if (c1 <= c0) return NODAL_OK;
        const int w = (int)(J1 - J0);
        if (sym) buf += (c0 - J1) * wmax;
        if (!copied) {
            copy_block<<<blocks_for(c1 - c0, 4), 256, 0, st>>>(A + c0 * lda + J0, lda, buf, wmax, w, c1 - c0);
            NODAL_HIP_TRY(h, hipGetLastError());
        }
        return gemm_f64(h, st, GEMM_SET, A + c0 * lda + J0, lda, Qk, wmax, buf, wmax, w, c1 - c0, w);
    };
    auto copy_cols = [&](hipStream_t st, double *buf, int64_t J0, int64_t J1, int64_t c0, int64_t c1) -> int {
        if (c1 <= c0) return NODAL_OK;
        copy_block<<<blocks_for(c1 - c0, 4), 256, 0, st>>>(A + c0 * lda + J0, lda, buf + (c0 - J1) * wmax, wmax, (int)(J1 - J0), c1 - c0);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    };
    static const bool early_copy_env = !(getenv("NODAL_BI_EARLY_COPY") && atoi(getenv("NODAL_BI_EARLY_COPY")) == 0);
    const bool early_copy = sym && early_copy_env;
    // sym: Lt(k) = (first w_next columns of V(k))^T, the rows J1:J2 of the never-formed block column k
    auto head_transpose = [&](hipStream_t st, int k) -> int {
        if (!sym || k + 2 > nb) return NODAL_OK;
        const int w = (int)(bnd[k + 1] - bnd[k]), wn = (int)(bnd[k + 2] - bnd[k + 1]);
        transpose_block<<<dim3((unsigned)((w + 31) / 32), (unsigned)((wn + 31) / 32)), 256, 0, st>>>(
            V[k & 1], wmax, Lt[k & 1], wmax, w, wn);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    };
    // W(k)'s first columns: those the next two diagonal blocks (k + 1, k + 2) read
    static const int first_blocks = getenv("NODAL_BI_FIRST_BLOCKS") ? atoi(getenv("NODAL_BI_FIRST_BLOCKS")) : 2;
    auto first_end = [&](int k) {
        const int to = k + 1 + (first_blocks < 1 ? 1 : (first_blocks > 2 ? 2 : first_blocks));
        const int64_t e = bnd[to < nb ? to : nb];
        return e < ncols ? e : ncols;
    };

    StreamJoinGuard join(sp);  // (a failed call below must not leave the bulk / strip streams running unjoined)
    join.add(sg, ev_done);
    join.add(s3, ev_strip);
    NODAL_HIP_TRY(h, hipEventRecord(ev_start, sp));  // the matrix was prepared on the main stream
    NODAL_HIP_TRY(h, hipStreamWaitEvent(sg, ev_start, 0));
    NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_start, 0));
    {
        const int64_t J1 = bnd[1];
        NODAL_TRY(invert_diag(h, sp, A, lda, (int)J1, Q[0], wmax, T, dinfo, 0));
        NODAL_HIP_TRY(h, hipEventRecord(ev_q, sp));
        NODAL_TRY(scale_cols(sp, Q[0], sym ? V[0] : S1, 0, J1, J1, first_end(0)));
        NODAL_TRY(head_transpose(sp, 0));
        NODAL_HIP_TRY(h, hipEventRecord(ev_wfirst, sp));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_q, 0));
        NODAL_TRY(scale_cols(s3, Q[0], sym ? V[0] : S, 0, J1, first_end(0), ncols));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_wfirst, 0));  // ev_wrest then stands for ALL of W(k):
        NODAL_HIP_TRY(h, hipEventRecord(ev_wrest, s3));          // one barrier packet less in front of the bulk GEMM
    }
    for (int blk = 0; blk + 1 < nb; ++blk) {
        const int64_t J0 = bnd[blk], J1 = bnd[blk + 1], J2 = bnd[blk + 2];
        const int w = (int)(J1 - J0);
        const double *L = A + J0 * lda, *U = A + J1 * lda + J0;  // A[:, J0:J1] and W(k)
        // rows J1:J2 of the block column: in place, or (sym) the transposed head of V(k)
        const double *Lhead = sym ? Lt[blk & 1] : L + J1;
        const int64_t ldh = sym ? wmax : lda;
        double *Qn = Q[(blk + 1) & 1];
        if (adaptive && !swapped && n - J2 <= unmask_rows) {  // from here on the bulk update leaves CUs to the chain
            std::swap(sg, s3);
            swapped = true;
            if (ev_rest) NODAL_HIP_TRY(h, hipStreamWaitEvent(sg, ev_rest, 0));  // (bulk updates stay in order)
            // (the new strip stream did not scale W(k)'s wide part, nor transpose its head: ev_wrest lies behind both)
            NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_wrest, 0));
        }
        if (ev_rest) {  // block row k+1 was last written by the previous bulk update
            NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_rest, 0));
            NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_rest, 0));
        }
        NODAL_HIP_TRY(h, hipStreamWaitEvent(sg, ev_wrest, 0));  // all of W(k) (s3 waited for the first columns)
        // sp: diag + inverse chain
        NODAL_TRY(gemm_sub_f64(h, sp, A + J1 * lda + J1, lda, Lhead, ldh, U, lda, J2 - J1, J2 - J1, w));
        NODAL_TRY(invert_diag(h, sp, A + J1 * lda + J1, lda, (int)(J2 - J1), Qn, wmax, T, dinfo, (int)J1));
        NODAL_HIP_TRY(h, hipEventRecord(ev_q, sp));
        // s3: strip
        NODAL_TRY(gemm_sub_f64(h, s3, A + J2 * lda + J1, lda, Lhead, ldh, U + (J2 - J1) * lda, lda,
                               J2 - J1, ncols - J2, w));
        if (early_copy) {  // V(k+1)'s first columns and its transposed head need the strip only: not the chain's business
            NODAL_TRY(copy_cols(s3, V[(blk + 1) & 1], J1, J2, J2, first_end(blk + 1)));
            NODAL_TRY(head_transpose(s3, blk + 1));
        }
        NODAL_HIP_TRY(h, hipEventRecord(ev_strip, s3));
        // sg: rest
        if (J2 < n) {
            NODAL_TRY(tm.begin(sg));
            if (sym) {  // upper tiles of A[J2:, J2:] -= V(k)[:, J2:]^T W(k)[:, J2:]
                // the later diagonal blocks must come out whole: they start at multiples of the widest
                // remaining block width from J2 (512-wide blocks come first, then 256-wide ones)
                int64_t band = 128;
                for (int q = blk + 2; q < nb; ++q) band = std::max(band, (bnd[q + 1] - bnd[q] + 127) / 128 * 128);
                NODAL_TRY(gemm_sub_tn_upper_f64(h, sg, A + J2 * lda + J2, lda, V[blk & 1] + (J2 - J1) * wmax, wmax,
                                                U + (J2 - J1) * lda, lda, n - J2, ncols - J2, w, (int)band));
                // (the diagonal blocks are computed whole)
                const double m = (double)(n - J2);
                NODAL_TRY(tm.end(sg, 2.0 * (double)w * (m * (double)(ncols - J2) - 0.5 * m * std::max(0.0, m - (double)band))));
            } else {
                NODAL_TRY(gemm_sub_f64(h, sg, A + J1 * lda + J2, lda, L + J2, lda, U, lda, n - J2, ncols - J1, w));
                NODAL_TRY(tm.end(sg, 2.0 * (double)w * (double)(n - J2) * (double)(ncols - J1)));
            }
            ev_rest = tm.last_end();  // the timing event doubles as the dependency (one packet less;
                                      // dropping the timing events altogether was measured: no change)
        } else ev_rest = nullptr;
        // W(k+1): the first columns on the critical stream, the wide remainder beside it
        NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_strip, 0));
        NODAL_TRY(scale_cols(sp, Qn, sym ? V[(blk + 1) & 1] : S1, J1, J2, J2, first_end(blk + 1), early_copy));
        if (!early_copy) NODAL_TRY(head_transpose(sp, blk + 1));
        NODAL_HIP_TRY(h, hipEventRecord(ev_wfirst, sp));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_q, 0));
        NODAL_TRY(scale_cols(s3, Qn, sym ? V[(blk + 1) & 1] : S, J1, J2, first_end(blk + 1), ncols));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_wfirst, 0));
        NODAL_HIP_TRY(h, hipEventRecord(ev_wrest, s3));
    }
    NODAL_HIP_TRY(h, hipEventRecord(ev_done, sg));
    NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_done, 0));
    NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_wrest, 0));
    join.disarm();
    return NODAL_OK;
}

}  // namespace

// Factor-and-solve by block elimination: A is the column-major augmented matrix (n + nrhs
// columns, leading dimension lda); the solutions go to xout (column c at xout + c * ldx).
int dense_block_elimination(nodal_ctx *h, double *A, int64_t n, int64_t lda, int32_t nrhs, double *xout,
                            int64_t ldx, int32_t *dinfo) {
    NODAL_TRY(nodal_ensure_aux_streams(h));
    const int64_t ncols = n + nrhs;
    hipStream_t st = h->stream;
    GemmTimer tm{h};  // (reset and collected by the caller, dense_factor_solve_multi)
    // Block widths.  While the bulk update of a block is long (many rows left) K = 512 updates run at
    // 43 instead of 38 TFLOP/s (two K = 256 launches read and write C twice); narrow blocks keep
    // the inverse chain short where it is the longer of the two.  NODAL_BI_WIDTH=256 / 512 forces
    // one width; NODAL_BI_SWITCH the number of rows left at which the width drops to 256 (5632 until round 5's third
    // session; re-scanned after the chain of a 256-block went from 245 to 198 us: 3584 / 4096 / 4608 / 5120 / 5632 / 6656 /
    // 7680 rows: 12.91 / 12.89 / 12.83 / 13.00 / 13.02 / 13.18 / 13.66 ms for config 2 on one box; grid(80) .. grid(140) 0.3-2 % faster).
    std::vector<int64_t> bnd;
    {
        int64_t sw = 4608;
        if (const char *e = getenv("NODAL_BI_SWITCH")) sw = atoll(e);
        // below sw2 rows left the chain of small kernels is far longer than the bulk update: 128-wide blocks need
        // ONE Gauss-Jordan inversion each instead of the 2 x 2 Schur formula's two inversions + four products
        int64_t sw2 = 0;
        if (const char *e = getenv("NODAL_BI_SWITCH2")) sw2 = atoll(e);
        int forced = 0;
        if (const char *e = getenv("NODAL_BI_WIDTH")) {
            const int v = atoi(e);
            forced = v == 512 ? 512 : (v == 128 ? 128 : 256);
        }
        bnd.push_back(0);
        for (int64_t j = 0; j < n;) {
            const int64_t left = n - j;
            const int64_t wb = forced ? forced : (left > sw + 512 ? 512 : (left <= sw2 ? 128 : W));
            j = j + wb < n ? j + wb : n;
            bnd.push_back(j);
        }
    }
    // a passive network's matrix is symmetric bit for bit (every off-diagonal pair is the same sum of the
    // same -1/R terms in the same order); a presolved system with transconductance stamps is not
    static const bool sym_env = !(getenv("NODAL_BI_SYM") && atoi(getenv("NODAL_BI_SYM")) == 0);
    const bool sym = sym_env && h->passive_network && !h->optimistic_nopivot;
    if (sym && getenv("NODAL_TRACE")) {
        NODAL_HIP_TRY(h, h->work3.reserve(256));
        unsigned long long *o = h->work3.as<unsigned long long>();
        NODAL_HIP_TRY(h, hipMemsetAsync(o, 0, 16, st));
        asymmetry<<<1024, 256, 0, st>>>(A, lda, n, o);
        double host[2];
        NODAL_HIP_TRY(h, hipMemcpyAsync(host, o, 16, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        fprintf(stderr, "[dense] symmetric block elimination, n %lld: max |a_ij - a_ji| = %.3e, max |a_ij| = %.3e\n",
                (long long)n, host[0], host[1]);
    }
    NODAL_TRY(factor_blockinv(h, A, n, lda, ncols, dinfo, tm, bnd, sym));
    double *y = A + n * lda;
    for (int k = (int)bnd.size() - 2; k >= 0; --k) {
        const int64_t j0 = bnd[k], j1 = bnd[k + 1];
        dim3 grid(blocks_for(j0 > 0 ? j0 : 1, 64), (unsigned)nrhs);
        if (grid.x > 256 && nrhs > 1) grid.x = 256;  // many columns: fewer workgroups per column
        bs_block<<<grid, 256, 0, st>>>(A, lda, y, xout, ldx, (int)j0, (int)j1);
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}
