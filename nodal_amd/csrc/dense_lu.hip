// Dense path of Circuit.solve: LU with row pivoting + triangular solves, fp64 --
// the arithmetic LAPACK dgesv performs behind np.linalg.solve(G, A)
// (reference nodal/nodal.py:327).
//
// Layout: column-major, leading dimension lda = dense_lda(n) (padded), n + 1
// columns: column n is the right-hand side, so the row interchanges and the
// forward substitution L y = P b happen as part of the blocked factorisation.
//
// Two regimes:
//   n <= GEPP_MAX  classic partial pivoting, one pivot search per column with the
//                  LAPACK idamax rule (first row of maximal |a|): the same pivot
//                  sequence as dgetrf, for the small circuits whose printed digits
//                  users compare with the reference.
//   n >  GEPP_MAX  two-level blocking for the matrix cores.  Outer panels of W = 256
//                  columns feed a K = 256 trailing update (gemm_f64.hip; 32 flop per
//                  byte of C traffic, MFMA-bound).  Inside an outer panel, blocks of
//                  32 columns are pivoted by a tournament (communication-avoiding
//                  LU, Grigori/Demmel/Xiang): every 256-row slab elects 32 candidate
//                  rows by partial pivoting on a register-resident copy, candidates
//                  are merged 8 slabs at a time, and the winners' LU is the block's
//                  L11/U11.  That replaces 2 launches per COLUMN by ~7 per 32 columns;
//                  its stability is that of partial pivoting in practice and the
//                  solution is checked by the scaled residual.
//   passive        resistors + current sources only, all R > 0 (B == 0): G is column
//   networks       diagonally dominant, and on such matrices partial pivoting never
//   (n > GEPP_MAX) interchanges -- the diagonal is a maximal entry of its column, idamax
//                  resolves ties to the first row, which IS the diagonal, and Schur
//                  complements of column diagonally dominant matrices stay so.  The
//                  pivot search is skipped: same pivot sequence as dgetrf, none of its
//                  latency.  (BASELINE.json configs 2 and 4.)
// An exactly zero pivot sets info = column + 1 (LAPACK convention).
#include <cstdlib>

#include "ctx.h"

namespace {

constexpr int NB = 32;          // inner block
constexpr int W = 256;          // outer panel (K of the trailing update)
constexpr int GEPP_MAX = 2048;  // above this, tournament pivoting
constexpr int SLAB = 256;       // rows per tournament workgroup

struct MaxLoc {
    double v;
    int i;
};

typedef const __attribute__((address_space(3))) double *lds_cptr;

// Launder an LDS pointer so the optimiser can no longer prove that the
// `asm volatile("" ::: "memory")` fences below leave the array untouched.  Without
// it LICM hoists every (loop-invariant) LDS read of a fully unrolled triangular
// solve out of the loop: ~500 live doubles, 256 VGPRs and 2.5 KB of scratch per lane.
__device__ __forceinline__ lds_cptr opaque_lds(const double *p) {
    lds_cptr q = (lds_cptr)p;
    asm volatile("" : "+v"(q)::"memory");
    return q;
}

__device__ __forceinline__ MaxLoc better(MaxLoc a, MaxLoc b) {
    // larger magnitude wins; on ties the smaller index (idamax semantics)
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}

// ---------------------------------------------------------------------------------
// classic partial pivoting (small n)
// ---------------------------------------------------------------------------------

// One workgroup: pivot of column c among rows c..n-1, then swap rows c and pivot
// inside the panel columns [j0, j1).
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
        if (v > best.v) best = MaxLoc{v, i};  // NaN never compares greater
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

// rows below the diagonal of column c: l = a / pivot, rank-1 update of (c, j1)
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

// ---------------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------------

// apply interchanges piv[c], c in [c0, c1), to the columns [q0, q1) except [x0, x1)
__global__ __launch_bounds__(256) void lu_swap_cols(double *__restrict__ A, int64_t lda,
                                                    int64_t q0, int64_t q1, int64_t x0,
                                                    int64_t x1, int c0, int c1,
                                                    const int32_t *__restrict__ piv) {
    for (int64_t q = q0 + (int64_t)blockIdx.x * 256 + threadIdx.x; q < q1;
         q += (int64_t)gridDim.x * 256) {
        if (q >= x0 && q < x1) continue;
        double *cq = A + q * lda;
        for (int c = c0; c < c1; ++c) {
            const int p = piv[c];
            if (p != c) {
                const double t = cq[c];
                cq[c] = cq[p];
                cq[p] = t;
            }
        }
    }
}

// U12 = L11^-1 A12 for columns [q0, q1), L11 = unit lower nb x nb block at (j0, j0)
template <bool FULL>  // FULL: nb == NB, no per-element predicates (the hot case)
__global__ __launch_bounds__(256) void lu_trsm32(double *__restrict__ A, int64_t lda, int64_t q0,
                                                 int64_t q1, int j0, int nb_) {
    const int nb = FULL ? NB : nb_;
    __shared__ double L[NB][NB + 1];
    for (int t = threadIdx.x; t < NB * NB; t += 256) {
        const int r = t % NB, s = t / NB;
        L[r][s] = (r < nb && s < nb) ? A[(int64_t)(j0 + s) * lda + j0 + r] : 0.0;
    }
    __syncthreads();
    lds_cptr Lp = opaque_lds(&L[0][0]);
    for (int64_t q = q0 + (int64_t)blockIdx.x * 256 + threadIdx.x; q < q1;
         q += (int64_t)gridDim.x * 256) {
        double *cq = A + q * lda + j0;
        double u[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) u[r] = (FULL || r < nb) ? cq[r] : 0.0;
#pragma unroll
        for (int r = 0; r < NB; ++r) {
#pragma unroll
            for (int s = r + 1; s < NB; ++s) u[s] = fma(-Lp[s * (NB + 1) + r], u[r], u[s]);
            asm volatile("" ::: "memory");  // with the laundered pointer: no hoisting (IR level)
            __builtin_amdgcn_sched_barrier(0);  // ... nor clustering by the machine scheduler
        }
#pragma unroll
        for (int r = 0; r < NB; ++r)
            if (FULL || r < nb) cq[r] = u[r];
    }
}

// ---------------------------------------------------------------------------------
// tournament pivoting
// ---------------------------------------------------------------------------------

// Wave-wide unsigned max with DPP (row_shr 1,2,4,8 inside rows of 16, then
// row_bcast15 / row_bcast31): six v_max_u32_dpp + one v_readlane instead of a
// chain of ds_bpermute shuffles, whose latency dominated the pivot search.
__device__ __forceinline__ unsigned wave_umax(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// One workgroup = up to SLAB rows, one row per lane held in registers (current
// and original values).  Partial pivoting on that copy elects nbc rows; their
// indices and ORIGINAL values go to the candidate buffers in pivot order, so the
// next level reads a compact, coalesced image instead of gathering matrix rows.
//
// One barrier per column: each wave finds its best row (DPP max on the top 32 bits
// of |a|: 20 mantissa bits, plenty for choosing a tournament pivot) and that lane
// speculatively publishes its row and 1/pivot to LDS; after the barrier every
// lane reads the four wave keys, picks the winning wave (lowest on ties) and
// eliminates with its row.  LDS images are double buffered across columns.
//
// At the final level (one workgroup) the winners' eliminated rows ARE the block's
// L11\U11: they are written to lu11 (32 x 32, column-major) and the LAPACK-style
// interchange list piv[c0 .. c0+nbc) is derived from the winners.
template <bool FROM_LIST, bool FULL>
__global__ __launch_bounds__(SLAB) void tslu_select(
    const double *__restrict__ A, int64_t lda, int n, int c0, int nbc_,
    const int32_t *__restrict__ cand_in, const double *__restrict__ val_in, int ncand_in,
    int in_stride, int32_t *__restrict__ cand_out, double *__restrict__ val_out, int out_stride,
    int final_level, double *__restrict__ lu11, int32_t *__restrict__ piv,
    int32_t *__restrict__ info) {
    constexpr int WAVES = SLAB / 64;
    __shared__ double prow[2][WAVES][NB];
    __shared__ double prcp[2][WAVES];
    __shared__ unsigned keys[2][WAVES];
    __shared__ int sel[NB];
    const int nbc = FULL ? NB : nbc_;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int row = -1;
    double a[NB], orig[NB];
    if (FROM_LIST) {
        const int idx = blockIdx.x * SLAB + t;
        if (idx < ncand_in) row = cand_in[idx];
#pragma unroll
        for (int q = 0; q < NB; ++q)
            a[q] = (row >= 0 && (FULL || q < nbc)) ? val_in[(int64_t)q * in_stride + idx] : 0.0;
    } else {
        const int r = c0 + blockIdx.x * SLAB + t;
        if (r < n) row = r;
#pragma unroll
        for (int q = 0; q < NB; ++q)
            a[q] = (row >= 0 && (FULL || q < nbc)) ? A[(int64_t)(c0 + q) * lda + row] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < NB; ++q) orig[q] = a[q];
    bool active = row >= 0;

#pragma unroll
    for (int c = 0; c < NB; ++c) {
        if (FULL || c < nbc) {  // uniform
            const int buf = c & 1;
            unsigned key = 0;
            if (active) {
                const double v = fabs(a[c]);
                key = (v == v ? (unsigned)__double2hiint(v) : 0u) + 1u;  // NaN ranks last
            }
            const unsigned wmax = wave_umax(key);
            const unsigned long long mine = __ballot(key == wmax && key != 0);
            const int lane_w = mine ? __ffsll((long long)mine) - 1 : -1;
            if (lane == lane_w) {
                keys[buf][wave] = wmax;
                prcp[buf][wave] = a[c] != 0.0 ? 1.0 / a[c] : 0.0;
#pragma unroll
                for (int q = 0; q < NB; ++q) prow[buf][wave][q] = a[q];
            }
            if (lane_w < 0 && lane == 0) keys[buf][wave] = 0;
            __syncthreads();
            unsigned kbest = keys[buf][0];
            int wb = 0;
#pragma unroll
            for (int w = 1; w < WAVES; ++w) {
                const unsigned kw = keys[buf][w];
                if (kw > kbest) { kbest = kw; wb = w; }
            }
            const bool any = kbest != 0;
            if (any && wave == wb && lane == lane_w) {
                sel[c] = row;
                active = false;
                const int slot = blockIdx.x * NB + c;
#pragma unroll
                for (int q = 0; q < NB; ++q) val_out[(int64_t)q * out_stride + slot] = orig[q];
                if (final_level) {
#pragma unroll
                    for (int q = 0; q < NB; ++q) lu11[q * NB + c] = a[q];
                    if (a[c] == 0.0) atomicCAS(info, 0, c0 + c + 1);  // exact zero pivot
                }
            } else if (any && active) {
                const double rc = prcp[buf][wb];
                if (rc != 0.0) {
                    const double l = a[c] * rc;
                    a[c] = l;
#pragma unroll
                    for (int q = c + 1; q < NB; ++q) a[q] = fma(-l, prow[buf][wb][q], a[q]);
                }
            }
            if (!any && t == 0) sel[c] = -1;
        }
    }
    __syncthreads();
    if (t < NB) cand_out[blockIdx.x * NB + t] = t < nbc ? sel[t] : -1;

    if (final_level && t < 64) {
        // Turn "row sel[c] becomes row c0+c" into sequential interchanges.  Only the
        // positions c0..c0+nbc-1 and the winners' home positions are ever touched:
        // lane i of wave 0 tracks one such position and what currently sits there.
        __shared__ int s_pos[2 * NB], s_content[2 * NB];
        const int r = t < nbc ? sel[t] : -1;
        const bool extra = t < nbc && r >= c0 + nbc;
        const unsigned long long em = __ballot(extra);
        if (t < nbc) { s_pos[t] = c0 + t; s_content[t] = c0 + t; }
        if (extra) {
            const int at = nbc + __popcll(em & ((1ull << t) - 1ull));
            s_pos[at] = r;
            s_content[at] = r;
        }
        const int cnt = nbc + __popcll(em);
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): one wave, LDS writes done
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < nbc; ++c) {
            const int want = sel[c];
            const bool hit = t < cnt && want >= 0 && s_content[t] == want;
            const unsigned long long hm = __ballot(hit);
            const int at = hm ? __ffsll((long long)hm) - 1 : c;
            if (t == 0) {
                piv[c0 + c] = s_pos[at];
                const int tmp = s_content[c];
                s_content[c] = s_content[at];
                s_content[at] = tmp;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Unpivoted LU of the nbc x nbc diagonal block at (c0, c0) by ONE wave: lane r holds
// row r in registers, the pivot row is broadcast through a double-buffered LDS line,
// and there is no workgroup barrier (a wave executes in lock step).  The factors go
// to lu11 (tslu_apply writes them back into the matrix).  piv = identity.
__global__ __launch_bounds__(64) void diag_lu_nopivot(const double *__restrict__ A, int64_t lda,
                                                      int c0, int nbc, double *__restrict__ lu11,
                                                      int32_t *__restrict__ piv,
                                                      int32_t *__restrict__ info) {
    __shared__ double prow[2][NB];
    const int r = threadIdx.x;
    double a[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q)
        a[q] = (r < nbc && q < nbc) ? A[(int64_t)(c0 + q) * lda + c0 + r] : (r == q ? 1.0 : 0.0);
    if (r < nbc) piv[c0 + r] = c0 + r;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int buf = k & 1;
        if (r == k) {
#pragma unroll
            for (int q = 0; q < NB; ++q)
                if (q >= k) prow[buf][q] = a[q];
            if (a[k] == 0.0 && k < nbc) atomicCAS(info, 0, c0 + k + 1);  // exact zero pivot
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the LDS line is written
        __builtin_amdgcn_wave_barrier();
        if (r > k && r < NB) {
            const double pv = prow[buf][k];
            if (pv != 0.0) {
                const double l = a[k] * (1.0 / pv);
                a[k] = l;
#pragma unroll
                for (int q = k + 1; q < NB; ++q) a[q] = fma(-l, prow[buf][q], a[q]);
            }
        }
    }
    if (r < NB) {
#pragma unroll
        for (int q = 0; q < NB; ++q) lu11[q * NB + r] = a[q];
    }
}

// After the interchanges: rows [c0, c0+nbc) of the block receive L11\U11, every row
// below solves x U11 = a (its row of L21).  One lane per row.
template <bool FULL>
__global__ __launch_bounds__(256) void tslu_apply(double *__restrict__ A, int64_t lda, int n,
                                                  int c0, int nbc_,
                                                  const double *__restrict__ lu11) {
    const int nbc = FULL ? NB : nbc_;
    __shared__ double U[NB][NB + 1];  // U[s][c]
    __shared__ double rcp[NB];
    for (int t = threadIdx.x; t < NB * NB; t += 256) {
        const int s = t % NB, c = t / NB;
        U[s][c] = lu11[c * NB + s];
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        const double d = U[threadIdx.x][threadIdx.x];
        rcp[threadIdx.x] = d != 0.0 ? 1.0 / d : 0.0;
    }
    __syncthreads();
    lds_cptr Up = opaque_lds(&U[0][0]);
    // one row per lane, no grid-stride loop: with a loop the compiler hoists every
    // (loop-invariant) LDS read of U out of it -- 256 VGPRs, and then the kernel cannot
    // co-reside with the trailing update's waves on the second stream
    const int64_t r = c0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    double *ar = A + (int64_t)c0 * lda + r;
    if (r < c0 + nbc) {
        const int rr = (int)(r - c0);
        for (int q = 0; q < nbc; ++q) ar[(int64_t)q * lda] = lu11[q * NB + rr];
        return;
    }
    double x[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) x[q] = (FULL || q < nbc) ? ar[(int64_t)q * lda] : 0.0;
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        const double xc = x[c] * rcp[c];
        x[c] = xc;
#pragma unroll
        for (int q = c + 1; q < NB; ++q) x[q] = fma(-xc, Up[c * (NB + 1) + q], x[q]);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int q = 0; q < NB; ++q)
        if (FULL || q < nbc) ar[(int64_t)q * lda] = x[q];
}

// U12 = L11^-1 A12 with a W x W unit lower triangular L11 at (j0, j0): one workgroup
// per 32 columns; the W x 32 tile of A12 lives in LDS for the whole solve and L11 is
// streamed through LDS in 32 x 32 blocks (diagonal block: substitution; blocks
// below: rank-32 update with the just-solved rows held in registers).
__global__ __launch_bounds__(256) void trsm_outer(double *__restrict__ A, int64_t lda, int64_t q0,
                                                  int64_t q1, int j0, int w) {
    __shared__ double T[32][W + 1];
    __shared__ double Lb[NB][NB + 1];  // Lb[r][k]
    const int col = threadIdx.x & 31, grp = threadIdx.x >> 5;  // 8 row groups
    const int64_t q = q0 + (int64_t)blockIdx.x * 32 + col;
    const bool live = q < q1;
    for (int r = grp; r < w; r += 8) T[col][r] = live ? A[q * lda + j0 + r] : 0.0;
    const int nblk = (w + NB - 1) / NB;
    for (int kb = 0; kb < nblk; ++kb) {
        const int base = kb * NB;
        // diagonal block -> LDS
        __syncthreads();
        for (int t = threadIdx.x; t < NB * NB; t += 256) {
            const int r = t % NB, k = t / NB;
            Lb[r][k] = (base + r < w && base + k < w)
                           ? A[(int64_t)(j0 + base + k) * lda + j0 + base + r] : 0.0;
        }
        __syncthreads();
        for (int k = 0; k < NB - 1; ++k) {
            const double xk = T[col][base + k];
            for (int r = k + 1 + grp; r < NB; r += 8)
                if (base + r < w) T[col][base + r] = fma(-Lb[r][k], xk, T[col][base + r]);
            __syncthreads();
        }
        double x[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) x[k] = base + k < w ? T[col][base + k] : 0.0;
        // blocks below the diagonal
        for (int rb = kb + 1; rb < nblk; ++rb) {
            const int rbase = rb * NB;
            __syncthreads();
            for (int t = threadIdx.x; t < NB * NB; t += 256) {
                const int r = t % NB, k = t / NB;
                Lb[r][k] = (rbase + r < w && base + k < w)
                               ? A[(int64_t)(j0 + base + k) * lda + j0 + rbase + r] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NB / 8; ++i) {
                const int r = grp + 8 * i;
                if (rbase + r < w) {
                    double acc = T[col][rbase + r];
#pragma unroll
                    for (int k = 0; k < NB; ++k) acc = fma(-Lb[r][k], x[k], acc);
                    T[col][rbase + r] = acc;
                }
            }
        }
    }
    __syncthreads();
    if (live)
        for (int r = grp; r < w; r += 8) A[q * lda + j0 + r] = T[col][r];
}

// ---------------------------------------------------------------------------------
// back substitution
// ---------------------------------------------------------------------------------

// One back-substitution step, fused: every workgroup solves the nb x nb diagonal
// block U[j0:j1, j0:j1] x = y redundantly in LDS (32 cheap steps), then updates its
// share of y[0:j0] -= U[0:j0, j0:j1] x.  Workgroup 0 also stores x.  One launch per
// block instead of two.  The solved x_J goes to xout, y[0:j0] is updated in place.
__global__ __launch_bounds__(256) void bs_step(const double *__restrict__ A, int64_t lda,
                                               double *__restrict__ y, double *__restrict__ xout,
                                               int64_t ldx, int j0, int j1) {
    __shared__ double U[NB][NB + 1];
    __shared__ double x[NB];
    const int nb = j1 - j0;
    y += (int64_t)blockIdx.y * lda;      // right-hand side column blockIdx.y
    xout += (int64_t)blockIdx.y * ldx;
    for (int t = threadIdx.x; t < NB * NB; t += 256) {
        const int r = t % NB, s = t / NB;
        U[r][s] = (r < nb && s < nb) ? A[(int64_t)(j0 + s) * lda + j0 + r] : (r == s ? 1.0 : 0.0);
    }
    if (threadIdx.x < NB) x[threadIdx.x] = (int)threadIdx.x < nb ? y[j0 + threadIdx.x] : 0.0;
    __syncthreads();
    if (threadIdx.x < 64) {  // one wave: no workgroup barrier inside the 32 steps
        for (int r = nb - 1; r >= 0; --r) {
            if (threadIdx.x == 0) x[r] = x[r] / U[r][r];
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if ((int)threadIdx.x < r) x[threadIdx.x] = fma(-U[threadIdx.x][r], x[r], x[threadIdx.x]);
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < j0; i += (int64_t)gridDim.x * 256) {
        double acc = y[i];
        for (int s = 0; s < nb; ++s) acc = fma(-A[(int64_t)(j0 + s) * lda + i], x[s], acc);
        y[i] = acc;
    }
    // x goes to a separate vector: y[j0:j1] must stay intact, workgroups that start
    // late still read it
    if (blockIdx.x == 0 && (int)threadIdx.x < nb) xout[j0 + threadIdx.x] = x[threadIdx.x];
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

struct GemmTimer {
    nodal_ctx *h;
    size_t used = 0;
    double flops = 0;
    int begin(hipStream_t st) {
        while (h->evpool.size() < 2 * (used + 1)) {
            hipEvent_t e;
            NODAL_HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventReleaseToDevice));
            h->evpool.push_back(e);
        }
        NODAL_HIP_TRY(h, hipEventRecord(h->evpool[2 * used], st));
        return NODAL_OK;
    }
    int end(hipStream_t st, double f) {
        NODAL_HIP_TRY(h, hipEventRecord(h->evpool[2 * used + 1], st));
        ++used;
        flops += f;
        return NODAL_OK;
    }
    hipEvent_t last_end() const { return h->evpool[2 * used - 1]; }
    void collect() {
        h->kern_ms = 0;
        for (size_t i = 0; i < used; ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->evpool[2 * i], h->evpool[2 * i + 1]) == hipSuccess)
                h->kern_ms += ms;
        }
        h->kern_launches = (int64_t)used;
        h->kern_alg = used ? flops / (double)used : 0.0;  // average flops per launch
    }
};

// After the outer panel [J0, J1) is factored on the panel stream: apply it to the rest
// of the matrix with one panel of LOOKAHEAD.  The next panel's columns [J1, LA1) are
// updated on the panel stream right away, so its factorisation (a chain of small,
// latency-bound kernels) can start while the big trailing update of everything to
// the right, [LA1, ncols), runs on the second stream.  ev_la[0]: panel done;
// ev_la[1]: trailing update done.  `piv` != nullptr: apply the panel's interchanges.
int trailing_update(nodal_ctx *h, double *A, int64_t n, int64_t lda, int64_t ncols, int64_t J0,
                    int64_t J1, const int32_t *piv, GemmTimer &tm, bool &trail_pending) {
    hipStream_t sa = h->stream, sb = h->stream2;
    const int w = (int)(J1 - J0);
    const int64_t LA1 = J1 + W < n ? J1 + W : n;  // end of the lookahead columns
    NODAL_HIP_TRY(h, hipEventRecord(h->ev_la[0], sa));
    // the previous trailing update also touched [J1, LA1): wait for it
    if (trail_pending) NODAL_HIP_TRY(h, hipStreamWaitEvent(sa, h->ev_la[1], 0));
    if (piv)  // columns left of the panel and the lookahead columns
        lu_swap_cols<<<blocks_for(LA1, 256), 256, 0, sa>>>(A, lda, 0, LA1, J0, J1, (int)J0, (int)J1, piv);
    if (LA1 > J1) {
        trsm_outer<<<(unsigned)((LA1 - J1 + 31) / 32), 256, 0, sa>>>(A, lda, J1, LA1, (int)J0, w);
        NODAL_TRY(gemm_sub_f64(h, sa, A + J1 * lda + J1, lda, A + J0 * lda + J1, lda,
                               A + J1 * lda + J0, lda, n - J1, LA1 - J1, w));
    }
    // everything to the right (including the rhs column) on the second stream
    NODAL_HIP_TRY(h, hipStreamWaitEvent(sb, h->ev_la[0], 0));
    if (piv)
        lu_swap_cols<<<blocks_for(ncols - LA1, 256), 256, 0, sb>>>(A, lda, LA1, ncols, 0, 0, (int)J0,
                                                                  (int)J1, piv);
    trsm_outer<<<(unsigned)((ncols - LA1 + 31) / 32), 256, 0, sb>>>(A, lda, LA1, ncols, (int)J0, w);
    NODAL_HIP_TRY(h, hipGetLastError());
    if (J1 < n) {
        NODAL_TRY(tm.begin(sb));
        NODAL_TRY(gemm_sub_f64(h, sb, A + LA1 * lda + J1, lda, A + J0 * lda + J1, lda,
                               A + LA1 * lda + J0, lda, n - J1, ncols - LA1, w));
        NODAL_TRY(tm.end(sb, 2.0 * (double)w * (double)(n - J1) * (double)(ncols - LA1)));
    }
    NODAL_HIP_TRY(h, hipEventRecord(h->ev_la[1], sb));
    trail_pending = true;
    return NODAL_OK;
}

int factor_gepp(nodal_ctx *h, double *A, int64_t n, int64_t lda, int64_t ncols, int32_t *piv,
                int32_t *dinfo, GemmTimer &tm) {
    hipStream_t st = h->stream;
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        const int64_t j1 = j0 + NB < n ? j0 + NB : n;
        for (int64_t c = j0; c < j1; ++c) {
            lu_pivot_swap<<<1, 1024, 0, st>>>(A, n, lda, (int)c, (int)j0, (int)j1, piv, dinfo);
            if (c + 1 < n)
                lu_scale_update<<<blocks_for(n - c - 1, 256), 256, 0, st>>>(A, n, lda, (int)c,
                                                                           (int)j1);
        }
        lu_swap_cols<<<blocks_for(ncols, 256), 256, 0, st>>>(A, lda, 0, ncols, j0, j1, (int)j0,
                                                            (int)j1, piv);
        if (j1 - j0 == NB)
            lu_trsm32<true><<<blocks_for(ncols - j1, 256), 256, 0, st>>>(A, lda, j1, ncols, (int)j0, NB);
        else
            lu_trsm32<false><<<blocks_for(ncols - j1, 256), 256, 0, st>>>(A, lda, j1, ncols,
                                                                         (int)j0, (int)(j1 - j0));
        NODAL_HIP_TRY(h, hipGetLastError());
        if (j1 < n) {
            NODAL_TRY(tm.begin(st));
            NODAL_TRY(gemm_sub_f64(h, st, A + j1 * lda + j1, lda, A + j0 * lda + j1, lda,
                                   A + j1 * lda + j0, lda, n - j1, ncols - j1, j1 - j0));
            NODAL_TRY(tm.end(st, 2.0 * (double)(j1 - j0) * (double)(n - j1) * (double)(ncols - j1)));
        }
    }
    return NODAL_OK;
}

int factor_tournament(nodal_ctx *h, double *A, int64_t n, int64_t lda, int64_t ncols,
                      int32_t *piv, int32_t *dinfo, GemmTimer &tm) {
    hipStream_t st = h->stream;
    // scratch: two candidate sets (row indices + original row values, [q][slot]) and
    // the 32 x 32 winners' LU
    const int64_t max_slabs = (n + SLAB - 1) / SLAB;
    const int stride = (int)(max_slabs * NB);
    const size_t idx_bytes = ((size_t)stride * 4 + 255) & ~(size_t)255;
    const size_t val_bytes = ((size_t)stride * NB * 8 + 255) & ~(size_t)255;
    NODAL_HIP_TRY(h, h->work.reserve(2 * (idx_bytes + val_bytes) + NB * NB * 8 + 256));
    char *wbase = h->work.as<char>();
    int32_t *cand[2] = {reinterpret_cast<int32_t *>(wbase),
                        reinterpret_cast<int32_t *>(wbase + idx_bytes)};
    double *cval[2] = {reinterpret_cast<double *>(wbase + 2 * idx_bytes),
                       reinterpret_cast<double *>(wbase + 2 * idx_bytes + val_bytes)};
    double *lu11 = reinterpret_cast<double *>(wbase + 2 * (idx_bytes + val_bytes));
    bool trail_pending = false;

    for (int64_t J0 = 0; J0 < n; J0 += W) {
        const int64_t J1 = J0 + W < n ? J0 + W : n;
        for (int64_t c0 = J0; c0 < J1; c0 += NB) {
            const int nbc = (int)(c0 + NB < J1 ? NB : J1 - c0);
            // tournament: slabs of the panel, then 8-way merges down to one workgroup
            int64_t groups = (n - c0 + SLAB - 1) / SLAB;
            int cur = 0;
            if (nbc == NB)
                tslu_select<false, true><<<(unsigned)groups, SLAB, 0, st>>>(
                    A, lda, (int)n, (int)c0, nbc, nullptr, nullptr, 0, 0, cand[cur], cval[cur],
                    stride, groups == 1, lu11, piv, dinfo);
            else
                tslu_select<false, false><<<(unsigned)groups, SLAB, 0, st>>>(
                    A, lda, (int)n, (int)c0, nbc, nullptr, nullptr, 0, 0, cand[cur], cval[cur],
                    stride, groups == 1, lu11, piv, dinfo);
            while (groups > 1) {
                const int ncand = (int)groups * NB;
                const int64_t next = (ncand + SLAB - 1) / SLAB;
                if (nbc == NB)
                    tslu_select<true, true><<<(unsigned)next, SLAB, 0, st>>>(
                        A, lda, (int)n, (int)c0, nbc, cand[cur], cval[cur], ncand, stride,
                        cand[cur ^ 1], cval[cur ^ 1], stride, next == 1, lu11, piv, dinfo);
                else
                    tslu_select<true, false><<<(unsigned)next, SLAB, 0, st>>>(
                        A, lda, (int)n, (int)c0, nbc, cand[cur], cval[cur], ncand, stride,
                        cand[cur ^ 1], cval[cur ^ 1], stride, next == 1, lu11, piv, dinfo);
                cur ^= 1;
                groups = next;
            }
            // interchanges inside the outer panel, L11\U11 + L21, then the rest of the panel
            lu_swap_cols<<<blocks_for(J1 - J0, 256), 256, 0, st>>>(A, lda, J0, J1, 0, 0, (int)c0,
                                                                  (int)c0 + nbc, piv);
            if (nbc == NB)
                tslu_apply<true><<<blocks_for(n - c0, 256), 256, 0, st>>>(A, lda, (int)n, (int)c0, NB, lu11);
            else
                tslu_apply<false><<<blocks_for(n - c0, 256), 256, 0, st>>>(A, lda, (int)n, (int)c0, nbc, lu11);
            const int64_t c1 = c0 + nbc;
            if (c1 < J1) {
                if (nbc == NB)
                    lu_trsm32<true><<<blocks_for(J1 - c1, 256), 256, 0, st>>>(A, lda, c1, J1, (int)c0, NB);
                else
                    lu_trsm32<false><<<blocks_for(J1 - c1, 256), 256, 0, st>>>(A, lda, c1, J1, (int)c0, nbc);
                NODAL_TRY(gemm_sub_f64(h, st, A + c1 * lda + c1, lda, A + c0 * lda + c1, lda,
                                       A + c1 * lda + c0, lda, n - c1, J1 - c1, nbc));
            }
            NODAL_HIP_TRY(h, hipGetLastError());
        }
        // outside the panel: interchanges, U12, trailing update (with lookahead)
        NODAL_TRY(trailing_update(h, A, n, lda, ncols, J0, J1, piv, tm, trail_pending));
    }
    if (trail_pending) NODAL_HIP_TRY(h, hipStreamWaitEvent(st, h->ev_la[1], 0));
    return NODAL_OK;
}

// Passive networks: same blocking as the tournament path, no pivot search, no swaps.
int factor_nopivot(nodal_ctx *h, double *A, int64_t n, int64_t lda, int64_t ncols, int32_t *piv,
                   int32_t *dinfo, GemmTimer &tm) {
    hipStream_t st = h->stream;
    NODAL_HIP_TRY(h, h->work.reserve(NB * NB * 8 + 256));
    double *lu11 = h->work.as<double>();
    bool trail_pending = false;
    for (int64_t J0 = 0; J0 < n; J0 += W) {
        const int64_t J1 = J0 + W < n ? J0 + W : n;
        for (int64_t c0 = J0; c0 < J1; c0 += NB) {
            const int nbc = (int)(c0 + NB < J1 ? NB : J1 - c0);
            diag_lu_nopivot<<<1, 64, 0, st>>>(A, lda, (int)c0, nbc, lu11, piv, dinfo);
            if (nbc == NB)
                tslu_apply<true><<<blocks_for(n - c0, 256), 256, 0, st>>>(A, lda, (int)n, (int)c0, NB, lu11);
            else
                tslu_apply<false><<<blocks_for(n - c0, 256), 256, 0, st>>>(A, lda, (int)n, (int)c0, nbc, lu11);
            const int64_t c1 = c0 + nbc;
            if (c1 < J1) {
                if (nbc == NB)
                    lu_trsm32<true><<<blocks_for(J1 - c1, 256), 256, 0, st>>>(A, lda, c1, J1, (int)c0, NB);
                else
                    lu_trsm32<false><<<blocks_for(J1 - c1, 256), 256, 0, st>>>(A, lda, c1, J1, (int)c0, nbc);
                NODAL_TRY(gemm_sub_f64(h, st, A + c1 * lda + c1, lda, A + c0 * lda + c1, lda,
                                       A + c1 * lda + c0, lda, n - c1, J1 - c1, nbc));
            }
            NODAL_HIP_TRY(h, hipGetLastError());
        }
        NODAL_TRY(trailing_update(h, A, n, lda, ncols, J0, J1, nullptr, tm, trail_pending));
    }
    if (trail_pending) NODAL_HIP_TRY(h, hipStreamWaitEvent(st, h->ev_la[1], 0));
    return NODAL_OK;
}

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
constexpr int BLOCKINV_MIN = 256;  // passive systems larger than this: block elimination
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

__global__ __launch_bounds__(1024) void gj128_mfma(const double *__restrict__ src, int64_t lds_, int m,
                                                    double *__restrict__ dst, int64_t ldd,
                                                    int32_t *__restrict__ dinfo, int base) {
    __shared__ double rowraw[4][128], colraw[4][128], pinv[16];
    __shared__ double rowpan[4][RP_S], colpan[4][CP_S];
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
        if (h->gj_scalar) gj128<<<1, 1024, 0, sp>>>(D, lda, w, Q, ldq, dinfo, base);
        else gj128_mfma<<<1, 1024, 0, sp>>>(D, lda, w, Q, ldq, dinfo, base);
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

int factor_blockinv(nodal_ctx *h, double *A, int64_t n, int64_t lda, int64_t ncols, int32_t *dinfo,
                    GemmTimer &tm, int64_t wb) {
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
    static const bool full_mask = getenv("NODAL_BI_MASKED") == nullptr;  // bulk updates on all CUs
    hipStream_t sp = h->stream, sg = full_mask ? h->stream3 : h->stream2;
    hipStream_t s3 = full_mask ? h->stream2 : h->stream3;
    hipEvent_t ev_wfirst = h->ev_bi[0], ev_strip = h->ev_bi[1], ev_wrest = h->ev_bi[2],
               ev_start = h->ev_bi[3], ev_done = h->ev_bi[4], ev_q = h->ev_bi[5], ev_rest = nullptr;
    const int64_t FIRST = 2 * wb;  // columns of W(k) that the next two diagonal blocks need
    // scratch: Q[2] (wb x wb), the inverse's T1 / T2 per recursion level, S1 (wb x FIRST), S (wb x ncols)
    const size_t qb = (size_t)wb * wb, tb = 2 * (size_t)(2 * GJ) * (2 * GJ) + 2 * (size_t)GJ * GJ;
    NODAL_HIP_TRY(h, h->work.reserve((2 * qb + tb + (size_t)wb * FIRST + (size_t)wb * (size_t)ncols) * 8 + 256));
    double *Q[2] = {h->work.as<double>(), h->work.as<double>() + qb};
    double *T = Q[1] + qb, *S1 = T + tb, *S = S1 + (size_t)wb * FIRST;

    // A12 <- Q A12 for the block [J0, J1): columns [c0, c1) on stream st through scratch buf
    auto scale_cols = [&](hipStream_t st, const double *Qk, double *buf, int64_t J0, int64_t J1, int64_t c0,
                          int64_t c1) -> int {
        if (c1 <= c0) return NODAL_OK;
        const int w = (int)(J1 - J0);
        copy_block<<<blocks_for(c1 - c0, 4), 256, 0, st>>>(A + c0 * lda + J0, lda, buf, wb, w, c1 - c0);
        NODAL_HIP_TRY(h, hipGetLastError());
        return gemm_f64(h, st, GEMM_SET, A + c0 * lda + J0, lda, Qk, wb, buf, wb, w, c1 - c0, w);
    };
    auto first_end = [&](int64_t J1) { return J1 + FIRST < ncols ? J1 + FIRST : ncols; };

    NODAL_HIP_TRY(h, hipEventRecord(ev_start, sp));  // the matrix was prepared on the main stream
    NODAL_HIP_TRY(h, hipStreamWaitEvent(sg, ev_start, 0));
    NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_start, 0));
    {
        const int64_t J1 = n < wb ? n : wb;
        NODAL_TRY(invert_diag(h, sp, A, lda, (int)J1, Q[0], wb, T, dinfo, 0));
        NODAL_HIP_TRY(h, hipEventRecord(ev_q, sp));
        NODAL_TRY(scale_cols(sp, Q[0], S1, 0, J1, J1, first_end(J1)));
        NODAL_HIP_TRY(h, hipEventRecord(ev_wfirst, sp));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_q, 0));
        NODAL_TRY(scale_cols(s3, Q[0], S, 0, J1, first_end(J1), ncols));
        NODAL_HIP_TRY(h, hipEventRecord(ev_wrest, s3));
    }
    int blk = 0;
    for (int64_t J0 = 0; J0 < n; J0 += wb, ++blk) {
        const int64_t J1 = J0 + wb < n ? J0 + wb : n;
        const int w = (int)(J1 - J0);
        if (J1 >= n) break;
        const int64_t J2 = J1 + wb < n ? J1 + wb : n;
        const double *L = A + J0 * lda, *U = A + J1 * lda + J0;  // A[:, J0:J1] and W(k)
        double *Qn = Q[(blk + 1) & 1];
        if (ev_rest) {  // block row k+1 was last written by the previous bulk update
            NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_rest, 0));
            NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_rest, 0));
        }
        NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_wfirst, 0));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(sg, ev_wfirst, 0));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(sg, ev_wrest, 0));
        // sp: diag + inverse chain
        NODAL_TRY(gemm_sub_f64(h, sp, A + J1 * lda + J1, lda, L + J1, lda, U, lda, J2 - J1, J2 - J1, w));
        NODAL_TRY(invert_diag(h, sp, A + J1 * lda + J1, lda, (int)(J2 - J1), Qn, wb, T, dinfo, (int)J1));
        NODAL_HIP_TRY(h, hipEventRecord(ev_q, sp));
        // s3: strip
        NODAL_TRY(gemm_sub_f64(h, s3, A + J2 * lda + J1, lda, L + J1, lda, U + (J2 - J1) * lda, lda,
                               J2 - J1, ncols - J2, w));
        NODAL_HIP_TRY(h, hipEventRecord(ev_strip, s3));
        // sg: rest
        if (J2 < n) {
            NODAL_TRY(tm.begin(sg));
            NODAL_TRY(gemm_sub_f64(h, sg, A + J1 * lda + J2, lda, L + J2, lda, U, lda, n - J2, ncols - J1, w));
            NODAL_TRY(tm.end(sg, 2.0 * (double)w * (double)(n - J2) * (double)(ncols - J1)));
            ev_rest = tm.last_end();  // the timing event doubles as the dependency (one packet less)
        } else ev_rest = nullptr;
        // W(k+1): the first columns on the critical stream, the wide remainder beside it
        NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_strip, 0));
        NODAL_TRY(scale_cols(sp, Qn, S1, J1, J2, J2, first_end(J2)));
        NODAL_HIP_TRY(h, hipEventRecord(ev_wfirst, sp));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(s3, ev_q, 0));
        NODAL_TRY(scale_cols(s3, Qn, S, J1, J2, first_end(J2), ncols));
        NODAL_HIP_TRY(h, hipEventRecord(ev_wrest, s3));
    }
    NODAL_HIP_TRY(h, hipEventRecord(ev_done, sg));
    NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_done, 0));
    NODAL_HIP_TRY(h, hipStreamWaitEvent(sp, ev_wrest, 0));
    return NODAL_OK;
}

}  // namespace

int dense_fill_nan(nodal_ctx *h, double *x, int64_t n) {
    fill_nan<<<blocks_for(n, 256), 256, 0, h->stream>>>(x, n);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

int dense_factor_solve(nodal_ctx *h, int32_t *info) {
    NODAL_TRY(dense_factor_solve_multi(h, 1, h->x.as<double>(), h->n, info));
    // Passive systems are eliminated without pivoting.  A floating sub-network makes G exactly
    // singular, but rounding can hide the zero pivot; a solution that does not satisfy the
    // equations triggers the structural test (a connected component without a path to ground),
    // which then reports the matrix as singular like the reference's dgesv would.
    const bool passive = h->passive_network && !h->force_pivoting;
    if (*info == 0 && passive && h->n > BLOCKINV_MIN) {
        const bool had_x = h->have_x;
        h->have_x = true;
        double scaled = 0.0;
        int s = sparse_residual(h, &scaled);
        h->have_x = had_x;
        if (s != NODAL_OK) return s;
        if (!(scaled <= 1e-9)) {
            int32_t floating = 0;
            NODAL_HIP_TRY(h, h->work3.reserve((size_t)h->n + 256));
            NODAL_TRY(stamp_grounded_flags(h, h->work3.as<uint8_t>()));
            NODAL_TRY(csr_has_floating_component(h, h->work3.as<uint8_t>(), &floating));
            if (floating) *info = (int32_t)h->n;
        }
    }
    return NODAL_OK;
}

// Factor the column-major augmented matrix in h->dense (lda = dense_lda(n), n + nrhs
// columns) and leave the solutions in xout (column c at xout + c * ldx).  *info as
// LAPACK dgesv.
int dense_factor_solve_multi(nodal_ctx *h, int32_t nrhs, double *xout, int64_t ldx, int32_t *info) {
    const int64_t n = h->n, lda = dense_lda(n), ncols = n + nrhs;
    hipStream_t st = h->stream;
    double *A = h->dense.as<double>();
    NODAL_HIP_TRY(h, h->piv.reserve((size_t)n * 4 + 64));
    int32_t *piv = h->piv.as<int32_t>();
    int32_t *dinfo = piv + n;  // one spare word after the pivots
    NODAL_HIP_TRY(h, hipMemsetAsync(dinfo, 0, 4, st));

    GemmTimer tm{h};
    bool block_form = false;
    // Small systems keep LAPACK's pivot order exactly (reference parity down to the
    // exact-zero-pivot test of singular circuits); passive ones above BLOCKINV_MIN take
    // the block elimination whatever their size.
    const bool passive = (h->passive_network || h->optimistic_nopivot) && !h->force_pivoting;
    if (passive && h->dense_blockinv && n > BLOCKINV_MIN) block_form = true;
    else if (n <= GEPP_MAX) NODAL_TRY(factor_gepp(h, A, n, lda, ncols, piv, dinfo, tm));
    else if (passive) NODAL_TRY(factor_nopivot(h, A, n, lda, ncols, piv, dinfo, tm));
    else NODAL_TRY(factor_tournament(h, A, n, lda, ncols, piv, dinfo, tm));

    // back substitution on the transformed rhs (column n)
    double *y = A + n * lda;
    if (block_form) {
        // Block width 256.  (512 is implemented -- NODAL_BI_WIDTH=512 -- and was measured on
        // config 2: the K = 512 bulk updates run at 43 instead of 38 TFLOP/s, 13.1 instead of
        // 15.9 ms in total, but their 150-us tiles make every launch of the inverse chain wait
        // longer for a free CU: 20.4 ms per solve against 20.5.)
        int64_t wb = 256;
        if (const char *e = getenv("NODAL_BI_WIDTH")) wb = atoi(e) == 512 ? 512 : 256;
        NODAL_TRY(factor_blockinv(h, A, n, lda, ncols, dinfo, tm, wb));
        for (int64_t j1 = n; j1 > 0;) {
            const int64_t j0 = ((j1 - 1) / wb) * wb;
            dim3 grid(blocks_for(j0 > 0 ? j0 : 1, 64), (unsigned)nrhs);
            if (grid.x > 256 && nrhs > 1) grid.x = 256;  // many columns: fewer workgroups per column
            bs_block<<<grid, 256, 0, st>>>(A, lda, y, xout, ldx, (int)j0, (int)j1);
            j1 = j0;
        }
    } else
    for (int64_t j1 = n; j1 > 0;) {
        int64_t j0 = ((j1 - 1) / NB) * NB;
        dim3 grid(blocks_for(j0 > 0 ? j0 : 1, 256), (unsigned)nrhs);
        if (grid.x > 64 && nrhs > 1) grid.x = 64;  // many columns: fewer workgroups per column
        bs_step<<<grid, 256, 0, st>>>(A, lda, y, xout, ldx, (int)j0, (int)j1);
        j1 = j0;
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    int32_t hinfo = 0;
    NODAL_HIP_TRY(h, hipMemcpyAsync(&hinfo, dinfo, 4, hipMemcpyDeviceToHost, st));
    NODAL_HIP_TRY(h, hipStreamSynchronize(st));
    *info = hinfo;
    tm.collect();
    return NODAL_OK;
}
