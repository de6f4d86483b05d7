// Dense path of Circuit.solve: LU with row pivoting + triangular solves, fp64 --
// the arithmetic LAPACK dgesv performs behind np.linalg.solve(G, A)
// (reference nodal/nodal.py:327).
//
// Layout: column-major, leading dimension lda = dense_lda(n) (padded), n + 1
// columns: column n is the right-hand side, so the row interchanges and the
// forward substitution L y = P b happen as part of the blocked factorisation.
//
// Dispatch (dense_factor_solve_multi):
//   passive, n > 256   conductance networks (resistors + current sources, all R > 0, B == 0) are
//                      symmetric positive definite: block elimination with inverted diagonal
//                      blocks, no pivoting -- block_elim.hip.  Systems with voltage-defined
//                      branches are first reduced to such a network by presolve.hip (api.hip).
//   n <= GEPP_MAX      (1280) classic partial pivoting, one pivot search per column with the
//                      LAPACK idamax rule (first row of maximal |a|): the same pivot
//                      sequence as dgetrf, for the small circuits whose printed digits
//                      users compare with the reference.
//   n >  GEPP_MAX      two-level blocking for the matrix cores.  Outer panels of W = 256
//                      columns feed a K = 256 trailing update (gemm_f64.hip).  Inside an outer
//                      panel, blocks of 32 columns are pivoted by a tournament
//                      (communication-avoiding LU, Grigori/Demmel/Xiang): every 256-row slab
//                      elects 32 candidate rows by partial pivoting on a register-resident
//                      copy, candidates are merged 8 slabs at a time, and the winners' LU is
//                      the block's L11/U11.  That replaces 2 launches per COLUMN by ~7 per 32
//                      columns; its stability is that of partial pivoting in practice and the
//                      solution is checked by the scaled residual.
//   NODAL_DENSE_BLOCKINV=0  passive networks through a no-pivot LU with the same blocking (on
//                      column diagonally dominant matrices partial pivoting never
//                      interchanges: the diagonal is a maximal entry of its column, idamax
//                      resolves ties to the first row, which IS the diagonal, and Schur
//                      complements stay dominant).  Kept as a cross-check of block_elim.hip.
// An exactly zero pivot sets info = column + 1 (LAPACK convention).
#include <cstdlib>

#include <chrono>
#include "dense_common.h"

namespace {

constexpr int NB = 32;          // inner block
constexpr int W = 256;          // outer panel (K of the trailing update)
// above this, tournament pivoting (2048 until round 3; with the panel kernel's 1024-row limit the tournament wins
// from ~1250 unknowns on: n = 1338 6.0 instead of 6.8 ms, n = 1989 9.1 instead of 14.8 ms)
static const int GEPP_MAX = getenv("NODAL_GEPP_MAX") ? atoi(getenv("NODAL_GEPP_MAX")) : 1280;
constexpr int BLOCKINV_MIN = 256;  // passive systems larger than this: block elimination (block_elim.hip)
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

// The same panel in ONE launch: rows [j0, n) x columns [j0, j1) (at most NB columns, at most
// GEPP_PANEL_ROWS rows) live in registers, RPT rows per thread.  Per column ONE workgroup barrier: every
// wave finds its best row ((|a|, position) pairs: larger magnitude wins, on ties the smaller position --
// idamax), the owning lane publishes the pair, the rest of the row and 1 / pivot to LDS (double buffered by
// the column's parity), after the barrier every wave picks the winning wave from the published pairs and
// every thread eliminates its rows with the winner's.  Rows do not move while the panel is in registers: a
// thread tracks the POSITION its row has after the interchanges LAPACK would have made so far (pivot row <->
// row at position c), which is what the tie rule looks at, and writes the row there at the end.  Arithmetic
// per element exactly as lu_scale_update (l = a * (1 / pivot), then fma(-l, u, a) column by column): the two
// forms give the same bits (`NODAL_OPT_GEPP_PANEL` / `NODAL_GEPP_PANEL=0` selects the per-column kernels;
// tested equal).
// Clock stamps (`NODAL_GEPP_PROBE`) for n = 992, cycles per column at 2.4 GHz: search 510 (a DPP reduction +
// readlane is 108 cycles -- tools/clock_probe.hip --, three of them when several lanes hold the maximal high
// word), publish + barrier 1060 (the wait for the slowest wave included), winner 500, elimination 920
// (instruction issue: every wave of a SIMD takes 4 cycles per fp64 FMA, and every wave repeats the search and
// the selection -- hence few waves, two rows per thread above 256 rows): 1.25 us per column, 45 us per panel
// where the per-column kernels took 2 x 32 launches = 300 us.  Measured and not kept: the column loop rolled up, rows shifting through the registers
// instead of being indexed by the unrolled loop's counter (2 KB of code instead of 75 KB: 1.2 instead of
// 0.8 us per column -- the always-full-length update and the rotation of finished rows cost more than the
// instruction fetch of straight-line code).
constexpr int GEPP_PANEL_ROWS = 1024;

__device__ __forceinline__ unsigned wave_umax(unsigned v);  // (below, with the tournament kernels)

// maximum over the first row of 16 lanes (DPP row_shr 1, 2, 4, 8), uniform
__device__ __forceinline__ unsigned row16_umax(unsigned v) {
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 15);
}

// key of a candidate: 0 none, 1 a NaN (chosen only if nothing else is left), bits(|a|) + 2 otherwise (an
// order-preserving image of the magnitude)
__device__ __forceinline__ unsigned long long pivot_key(double a, bool done) {
    const double av = fabs(a);
    return done ? 0ull : (av != av ? 1ull : (unsigned long long)__double_as_longlong(av) + 2ull);
}

template <int RPT>
__global__ __launch_bounds__(GEPP_PANEL_ROWS / 2) void gepp_panel(double *__restrict__ A, int64_t n, int64_t lda,
                                                                  int j0, int j1, int32_t *__restrict__ piv,
                                                                  int32_t *__restrict__ info,
                                                                  long long *__restrict__ probe) {
    constexpr int MAXW = GEPP_PANEL_ROWS / 2 / 64;  // 8 waves at most
    __shared__ __attribute__((aligned(16))) unsigned long long wkey[2][16];  // (entries MAXW .. 15 never win)
    __shared__ __attribute__((aligned(16))) int wpos[2][16];
    // the wave's candidate row; slot k (the pivot itself) carries 1 / pivot, computed by the row's owner
    // before the search: off the critical path
    __shared__ __attribute__((aligned(16))) double wrow[2][MAXW][NB];
    __shared__ int spiv[NB];
    const int wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int wp = j1 - j0;
    if ((int)threadIdx.x < 32 && (int)(threadIdx.x & 15) >= nwaves) {  // waves that do not exist never win
        wkey[threadIdx.x >> 4][threadIdx.x & 15] = 0ull;
        wpos[threadIdx.x >> 4][threadIdx.x & 15] = 0x7fffffff;
    }
    // (two separate arrays, not a[RPT][NB]: the two-dimensional form ended up in scratch memory)
    double a0[NB], a1[NB];
    const int row0 = j0 + (int)threadIdx.x, row1 = row0 + (int)blockDim.x;
    int pos0 = row0, pos1 = row1;
    bool done0 = !(row0 < n), done1 = !(RPT == 2 && row1 < n);
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        a0[q] = (row0 < n && q < wp) ? A[(int64_t)(j0 + q) * lda + row0] : 0.0;
        a1[q] = (RPT == 2 && row1 < n && q < wp) ? A[(int64_t)(j0 + q) * lda + row1] : 0.0;
    }
    int zero_at = 0;
    long long ph[5] = {0, 0, 0, 0, 0}, tlast = probe ? clock64() : 0;  // NODAL_GEPP_PROBE: cycles per phase
#define GEPP_STAMP(i)                        \
    if (probe) {                             \
        const long long tnow = clock64();    \
        ph[i] += tnow - tlast;               \
        tlast = tnow;                        \
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        if (k < wp) {  // (uniform)
            const int c = j0 + k, par = k & 1;
            // the thread's candidate
            unsigned long long key = pivot_key(a0[k], done0);
            int mypos = pos0;
            bool second = false;
            if (RPT == 2) {
                const unsigned long long k1 = pivot_key(a1[k], done1);
                second = k1 > key || (k1 == key && pos1 < mypos);
                key = second ? k1 : key;
                mypos = second ? pos1 : mypos;
            }
            const double myrcp = 1.0 / (second ? a1[k] : a0[k]);
            // the wave's maximum by 32-bit DPP reductions: high word; if one lane holds it, that lane is the
            // winner, otherwise low word among the lanes that hold the high maximum, then the smallest position
            // among the lanes that hold the maximum (idamax)
            const unsigned khi = (unsigned)(key >> 32);
            const unsigned mhi = wave_umax(khi);
            const unsigned long long at_hi = __ballot(khi == mhi);
            unsigned long long best;
            int bpos;
            if (__popcll(at_hi) == 1) {  // (uniform)
                const int src = __ffsll(at_hi) - 1;
                best = ((unsigned long long)mhi << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)key, src);
                bpos = __builtin_amdgcn_readlane(mypos, src);
            } else {
                const unsigned mlo = wave_umax(khi == mhi ? (unsigned)key : 0u);
                best = ((unsigned long long)mhi << 32) | mlo;
                bpos = (int)~wave_umax(key == best ? ~(unsigned)mypos : 0u);
            }
            GEPP_STAMP(0)  // candidate + the wave's search
            if (best == 0ull) {
                if ((threadIdx.x & 63) == 0) {
                    wkey[par][wave] = 0ull;
                    wpos[par][wave] = 0x7fffffff;
                }
            } else if (key == best && mypos == bpos) {  // one lane: positions are distinct
                wkey[par][wave] = best;
                wpos[par][wave] = bpos;
                wrow[par][wave][k] = myrcp;
#pragma unroll
                for (int q = k + 1; q < NB; ++q) wrow[par][wave][q] = second ? a1[q] : a0[q];
            }
            __syncthreads();
            GEPP_STAMP(1)  // publish + barrier
            // the winner among the waves: lane l looks at wave l & 15, the same reductions inside a row of 16
            // lanes (every wave does this for itself: one barrier per column)
            const unsigned long long ek = wkey[par][threadIdx.x & 15];
            const int ep = wpos[par][threadIdx.x & 15];
            const unsigned ehi = (unsigned)(ek >> 32);
            const unsigned whi = row16_umax(ehi);
            const unsigned long long w_hi = __ballot(ehi == whi) & 0xffffull;
            unsigned long long win;
            int winpos, ww;
            if (__popcll(w_hi) == 1) {
                ww = __ffsll(w_hi) - 1;
                win = ((unsigned long long)whi << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)ek, ww);
                winpos = __builtin_amdgcn_readlane(ep, ww);
            } else {
                const unsigned wlo = row16_umax(ehi == whi ? (unsigned)ek : 0u);
                win = ((unsigned long long)whi << 32) | wlo;
                winpos = (int)~row16_umax(ek == win ? ~(unsigned)ep : 0u);
                ww = __ffsll((unsigned long long)__ballot(ek == win && ep == winpos)) - 1;
            }
            // (position c is always among the candidates, so there is a winner; no global store inside the
            // loop: the next step's loads would wait for it)
            if (threadIdx.x == 0) {
                spiv[k] = winpos;
                if (win <= 2ull && zero_at == 0) zero_at = c + 1;  // exact zero pivot (or a column of NaNs)
            }
            lds_cptr u = opaque_lds(&wrow[par][ww][0]);  // (a VGPR base: the reads take immediate offsets)
            const double rcp = u[k];
            const bool eliminate = win != 2ull;  // not an exact zero (a NaN pivot spreads, as in the per-column form)
            GEPP_STAMP(2)  // the winner among the waves
#define GEPP_ROW(a_, pos_, done_)                                                                        \
            if (!done_) {                                                                                \
                if (pos_ == winpos) {                                                                    \
                    done_ = true;                                                                        \
                    pos_ = c;                                                                            \
                } else {                                                                                 \
                    if (pos_ == c) pos_ = winpos;                                                        \
                    if (eliminate) {                                                                     \
                        const double l = a_[k] * rcp;                                                    \
                        a_[k] = l;                                                                       \
                        _Pragma("unroll") for (int q = k + 1; q < NB; ++q) a_[q] = fma(-l, u[q], a_[q]); \
                    }                                                                                    \
                }                                                                                        \
            }
            GEPP_ROW(a0, pos0, done0)
            if (RPT == 2) { GEPP_ROW(a1, pos1, done1) }
#undef GEPP_ROW
            GEPP_STAMP(3)  // elimination
        }
    }
    if (row0 < n) {
#pragma unroll
        for (int q = 0; q < NB; ++q)
            if (q < wp) A[(int64_t)(j0 + q) * lda + pos0] = a0[q];
    }
    if (RPT == 2 && row1 < n) {
#pragma unroll
        for (int q = 0; q < NB; ++q)
            if (q < wp) A[(int64_t)(j0 + q) * lda + pos1] = a1[q];
    }
    if (threadIdx.x == 0 && zero_at) atomicCAS(info, 0, zero_at);
    if ((int)threadIdx.x < wp) piv[j0 + threadIdx.x] = spiv[threadIdx.x];  // (written by thread 0 of the same wave)
    GEPP_STAMP(4)  // write-back
    if (probe && threadIdx.x == 0) {
        for (int i = 0; i < 5; ++i) atomicAdd((unsigned long long *)&probe[i], (unsigned long long)ph[i]);
        atomicAdd((unsigned long long *)&probe[5], 1ull);
    }
}
#undef GEPP_STAMP

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

// The same product with the 32 rows of a column spread over 32 LANES (two columns per wavefront): lane s keeps
// u[s] and its row of L in registers, step r broadcasts u[r] by a shuffle and every lane below takes its
// fma.  For the few columns of a small system (n <= 1280: at most 5 workgroups of lu_trsm32, each lane walking
// its 496 dependent fmas alone: 9.4 us) the chain is 32 shuffles long: 3 us.  Same fma per element, same order.
__global__ __launch_bounds__(256) void lu_trsm32w(double *__restrict__ A, int64_t lda, int64_t q0, int64_t q1, int j0,
                                                  int nb) {
    const int lane = threadIdx.x & 63, s = lane & 31, half = lane >> 5;
    double Lrow[NB];
#pragma unroll
    for (int r = 0; r < NB; ++r) Lrow[r] = (s < nb && r < s) ? A[(int64_t)(j0 + r) * lda + j0 + s] : 0.0;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t q = q0 + 2 * wave + half; q - half < q1; q += 2 * nwaves) {  // (both halves of a wave loop together)
        const bool live = q < q1 && s < nb;
        double u = live ? A[q * lda + j0 + s] : 0.0;
#pragma unroll
        for (int r = 0; r < NB - 1; ++r) {
            const double ur = __shfl(u, (lane & 32) | r, 64);
            if (s > r) u = fma(-Lrow[r], ur, u);
        }
        if (live) A[q * lda + j0 + s] = u;
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

// The whole back substitution of a system of at most BS_SMALL_MAX unknowns in ONE launch (one workgroup per
// right-hand side; thread i keeps y[i] in a register): block by block from the last, the first wave solves
// the 32 x 32 diagonal block in LDS, every thread above the block subtracts its row's share.  The arithmetic
// per element is bs_step's (x[r] / U[r][r]; fma(-U[t][r], x[r], x[t]); then fma(-A[i][j0 + s], x[s], y[i]) for
// s ascending): the same bits, 3 us per block instead of a 10.5-us launch.
constexpr int BS_SMALL_MAX = 1024;
__global__ __launch_bounds__(BS_SMALL_MAX) void bs_small(const double *__restrict__ A, int64_t lda,
                                                         const double *__restrict__ y_in, double *__restrict__ xout,
                                                         int64_t ldx, int n) {
    __shared__ double U[NB][NB + 1];
    __shared__ double x[NB];
    const double *y = y_in + (int64_t)blockIdx.x * lda;  // right-hand side column blockIdx.x
    xout += (int64_t)blockIdx.x * ldx;
    const int i = threadIdx.x;
    double acc = i < n ? y[i] : 0.0;
    for (int j1 = n; j1 > 0;) {
        const int j0 = ((j1 - 1) / NB) * NB, nb = j1 - j0;
        for (int t = threadIdx.x; t < NB * NB; t += blockDim.x) {
            const int r = t % NB, s = t / NB;
            U[r][s] = (r < nb && s < nb) ? A[(int64_t)(j0 + s) * lda + j0 + r] : (r == s ? 1.0 : 0.0);
        }
        if (i >= j0 && i < j0 + NB) x[i - j0] = i < j1 ? acc : 0.0;
        // this thread's row of the block column (the loads do not wait for the triangular solve)
        double a[NB];
#pragma unroll
        for (int s = 0; s < NB; ++s) a[s] = (i < j0 && s < nb) ? A[(int64_t)(j0 + s) * lda + i] : 0.0;
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
        if (i < j0) {
#pragma unroll
            for (int s = 0; s < NB; ++s)
                if (s < nb) acc = fma(-a[s], x[s], acc);
        } else if (i < j1) {
            xout[i] = x[i - j0];
        }
        __syncthreads();  // (U and x are rewritten by the next block)
        j1 = j0;
    }
}

__global__ __launch_bounds__(256) void fill_nan(double *x, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        x[i] = __builtin_nan("");
}



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
    static const bool probing = getenv("NODAL_GEPP_PROBE") != nullptr;
    long long *probe = nullptr;
    if (probing) {
        NODAL_HIP_TRY(h, h->work3.reserve(256));
        probe = h->work3.as<long long>();
        NODAL_HIP_TRY(h, hipMemsetAsync(probe, 0, 64, st));
    }
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        const int64_t j1 = j0 + NB < n ? j0 + NB : n;
        if (h->gepp_panel && n - j0 <= GEPP_PANEL_ROWS) {
            const int64_t rows = n - j0;
            if (rows <= 256) gepp_panel<1><<<1, (int)((rows + 63) / 64) * 64, 0, st>>>(A, n, lda, (int)j0, (int)j1, piv, dinfo, probe);
            else gepp_panel<2><<<1, (int)((rows + 127) / 128) * 64, 0, st>>>(A, n, lda, (int)j0, (int)j1, piv, dinfo, probe);
        } else
        for (int64_t c = j0; c < j1; ++c) {
            lu_pivot_swap<<<1, 1024, 0, st>>>(A, n, lda, (int)c, (int)j0, (int)j1, piv, dinfo);
            if (c + 1 < n)
                lu_scale_update<<<blocks_for(n - c - 1, 256), 256, 0, st>>>(A, n, lda, (int)c,
                                                                           (int)j1);
        }
        lu_swap_cols<<<blocks_for(ncols, 256), 256, 0, st>>>(A, lda, 0, ncols, j0, j1, (int)j0,
                                                            (int)j1, piv);
        if (h->gepp_panel)
            lu_trsm32w<<<blocks_for(ncols - j1, 8), 256, 0, st>>>(A, lda, j1, ncols, (int)j0, (int)(j1 - j0));
        else if (j1 - j0 == NB)
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
    if (probe) {
        long long host[8];
        NODAL_HIP_TRY(h, hipMemcpyAsync(host, probe, 64, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        if (host[5] > 0)
            fprintf(stderr, "[gepp] n %lld: %lld panels, cycles per panel: search %lld, publish + barrier %lld, winner %lld, "
                            "elimination %lld, write-back %lld\n", (long long)n, host[5], host[0] / host[5], host[1] / host[5],
                    host[2] / host[5], host[3] / host[5], host[4] / host[5]);
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
    StreamJoinGuard join(st);  // (a failed call below must not leave the trailing-update stream running unjoined)
    join.add(h->stream2, h->ev_la[1]);

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
    join.disarm();
    return NODAL_OK;
}

// Passive networks: same blocking as the tournament path, no pivot search, no swaps.
int factor_nopivot(nodal_ctx *h, double *A, int64_t n, int64_t lda, int64_t ncols, int32_t *piv,
                   int32_t *dinfo, GemmTimer &tm) {
    hipStream_t st = h->stream;
    NODAL_HIP_TRY(h, h->work.reserve(NB * NB * 8 + 256));
    double *lu11 = h->work.as<double>();
    bool trail_pending = false;
    StreamJoinGuard join(st);  // (a failed call below must not leave the trailing-update stream running unjoined)
    join.add(h->stream2, h->ev_la[1]);
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
    join.disarm();
    return NODAL_OK;
}

}  // namespace

int dense_fill_nan(nodal_ctx *h, double *x, int64_t n) {
    fill_nan<<<blocks_for(n, 256), 256, 0, h->stream>>>(x, n);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

int dense_factor_solve(nodal_ctx *h, int32_t *info) {
    NODAL_TRY(nodal_ensure_aux_streams(h));
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
            NODAL_TRY(grounded_flags(h, h->work3.as<uint8_t>()));
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
    const auto t_start = std::chrono::steady_clock::now();
    NODAL_TRY(nodal_ensure_aux_streams(h));
    const int64_t n = h->n, lda = dense_lda(n), ncols = n + nrhs;
    hipStream_t st = h->stream;
    double *A = h->dense.as<double>();
    NODAL_HIP_TRY(h, h->piv.reserve((size_t)n * 4 + 64));
    int32_t *piv = h->piv.as<int32_t>();
    int32_t *dinfo = piv + n;  // one spare word after the pivots
    NODAL_HIP_TRY(h, hipMemsetAsync(dinfo, 0, 4, st));

    GemmTimer tm{h};
    tm.reset();
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
        NODAL_TRY(dense_block_elimination(h, A, n, lda, nrhs, xout, ldx, dinfo));
    } else if (n <= BS_SMALL_MAX && h->gepp_panel) {
        bs_small<<<(unsigned)nrhs, (unsigned)((n + 63) / 64 * 64), 0, st>>>(A, lda, y, xout, ldx, (int)n);
    } else
    for (int64_t j1 = n; j1 > 0;) {
        int64_t j0 = ((j1 - 1) / NB) * NB;
        dim3 grid(blocks_for(j0 > 0 ? j0 : 1, 256), (unsigned)nrhs);
        if (grid.x > 64 && nrhs > 1) grid.x = 64;  // many columns: fewer workgroups per column
        bs_step<<<grid, 256, 0, st>>>(A, lda, y, xout, ldx, (int)j0, (int)j1);
        j1 = j0;
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    static const bool trace_enq = getenv("NODAL_TRACE") != nullptr;
    const auto t_enq = std::chrono::steady_clock::now();
    int32_t hinfo = 0;
    NODAL_TRY(nodal_read_words(h, &hinfo, dinfo, 4));
    if (trace_enq)  // (is the host ahead of the device? a wait of ~0 means the launches are what takes the time)
        fprintf(stderr, "[dense] n %lld: enqueued in %.2f ms (%llu launches of this thread so far), then waited %.2f ms for the device\n",
                (long long)n, std::chrono::duration<double, std::milli>(t_enq - t_start).count(),
                (unsigned long long)nodal_launches_noted(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq).count());
    *info = hinfo;
    tm.collect();
    return NODAL_OK;
}
