// C[M x N] -= A[M x K] * B[K x N], fp64, column-major, on the CDNA4 matrix cores
// (v_mfma_f64_4x4x4_4b_f64).  The trailing update of the dense LU (dense_lu.hip).
// MODE selects the epilogue: GEMM_SUB  C -= A B,  GEMM_SET  C = A B,  GEMM_SETNEG  C = -A B
// (the two overwrite forms serve the block-inverse elimination of dense_lu.hip).
//
// Tiling for 64-wide wavefronts: a 256-thread workgroup (4 waves in a 2 x 2 grid)
// owns a 128 x 128 tile of C; each wave a 64 x 64 sub-tile = 4 x 4 tiles of 16 x 16,
// i.e. 64 accumulators of one f64 per lane (128 VGPRs; two workgroups per CU).  K is walked in chunks of 16
// staged through LDS, double buffered: while the matrix cores work on chunk k the
// global loads of chunk k+1 are in flight.
//
// Instruction: v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks per issue).
// On gfx950 it sustains 72-74 TFLOP/s, twice the 36 TFLOP/s of v_mfma_f64_16x16x4_f64
// (tools/fp64_mix.hip, profiles/r01_fp64_mix.txt).  Register layout, probed with one-hot
// operands (tools/mfma444_probe.hip): with lane l = 16 k + 4 b + t,
//   A operand holds A_b[i = t][k],  B operand holds B_b[k][j = t],
//   D lane 16 i + 4 b + j holds D_b[i][j] = sum_k A_b[i][k] B_b[k][j].
// It is issued here as  D = mfma(a := B fragment, b := A fragment):
//   b operand: lane l holds A[row = R0 + (l & 15)][k = l >> 4]      (16 contiguous rows)
//   a operand: lane l holds B[k = l >> 4][col = C0 + (l & 3)]       (4 columns, the same
//              for all four blocks: a broadcast read of 16 values)
//   D        : lane l holds C[row = R0 + (l & 15)][col = C0 + (l >> 4)]
// i.e. one issue updates a 16 x 4 strip of C over 4 k; four issues (C0 = 0, 4, 8, 12) are
// one 16 x 16 x 4 step with accumulator register r <-> column (l >> 4) + 4 r.  The 16
// lanes of a quarter-wave cover 16 CONTIGUOUS rows of column-major C, so every
// epilogue load / store moves whole 128-byte lines.
// LDS images are k-major ([k][i] and [k][j]); the row stride 144 doubles puts the
// two k values a 32-lane half reads on disjoint bank halves (ds_read_b64: 64 banks).
#include "ctx.h"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int NT = 256;  // threads per workgroup of the big kernel
constexpr int LDA_S = 144;  // A image row stride (doubles)
constexpr int LDB_S = 145;  // B image row stride: odd/2 -> conflict-free transposing writes

// TA: the A operand is given TRANSPOSED -- `A` points to a K x M column-major panel (leading dimension
// lda) and the product is C -= A^T B: both operands are then read along k, the contiguous direction.
// This is the bulk update of the SYMMETRIC block elimination (block_elim.hip), where the block column
// below the diagonal is the transpose of the block row right of it and is never formed.
// UPPER: C is square-leading (row i <-> column i) and only its upper block triangle is wanted: tiles
// entirely below the diagonal blocks of `band` rows return at once.
template <int MODE, bool TA = false, bool UPPER = false>
__global__ __launch_bounds__(NT, 2) void gemm_sub_kernel(double *__restrict__ C, int64_t ldc,
                                                         const double *__restrict__ A, int64_t lda,
                                                         const double *__restrict__ B, int64_t ldb,
                                                         int M, int N, int K, int band = BM) {
    constexpr int LA = TA ? LDB_S : LDA_S;  // (transposing writes of the A image want the odd stride too)
    __shared__ double As[2][BK][LA];
    __shared__ double Bs[2][BK][LDB_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;  // 2 x 2 waves, 64 x 64 each
    // Plain column-major tile order.  (An XCD-aware 8 x 8 super-tile order was measured:
    // HBM operand traffic drops, the rate does not change -- 47.4 vs 47.5 TFLOP/s at 8192^2 --
    // and partial super-tiles unbalance the XCDs, so it is not used.)
    const int tiles_m = (M + BM - 1) / BM;
    const int tm = (int)(blockIdx.x % (unsigned)tiles_m), tn = (int)(blockIdx.x / (unsigned)tiles_m);
    const int row0 = tm * BM, col0 = tn * BN;
    // (uniform) strictly below the diagonal BLOCK the tile's rows belong to: the diagonal blocks (`band`
    // rows and columns each, aligned with the origin of C) are wanted whole
    if (UPPER && col0 + BN <= (row0 / band) * band) return;
    const int li = lane & 15, lk = lane >> 4, lq = lane & 3;

    double acc[4][4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;

    // global -> register staging of one K chunk: 8 doubles of A and 8 of B per thread.
    // Rows >= M and columns >= N are read from the clamped (last valid) row / column:
    // they only feed C entries that are never stored.  k >= K is read from k = K - 1
    // and multiplied by zero.  No divergent branches; uniform 64-bit bases (SGPRs) plus
    // 32-bit per-thread byte offsets.
    double ra[8], rb[8];
    const int ai = tid & 127, ak = tid >> 7;  // A image [k][i]: i contiguous in memory; k = ak + 2 r
    const int bk = tid & 15, bj = tid >> 4;   // B: 16 contiguous k of one column; j = bj + 16 r
    const char *Abase = reinterpret_cast<const char *>(TA ? A + (int64_t)row0 * lda : A + row0);
    const char *Bbase = reinterpret_cast<const char *>(B + (int64_t)col0 * ldb);
    const int64_t lda8 = lda * 8, ldb8 = ldb * 8;
    const uint32_t aoff = (uint32_t)((row0 + ai < M ? ai : M - 1 - row0) * 8);
    const int jlast = N - 1 - col0;  // uniform
    const int ilast = M - 1 - row0;  // uniform (TA)
    auto load_chunk = [&](int k0) {
        const int gkb = k0 + bk;
        const uint32_t kb8 = (uint32_t)(gkb < K ? gkb : K - 1) * 8u;
        const char *Ak = Abase + (int64_t)k0 * lda8;  // uniform
        const int klast = K - 1 - k0;                 // uniform
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (TA) {  // like B: 16 contiguous k of one column i of the K x M panel; i = bj + 16 r
                const int ii = bj + 16 * r < ilast ? bj + 16 * r : ilast;
                ra[r] = *reinterpret_cast<const double *>(Abase + ((uint32_t)ii * (uint32_t)lda8 + kb8));
            } else {
                const int kk = ak + 2 * r < klast ? ak + 2 * r : klast;
                ra[r] = *reinterpret_cast<const double *>(Ak + ((uint32_t)kk * (uint32_t)lda8 + aoff));
            }
            const int jj = bj + 16 * r < jlast ? bj + 16 * r : jlast;
            rb[r] = *reinterpret_cast<const double *>(Bbase + ((uint32_t)jj * (uint32_t)ldb8 + kb8));
        }
    };
    // (the zero mask is applied here, after the MFMA section, so that nothing waits on the
    // global loads before the matrix cores have their work)
    auto store_chunk = [&](int buf, int k0) {
        const double mb = k0 + bk < K ? 1.0 : 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (TA) As[buf][bk][bj + 16 * r] = ra[r] * mb;
            else As[buf][ak + 2 * r][ai] = ra[r] * (k0 + ak + 2 * r < K ? 1.0 : 0.0);
            Bs[buf][bk][bj + 16 * r] = rb[r] * mb;
        }
    };

    auto mfma_chunk = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double af[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = As[buf][ks * 4 + lk][wm + mi * 16 + li];
#pragma unroll
            for (int np = 0; np < 2; ++np) {  // two 16-column groups at a time
                double bq[2][4];
#pragma unroll
                for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        bq[nh][r] = Bs[buf][ks * 4 + lk][wn + (2 * np + nh) * 16 + 4 * r + lq];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            acc[mi][2 * np + nh][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(
                                bq[nh][r], af[mi], acc[mi][2 * np + nh][r], 0, 0, 0);
                // do not hoist the later LDS reads above these MFMAs (register budget)
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    const int nchunks = (K + BK - 1) / BK;
    load_chunk(0);
    store_chunk(0, 0);
    __syncthreads();
    for (int ch = 0; ch + 1 < nchunks; ++ch) {
        load_chunk((ch + 1) * BK);
        mfma_chunk(ch & 1);
        store_chunk((ch & 1) ^ 1, (ch + 1) * BK);
        __syncthreads();
    }

    const bool interior = row0 + BM <= M && col0 + BN <= N;
    if (interior) {
        // Interior tile, no predicates.  The C values of the first 16-column group are
        // requested BEFORE the last chunk's MFMAs (the staging registers are free by
        // then) and each later group while the previous one is subtracted and stored.
        double *cw = C + (int64_t)(col0 + wn + lk) * ldc + row0 + wm + li;
        double cv[2][4][4];
        auto request = [&](int ni, int slot) {
            if (MODE != GEMM_SUB) return;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) cv[slot][r][mi] = cw[(int64_t)(ni * 16 + 4 * r) * ldc + mi * 16];
        };
        request(0, 0);
        mfma_chunk((nchunks - 1) & 1);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            if (ni + 1 < 4) request(ni + 1, (ni + 1) & 1);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    double v;
                    if (MODE == GEMM_SUB) v = cv[ni & 1][r][mi] - acc[mi][ni][r];
                    else if (MODE == GEMM_SET) v = acc[mi][ni][r];
                    else v = -acc[mi][ni][r];
                    cw[(int64_t)(ni * 16 + 4 * r) * ldc + mi * 16] = v;
                }
        }
        return;
    }
    mfma_chunk((nchunks - 1) & 1);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gc = col0 + wn + ni * 16 + lk + 4 * r;
            if (gc >= N) continue;
            double *cc = C + (int64_t)gc * ldc;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int gr = row0 + wm + mi * 16 + li;
                if (gr < M) {
                    if (MODE == GEMM_SUB) cc[gr] -= acc[mi][ni][r];
                    else if (MODE == GEMM_SET) cc[gr] = acc[mi][ni][r];
                    else cc[gr] = -acc[mi][ni][r];
                }
            }
        }
    }
}

// Small products (the 128 / 256-wide blocks of the block-inverse chain): the big kernel
// would run them on one to four workgroups, a chain of K / 16 dependent global loads
// each.  Here a workgroup owns a 32 x 32 tile of C, its four waves split K, every wave
// loads its MFMA fragments straight from global memory (L2-resident operands) with all
// loads of 16 k-steps in flight, and the four partial tiles meet in LDS.
template <int MODE>
__device__ __forceinline__ void gemm_small_body(double *__restrict__ C, int64_t ldc,
                                                const double *__restrict__ A, int64_t lda,
                                                const double *__restrict__ B, int64_t ldb, int M,
                                                int N, int K) {
    __shared__ double part[3][4][64][4];
    __builtin_amdgcn_s_setprio(3);  // chain kernels: ahead of the bulk update's waves on a shared CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int row0 = blockIdx.x * 32, col0 = blockIdx.y * 32;
    // k-steps (of 4) split evenly over the four waves
    const int ksteps = (K + 3) / 4, per = (ksteps + 3) / 4;
    const int s0 = wave * per, s1 = s0 + per < ksteps ? s0 + per : ksteps;
    v4f64 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = v4f64{0.0, 0.0, 0.0, 0.0};
    const int r0 = row0 + li, r1 = row0 + 16 + li, c0 = col0 + li, c1 = col0 + 16 + li;
    // Rows >= M and columns >= N are read from the last valid one (they feed entries of C that are never stored) and
    // k >= K from k = K - 1, times zero: every load is unconditional, and the loads of KG k-steps are ALL requested
    // before the first matrix instruction -- one round trip per group.  (Until round 5 the loop kept its bounds tests
    // as branches around the loads and waited for each k-step's four loads before its four MFMAs: 1.6 us per k-step,
    // 27 us for a 256^3 product, 115 us of the 245-us chain of a 256-block.)
    const double *a0 = A + (r0 < M ? r0 : M - 1), *a1 = A + (r1 < M ? r1 : M - 1);
    const double *b0 = B + (int64_t)(c0 < N ? c0 : N - 1) * ldb, *b1 = B + (int64_t)(c1 < N ? c1 : N - 1) * ldb;
    constexpr int KG = 8;
    for (int sg = s0; sg < s1; sg += KG) {
        double af0[KG], af1[KG], bf0[KG], bf1[KG];
#pragma unroll
        for (int u = 0; u < KG; ++u) {
            const int k = 4 * (sg + u) + lk, kc = k < K ? k : K - 1;
            af0[u] = a0[(int64_t)kc * lda];
            af1[u] = a1[(int64_t)kc * lda];
            bf0[u] = b0[kc];
            bf1[u] = b1[kc];
        }
#pragma unroll
        for (int u = 0; u < KG; ++u) {
            const int k = 4 * (sg + u) + lk;
            const double z = (sg + u < s1 && k < K) ? 1.0 : 0.0;  // (one operand of each product is enough)
            const double x0 = af0[u] * z, x1 = af1[u] * z;
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf0[u], x0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf1[u], x0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf0[u], x1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf1[u], x1, acc[1][1], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave - 1][mi * 2 + ni][lane][r] = acc[mi][ni][r];
    }
    // (wave 0 requests its entries of C while the others write their partial tiles)
    double cold[2][4][2];
    if (MODE == GEMM_SUB && wave == 0) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gc = col0 + ni * 16 + lk + 4 * r, gcc = gc < N ? gc : N - 1;
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const int gr = row0 + mi * 16 + li, grc = gr < M ? gr : M - 1;
                    cold[ni][r][mi] = C[(int64_t)gcc * ldc + grc];
                }
            }
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gc = col0 + ni * 16 + lk + 4 * r;
            if (gc >= N) continue;
            double *cc = C + (int64_t)gc * ldc;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int gr = row0 + mi * 16 + li;
                if (gr >= M) continue;
                const double v = ((acc[mi][ni][r] + part[0][mi * 2 + ni][lane][r]) +
                                  (part[1][mi * 2 + ni][lane][r] + part[2][mi * 2 + ni][lane][r]));
                if (MODE == GEMM_SUB) cc[gr] = cold[ni][r][mi] - v;
                else if (MODE == GEMM_SET) cc[gr] = v;
                else cc[gr] = -v;
            }
        }
}

template <int MODE>
__global__ __launch_bounds__(256) void gemm_small_kernel(double *__restrict__ C, int64_t ldc,
                                                         const double *__restrict__ A, int64_t lda,
                                                         const double *__restrict__ B, int64_t ldb,
                                                         int M, int N, int K) {
    gemm_small_body<MODE>(C, ldc, A, lda, B, ldb, M, N, K);
}

// two independent small products in one launch (blockIdx.z picks the problem)
template <int MODE>
__global__ __launch_bounds__(256) void gemm_small_pair_kernel(GemmProblem p0, GemmProblem p1) {
    const GemmProblem &p = blockIdx.z == 0 ? p0 : p1;
    if ((int)blockIdx.x * 32 >= p.M || (int)blockIdx.y * 32 >= p.N) return;
    gemm_small_body<MODE>(p.C, p.ldc, p.A, p.lda, p.B, p.ldb, p.M, p.N, p.K);
}

}  // namespace

// C (op)= A * B on `stream`.  All matrices column-major, device pointers.
int gemm_f64(nodal_ctx *h, hipStream_t stream, int mode, double *C, int64_t ldc, const double *A,
             int64_t lda, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return NODAL_OK;
    if (M * N <= 512 * 1024 || (M <= 256 && K <= 256)) {
        dim3 grid((unsigned)((M + 31) / 32), (unsigned)((N + 31) / 32));
        if (mode == GEMM_SUB)
            gemm_small_kernel<GEMM_SUB><<<grid, 256, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
        else if (mode == GEMM_SET)
            gemm_small_kernel<GEMM_SET><<<grid, 256, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
        else
            gemm_small_kernel<GEMM_SETNEG><<<grid, 256, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    }
    dim3 grid((unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN)));
    if (mode == GEMM_SUB)
        gemm_sub_kernel<GEMM_SUB><<<grid, NT, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
    else if (mode == GEMM_SET)
        gemm_sub_kernel<GEMM_SET><<<grid, NT, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
    else
        gemm_sub_kernel<GEMM_SETNEG><<<grid, NT, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// C -= At^T B, upper block triangle of the square-leading C only (At: K x M, B: K x N, N >= M): the
// diagonal blocks of `band` rows (a multiple of 128, aligned with C's origin) are updated whole.
int gemm_sub_tn_upper_f64(nodal_ctx *h, hipStream_t stream, double *C, int64_t ldc, const double *At,
                          int64_t ldat, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K, int band) {
    if (M <= 0 || N <= 0 || K <= 0) return NODAL_OK;
    if (band < BM || band % BM) return nodal_fail(h, NODAL_E_INVALID, "gemm_sub_tn_upper: band must be a multiple of 128");
    dim3 grid((unsigned)(((M + BM - 1) / BM) * ((N + BN - 1) / BN)));
    gemm_sub_kernel<GEMM_SUB, true, true><<<grid, NT, 0, stream>>>(C, ldc, At, ldat, B, ldb, (int)M, (int)N, (int)K, band);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

int gemm_sub_f64(nodal_ctx *h, hipStream_t stream, double *C, int64_t ldc, const double *A,
                 int64_t lda, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    return gemm_f64(h, stream, GEMM_SUB, C, ldc, A, lda, B, ldb, M, N, K);
}

// Two independent small products (M, N <= 256 each) in ONE launch.
int gemm_pair_f64(nodal_ctx *h, hipStream_t stream, int mode, const GemmProblem &p0,
                  const GemmProblem &p1) {
    const int mx = p0.M > p1.M ? p0.M : p1.M, nx = p0.N > p1.N ? p0.N : p1.N;
    if (mx <= 0 || nx <= 0) return NODAL_OK;
    dim3 grid((unsigned)((mx + 31) / 32), (unsigned)((nx + 31) / 32), 2);
    if (mode == GEMM_SUB) gemm_small_pair_kernel<GEMM_SUB><<<grid, 256, 0, stream>>>(p0, p1);
    else if (mode == GEMM_SET) gemm_small_pair_kernel<GEMM_SET><<<grid, 256, 0, stream>>>(p0, p1);
    else gemm_small_pair_kernel<GEMM_SETNEG><<<grid, 256, 0, stream>>>(p0, p1);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}
