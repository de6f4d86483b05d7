// C[M x N] -= A[M x K] * B[K x N], fp64, column-major, on the CDNA4 matrix cores
// (v_mfma_f64_16x16x4_f64).  The trailing update of the dense LU (dense_lu.hip).
// MODE selects the epilogue: GEMM_SUB  C -= A B,  GEMM_SET  C = A B,  GEMM_SETNEG  C = -A B
// (the two overwrite forms serve the block-inverse elimination of dense_lu.hip).
//
// Tiling for 64-wide wavefronts: a 256-thread workgroup (4 waves in a 2 x 2 grid)
// owns a 128 x 128 tile of C; each wave a 64 x 64 sub-tile = 4 x 4 MFMA tiles,
// i.e. 16 accumulators of 4 f64 per lane (128 VGPRs).  K is walked in chunks of 16
// staged through LDS, double buffered: while the matrix cores work on chunk k the
// global loads of chunk k+1 are in flight.
//
// MFMA f64 16x16x4 fragment layout (cdna_hip_programming.md section 3):
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D      : lane l, register r holds D[row = (l >> 4) + 4 r][col = l & 15]
// The instruction is issued with the operands SWAPPED (a = B fragment, b = A
// fragment), i.e. it accumulates the transposed tile D = (A B)^T.  Lane l,
// register r then holds C[row = l & 15][col = (l >> 4) + 4 r]: the 16 lanes of a
// quarter-wave cover 16 CONTIGUOUS rows of column-major C, so every epilogue load /
// store moves whole 128-byte lines.  (Un-swapped, each access touched 16 columns x
// 32 bytes and HBM over-fetched C about 2x: 36 TFLOP/s instead of the MFMA rate.)
// LDS images are k-major ([k][i] and [k][j]); the row stride 144 doubles puts the
// two k values a 32-lane half reads on disjoint bank halves (ds_read_b64: 64 banks).
#include "ctx.h"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA_S = 144;  // A image row stride (doubles)
constexpr int LDB_S = 145;  // B image row stride: odd/2 -> conflict-free transposing writes

template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_sub_kernel(double *__restrict__ C, int64_t ldc,
                                                          const double *__restrict__ A,
                                                          int64_t lda,
                                                          const double *__restrict__ B,
                                                          int64_t ldb, int M, int N, int K) {
    __shared__ double As[2][BK][LDA_S];
    __shared__ double Bs[2][BK][LDB_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave & 1) * 64, wn = (wave >> 1) * 64;
    const int row0 = blockIdx.x * BM, col0 = blockIdx.y * BN;
    const int li = lane & 15, lk = lane >> 4;

    v4f64 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = v4f64{0.0, 0.0, 0.0, 0.0};

    // global -> register staging of one K chunk: 8 doubles of A and 8 of B per thread
    double ra[8], rb[8];
    auto load_chunk = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int p = tid + 256 * r;           // 0 .. 2047
            const int i = p & 127, k = p >> 7;     // A image [k][i]: i contiguous in memory
            const int gi = row0 + i, gk = k0 + k;
            ra[r] = (gi < M && gk < K) ? A[(int64_t)gk * lda + gi] : 0.0;
            const int kb = p & 15, j = p >> 4;     // B: 16 contiguous k of one column
            const int gj = col0 + j, gkb = k0 + kb;
            rb[r] = (gj < N && gkb < K) ? B[(int64_t)gj * ldb + gkb] : 0.0;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int p = tid + 256 * r;
            As[buf][p >> 7][p & 127] = ra[r];
            Bs[buf][p & 15][p >> 4] = rb[r];
        }
    };

    const int nchunks = (K + BK - 1) / BK;
    load_chunk(0);
    store_chunk(0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) load_chunk((ch + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
            double af[4], bf[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) af[mi] = As[buf][ks * 4 + lk][wm + mi * 16 + li];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) bf[ni] = Bs[buf][ks * 4 + lk][wn + ni * 16 + li];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[ni], af[mi], acc[mi][ni],
                                                                      0, 0, 0);
        }
        if (ch + 1 < nchunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

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
__global__ __launch_bounds__(256) void gemm_small_kernel(double *__restrict__ C, int64_t ldc,
                                                         const double *__restrict__ A, int64_t lda,
                                                         const double *__restrict__ B, int64_t ldb,
                                                         int M, int N, int K) {
    __shared__ double part[3][4][64][4];
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
    const bool r0ok = r0 < M, r1ok = r1 < M, c0ok = c0 < N, c1ok = c1 < N;
    const double *a0 = A + r0, *a1 = A + r1;
    const double *b0 = B + (int64_t)c0 * ldb, *b1 = B + (int64_t)c1 * ldb;
#pragma unroll 16
    for (int s = s0; s < s1; ++s) {
        const int k = 4 * s + lk;
        const bool kok = k < K;
        const double af0 = (kok && r0ok) ? a0[(int64_t)k * lda] : 0.0;
        const double af1 = (kok && r1ok) ? a1[(int64_t)k * lda] : 0.0;
        const double bf0 = (kok && c0ok) ? b0[k] : 0.0;
        const double bf1 = (kok && c1ok) ? b1[k] : 0.0;
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf0, af0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf1, af0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf0, af1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf1, af1, acc[1][1], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) part[wave - 1][mi * 2 + ni][lane][r] = acc[mi][ni][r];
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
                if (MODE == GEMM_SUB) cc[gr] -= v;
                else if (MODE == GEMM_SET) cc[gr] = v;
                else cc[gr] = -v;
            }
        }
}

}  // namespace

// C (op)= A * B on `stream`.  All matrices column-major, device pointers.
int gemm_f64(nodal_ctx *h, hipStream_t stream, int mode, double *C, int64_t ldc, const double *A,
             int64_t lda, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return NODAL_OK;
    if (M <= 256 && N <= 256) {
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
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
    if (mode == GEMM_SUB)
        gemm_sub_kernel<GEMM_SUB><<<grid, 256, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
    else if (mode == GEMM_SET)
        gemm_sub_kernel<GEMM_SET><<<grid, 256, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
    else
        gemm_sub_kernel<GEMM_SETNEG><<<grid, 256, 0, stream>>>(C, ldc, A, lda, B, ldb, (int)M, (int)N, (int)K);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

int gemm_sub_f64(nodal_ctx *h, hipStream_t stream, double *C, int64_t ldc, const double *A,
                 int64_t lda, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K) {
    return gemm_f64(h, stream, GEMM_SUB, C, ldc, A, lda, B, ldb, M, N, K);
}
