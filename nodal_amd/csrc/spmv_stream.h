// CSR-stream row kernels (Greathouse & Daga's CSR-Stream idea, wave64 flavour).
//
// Circuit matrices have short rows (5 entries per row on a grid).  Giving a row to
// a lane -- or to a sub-wave group -- makes the wave touch the value/index arrays
// in 40-60 byte pieces and HBM over-fetches 1.3-2.8x (rocprofv3 FETCH_SIZE, see
// DESIGN.md).  Here a 256-thread workgroup owns 256 consecutive rows = ONE
// contiguous range of entries: every lane multiplies consecutive entries (fully
// coalesced 8-byte value and 4-byte index loads), the products are staged in LDS,
// and after a barrier lane t sums row t's products from LDS and runs the row
// epilogue.  Row blocks with more than STREAM_CAP entries (hub nodes) are walked in chunks,
// long rows summed by the whole workgroup.
//
//   entry(e, col, val) -> double   value staged for entry e (e.g. val * x[col]); `data` may be
//                                  null (pattern-only sums: val = 0)
//   row(r, sum)                    consumes the sum of the staged values of row r
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
namespace stream {

constexpr int TB = 256;           // threads = rows per block
constexpr int STREAM_CAP = 3072;  // staged entries per row block (24 KB of LDS)
constexpr int LONG_PART = 96;     // longer parts of a row inside a chunk are summed by the whole workgroup

inline unsigned grid_for_rows(int64_t n, unsigned cap = 8192) {  // (a multiple of 8 from 64 blocks on: below)
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    if (g >= 64) g = (g + 7) & ~(int64_t)7;
    return (unsigned)(g > cap ? cap : g);
}

// Workgroup b runs on XCD b % 8 (one L2 each): row blocks are walked under a virtual block number
// that gives XCD k the contiguous eighth [k G/8, (k+1) G/8) of the grid, so the vector entries a
// row block gathers from its neighbours are mostly lines its own L2 already holds.
__device__ __forceinline__ unsigned xcd_block() {
    const unsigned g = gridDim.x, b = blockIdx.x;
    return (g & 7u) ? b : (b & 7u) * (g >> 3) + (b >> 3);
}

template <class EntryF, class RowF>
__device__ __forceinline__ void for_rows(const int32_t *__restrict__ indptr,
                                         const int32_t *__restrict__ indices,
                                         const double *__restrict__ data, int64_t n,
                                         EntryF entry, RowF row) {
    __shared__ double staged[STREAM_CAP];
    const int64_t nblocks = (n + TB - 1) / TB;
    for (int64_t blk = xcd_block(); blk < nblocks; blk += gridDim.x) {
        const int64_t r0 = blk * TB;
        const int64_t r1 = r0 + TB < n ? r0 + TB : n;
        const int32_t e0 = indptr[r0], e1 = indptr[r1];
        const int64_t r = r0 + threadIdx.x;
        if (e1 - e0 <= STREAM_CAP) {  // uniform over the workgroup
            for (int32_t e = e0 + (int32_t)threadIdx.x; e < e1; e += TB)
                staged[e - e0] = entry(e, indices[e], data ? data[e] : 0.0);
            __syncthreads();
            if (r < r1) {
                double s = 0.0;
                const int32_t a = indptr[r] - e0, b = indptr[r + 1] - e0;
                for (int32_t p = a; p < b; ++p) s += staged[p];
                row(r, s);
            }
            __syncthreads();
        } else {
            // A row block with more than STREAM_CAP entries holds a hub (a node with thousands of
            // neighbours).  The entry range is walked in chunks of STREAM_CAP: lane t keeps the
            // running sum of row t; the part of a row inside a chunk is summed by its lane if it
            // is short and by the whole workgroup if it is long (a lane walking 5000 staged values
            // alone made every row kernel ~1 ms).
            constexpr int MAX_LONG = STREAM_CAP / LONG_PART;  // long parts that fit in one chunk
            __shared__ int long_lane[MAX_LONG], long_a[MAX_LONG], long_b[MAX_LONG];  // (small: LDS sets the occupancy)
            __shared__ int long_count;
            __shared__ double wave_part[TB / 64];
            const int32_t ra = r < r1 ? indptr[r] : e1, rb = r < r1 ? indptr[r + 1] : e1;
            double acc = 0.0;
            for (int32_t c0 = e0; c0 < e1; c0 += STREAM_CAP) {
                const int32_t c1 = c0 + STREAM_CAP < e1 ? c0 + STREAM_CAP : e1;
                if (threadIdx.x == 0) long_count = 0;
                for (int32_t e = c0 + (int32_t)threadIdx.x; e < c1; e += TB)
                    staged[e - c0] = entry(e, indices[e], data ? data[e] : 0.0);
                __syncthreads();
                const int32_t a = ra > c0 ? ra : c0, b = rb < c1 ? rb : c1;
                if (b - a > LONG_PART) {
                    const int k = atomicAdd(&long_count, 1);  // (order irrelevant: every row is summed on its own)
                    long_lane[k] = (int)threadIdx.x;
                    long_a[k] = a - c0;
                    long_b[k] = b - c0;
                } else {
                    for (int32_t p = a; p < b; ++p) acc += staged[p - c0];
                }
                __syncthreads();
                const int nlong = long_count;
                for (int k = 0; k < nlong; ++k) {
                    double part = 0.0;
                    for (int p = long_a[k] + (int)threadIdx.x; p < long_b[k]; p += TB) part += staged[p];
#pragma unroll
                    for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
                    if ((threadIdx.x & 63) == 0) wave_part[threadIdx.x >> 6] = part;
                    __syncthreads();
                    if ((int)threadIdx.x == long_lane[k]) {
#pragma unroll
                        for (int w = 0; w < TB / 64; ++w) acc += wave_part[w];
                    }
                    __syncthreads();
                }
                __syncthreads();  // everyone has read long_count / staged before the next chunk resets them
            }
            if (r < r1) row(r, acc);
        }
    }
}

}  // namespace stream
}  // namespace
