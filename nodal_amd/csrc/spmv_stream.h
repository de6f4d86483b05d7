// CSR-stream row kernels (Greathouse & Daga's CSR-Stream idea, wave64 flavour).
//
// Circuit matrices have short rows (5 entries per row on a grid).  Giving a row to
// a lane -- or to a sub-wave group -- makes the wave touch the value/index arrays
// in 40-60 byte pieces and HBM over-fetches 1.3-2.8x (rocprofv3 FETCH_SIZE, see
// DESIGN.md).  Here a 256-thread workgroup owns 256 consecutive rows = ONE
// contiguous range of entries: every lane multiplies consecutive entries (fully
// coalesced 8-byte value and 4-byte index loads), the products are staged in LDS,
// and after a barrier lane t sums row t's products from LDS and runs the row
// epilogue.  Row blocks with more than STREAM_CAP entries (hub nodes) fall back to
// a lane-per-row loop.
//
//   entry(e, col, val) -> double   value staged for entry e (e.g. val * x[col])
//   row(r, sum)                    consumes the sum of the staged values of row r
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
namespace stream {

constexpr int TB = 256;           // threads = rows per block
constexpr int STREAM_CAP = 3072;  // staged entries per row block (24 KB of LDS)

inline unsigned grid_for_rows(int64_t n, unsigned cap = 8192) {
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > cap ? cap : g);
}

template <class EntryF, class RowF>
__device__ __forceinline__ void for_rows(const int32_t *__restrict__ indptr,
                                         const int32_t *__restrict__ indices,
                                         const double *__restrict__ data, int64_t n,
                                         EntryF entry, RowF row) {
    __shared__ double staged[STREAM_CAP];
    const int64_t nblocks = (n + TB - 1) / TB;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t r0 = blk * TB;
        const int64_t r1 = r0 + TB < n ? r0 + TB : n;
        const int32_t e0 = indptr[r0], e1 = indptr[r1];
        const int64_t r = r0 + threadIdx.x;
        if (e1 - e0 <= STREAM_CAP) {  // uniform over the workgroup
            for (int32_t e = e0 + (int32_t)threadIdx.x; e < e1; e += TB)
                staged[e - e0] = entry(e, indices[e], data[e]);
            __syncthreads();
            if (r < r1) {
                double s = 0.0;
                const int32_t a = indptr[r] - e0, b = indptr[r + 1] - e0;
                for (int32_t p = a; p < b; ++p) s += staged[p];
                row(r, s);
            }
            __syncthreads();
        } else if (r < r1) {
            double s = 0.0;
            for (int32_t e = indptr[r]; e < indptr[r + 1]; ++e) s += entry(e, indices[e], data[e]);
            row(r, s);
        }
    }
}

}  // namespace stream
}  // namespace
