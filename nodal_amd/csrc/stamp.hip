// Assembly of G (CSR) and A (rhs) from the structure-of-arrays component table.
//
// Replaces Circuit.build_model and models.write_{R,A,E,VCVS,CCVS,CCCS}
// (reference nodal/nodal.py:338-398, nodal/models.py:13-214).  The reference
// walks the components in file order and mutates G with `+=` / `=`.  Here the
// same result is produced without any floating-point atomics:
//
//   symbolic (topology only)
//     1. every component enumerates its stamps as (row, col, slot) -- at most
//        six matrix stamps and two rhs stamps, in the reference's write order;
//     2. stamps are bucketed by row (integer histogram + exclusive scan) and
//        each row bucket is sorted by (col, component, slot);
//     3. runs of equal (row, col) become one CSR entry; the run itself is the
//        entry's ordered contribution list.
//   numeric (per set of values)
//     one lane per CSR entry folds its contribution list in order with the
//     reference's semantics: ADD (x += v), SET (x = v) and the reference's
//     `assert G[i, j] == 0` checks.  Summation order = component order, so G
//     and A are bit-identical to the reference's.
//
// Stamp tables (ia, ib, ic, id = node rows, -1 = ground; m = K + k):
//   R    s0 (ia,ia)+g  s1 (ib,ib)+g  s2 (ia,ib)-g  s3 (ib,ia)-g        ADD   g = 1/v
//   E    s0 (m,ia)=1*  s1 (ia,m)=-1  s2 (m,ib)=-1* s3 (ib,m)=1         SET   rhs (m)+v
//   VCVS as E, then s4 (m,ic)+=-v  s5 (m,id)+=v                        (also VCCS)
//   CCVS as E without asserts, then s4 (m,ic)=v/Rd  s5 (m,id)=(-v)/Rd  SET
//   CCCS s0 (ia,m)=-1* s1 (ib,m)=1* s2 (m,m)=1* s3 (m,ic)=v/Rd* s4 (m,id)=(-v)/Rd*
//   A    rhs s0 (ia)+v  s1 (ib)-v
//   (* = the reference asserts the entry is zero before writing it)
#include "ctx.h"

namespace {

constexpr int TB = 256;
constexpr unsigned MAX_GRID = 4096;

struct Table {
    const uint8_t *type;
    const double *value;
    const int32_t *a, *b, *c, *d, *drv, *k;
    int64_t ncomp;
    int32_t K;
};

__host__ __device__ inline unsigned grid_for(int64_t n) {
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > MAX_GRID ? MAX_GRID : g);
}

// ---- stamp enumeration -------------------------------------------------------

struct MatrixStamps {
    static constexpr int SLOTS = 6;
    // position of slot `s` of a component, false if that stamp does not exist
    __device__ static bool at(int t, int s, int ia, int ib, int ic, int id, int m, int &row,
                              int &col) {
        switch (t) {
        case NODAL_T_R:
            if (s == 0) { row = ia; col = ia; return ia >= 0; }
            if (s == 1) { row = ib; col = ib; return ib >= 0; }
            if (s == 2) { row = ia; col = ib; return ia >= 0 && ib >= 0; }
            if (s == 3) { row = ib; col = ia; return ia >= 0 && ib >= 0; }
            return false;
        case NODAL_T_E:
        case NODAL_T_VCVS:
        case NODAL_T_CCVS:
            if (s == 0) { row = m; col = ia; return ia >= 0; }
            if (s == 1) { row = ia; col = m; return ia >= 0; }
            if (s == 2) { row = m; col = ib; return ib >= 0; }
            if (s == 3) { row = ib; col = m; return ib >= 0; }
            if (t == NODAL_T_E) return false;
            if (s == 4) { row = m; col = ic; return ic >= 0; }
            if (s == 5) { row = m; col = id; return id >= 0; }
            return false;
        case NODAL_T_CCCS:
            if (s == 0) { row = ia; col = m; return ia >= 0; }
            if (s == 1) { row = ib; col = m; return ib >= 0; }
            if (s == 2) { row = m; col = m; return true; }
            if (s == 3) { row = m; col = ic; return ic >= 0; }
            if (s == 4) { row = m; col = id; return id >= 0; }
            return false;
        default:
            return false;
        }
    }
};

struct RhsStamps {
    static constexpr int SLOTS = 2;
    __device__ static bool at(int t, int s, int ia, int ib, int, int, int m, int &row, int &col) {
        col = 0;
        if (t == NODAL_T_A) {
            if (s == 0) { row = ia; return ia >= 0; }
            row = ib;
            return ib >= 0;
        }
        if (t == NODAL_T_E && s == 0) { row = m; return true; }
        return false;
    }
};

constexpr unsigned F_SET = 1u, F_ASSERT0 = 2u;

// value and semantics of matrix slot s (v = component value, Rd = driver's)
__device__ __forceinline__ double matrix_value(int t, int s, double v, double Rd,
                                               unsigned &flags) {
    switch (t) {
    case NODAL_T_R: {
        const double g = 1.0 / v;
        flags = 0;
        return s < 2 ? g : -g;
    }
    case NODAL_T_E:
    case NODAL_T_VCVS:
        if (s < 4) {
            flags = F_SET | ((s == 0 || s == 2) ? F_ASSERT0 : 0u);
            return (s == 0 || s == 3) ? 1.0 : -1.0;
        }
        flags = 0;
        return s == 4 ? -v : v;
    case NODAL_T_CCVS:
        flags = F_SET;
        if (s < 4) return (s == 0 || s == 3) ? 1.0 : -1.0;
        return s == 4 ? v / Rd : (-v) / Rd;
    default:  // CCCS
        flags = F_SET | F_ASSERT0;
        if (s == 0) return -1.0;
        if (s == 1 || s == 2) return 1.0;
        return s == 3 ? v / Rd : (-v) / Rd;
    }
}

__device__ __forceinline__ double rhs_value(int t, int s, double v) {
    if (t == NODAL_T_A) return s == 0 ? v : -v;
    return v;  // E
}

template <class Stamps>
__global__ __launch_bounds__(TB) void count_rows(Table tb, uint32_t *__restrict__ rowcount) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < tb.ncomp;
         i += (int64_t)gridDim.x * TB) {
        const int t = tb.type[i];
        const int ia = tb.a[i], ib = tb.b[i], ic = tb.c[i], id = tb.d[i];
        const int kk = tb.k[i];
        const int m = kk >= 0 ? tb.K + kk : -1;
#pragma unroll
        for (int s = 0; s < Stamps::SLOTS; ++s) {
            int row, col;
            if (Stamps::at(t, s, ia, ib, ic, id, m, row, col)) atomicAdd(&rowcount[row], 1u);
        }
    }
}

// key = col << 32 | comp << 3 | slot : sorting a row bucket by key orders it by
// column, then by the reference's stamping order.
template <class Stamps>
__global__ __launch_bounds__(TB) void emit_stamps(Table tb, const uint32_t *__restrict__ rowstart,
                                                  uint32_t *__restrict__ fill,
                                                  uint64_t *__restrict__ skey,
                                                  int32_t *__restrict__ srow) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < tb.ncomp;
         i += (int64_t)gridDim.x * TB) {
        const int t = tb.type[i];
        const int ia = tb.a[i], ib = tb.b[i], ic = tb.c[i], id = tb.d[i];
        const int kk = tb.k[i];
        const int m = kk >= 0 ? tb.K + kk : -1;
#pragma unroll
        for (int s = 0; s < Stamps::SLOTS; ++s) {
            int row, col;
            if (Stamps::at(t, s, ia, ib, ic, id, m, row, col)) {
                const uint32_t p = rowstart[row] + atomicAdd(&fill[row], 1u);
                skey[p] = ((uint64_t)(uint32_t)col << 32) | ((uint64_t)i << 3) | (uint64_t)s;
                srow[p] = row;
            }
        }
    }
}

// ---- per-row sorts -------------------------------------------------------------

template <int N>
__device__ __forceinline__ void sort_network(uint64_t (&k)[N]) {
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ stride;
                if (l > i) {
                    const bool up = (i & size) == 0;
                    const uint64_t lo = k[i] < k[l] ? k[i] : k[l];
                    const uint64_t hi = k[i] < k[l] ? k[l] : k[i];
                    k[i] = up ? lo : hi;
                    k[l] = up ? hi : lo;
                }
            }
        }
    }
}

template <int N>
__device__ __forceinline__ void sort_short_row(uint64_t *seg, int len) {
    uint64_t k[N];
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = i < len ? seg[i] : ~0ull;
    sort_network<N>(k);
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < len) seg[i] = k[i];
}

constexpr int SHORT_MAX = 16;
constexpr int MEDIUM_MAX = 2048;

// one lane per row; rows longer than SHORT_MAX are appended to work lists
__global__ __launch_bounds__(TB) void sort_rows_short(const uint32_t *__restrict__ rowstart,
                                                      uint64_t *__restrict__ skey, int64_t nrows,
                                                      int32_t *__restrict__ medium_list,
                                                      int32_t *__restrict__ long_list,
                                                      uint32_t *__restrict__ list_counts) {
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < nrows;
         r += (int64_t)gridDim.x * TB) {
        const uint32_t s = rowstart[r];
        const int len = (int)(rowstart[r + 1] - s);
        if (len < 2) continue;
        if (len <= 4) sort_short_row<4>(skey + s, len);
        else if (len <= 8) sort_short_row<8>(skey + s, len);
        else if (len <= SHORT_MAX) sort_short_row<16>(skey + s, len);
        else if (len <= MEDIUM_MAX) medium_list[atomicAdd(&list_counts[0], 1u)] = (int32_t)r;
        else long_list[atomicAdd(&list_counts[1], 1u)] = (int32_t)r;
    }
}

// one workgroup per listed row, bitonic sort in LDS (len <= MEDIUM_MAX)
__global__ __launch_bounds__(TB) void sort_rows_medium(const uint32_t *__restrict__ rowstart,
                                                       uint64_t *__restrict__ skey,
                                                       const int32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ list_counts) {
    __shared__ uint64_t buf[MEDIUM_MAX];
    const uint32_t count = list_counts[0];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        const int32_t r = list[it];
        const uint32_t s = rowstart[r];
        const int len = (int)(rowstart[r + 1] - s);
        int P = 32;
        while (P < len) P <<= 1;
        for (int i = threadIdx.x; i < P; i += TB) buf[i] = i < len ? skey[s + i] : ~0ull;
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = threadIdx.x; i < P; i += TB) {
                    const int l = i ^ stride;
                    if (l > i) {
                        const bool up = (i & size) == 0;
                        const uint64_t x = buf[i], y = buf[l];
                        if ((x > y) == up) { buf[i] = y; buf[l] = x; }
                    }
                }
                __syncthreads();
            }
        }
        for (int i = threadIdx.x; i < len; i += TB) skey[s + i] = buf[i];
        __syncthreads();
    }
}

// one workgroup per listed row, bitonic sort through a padded global scratch
// region [2*rowstart, 2*rowstart + P): regions of different rows are disjoint
// because P < 2*len.  Only hub nodes with > MEDIUM_MAX stamps get here.
__global__ __launch_bounds__(1024) void sort_rows_long(const uint32_t *__restrict__ rowstart,
                                                       uint64_t *__restrict__ skey,
                                                       uint64_t *__restrict__ scratch,
                                                       const int32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ list_counts) {
    const uint32_t count = list_counts[1];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        const int32_t r = list[it];
        const uint32_t s = rowstart[r];
        const int64_t len = (int64_t)rowstart[r + 1] - s;
        int64_t P = 1;
        while (P < len) P <<= 1;
        uint64_t *buf = scratch + 2 * (int64_t)s;
        for (int64_t i = threadIdx.x; i < P; i += 1024) buf[i] = i < len ? skey[s + i] : ~0ull;
        __syncthreads();
        for (int64_t size = 2; size <= P; size <<= 1) {
            for (int64_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (int64_t i = threadIdx.x; i < P; i += 1024) {
                    const int64_t l = i ^ stride;
                    if (l > i) {
                        const bool up = (i & size) == 0;
                        const uint64_t x = buf[i], y = buf[l];
                        if ((x > y) == up) { buf[i] = y; buf[l] = x; }
                    }
                }
                __syncthreads();  // one workgroup: global writes are visible after the barrier
            }
        }
        for (int64_t i = threadIdx.x; i < len; i += 1024) skey[s + i] = buf[i];
        __syncthreads();
    }
}

// ---- runs of equal (row, col) -> CSR entries -------------------------------------

__global__ __launch_bounds__(TB) void mark_heads(const uint64_t *__restrict__ skey,
                                                 const int32_t *__restrict__ srow,
                                                 const uint32_t *__restrict__ rowstart,
                                                 uint32_t *__restrict__ head, int64_t C) {
    for (int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x; p < C; p += (int64_t)gridDim.x * TB) {
        const bool first = p == (int64_t)rowstart[srow[p]];
        head[p] = (first || (skey[p] >> 32) != (skey[p - 1] >> 32)) ? 1u : 0u;
    }
}

__global__ __launch_bounds__(TB) void fill_entries(const uint64_t *__restrict__ skey,
                                                   const int32_t *__restrict__ srow,
                                                   const uint32_t *__restrict__ head,
                                                   const uint32_t *__restrict__ eidx, int64_t C,
                                                   int32_t *__restrict__ indices,
                                                   int32_t *__restrict__ rowidx,
                                                   int32_t *__restrict__ cptr,
                                                   uint32_t *__restrict__ contrib,
                                                   int32_t *__restrict__ diag_pos) {
    for (int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x; p < C; p += (int64_t)gridDim.x * TB) {
        const uint64_t key = skey[p];
        contrib[p] = (uint32_t)key;
        if (head[p]) {
            const uint32_t e = eidx[p];
            const int32_t col = (int32_t)(key >> 32), row = srow[p];
            if (indices) indices[e] = col;
            rowidx[e] = row;
            cptr[e] = (int32_t)p;
            if (diag_pos && col == row) diag_pos[row] = (int32_t)e;
        }
    }
}

// indptr[r] = entry index of the first stamp of row r (eidx has C+1 values)
__global__ __launch_bounds__(TB) void fill_indptr(const uint32_t *__restrict__ rowstart,
                                                  const uint32_t *__restrict__ eidx,
                                                  int32_t *__restrict__ indptr, int64_t nrows) {
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r <= nrows;
         r += (int64_t)gridDim.x * TB)
        indptr[r] = (int32_t)eidx[rowstart[r]];
}

__global__ void set_tail(int32_t *cptr, int64_t nent, int64_t C) { cptr[nent] = (int32_t)C; }

__global__ __launch_bounds__(TB) void fill_i32(int32_t *p, int32_t v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        p[i] = v;
}

// ---- numeric folds ---------------------------------------------------------------

__device__ __forceinline__ void note_min(unsigned long long *slot, int64_t comp) {
    atomicMin(slot, (unsigned long long)comp);
}

__global__ __launch_bounds__(TB) void fold_matrix(Table tb, const double *__restrict__ value,
                                                  const int32_t *__restrict__ cptr,
                                                  const uint32_t *__restrict__ contrib,
                                                  double *__restrict__ data, int64_t nnz,
                                                  unsigned long long *__restrict__ status) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < nnz;
         e += (int64_t)gridDim.x * TB) {
        const int32_t p0 = cptr[e], p1 = cptr[e + 1];
        double x = 0.0;
        for (int32_t p = p0; p < p1; ++p) {
            const uint32_t u = contrib[p];
            const int64_t comp = u >> 3;
            const int s = (int)(u & 7u);
            const int t = tb.type[comp];
            const double v = value[comp];
            double Rd = 1.0;
            if (t >= NODAL_T_CCVS && s >= 3) {
                const int32_t dr = tb.drv[comp];
                if (dr >= 0) Rd = value[dr];
            }
            if (t == NODAL_T_R && v == 0.0) note_min(&status[0], comp);
            if (t == NODAL_T_R && !(v > 0.0)) status[2] = 1;  // not a passive network (benign race)
            unsigned flags;
            const double val = matrix_value(t, s, v, Rd, flags);
            if ((flags & F_ASSERT0) && x != 0.0) note_min(&status[1], comp);
            x = (flags & F_SET) ? val : x + val;
        }
        data[e] = x;
    }
}

__global__ __launch_bounds__(TB) void fold_rhs(Table tb, const double *__restrict__ value,
                                               const int32_t *__restrict__ rhs_row,
                                               const int32_t *__restrict__ cptr,
                                               const uint32_t *__restrict__ contrib,
                                               double *__restrict__ rhs, int64_t nent) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < nent;
         e += (int64_t)gridDim.x * TB) {
        const int32_t p0 = cptr[e], p1 = cptr[e + 1];
        double x = 0.0;
        for (int32_t p = p0; p < p1; ++p) {
            const uint32_t u = contrib[p];
            const int64_t comp = u >> 3;
            x += rhs_value(tb.type[comp], (int)(u & 7u), value[comp]);
        }
        rhs[rhs_row[e]] = x;
    }
}

__global__ __launch_bounds__(TB) void scatter_dense(const int32_t *__restrict__ rowidx,
                                                    const int32_t *__restrict__ indices,
                                                    const double *__restrict__ data,
                                                    double *__restrict__ G, int64_t ld, int64_t nnz,
                                                    bool col_major) {
    for (int64_t e = (int64_t)blockIdx.x * TB + threadIdx.x; e < nnz;
         e += (int64_t)gridDim.x * TB) {
        const int64_t r = rowidx[e], c = indices[e];
        G[col_major ? c * ld + r : r * ld + c] = data[e];
    }
}

Table table_of(nodal_ctx *h) {
    Table tb;
    tb.type = h->type.as<uint8_t>();
    tb.value = h->value.as<double>();
    tb.a = h->a.as<int32_t>();
    tb.b = h->b.as<int32_t>();
    tb.c = h->c.as<int32_t>();
    tb.d = h->d.as<int32_t>();
    tb.drv = h->drv.as<int32_t>();
    tb.k = h->k.as<int32_t>();
    tb.ncomp = h->ncomp;
    tb.K = h->K;
    return tb;
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Group one family of stamps (matrix or rhs) into entries.  On return:
//   *nent entries, *ncon contributions; rowidx / cptr / contrib filled;
//   indices, indptr, diag_pos filled when non-null.
template <class Stamps>
int build_lists(nodal_ctx *h, int64_t nrows, int64_t *nent_out, int64_t *ncon_out, DevBuf &indices,
                DevBuf &rowidx, DevBuf &cptr, DevBuf &contrib, DevBuf *indptr, DevBuf *diag_pos) {
    const Table tb = table_of(h);
    hipStream_t st = h->stream;

    // work layout: rowcount/rowstart [nrows+1] | fill [nrows] | counts[4] | scan tmp
    const size_t off_start = 0;
    const size_t off_fill = align_up((size_t)(nrows + 1) * 4);
    const size_t off_counts = off_fill + align_up((size_t)nrows * 4);
    const size_t off_scan = off_counts + 256;
    const size_t scan_bytes = scan_tmp_bytes(nrows + 1);
    NODAL_HIP_TRY(h, h->work.reserve(off_scan + scan_bytes));
    char *w = h->work.as<char>();
    uint32_t *rowstart = reinterpret_cast<uint32_t *>(w + off_start);
    uint32_t *fill = reinterpret_cast<uint32_t *>(w + off_fill);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w + off_counts);  // [0] medium [1] long [2] C [3] nent
    NODAL_HIP_TRY(h, hipMemsetAsync(w, 0, off_scan, st));

    count_rows<Stamps><<<grid_for(tb.ncomp), TB, 0, st>>>(tb, rowstart);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_TRY(scan_exclusive_u32(h, rowstart, rowstart, nrows + 1, &counts[2], w + off_scan));
    uint32_t C32 = 0;
    NODAL_HIP_TRY(h, hipMemcpyAsync(&C32, &counts[2], 4, hipMemcpyDeviceToHost, st));
    NODAL_HIP_TRY(h, hipStreamSynchronize(st));
    const int64_t C = C32;
    if (C > 0x7fffffffll)
        return nodal_fail(h, NODAL_E_UNSUPPORTED, "more than 2^31 stamps");
    *ncon_out = C;
    if (C == 0) {
        *nent_out = 0;
        NODAL_HIP_TRY(h, cptr.reserve(4));
        NODAL_HIP_TRY(h, hipMemsetAsync(cptr.p, 0, 4, st));
        if (indptr) {
            NODAL_HIP_TRY(h, indptr->reserve((size_t)(nrows + 1) * 4));
            NODAL_HIP_TRY(h, hipMemsetAsync(indptr->p, 0, (size_t)(nrows + 1) * 4, st));
        }
        if (diag_pos) {
            NODAL_HIP_TRY(h, diag_pos->reserve((size_t)nrows * 4 + 4));
            fill_i32<<<grid_for(nrows), TB, 0, st>>>(diag_pos->as<int32_t>(), -1, nrows);
        }
        return NODAL_OK;
    }

    // work2 layout: skey [C] u64 | srow [C] i32 | head/eidx [C+1] u32 | lists 2 x [nrows] | scan tmp
    const size_t o_key = 0;
    const size_t o_row = o_key + align_up((size_t)C * 8);
    const size_t o_head = o_row + align_up((size_t)C * 4);
    const size_t o_eidx = o_head + align_up((size_t)(C + 1) * 4);
    const size_t o_med = o_eidx + align_up((size_t)(C + 1) * 4);
    const size_t o_long = o_med + align_up((size_t)nrows * 4);
    const size_t o_scan2 = o_long + align_up((size_t)nrows * 4);
    NODAL_HIP_TRY(h, h->work2.reserve(o_scan2 + scan_tmp_bytes(C + 1)));
    char *w2 = h->work2.as<char>();
    uint64_t *skey = reinterpret_cast<uint64_t *>(w2 + o_key);
    int32_t *srow = reinterpret_cast<int32_t *>(w2 + o_row);
    uint32_t *head = reinterpret_cast<uint32_t *>(w2 + o_head);
    uint32_t *eidx = reinterpret_cast<uint32_t *>(w2 + o_eidx);
    int32_t *medium_list = reinterpret_cast<int32_t *>(w2 + o_med);
    int32_t *long_list = reinterpret_cast<int32_t *>(w2 + o_long);

    emit_stamps<Stamps><<<grid_for(tb.ncomp), TB, 0, st>>>(tb, rowstart, fill, skey, srow);
    NODAL_HIP_TRY(h, hipGetLastError());
    sort_rows_short<<<grid_for(nrows), TB, 0, st>>>(rowstart, skey, nrows, medium_list, long_list,
                                                   counts);
    NODAL_HIP_TRY(h, hipGetLastError());
    uint32_t lc[2] = {0, 0};
    NODAL_HIP_TRY(h, hipMemcpyAsync(lc, counts, 8, hipMemcpyDeviceToHost, st));
    NODAL_HIP_TRY(h, hipStreamSynchronize(st));
    if (lc[0] > 0) {
        sort_rows_medium<<<lc[0] > 2048 ? 2048 : lc[0], TB, 0, st>>>(rowstart, skey, medium_list,
                                                                    counts);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    if (lc[1] > 0) {
        NODAL_HIP_TRY(h, h->solver.reserve((size_t)C * 16));  // borrowed as padded scratch
        sort_rows_long<<<lc[1] > 256 ? 256 : lc[1], 1024, 0, st>>>(
            rowstart, skey, h->solver.as<uint64_t>(), long_list, counts);
        NODAL_HIP_TRY(h, hipGetLastError());
    }

    mark_heads<<<grid_for(C), TB, 0, st>>>(skey, srow, rowstart, head, C);
    NODAL_HIP_TRY(h, hipGetLastError());
    // scanned over C+1 slots: slot C (zero) receives the number of entries
    NODAL_HIP_TRY(h, hipMemsetAsync(head + C, 0, 4, st));
    NODAL_TRY(scan_exclusive_u32(h, head, eidx, C + 1, nullptr, w2 + o_scan2));
    uint32_t nent32 = 0;
    NODAL_HIP_TRY(h, hipMemcpyAsync(&nent32, eidx + C, 4, hipMemcpyDeviceToHost, st));
    NODAL_HIP_TRY(h, hipStreamSynchronize(st));
    const int64_t nent = nent32;
    *nent_out = nent;

    NODAL_HIP_TRY(h, indices.reserve((size_t)nent * 4 + 4));
    NODAL_HIP_TRY(h, rowidx.reserve((size_t)nent * 4 + 4));
    NODAL_HIP_TRY(h, cptr.reserve((size_t)(nent + 1) * 4));
    NODAL_HIP_TRY(h, contrib.reserve((size_t)C * 4));
    if (diag_pos) {
        NODAL_HIP_TRY(h, diag_pos->reserve((size_t)nrows * 4 + 4));
        fill_i32<<<grid_for(nrows), TB, 0, st>>>(diag_pos->as<int32_t>(), -1, nrows);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    fill_entries<<<grid_for(C), TB, 0, st>>>(skey, srow, head, eidx, C, indices.as<int32_t>(),
                                            rowidx.as<int32_t>(), cptr.as<int32_t>(),
                                            contrib.as<uint32_t>(),
                                            diag_pos ? diag_pos->as<int32_t>() : nullptr);
    NODAL_HIP_TRY(h, hipGetLastError());
    set_tail<<<1, 1, 0, st>>>(cptr.as<int32_t>(), nent, C);
    NODAL_HIP_TRY(h, hipGetLastError());
    if (indptr) {
        NODAL_HIP_TRY(h, indptr->reserve((size_t)(nrows + 1) * 4));
        fill_indptr<<<grid_for(nrows + 1), TB, 0, st>>>(rowstart, eidx, indptr->as<int32_t>(), nrows);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    return NODAL_OK;
}

}  // namespace

int stamp_symbolic(nodal_ctx *h) {
    if (!h->have_table) return nodal_fail(h, NODAL_E_INVALID, "no component table uploaded");
    h->have_symbolic = h->have_numeric = h->have_x = false;
    const int64_t n = h->n;
    if (h->ncomp >= (1ll << 29) || n >= (1ll << 31) - 2)
        return nodal_fail(h, NODAL_E_UNSUPPORTED, "component table too large for 32-bit stamp keys");
    NODAL_TRY(build_lists<MatrixStamps>(h, n, &h->nnz, &h->ncontrib, h->indices, h->rowidx, h->cptr,
                                        h->contrib, &h->indptr, &h->diag_pos));
    DevBuf none;  // rhs entries have no column index
    NODAL_TRY(build_lists<RhsStamps>(h, n, &h->nrhs, &h->nrhs_contrib, none, h->rhs_row,
                                     h->rhs_cptr, h->rhs_contrib, nullptr, nullptr));
    none.release();
    NODAL_HIP_TRY(h, h->data.reserve((size_t)h->nnz * 8 + 8));
    NODAL_HIP_TRY(h, h->rhs.reserve((size_t)n * 8 + 8));
    NODAL_HIP_TRY(h, h->x.reserve((size_t)n * 8 + 8));
    NODAL_HIP_TRY(h, h->status.reserve(64));
    h->have_symbolic = true;
    return NODAL_OK;
}

int stamp_numeric(nodal_ctx *h, int32_t member, int64_t *bad_component) {
    if (!h->have_symbolic) return nodal_fail(h, NODAL_E_INVALID, "assemble_symbolic not called");
    if (member < 0 || (h->batch > 0 && member >= h->batch) || (h->batch == 0 && member != 0))
        return nodal_fail(h, NODAL_E_INVALID, "batch member out of range");
    hipStream_t st = h->stream;
    const Table tb = table_of(h);
    const double *value =
        h->batch > 0 ? h->values_batch.as<double>() + (int64_t)member * h->ncomp : tb.value;
    unsigned long long *status = h->status.as<unsigned long long>();
    NODAL_HIP_TRY(h, hipMemsetAsync(status, 0xff, 16, st));
    NODAL_HIP_TRY(h, hipMemsetAsync(status + 2, 0, 8, st));
    NODAL_HIP_TRY(h, hipMemsetAsync(h->rhs.p, 0, (size_t)h->n * 8, st));
    if (h->nnz > 0) {
        fold_matrix<<<grid_for(h->nnz), TB, 0, st>>>(tb, value, h->cptr.as<int32_t>(),
                                                    h->contrib.as<uint32_t>(),
                                                    h->data.as<double>(), h->nnz, status);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    if (h->nrhs > 0) {
        fold_rhs<<<grid_for(h->nrhs), TB, 0, st>>>(tb, value, h->rhs_row.as<int32_t>(),
                                                  h->rhs_cptr.as<int32_t>(),
                                                  h->rhs_contrib.as<uint32_t>(),
                                                  h->rhs.as<double>(), h->nrhs);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    unsigned long long st_host[3];
    NODAL_HIP_TRY(h, hipMemcpyAsync(st_host, status, 24, hipMemcpyDeviceToHost, st));
    NODAL_HIP_TRY(h, hipStreamSynchronize(st));
    h->have_numeric = true;
    h->have_x = false;
    // resistors and current sources only, every resistance positive: G is a column
    // diagonally dominant M-matrix (used by the dense LU to skip the pivot search)
    h->passive_network = (h->B == 0) && st_host[2] == 0;
    const unsigned long long none = ~0ull;
    if (st_host[0] != none || st_host[1] != none) {
        // the reference stops at the first offending component in file order
        const bool zero_first = st_host[0] <= st_host[1];
        if (bad_component) *bad_component = (int64_t)(zero_first ? st_host[0] : st_host[1]);
        h->have_numeric = false;
        return nodal_fail(h, zero_first ? NODAL_E_ZERO_RESISTANCE : NODAL_E_STAMP_COLLISION,
                          zero_first ? "resistor with null resistance"
                                     : "stamp collision: entry asserted zero was already written");
    }
    return NODAL_OK;
}

int stamp_to_dense(nodal_ctx *h, double *G_dev, int64_t ld, bool col_major) {
    if (!h->have_numeric) return nodal_fail(h, NODAL_E_INVALID, "assemble_numeric not called");
    const int64_t n = h->n;
    NODAL_HIP_TRY(h, hipMemsetAsync(G_dev, 0, (size_t)ld * n * 8, h->stream));
    if (h->nnz > 0) {
        scatter_dense<<<grid_for(h->nnz), TB, 0, h->stream>>>(h->rowidx.as<int32_t>(),
                                                             h->indices.as<int32_t>(),
                                                             h->data.as<double>(), G_dev, ld, h->nnz,
                                                             col_major);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    return NODAL_OK;
}
