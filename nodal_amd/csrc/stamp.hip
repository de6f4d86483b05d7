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
//   GM   s0 (ia,ic)+v  s1 (ia,id)-v  s2 (ib,ic)-v  s3 (ib,id)+v  ADD  (internal, presolve.hip)
//   (* = the reference asserts the entry is zero before writing it)
#include "ctx.h"

#include "group.h"

namespace {

using grp::TB;
using grp::grid_for;

struct Table {
    const uint8_t *type;
    const double *value;
    const int32_t *a, *b, *c, *d, *drv, *k;
    int64_t ncomp;
    int32_t K;
};

// ---- stamp enumeration -------------------------------------------------------

struct MatrixStampRule {
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
        case NODAL_T_GM:  // current v (e_c - e_d) leaves a, enters b
            if (s == 0) { row = ia; col = ic; return ia >= 0 && ic >= 0; }
            if (s == 1) { row = ia; col = id; return ia >= 0 && id >= 0; }
            if (s == 2) { row = ib; col = ic; return ib >= 0 && ic >= 0; }
            if (s == 3) { row = ib; col = id; return ib >= 0 && id >= 0; }
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

struct RhsStampRule {
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

// grouping enumerator (group.h): the stamps of component `item`, in write order
template <class Rule>
struct StampEnum {
    static constexpr int SLOTS = Rule::SLOTS;
    Table tb;
    int64_t nitems;
    template <class F>
    __device__ void for_each(int64_t i, F f) const {
        const int t = tb.type[i];
        const int ia = tb.a[i], ib = tb.b[i], ic = tb.c[i], id = tb.d[i];
        const int kk = tb.k[i];
        const int m = kk >= 0 ? tb.K + kk : -1;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            int row, col;
            if (Rule::at(t, s, ia, ib, ic, id, m, row, col)) f(s, row, col);
        }
    }
};
using MatrixStamps = StampEnum<MatrixStampRule>;
using RhsStamps = StampEnum<RhsStampRule>;

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
    case NODAL_T_GM:
        flags = 0;
        return (s == 0 || s == 3) ? v : -v;
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
            if ((t == NODAL_T_CCVS || t == NODAL_T_CCCS) && s >= 3) {
                const int32_t dr = tb.drv[comp];
                if (dr >= 0) Rd = value[dr];
            }
            if (t == NODAL_T_R && v == 0.0) note_min(&status[0], comp);
            if ((t == NODAL_T_R && !(v > 0.0)) || t == NODAL_T_GM)
                status[2] = 1;  // not a passive network (benign race)
            unsigned flags;
            const double val = matrix_value(t, s, v, Rd, flags);
            if ((flags & F_ASSERT0) && x != 0.0) note_min(&status[1], comp);
            x = (flags & F_SET) ? val : x + val;
        }
        data[e] = x;
    }
}

// The same fold in the shape north_star describes: a workgroup owns FOLD_EPB consecutive CSR entries, i.e. ONE
// contiguous range of the contribution list (the lists are in entry order).  Its lanes stream that range --
// coalesced 4-byte reads of `contrib`, then the component's record (type, value, the driver's value: consecutive
// contributions name components that sit close together in the table) --, evaluate the stamp and stage value +
// semantics in LDS; after a barrier lane e folds the run of entry e FROM LDS, in list order, with the reference's
// `+=` / `=` / assert semantics: the same operations in the same order as fold_matrix, so the same bits (the two
// are cross-checked, tests/test_gpu_parity.py), but every gather of a workgroup is in flight at once instead of
// one dependent chain per entry.  Ranges beyond FOLD_CAP contributions (a hub node's diagonal) are walked in
// chunks; a lane keeps its running value across them.
constexpr int FOLD_EPB = 256;   // entries per workgroup (one per lane)
constexpr int FOLD_CAP = 2048;  // staged contributions per chunk (18 KB of LDS)
__global__ __launch_bounds__(TB) void fold_matrix_stream(Table tb, const double *__restrict__ value,
                                                         const int32_t *__restrict__ cptr,
                                                         const uint32_t *__restrict__ contrib,
                                                         double *__restrict__ data, int64_t nnz,
                                                         unsigned long long *__restrict__ status) {
    __shared__ double sval[FOLD_CAP];
    __shared__ uint8_t sflag[FOLD_CAP];  // F_SET | F_ASSERT0
    const int64_t nblocks = (nnz + FOLD_EPB - 1) / FOLD_EPB;
    for (int64_t blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const int64_t e0 = blk * FOLD_EPB;
        const int64_t e1 = e0 + FOLD_EPB < nnz ? e0 + FOLD_EPB : nnz;
        const int64_t e = e0 + threadIdx.x;
        const int32_t c_begin = cptr[e0], c_end = cptr[e1];
        const int32_t p0 = e < e1 ? cptr[e] : c_end, p1 = e < e1 ? cptr[e + 1] : c_end;
        double x = 0.0;
        for (int32_t c0 = c_begin; c0 < c_end; c0 += FOLD_CAP) {
            const int32_t c1 = c0 + FOLD_CAP < c_end ? c0 + FOLD_CAP : c_end;
            for (int32_t p = c0 + (int32_t)threadIdx.x; p < c1; p += TB) {
                const uint32_t u = contrib[p];
                const int64_t comp = u >> 3;
                const int s = (int)(u & 7u);
                const int t = tb.type[comp];
                const double v = value[comp];
                double Rd = 1.0;
                if ((t == NODAL_T_CCVS || t == NODAL_T_CCCS) && s >= 3) {
                    const int32_t dr = tb.drv[comp];
                    if (dr >= 0) Rd = value[dr];
                }
                if (t == NODAL_T_R && v == 0.0) note_min(&status[0], comp);
                if ((t == NODAL_T_R && !(v > 0.0)) || t == NODAL_T_GM) status[2] = 1;  // not a passive network (benign race)
                unsigned flags;
                sval[p - c0] = matrix_value(t, s, v, Rd, flags);
                sflag[p - c0] = (uint8_t)flags;
            }
            __syncthreads();
            const int32_t a = p0 > c0 ? p0 : c0, b = p1 < c1 ? p1 : c1;
            for (int32_t p = a; p < b; ++p) {
                const unsigned flags = sflag[p - c0];
                const double val = sval[p - c0];
                if ((flags & F_ASSERT0) && x != 0.0) note_min(&status[1], (int64_t)(contrib[p] >> 3));
                x = (flags & F_SET) ? val : x + val;
            }
            __syncthreads();
        }
        if (e < e1) data[e] = x;
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

__global__ __launch_bounds__(TB) void grounded_flags(Table tb, uint8_t *__restrict__ flags) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < tb.ncomp;
         i += (int64_t)gridDim.x * TB) {
        if (tb.type[i] != NODAL_T_R) continue;
        const int ia = tb.a[i], ib = tb.b[i];
        if (ia >= 0 && ib < 0) flags[ia] = 1;  // benign race: every writer stores 1
        if (ib >= 0 && ia < 0) flags[ib] = 1;
    }
}

// status words of a numeric phase ([0], [1] = ~0: no offending component yet; [2] = 0: passive so far) and a zero
// right-hand side, in ONE launch (three hipMemsetAsync calls cost 7-8 us of host launch latency each)
__global__ __launch_bounds__(TB) void init_numeric(unsigned long long *__restrict__ status, double *__restrict__ rhs,
                                                   int64_t n) {
    if (blockIdx.x == 0 && threadIdx.x < 3) status[threadIdx.x] = threadIdx.x < 2 ? ~0ull : 0ull;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) rhs[i] = 0.0;
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

}  // namespace

int stamp_symbolic(nodal_ctx *h) {
    if (!h->have_table) return nodal_fail(h, NODAL_E_INVALID, "no component table uploaded");
    h->have_symbolic = h->have_numeric = h->have_x = false;
    ++h->struct_epoch;
    const int64_t n = h->n;
    const Table tb = table_of(h);
    // The sizes of the lists (entries, contributions) depend on the topology only: known from the last
    // symbolic phase of the SAME uploaded table (table_epoch), they spare the four size read-backs --
    // each one drains the stream -- of a repeated assembly.
    const bool known = h->sym_sizes_epoch == h->table_epoch;
    int long_rows = known ? h->sym_long_rows : -1;  // (0: no node with more than 16 stamps in a row last time)
    NODAL_TRY(grp::build_lists(h, MatrixStamps{tb, tb.ncomp}, n, &h->nnz, &h->ncontrib, h->indices,
                               h->rowidx, h->cptr, h->contrib, &h->indptr, &h->diag_pos,
                               known ? h->sym_sizes[0] : -1, known ? h->sym_sizes[1] : -1, &long_rows, &h->sym_low_rows));
    if (!known || long_rows >= 0) h->sym_long_rows = long_rows;
    // (rows a low-degree elimination could take, an upper bound: read back with the sizes, kept with them for a
    // repeated grouping of the same table)
    h->low_rows = h->sym_low_rows;
    h->low_rows_epoch = h->struct_epoch;
    // (rhs entries have no column index: the column list lands in a scratch buffer the context keeps --
    // a local one meant a hipMalloc and a hipFree, which waits for the whole device, per symbolic phase)
    // A handful of sources (the usual netlist; the bound is the last grouping's count or twice the number of
    // source components counted at upload) are grouped by two launches.
    const int64_t rhs_bound = known ? h->sym_sizes[3] : (h->rhs_items >= 0 ? 2 * h->rhs_items : -1);
    if (rhs_bound >= 0 && rhs_bound <= grp::FEW_MAX)
        NODAL_TRY(grp::build_lists_few(h, RhsStamps{tb, tb.ncomp}, rhs_bound, &h->nrhs, &h->nrhs_contrib, h->rhs_none,
                                       h->rhs_row, h->rhs_cptr, h->rhs_contrib, known ? h->sym_sizes[2] : -1,
                                       known ? h->sym_sizes[3] : -1));
    else
    NODAL_TRY(grp::build_lists(h, RhsStamps{tb, tb.ncomp}, n, &h->nrhs, &h->nrhs_contrib, h->rhs_none,
                               h->rhs_row, h->rhs_cptr, h->rhs_contrib, nullptr, nullptr,
                               known ? h->sym_sizes[2] : -1, known ? h->sym_sizes[3] : -1));
    h->sym_sizes[0] = h->nnz;
    h->sym_sizes[1] = h->ncontrib;
    h->sym_sizes[2] = h->nrhs;
    h->sym_sizes[3] = h->nrhs_contrib;
    h->sym_sizes_epoch = h->table_epoch;
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
    init_numeric<<<grid_for(h->n), TB, 0, st>>>(status, h->rhs.as<double>(), h->n);
    NODAL_HIP_TRY(h, hipGetLastError());
    if (h->nnz > 0) {
        // NODAL_FOLD_STREAM=0: one lane per entry walking its own run (rounds 1-3; kept as the cross-check)
        const bool stream_fold = !(getenv("NODAL_FOLD_STREAM") && atoi(getenv("NODAL_FOLD_STREAM")) == 0);
        if (stream_fold) {
            const int64_t blocks = (h->nnz + FOLD_EPB - 1) / FOLD_EPB;
            fold_matrix_stream<<<(unsigned)(blocks > 65536 ? 65536 : blocks), TB, 0, st>>>(
                tb, value, h->cptr.as<int32_t>(), h->contrib.as<uint32_t>(), h->data.as<double>(), h->nnz, status);
        } else {
            fold_matrix<<<grid_for(h->nnz), TB, 0, st>>>(tb, value, h->cptr.as<int32_t>(),
                                                        h->contrib.as<uint32_t>(),
                                                        h->data.as<double>(), h->nnz, status);
        }
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    if (h->nrhs > 0) {
        fold_rhs<<<grid_for(h->nrhs), TB, 0, st>>>(tb, value, h->rhs_row.as<int32_t>(),
                                                  h->rhs_cptr.as<int32_t>(),
                                                  h->rhs_contrib.as<uint32_t>(),
                                                  h->rhs.as<double>(), h->nrhs);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    // (host work that needs nothing from the device goes here, between the launches and the wait for their status)
    if (h->B > 0 && !h->csr_only) presolve_plan_ahead(h);
    unsigned long long st_host[3];
    NODAL_TRY(nodal_read_words(h, st_host, status, 24));
    h->have_numeric = true;
    h->have_x = false;
    h->member = member;
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

int stamp_grounded_flags(nodal_ctx *h, uint8_t *flags_dev) {
    // (every caller reserves at least n + 64 bytes; a size that is not a multiple of 4 costs a second fill kernel)
    NODAL_HIP_TRY(h, hipMemsetAsync(flags_dev, 0, ((size_t)h->n + 63) & ~(size_t)63, h->stream));
    if (h->ncomp > 0) {
        grounded_flags<<<grid_for(h->ncomp), TB, 0, h->stream>>>(table_of(h), flags_dev);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    return NODAL_OK;
}
