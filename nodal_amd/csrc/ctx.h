// Internal context of libnodal_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/nodal_hip.h"

// A device allocation that grows on demand and is reused across calls, so the
// launch path never calls hipMalloc once sizes have settled.
// The stream of the handle whose API call is running on this thread (set by FillStreamScope at the
// entry points): a growing DevBuf zero-fills itself on it, in order with every kernel the call launches
// afterwards.  Null outside an API call: the fill then runs on the null stream and is waited for.
inline thread_local hipStream_t nodal_fill_stream = nullptr;
struct FillStreamScope {
    hipStream_t prev;
    explicit FillStreamScope(hipStream_t s) : prev(nodal_fill_stream) { nodal_fill_stream = s; }
    ~FillStreamScope() { nodal_fill_stream = prev; }
    FillStreamScope(const FillStreamScope &) = delete;
    FillStreamScope &operator=(const FillStreamScope &) = delete;
};

// 0: off; 1: growing buffers are filled with 0xFF; 2: scratch buffers too, at every solve entry
inline int nodal_poison_level() {
    static const int level = [] {
        const char *e = getenv("NODAL_POISON");
        if (!e) return 0;
        const int v = atoi(e);
        return v > 1 ? v : 1;
    }();
    return level;
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + (bytes >> 3) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        // A buffer that grows starts from zeros, whatever the allocator hands back: hardening only --
        // no kernel may depend on it (NODAL_POISON shows the ones that do).
        // NODAL_POISON=1 (debugging) fills with 0xFF bytes instead -- NaNs as doubles, -1 as integers;
        // NODAL_POISON=2 also re-poisons every scratch buffer of the handle at the entry of each solve
        // (nodal_poison_scratch, api.hip): what a pooled handle looks like after somebody else's solve.
        static const bool nofill = getenv("NODAL_NOFILL") != nullptr;
        const bool poison = nodal_poison_level() > 0;
        if (e != hipSuccess || nofill) return e;
        if (nodal_fill_stream) return hipMemsetAsync(p, poison ? 0xFF : 0, want, nodal_fill_stream);
        // (outside an API call the fill runs on the null stream; the contexts' streams do not wait for
        // that one by themselves)
        e = hipMemset(p, poison ? 0xFF : 0, want);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        return e;
    }
    // debugging (NODAL_POISON=2): the whole allocation becomes 0xFF bytes, in order on `st`
    void poison(hipStream_t st) {
        if (p) (void)hipMemsetAsync(p, 0xFF, cap, st);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

// A column of the host table: either a copy the library owns, or (NODAL_OPT_BORROW_TABLE) the caller's own column,
// read in place.  The subset of std::vector's interface the readers use.
template <class T>
struct HostCol {
    const T *p = nullptr;
    size_t n = 0;
    std::vector<T> own;
    bool empty() const { return n == 0; }
    size_t size() const { return n; }
    const T *data() const { return p; }
    const T &operator[](size_t i) const { return p[i]; }
    T &operator[](size_t i) { return own[i]; }  // (writers: a column they resize()d themselves)
    void assign(const T *first, const T *last) {
        own.assign(first, last);
        p = own.data();
        n = own.size();
    }
    void borrow(const T *q, size_t count) {
        std::vector<T>().swap(own);
        p = q;
        n = count;
    }
    void resize(size_t count) {
        own.resize(count);
        p = own.data();
        n = count;
    }
    void clear() {
        own.clear();
        p = nullptr;
        n = 0;
    }
};

// host view of the component table (the presolve of presolve.hip works on it)
struct HostTable {
    HostCol<uint8_t> type;
    HostCol<double> value;
    HostCol<int32_t> a, b, c, d, drv, k;
    std::vector<double> values_batch;
    std::vector<int64_t> branch_rows;  // rows of the components that own a branch unknown (types E .. CCCS),
                                       // in file order: what the presolve's planning pass looks at
    mutable std::vector<int32_t> node_slot;  // K entries, all -1 between uses: the presolve's direct map from a
                                             // node to its place in a short list (lead nodes, pivots)
};

// f(lo, hi) over [0, n) in contiguous chunks on up to `max_threads` host threads (the calling one included); chunks
// of at least `min_chunk` items, one chunk = a plain call.  The loops handed to it write disjoint ranges.
template <class F>
inline void nodal_parallel_chunks(int64_t n, int64_t min_chunk, int max_threads, F f) {
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *e = getenv("NODAL_HOST_THREADS")) hw = (unsigned)(atoi(e) > 1 ? atoi(e) : 1);
    int64_t T = hw ? (int64_t)hw : 1;
    if (T > max_threads) T = max_threads;
    if (min_chunk < 1) min_chunk = 1;
    if (T > n / min_chunk) T = n / min_chunk;
    if (T <= 1) {
        f((int64_t)0, n);
        return;
    }
    std::vector<std::thread> th;
    th.reserve((size_t)T - 1);
    for (int64_t c = 1; c < T; ++c) th.emplace_back(f, n * c / T, n * (c + 1) / T);
    f((int64_t)0, n / T);
    for (auto &t : th) t.join();
}

struct nodal_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // trailing updates of the dense LU (lookahead)
    hipEvent_t ev_la[2] = {nullptr, nullptr};
    hipStream_t stream3 = nullptr;   // bulk stream of the block-inverse elimination (all CUs)
    hipEvent_t ev_bi[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t gt_used = 0;              // GemmTimer (dense_common.h): timed launches, their flops
    double gt_flops = 0.0;
    bool hung = false;               // a bounded wait ran out (wait.hip): nothing more may run on this handle
    bool optimistic_nopivot = false; // dense: block elimination although not passive (caller verifies the answer)
    int gj_scalar = 0;               // NODAL_GJ_SCALAR=1: scalar Gauss-Jordan, 2: rank-4 MFMA steps; default rank-16
    bool dense_blockinv = true;      // passive dense systems: block elimination (NODAL_DENSE_BLOCKINV=0: LU)
    std::string err;

    // ---- component table (HBM, structure of arrays) ----
    int64_t ncomp = 0;
    int32_t K = 0, B = 0;
    int64_t n = 0;
    DevBuf type, value, a, b, c, d, drv, k;
    DevBuf values_batch;  // [batch][ncomp] doubles
    int32_t batch = 0;
    bool have_table = false;
    uint64_t table_epoch = 1;      // bumped by every upload_components (topology identity)
    // value sweeps as one block-diagonal system (batch.hip): child context + [count][n] results
    nodal_ctx *blocksys = nullptr;
    uint64_t block_epoch = 0;      // (child) table_epoch of the parent its table was replicated from
    DevBuf batch_x;
    DevBuf batch_scale;            // per-member right-hand-side scales of the last block solve (batch.hip)
    int32_t batch_count = 0;
    bool last_batch_block = false;  // the last solve was a block-diagonal batch (nodal_residual looks at it)
    hipEvent_t ev_batch[4] = {nullptr, nullptr, nullptr, nullptr};  // phase timing of nodal_run_batch
    HostTable host;
    bool keep_host_table = true;
    bool borrow_table = false;       // NODAL_OPT_BORROW_TABLE: the caller's columns stay valid until the next upload
    int32_t member = 0;            // batch member of the last numeric assembly
    nodal_ctx *reduced = nullptr;  // presolved (branch-free) system, see presolve.hip
    uint64_t reduced_key = 0;      // (on the reduced context) fingerprint of the topology its symbolic lists belong to
    bool owns_streams = true;
    nodal_ctx *stream_owner = nullptr;  // (child contexts) the context whose streams and events this one borrows
    void *pinned = nullptr;             // small page-locked scratch for read-backs of a few words (nodal_pinned)
    void *arena = nullptr;              // page-locked staging area that grows on demand (nodal_pinned_arena)
    size_t arena_bytes = 0;
    bool use_presolve = true;
    bool use_graphs = false;       // hipGraph replay of the FCG iteration: measured no gain (kernels are not host-bound)
    DevBuf ps_buf, ps_newidx, ps_hits, ps_stage;
    // exact elimination of nodes with <= 2 neighbours (lowdeg.hip): the reduced network is a
    // matrix-only context (no component table) that inherits the grounded-node flags
    nodal_ctx *lowdeg = nullptr;
    bool csr_only = false;
    DevBuf grounded;        // u8[n]
    DevBuf ld_newidx, ld_work;
    int ld_rounds = 0, ld_slow_rounds = 0;  // rounds that led to this context (all / those removing < 1/32)
    // The choice of the eliminated set and the pattern of the reduced matrix depend on the
    // STRUCTURE of this context's matrix only: kept while struct_epoch stands (value sweeps,
    // pair sweeps, repeated solves), so that a later round is three kernels and no read-back.
    uint64_t struct_epoch = 1;     // bumped whenever the CSR pattern of this context is rebuilt
    uint64_t ld_epoch = 0;         // struct_epoch the cached decision below belongs to
    int ld_state = 0;              // 1 too few candidates, 2 child built, 3 child built and singular (leftover)
    int ld_share = 0;              // the bar (n / share nodes) the decision was taken with
    int64_t ld_n = 0, ld_nnz = 0;

    // ---- symbolic assembly results ----
    bool have_symbolic = false;
    uint64_t sym_sizes_epoch = ~0ull;  // table_epoch the sizes below were read back for (stamp_symbolic); ~0: none yet
    int64_t sym_sizes[4] = {0, 0, 0, 0};  // nnz, ncontrib, nrhs, nrhs_contrib
    int sym_long_rows = -1;        // the matrix grouping of this table found rows of more than 16 stamps (1) / none (0)
    int64_t rhs_items = -1;        // components that stamp the right-hand side (counted at upload; -1: unknown)
    int64_t nnz = 0;        // matrix entries
    int64_t ncontrib = 0;   // matrix contributions
    int64_t nrhs = 0;       // rhs entries (rows with at least one contribution)
    int64_t nrhs_contrib = 0;
    DevBuf indptr;          // i32[n+1]
    DevBuf indices;         // i32[nnz]  (sorted inside each row)
    DevBuf rowidx;          // i32[nnz]  row of each entry (COO companion)
    DevBuf cptr;            // i32[nnz+1] contribution run of each entry
    DevBuf contrib;         // u32[ncontrib] comp<<3 | slot, in fold order
    DevBuf rhs_row;         // i32[nrhs]
    DevBuf rhs_cptr;        // i32[nrhs+1]
    DevBuf rhs_contrib;     // u32[nrhs_contrib]
    DevBuf rhs_none;        // scratch: the (unused) column list of the rhs grouping
    DevBuf diag_pos;        // i32[n] position of (i,i) in CSR or -1

    // ---- numeric assembly results ----
    bool have_numeric = false;
    bool force_pivoting = false;   // testing: use the tournament path even when passive
    bool gepp_panel = true;        // partial pivoting: one launch per 32-column panel (dense_lu.hip)
    bool passive_network = false;  // B == 0 and all R > 0 (set by stamp_numeric)
    DevBuf data;            // f64[nnz]
    DevBuf rhs;             // f64[n]
    DevBuf status;          // i64[4] device-side error words

    // ---- solve ----
    DevBuf x;               // f64[n]
    bool have_x = false;
    DevBuf dense;           // f64[n*n] column-major working copy for LU
    DevBuf piv;             // i32[n]
    DevBuf work;            // scratch (scans, sorts, solver vectors)
    DevBuf work2;
    DevBuf work3;           // grouping: padded scratch for hub-row sorts
    DevBuf solver;          // persistent solver vectors
    DevBuf krylov;          // FGMRES bases (sparse_general.hip)
    DevBuf gn_indptr, gn_indices, gn_rowidx, gn_data, gn_diag;  // node block of G
    DevBuf schur;           // diagonal Schur-complement approximation of the branch block

    // ---- timing ----
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double ms[3] = {0, 0, 0};
    double kern_ms = 0;
    int64_t kern_launches = 0;
    double kern_alg = 0;
    std::vector<hipEvent_t> evpool;  // HIP-event pairs around the dominant kernel

    void *amg = nullptr;  // multigrid hierarchy (amg.hip)
    void *sagg = nullptr; // smoothed-aggregation hierarchy (sagg.hip)
    int64_t sym_low_rows = -1;           // ... of the table grouped last (kept with sym_sizes)
    int64_t low_rows = -1;               // rows of two or three entries of the matrix pattern (upper bound; group.h), -1 unknown
    unsigned long long low_rows_epoch = ~0ull;  // struct_epoch it belongs to
    bool extra_streams = false;  // NODAL_OPT_EXTRA_STREAMS: the handle is used alone and may use streams of its own
    void *slu = nullptr;  // multifrontal LU of the direct route (sparse_direct.hip)
    void *ps_plan = nullptr;  // the presolve's plan, made ahead of the solve (presolve.hip: presolve_plan_ahead)
    bool slu_strict = false;  // refinement judged by |r| / |b| alone (the direct route's second opinion)
    int32_t last_iterations = 0;
    double last_relres = 0;
    int32_t amg_levels = 0;
    int64_t amg_min_n = 65;    // below this the sparse SPD path goes dense
};

#define NODAL_HIP_TRY(h, expr)                                                   \
    do {                                                                         \
        hipError_t _e = (expr);                                                  \
        if (_e != hipSuccess) {                                                  \
            char _buf[512];                                                      \
            snprintf(_buf, sizeof _buf, "%s failed at %s:%d: %s", #expr, __FILE__, \
                     __LINE__, hipGetErrorString(_e));                           \
            (h)->err = _buf;                                                     \
            return _e == hipErrorOutOfMemory ? NODAL_E_NOMEM : NODAL_E_HIP;      \
        }                                                                        \
    } while (0)

// ---- bounded host waits (wait.hip) ----
// Every host wait of the library: polls hipStreamQuery / hipEventQuery against NODAL_WAIT_TIMEOUT_S (default 60 s;
// 0: the runtime's blocking wait).  On a timeout: NODAL_E_HIP, nodal_last_error = the wait site + the last kernel
// enqueued on that stream, the handle marked hung.  `site` is a string literal ("file:line").
struct nodal_ctx;
int nodal_wait_stream(nodal_ctx *h, hipStream_t st, const char *site);
int nodal_wait_event(nodal_ctx *h, hipEvent_t ev, hipStream_t recorded_on, const char *site);
unsigned long long nodal_launches_noted();  // launches this thread has sent through the library's launch log
#define NODAL_STR2(x) #x
#define NODAL_STR(x) NODAL_STR2(x)
#define NODAL_SITE __FILE__ ":" NODAL_STR(__LINE__)
#define NODAL_WAIT_STREAM(h, st) NODAL_TRY(nodal_wait_stream((h), (st), NODAL_SITE))
#define NODAL_WAIT_EVENT(h, ev, st) NODAL_TRY(nodal_wait_event((h), (ev), (st), NODAL_SITE))

#define NODAL_TRY(expr)                  \
    do {                                 \
        int _s = (expr);                 \
        if (_s != NODAL_OK) return _s;   \
    } while (0)

// A section that spreads work over side streams joins them at its end; this makes the main stream wait for them on
// every OTHER way out as well (a failed call in the middle), so that no side stream keeps writing a handle's buffers
// behind the back of whatever the main stream does next.  disarm() on the normal way out, after the section's own join.
struct StreamJoinGuard {
    hipStream_t main;
    hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    int count = 0;
    bool armed = true;
    explicit StreamJoinGuard(hipStream_t m) : main(m) {}
    void add(hipStream_t s, hipEvent_t e) {
        if (s && e && s != main && count < 4) {
            side[count] = s;
            ev[count++] = e;
        }
    }
    void disarm() { armed = false; }
    ~StreamJoinGuard() {
        if (!armed) return;
        for (int i = 0; i < count; ++i)
            if (hipEventRecord(ev[i], side[i]) == hipSuccess) (void)hipStreamWaitEvent(main, ev[i], 0);
        (void)hipGetLastError();
    }
    StreamJoinGuard(const StreamJoinGuard &) = delete;
    StreamJoinGuard &operator=(const StreamJoinGuard &) = delete;
};

static inline int nodal_fail(nodal_ctx *h, int code, const char *msg) {
    h->err = msg;
    return code;
}

// ---- device-wide primitives (scan.hip) ----
// exclusive prefix sum of n uint32 values; out may alias in; total (optional,
// device pointer) receives the grand total.  `tmp` must hold scan_tmp_bytes(n).
size_t scan_tmp_bytes(int64_t n);
int scan_exclusive_u32(nodal_ctx *h, const uint32_t *in, uint32_t *out, int64_t n,
                       uint32_t *total_dev, void *tmp);

// ---- stamping (stamp.hip) ----
int stamp_symbolic(nodal_ctx *h);
int stamp_numeric(nodal_ctx *h, int32_t member, int64_t *bad_component);
int stamp_to_dense(nodal_ctx *h, double *G_dev, int64_t ld, bool col_major);

// ---- fp64 MFMA GEMM (gemm_f64.hip), column-major ----
enum { GEMM_SUB = 0, GEMM_SET = 1, GEMM_SETNEG = 2 };  // C -= A B | C = A B | C = -A B
int gemm_f64(nodal_ctx *h, hipStream_t stream, int mode, double *C, int64_t ldc, const double *A,
             int64_t lda, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K);
struct GemmProblem {
    double *C; int64_t ldc;
    const double *A; int64_t lda;
    const double *B; int64_t ldb;
    int M, N, K;
};
int gemm_pair_f64(nodal_ctx *h, hipStream_t stream, int mode, const GemmProblem &p0, const GemmProblem &p1);
int gemm_sub_f64(nodal_ctx *h, hipStream_t stream, double *C, int64_t ldc, const double *A,
                 int64_t lda, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K);
// C -= At^T B for the upper triangle of a square-leading C (At: K x M panel, B: K x N, column-major)
int gemm_sub_tn_upper_f64(nodal_ctx *h, hipStream_t stream, double *C, int64_t ldc, const double *At,
                          int64_t ldat, const double *B, int64_t ldb, int64_t M, int64_t N, int64_t K, int band);

// ---- dense LU (dense_lu.hip) ----
// leading dimension of the column-major dense panel: padded so that 32-row tiles
// are aligned and columns do not alias on the HBM channels
static inline int64_t dense_lda(int64_t n) {
    int64_t l = (n + 31) & ~(int64_t)31;
    if (l < 32) l = 32;
    if (l % 1024 == 0) l += 32;
    return l;
}
int dense_factor_solve(nodal_ctx *h, int32_t *info);
int dense_fill_nan(nodal_ctx *h, double *x, int64_t n);
// block elimination with inverted diagonal blocks (block_elim.hip): factor + back substitution
int dense_block_elimination(nodal_ctx *h, double *A, int64_t n, int64_t lda, int32_t nrhs, double *xout,
                            int64_t ldx, int32_t *dinfo);
int dense_factor_solve_multi(nodal_ctx *h, int32_t nrhs, double *xout, int64_t ldx, int32_t *info);
int dense_prepare(nodal_ctx *h);
int dense_prepare_pairs(nodal_ctx *h, int32_t nrhs, const int32_t *ia, const int32_t *ib);
int sparse_solve_pairs(nodal_ctx *h, int32_t npairs, const int32_t *ia, const int32_t *ib,
                       double *res_dev, int32_t *info);
int lowdeg_solve_pairs(nodal_ctx *h, int32_t npairs, const int32_t *ia, const int32_t *ib, double *res_dev,
                       bool *done, int32_t *info);

// ---- aggregation multigrid preconditioner (amg.hip) ----
int amg_setup(nodal_ctx *h, double *flag_dev);
int amg_setup_csr(nodal_ctx *h, int64_t n, int64_t nnz, const int32_t *indptr,
                  const int32_t *indices, const int32_t *rowidx, const double *data,
                  const int32_t *diag_pos, double *flag_dev);
int amg_apply(nodal_ctx *h, const double *r, double *z);
int amg_num_levels(nodal_ctx *h);
int64_t amg_level_size(nodal_ctx *h, int level);
void amg_destroy(nodal_ctx *h);
void sagg_destroy(nodal_ctx *h);
// smoothed-aggregation FCG (sagg.hip): NODAL_OK, -1 breakdown, -2 structurally singular,
// -3 declined (not this hierarchy's kind of network: use amg.hip)
int sagg_fcg_solve(nodal_ctx *h, const double *b, bool do_setup, int32_t *info, int32_t *iters, double *resid);
bool sagg_ready(nodal_ctx *h, int64_t n);  // a hierarchy for n unknowns is set up
// up to 16 probe pairs per iteration on that hierarchy (sagg_multi.h); -1: breakdown / no convergence
int sagg_pairs_block_width();  // pairs the block iteration takes per call (sagg_multi.h: MK)
int sagg_fcg_solve_pairs_block(nodal_ctx *h, int32_t count, const int32_t *ia_host, const int32_t *ib_host,
                               double *res_dev, int32_t *iters);
void sagg_invalidate(nodal_ctx *h);  // the hierarchy is not to be used again until the next setup
// the same hierarchy as a preconditioner for any device CSR matrix (general: a few off-diagonals
// of either sign, not symmetric); sagg_apply: z ~= A^-1 r with one cycle
int sagg_setup_csr(nodal_ctx *h, int64_t n, int64_t nnz, const int32_t *indptr, const int32_t *indices,
                   const double *data, bool general, bool check_floating, bool *accepted, int32_t *floating);
typedef float nodal_cyc_t;  // vectors inside the multigrid cycle (csrc/sagg.hip: cyc_t)
// x0_ready: the caller's last kernel already left the cycle's start iterate w D^-1 r (sagg_x0_slot) -- the launch that
// computes it is skipped
int sagg_apply(nodal_ctx *h, const double *r, double *z, bool x0_ready = false, int it = -1);
// where a producer of r can leave the start iterate of the next sagg_apply: x0[i] = (nodal_cyc_t)(omega * dinv[i] * r[i]),
// i < n0.  false: no hierarchy.
bool sagg_x0_slot(nodal_ctx *h, const double **dinv, nodal_cyc_t **x0, int64_t *n0, double *omega);
int sagg_levels(nodal_ctx *h);
int sagg_spmv(nodal_ctx *h, const double *x, double *y, hipEvent_t start = nullptr, hipEvent_t stop = nullptr);  // y = A x on the hierarchy's own (level-0, ELL) matrix
int amg_has_floating_component(nodal_ctx *h, const uint8_t *grounded0, int32_t *floating);
int csr_has_floating_component(nodal_ctx *h, const uint8_t *grounded, int32_t *floating);

// u8[n] flags: 1 where a resistor connects the node to ground (stamp.hip)
int stamp_grounded_flags(nodal_ctx *h, uint8_t *flags_dev);
// the same for any context, matrix-only ones included (lowdeg.hip)
int grounded_flags(nodal_ctx *h, uint8_t *flags_dev);
int csr_to_dense(nodal_ctx *h, double *G_dev, int64_t ld);
int csr_small_floating_check(nodal_ctx *h, int32_t *floating);
int csr_floating_check_small(nodal_ctx *h, int64_t n, const int32_t *indptr, const int32_t *indices,
                             const uint8_t *grounded, uint32_t *flag_dev);
int lowdeg_solve(nodal_ctx *h, int min_share, bool *done, int32_t *info, int32_t *iters, double *resid);

// ---- sparse solvers (sparse_*.hip) ----
int sparse_solve(nodal_ctx *h, int32_t method, int32_t *info, int32_t *iters, double *resid);
int sparse_residual(nodal_ctx *h, double *scaled);
// dense_child: solve the reduced system by the dense block elimination (only if it is passive)
void presolve_plan_ahead(nodal_ctx *h);  // host-only; called by stamp_numeric while its kernels run
void presolve_free_plan(nodal_ctx *h);
int presolve_solve(nodal_ctx *h, bool *done, int32_t *info, int32_t *iters, double *resid,
                   bool dense_child = false);
// ---- sparse direct route (sparse_direct.hip): multifrontal LU + FGMRES refinement ----
// tiny_factor scales (and signs) the value that replaces an unusable pivot
int slu_factor(nodal_ctx *h, int32_t *info, double tiny_factor = 1.0, double tiny_threshold = 0.0);
int slu_apply(nodal_ctx *h, const double *r, double *z);
int64_t slu_perturbed(nodal_ctx *h);  // pivots the last slu_factor replaced
constexpr int SLU_MULTI = 16;         // right-hand sides of slu_apply_multi (interleaved by row: element (i, c) at [i * 16 + c])
int slu_apply_multi(nodal_ctx *h, const double *r, double *z);
bool slu_analysis_kept(nodal_ctx *h); // an analysis for the context's present pattern is at hand (no host work to factor)
void slu_destroy(nodal_ctx *h);
void slu_poison(nodal_ctx *h);
int sparse_direct_solve(nodal_ctx *h, const double *b, double *x, int32_t *info, int32_t *iters, double *resid);
void nodal_free_buffers(nodal_ctx *h);  // api.hip
void nodal_poison_scratch(nodal_ctx *h);  // api.hip: NODAL_POISON=2, entry of every solve
// debugging (NODAL_NANCHECK=1): waits for the stream, copies n doubles to the host and reports on stderr how
// many are not finite and where the first one sits; a no-op otherwise
void nodal_nan_probe(nodal_ctx *h, const double *dev, int64_t n, const char *tag);  // api.hip
// The dense paths' two extra streams (one CU-masked) and their events, created on first use: a handle
// that only ever solves sparse systems holds ONE hardware queue, so that four of them in flight still
// get a queue each (the runtime multiplexes streams beyond its hardware queues: erratic throughput).
int nodal_ensure_aux_streams(nodal_ctx *h);  // api.hip
int nodal_calls_in_flight();  // host threads inside an API call right now, process-wide (api.hip)
int nodal_live_handles();     // handles alive in the process (api.hip)
bool nodal_extra_streams_ok(const nodal_ctx *h);  // NODAL_OPT_EXTRA_STREAMS is set and no other call is in flight (api.hip)
// NODAL_PINNED_BYTES of page-locked host memory of the handle (child contexts use their parent's): the
// destination of every read-back of a few words -- a copy to a stack variable is staged by the runtime
// through its own pinned buffer and costs 15-20 us more.  One API call at a time per handle, and every use is
// copy, wait, read: users need not coordinate.  Null if the allocation failed (callers fall back to the stack).
constexpr size_t NODAL_PINNED_BYTES = 4096;
void *nodal_pinned(nodal_ctx *h);  // api.hip
// At least `bytes` of page-locked host memory of the handle (child contexts: their parent's), grown on demand
// and kept: the staging area of host-built tables that go up in several pieces (a copy from a std::vector is
// staged by the runtime piece by piece, 30-60 us each).  The caller owns it until it has waited for its
// copies; contents do not survive a call that grows it.  Null if the allocation failed.
void *nodal_pinned_arena(nodal_ctx *h, size_t bytes);  // api.hip
// read `bytes` (<= NODAL_PINNED_BYTES) from the device into `dst` through the pinned scratch, and wait
int nodal_read_words(nodal_ctx *h, void *dst, const void *dev_src, size_t bytes);  // api.hip
void nodal_free_block_child(nodal_ctx *h);  // batch.hip
