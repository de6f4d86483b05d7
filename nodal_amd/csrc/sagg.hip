// Smoothed-aggregation multigrid for the sparse SPD path (passive networks: G is a weighted
// graph Laplacian plus ground conductances) -- the preconditioner of the flexible CG that
// replaces scipy.sparse.linalg.spsolve (reference nodal/nodal.py:325) on large resistor
// networks whose links are not graded (the benchmark grids; graded networks, hubs and
// expanders keep the plain-aggregation hierarchy of amg.hip, which this path hands over to
// by declining at setup).
//
// Why a second hierarchy: with piecewise-constant interpolation (amg.hip) the iteration
// count is set by the coarse correction -- 60 FCG iterations on the 1e6-node grid, each
// with a K-cycle of ~35 launches.  Smoothing the prolongator, P = (I - w D^-1 A) P_tent,
// halves the count (29) with K-cycling at the first coarse level only.
//
// Layout: every level stores its matrix in column-major ELL with a per-row length
// (slot s of row i at [s * ld + i]): one thread per row, every load of a wave is a
// contiguous 256/512-byte segment, no LDS staging, no barriers.  P has at most PW = 4
// entries per fine row (own aggregate + the neighbours' aggregates, the rest lumped into
// the own one: row sums are kept), R = P^T is stored by coarse row.
//
// Setup per level, all on the device, deterministic (no floating-point atomics):
//   1. aggregation: distance-2 maximal independent set by hashed priorities (Bell, Dalton,
//      Olson: two neighbour-max passes per round), roots + their neighbours form the
//      aggregates, the nodes at distance 2 join the aggregate of their strongest assigned
//      neighbour;
//   2. P rows in registers; R by integer counting + per-row sort of (node, slot) keys;
//   3. Galerkin product A_c = R A P: one wavefront per coarse row -- pass 1 collects the
//      row's column set in an LDS hash set and sorts it with wave shuffles, pass 2
//      regenerates the products in (R entry, A slot, P slot) order, stages them in LDS and
//      every lane sums the products of "its" column in that fixed order.
// One host round trip per level (the number of aggregates sizes the next level).
#include <hip/hip_ext.h>

#include <atomic>
#include <type_traits>

#include "group.h"

int grounded_flags(nodal_ctx *h, uint8_t *flags_dev);  // lowdeg.hip

namespace {

using grp::TB;
constexpr int PW = 4;            // prolongation entries per fine row
constexpr int ACAP = 64;         // row cap of a coarse matrix: one lane per column in the Galerkin kernel
constexpr int RCAP = 96;         // row cap of R on large levels
constexpr int RCAP_SMALL = 512;  // ... on levels of at most RCAP_SMALL_ROWS coarse rows (aggregates of a few
constexpr int RCAP_SMALL_ROWS = 16384;  // thousand-row level with 30-entry rows reach 100+ members)
inline int rcap_for(int64_t nc) { return nc <= RCAP_SMALL_ROWS ? RCAP_SMALL : RCAP; }
constexpr int W0_MAX = 32;       // level-0 rows longer than this (hubs): decline
constexpr int COARSEST = 64;     // dense inverse below this
constexpr int MAX_LEVELS = 12;
constexpr int MIS_ROUNDS = 6;     // 0.5 % of the nodes are still undecided then: they join or found small aggregates
constexpr int TAIL_MAX_N = 1024; // levels this small run inside the single-workgroup tail kernel
constexpr int TAIL_LEVELS = 6;
constexpr int TAIL_LDS_BUDGET = 150 * 1024;
constexpr int APAD = 32;         // the Galerkin kernel zero-pads every coarse row up to this many slots
constexpr int DOT_BLOCKS = 1024;   // partial sums per dot product of the K-cycle
#ifndef NODAL_SA_OMEGA
#define NODAL_SA_OMEGA 0.85
#endif
constexpr double OMEGA = NODAL_SA_OMEGA;      // damped-Jacobi smoother
// Prolongator smoothing weight (rho(D^-1 A) <= 2 for an M-matrix: anything below 1 damps).  The textbook 4 / (3 rho) = 2/3
// is for the untruncated P; with P cut to four entries per row and the rest lumped, a little more smoothing pays: on the
// 1e6-node grid 0.70 / 0.75 / 0.80 take 27 outer iterations, 0.72 and 0.85 take 28, 2/3 took 29 (A/B/A/B on one box: 8.36
// -> 7.99 ms); grids of other sizes, the batches, config 5, 3-D, wires, contrast and anisotropic grids keep their counts at
// 0.70, and the block systems keep their five levels (from 0.75 on a 64 x 19 600 batch coarsens to six).
// NODAL_SA_OMEGA_P=w: another weight (tools/shape_probe.py, tools/topologies.py under it).
constexpr double OMEGA_P = 0.70;
// Levels >= 1: the Galerkin operators of a smoothed hierarchy have rho(D^-1 A) ~ 1.4-1.6, not 2 (power iterations on
// the CPU prototype: 1.93 / 1.41 / 1.49 / 1.59 for the levels of the 1e6-node grid), so the textbook 4 / (3 rho) is
// 0.83-0.95 there, and the weight of level 0 was too timid for them: 0.80 / 0.90 / 1.00 on the coarse levels all take the
// 1e6-node grid from 27 to 26 outer iterations (grid(562) and grid(1200) 27 -> 26, the random-valued grid 30 -> 29),
// nothing else moves (batches, config 5, 3-D, wires, contrast, anisotropy: same counts, same levels).
constexpr double OMEGA_P_COARSE = 0.85;
// Vectors inside the cycle (the preconditioner: residuals, corrections and smoothing iterates of every level outside
// the tail, the start iterate w D^-1 r and the result z) are kept in f32 like the cycle's copies of A, P and R: the
// outer iterations are flexible ones and their own vectors (x, r, p, Ap; the Krylov basis of the general path) and
// every dot product stay f64.  A level-0 pass of the single-vector cycle moves 56 instead of 64 bytes per row (the
// matrix is most of it), one of the sixteen-column block iteration (sagg_multi.h) 64 instead of 128 per column.
using cyc_t = nodal_cyc_t;  // float (ctx.h)

inline unsigned grid_for(int64_t n, unsigned cap = 8192) {  // (caps are multiples of 8: see xcd_block)
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    if (g >= 64) g = (g + 7) & ~(int64_t)7;
    return (unsigned)(g > cap ? cap : g);
}
inline int64_t pad64(int64_t n) { return (n + 63) & ~(int64_t)63; }

// Workgroups are handed to the eight XCDs round-robin (workgroup b runs on XCD b % 8, each with its own
// L2).  The row kernels walk virtual block numbers instead: XCD k gets the contiguous eighth
// [k G/8, (k+1) G/8) of a grid of G blocks, so the vector lines a row block gathers (its neighbours'
// entries, the fine entries under a coarse row) are mostly lines the same L2 already holds, and the
// same rows stay on the same XCD from one kernel of the cycle to the next.  Grids are multiples of 8
// (grid_for rounds up; surplus blocks find no rows).
__device__ __forceinline__ unsigned xcd_block() {
    const unsigned g = gridDim.x, b = blockIdx.x;
    return (g & 7u) ? b : (b & 7u) * (g >> 3) + (b >> 3);
}


struct Ell {
    int64_t n = 0, ld = 0;
    int32_t width = 0;
    const int32_t *col = nullptr;
    const int16_t *dcol = nullptr;  // fixed-width levels whose columns all lie within 32767 of their row: column - row
                                    // (what the cycle's and the outer iteration's row kernels read instead of `col`)
    const double *val = nullptr;
    const float *valf = nullptr;  // the same values rounded to f32: what the preconditioner's sweeps read
    const int32_t *len = nullptr;
};

enum { ST_MAXLEN = 0, ST_GRADED = 1, ST_BADDIAG = 2, ST_OVERFLOW = 3, ST_NNZ = 4, ST_UNASSIGNED = 5,
       ST_MAXR = 6, ST_NC = 7 /* aggregates of the level (the scan's total) */,
       ST_FARCOL = 8 /* level 0: some column lies further than 32767 from its row */, ST_COUNT = 9 };

struct SLevel {
    int64_t n = 0, ld = 0, nc = 0;
    int32_t width = 0;       // slots allocated for A
    int32_t maxlen = 0;      // longest row of A (host copy)
    int32_t wfix = 0;        // > 0: every row is zero-padded to this many slots (unrolled row kernels)
    int64_t nnz = 0;
    DevBuf acol, aval, alen, dinv;
    DevBuf adcol;            // int16 column - row per slot (d16)
    bool d16 = false;
    DevBuf agg, pcol, pval;
    DevBuf rcol, rval, rlen;
    int64_t rld = 0;
    DevBuf vec, part, gflag;
    DevBuf avalf, pvalf, rvalf;  // f32 copies of A, P, R for the cycle kernels (levels outside the tail)
    Ell A() const {
        Ell e;
        e.n = n; e.ld = ld; e.width = width;
        e.col = acol.as<int32_t>(); e.val = aval.as<double>(); e.len = alen.as<int32_t>();
        e.valf = avalf.as<float>();
        e.dcol = d16 ? adcol.as<int16_t>() : nullptr;
        return e;
    }
    // (slots of ld doubles; a cyc_t vector uses the front of its slot)
    template <typename T = double> T *v(int which) const { return reinterpret_cast<T *>(vec.as<double>() + (int64_t)which * ld); }
};
enum { V_X = 0, V_R = 1, V_XP = 2, V_RC = 3, V_C1 = 4, V_C2 = 5, V_V1 = 6, V_V2 = 7, V_R2 = 8, V_X1 = 9, V_T = 10,
       V_COUNT = 11 };

struct TailLevelDesc {
    int n, nc, ld, rld, width, nq, lpr;  // width: longest row of A; nq: blocks of 8 per row of R; lpr: lanes per row
    const int32_t *acol, *pcol, *rcol, *rlen, *alen;
    const double *aval, *dinv, *pval, *rval;
    int o_aval, o_acol, o_dinv, o_pval, o_pcol, o_rval, o_rcol, o_B, o_X, o_Y, o_R;  // LDS offsets (bytes)
};
struct TailDesc {
    int nlev;  // tail levels; the last one is solved with its dense inverse / diagonal
    TailLevelDesc lv[TAIL_LEVELS];
    const double *inv;
    int nu;           // Jacobi sweeps before / after each coarse correction inside the tail
    int image_bytes;  // matrices and transfer operators (packed once per setup)
    int lds_bytes;    // image + vectors, less `skip`
    int skip;         // leading bytes of the image that never go to LDS (the first level's matrix: registers)
    int slots;        // register slots per lane the first level needs: ceil(width / lpr)
    long long *stamps;  // development probe (NODAL_TAIL_PROBE): wall-clock ticks at the phase boundaries
};

struct SHierarchy {
    std::vector<SLevel *> pool;
    int nlev = 0;
    int tail = -1;          // first level inside the tail kernel
    bool ready = false;
    bool kcycle = true;          // two inner FCG steps at the first coarse level
    // The coefficients of those two steps hardly move from one outer iteration to the next (the coarse correction of an
    // aggregation hierarchy is too small by the same factor every time): the flexible-CG driver calibrates them in its
    // first iterations and every fourth one after, and the cycles in between use the mean of the last three samples --
    // three launches fewer (sagg_fcg_solve).  kfrozen: this cycle uses the frozen ones; kslot / kcount: where an adaptive
    // cycle's sample goes (-1: nowhere -- every caller but that driver).
    bool kfrozen = false;
    int kslot = -1, kcount = 0, ksamples = 0;
    int klevels = 1;             // ... at the first `klevels` coarse levels (NODAL_SA_KLEVELS)
    int nu[3] = {2, 1, 2};       // Jacobi sweeps before and after the coarse correction: level 0 / 1 / deeper.
                                 // Two on the small levels outside the tail cost eight 4-us launches per iteration and
                                 // make the count independent of where the level sizes fall: grid(1600) / grid(2000),
                                 // whose 1050- / 1600-row level just misses the tail, 39 / 40 -> 31 iterations; grid(1000)
                                 // 30 -> 28 at the same time; the 128-member batch 32 -> 29.
                                 // Two at LEVEL 0 since round 5's third session (one until then: in round 2 a level-0
                                 // pass was 18 us and "212" cost 16.9 against 14.9 ms): a pass is 10 us now (f32 cycle
                                 // vectors, 16-bit columns) and an iteration is mostly its 22 coarse-level launches, so
                                 // two more level-0 passes (207 -> 227 us per iteration) for 26 -> 21 iterations on the
                                 // 1e6-node grid pay: config 3 7.40 -> 6.81 ms, the 128-member batch 29 -> 22 iterations,
                                 // grid(100) .. grid(1600) 8-18 % faster, grid(2000) 2 % slower (NODAL_SA_NU=112: one).
                                 // Two at level 1 as well ("222") is slower (22 iterations, 7.6 ms).
    bool dense_coarsest = true;  // last level: dense inverse; false: nothing but isolated nodes (diagonal)
    DevBuf tail_stamps, tail_image, apcol, apval, aplen, bstat;
    DevBuf mvec;  // vectors, partials and scalars of the block iteration (sagg_multi.h)
    hipEvent_t ev_copy = nullptr;  // behind the statistics read-back of a level (build_level)
    // R = P^T (count, scan, fill, two sorts, blocked layout: seven latency-bound launches) and A P (one long one) need
    // P only, not each other: R is built on a stream of the hierarchy's own while the main one computes A P
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_flags = nullptr;
    bool aux_pending = false;  // work was enqueued on `aux` that the main stream has not been made to wait for yet
                               // (sagg_setup_csr joins it on EVERY way out: declined, stopped, failed)
    bool aux_tried = false;
    // the "touches ground" flags go up the aggregate maps on that stream too, level by level behind R (they need the
    // aggregates only): flags_levels = how many levels' flags_up have been enqueued there in this setup
    bool want_flags = false;
    int flags_levels = 0;
    DevBuf stats, coarse_inv, mis_t, mis_m, mis_flag, mis_id, agg1, keys, rstart, cursor, lists;
    TailDesc td;
    uint64_t *host_stats = nullptr;  // pinned
    // The symbolic part of the setup -- aggregates, the patterns of P, R and of every coarse matrix, the
    // tail's layout, the structural-singularity verdict -- depends on the PATTERN of the level-0 matrix
    // only.  It is kept while the context's struct_epoch stands (a value sweep on one topology, a
    // repeated solve, the members of a batch): a later setup recomputes values only (sagg_refresh).
    bool sym_valid = false;
    uint64_t sym_epoch = 0;
    int64_t sym_n = 0, sym_nnz = 0;
    bool sym_general = false, sym_check = false;
    int32_t sym_floating = 0;
    // A pair of outer iterations (parities 0 and 1: ~66 launches with fixed arguments) captured as a
    // hipGraph and replayed while the hierarchy is kept: one API call instead of 66 (NODAL_SA_GRAPH=1).
    hipGraph_t it_graph = nullptr;
    hipGraphExec_t it_exec = nullptr;
    uint64_t it_key = 0;         // hash of every pointer and size the captured launches carry
    bool refreshed = false;      // the last setup was a values-only refresh
    int last_iters = 0;          // iterations of the last converged solve on this hierarchy (0: none yet)
    // A scheduling hint that survives a full setup: the iteration count of the last converged solve on this HANDLE
    // and the size of its system.  A fresh hierarchy for a system of the same size (the next circuit of a sweep over
    // netlists of one shape) will need about as many: the first look at the residual comes after three quarters of
    // them instead of after six (every look drains the queue: ~50 us).  It changes WHEN the host looks, nothing else.
    int hint_iters = 0;
    int64_t hint_n = 0, hint_nnz = 0;
    int mblock_iters = 0;  // the same for the block iteration of a pair sweep (sagg_multi.h): what the block before took
    int64_t mblock_n = 0;
    unsigned long long sym_stats[MAX_LEVELS * ST_COUNT] = {0};
    SLevel *level(int l) {
        while ((int)pool.size() <= l) pool.push_back(new SLevel());
        return pool[l];
    }
    void drop_graph() {
        if (it_exec) (void)hipGraphExecDestroy(it_exec);
        if (it_graph) (void)hipGraphDestroy(it_graph);
        it_exec = nullptr;
        it_graph = nullptr;
        it_key = 0;
    }
    ~SHierarchy() {
        drop_graph();
        if (ev_copy) (void)hipEventDestroy(ev_copy);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
        if (ev_flags) (void)hipEventDestroy(ev_flags);
        if (aux) (void)hipStreamDestroy(aux);
        for (SLevel *l : pool) {
            DevBuf *b[] = {&l->acol, &l->aval, &l->alen, &l->dinv, &l->agg, &l->pcol, &l->pval, &l->rcol,
                           &l->rval, &l->rlen, &l->vec, &l->part, &l->gflag, &l->avalf, &l->pvalf, &l->rvalf, &l->adcol};
            for (DevBuf *x : b) x->release();
            delete l;
        }
        tail_image.release();
        mvec.release();
        tail_stamps.release();
        bstat.release();
        apcol.release();
        apval.release();
        aplen.release();
        DevBuf *b[] = {&stats, &coarse_inv, &mis_t, &mis_m, &mis_flag, &mis_id, &agg1, &keys, &rstart, &cursor, &lists};
        for (DevBuf *x : b) x->release();
        if (host_stats) (void)hipHostFree(host_stats);
    }
};

// ---------------------------------------------------------------------------------
// level 0: CSR -> ELL
// ---------------------------------------------------------------------------------

// Statistics leave a kernel as one plain store per workgroup into bstat[0][block] (a maximum) and
// bstat[1][block] (a sum); reduce_bstat folds them into the level's statistics words.  (Tens of
// thousands of workgroups adding to ONE word serialise at ~11 ns each: 1.4 ms for the Galerkin
// kernel's 65 536 workgroups, ten times the kernel itself.)
constexpr int BSTAT_MAX = 65536;
__global__ __launch_bounds__(1024) void reduce_bstat(int nblocks, const uint32_t *__restrict__ bstat,
                                                     unsigned long long *__restrict__ stats, int slot_max,
                                                     int slot_sum) {
    __shared__ unsigned int s_max;
    __shared__ unsigned long long s_sum;
    if (threadIdx.x == 0) { s_max = 0; s_sum = 0; }
    __syncthreads();
    unsigned int m = 0;
    unsigned long long t = 0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) {
        m = max(m, bstat[i]);
        t += bstat[BSTAT_MAX + i];
    }
    atomicMax(&s_max, m);
    atomicAdd(&s_sum, t);
    __syncthreads();
    if (threadIdx.x == 0) {
        if (slot_max >= 0) stats[slot_max] = s_max;
        if (slot_sum >= 0) stats[slot_sum] = s_sum;
    }
}

// longest row, number of nodes with graded links (amg.hip's contrast criterion: one link above
// `share` of the diagonal, or the strongest more than `spread` times the weakest)
__global__ __launch_bounds__(TB) void row_stats(int64_t n, const int32_t *__restrict__ indptr,
                                                const int32_t *__restrict__ indices,
                                                const double *__restrict__ data, double share, double spread,
                                                bool general, unsigned long long *__restrict__ stats,
                                                uint32_t *__restrict__ bstat) {
    int32_t mlen = 0;
    uint32_t graded = 0, bad = 0, far = 0;
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const int32_t e0 = indptr[i], e1 = indptr[i + 1];
        mlen = max(mlen, e1 - e0);
        double d = 0.0, mx = 0.0, mn = 1e300;
        for (int32_t e = e0; e < e1; ++e) {
            const double v = data[e];
            const int64_t dc = (int64_t)indices[e] - i;
            if (dc > 32767 || dc < -32767) far = 1;
            if (indices[e] == (int32_t)i) d = v;
            else if (v != 0.0) {
                mx = fmax(mx, fabs(v));
                mn = fmin(mn, fabs(v));
                if (v > 0.0 && !general) bad = 1;  // positive off-diagonal: not an M-matrix
            }
        }
        if (!(d > 0.0)) bad = 1;
        graded += (d > 0.0 && (mx > share * d || mx > spread * mn)) ? 1u : 0u;
    }
    __shared__ unsigned int s_len, s_graded, s_bad;
    if (threadIdx.x == 0) s_len = s_graded = s_bad = 0;
    __syncthreads();
    atomicMax(&s_len, (unsigned)mlen);
    if (graded) atomicAdd(&s_graded, graded);
    if (bad) atomicOr(&s_bad, 1u);
    if (far) atomicOr(&s_bad, 2u);
    __syncthreads();
    if (threadIdx.x == 0) {
        bstat[blockIdx.x] = s_len;
        bstat[BSTAT_MAX + blockIdx.x] = s_graded;
        if (s_bad & 1u) atomicOr(&stats[ST_BADDIAG], 1ull);  // (rare)
        if (s_bad & 2u) stats[ST_FARCOL] = 1ull;             // (every writer stores the same word)
    }
}

// (valf: the f32 copy of the values the cycle's sweeps read, written by the producer of the f64 ones)
__global__ __launch_bounds__(TB) void csr_to_ell(int64_t n, int64_t ld, const int32_t *__restrict__ indptr,
                                                 const int32_t *__restrict__ indices,
                                                 const double *__restrict__ data, int32_t *__restrict__ col,
                                                 double *__restrict__ val, float *__restrict__ valf,
                                                 int32_t *__restrict__ len, double *__restrict__ dinv, int32_t wpad,
                                                 int16_t *__restrict__ dcol) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
        const int32_t e0 = indptr[i], e1 = indptr[i + 1];
        double d = 1.0;
        for (int32_t e = e0; e < e1; ++e) {
            const int32_t c = indices[e];
            const double v = data[e];
            col[(int64_t)(e - e0) * ld + i] = c;
            if (dcol) dcol[(int64_t)(e - e0) * ld + i] = (int16_t)(c - (int32_t)i);
            val[(int64_t)(e - e0) * ld + i] = v;
            valf[(int64_t)(e - e0) * ld + i] = (float)v;
            if (c == (int32_t)i) d = v;
        }
        for (int32_t s = e1 - e0; s < wpad; ++s) {  // zero padding up to the fixed width
            col[(int64_t)s * ld + i] = (int32_t)i;
            if (dcol) dcol[(int64_t)s * ld + i] = 0;
            val[(int64_t)s * ld + i] = 0.0;
            valf[(int64_t)s * ld + i] = 0.0f;
        }
        len[i] = e1 - e0;
        dinv[i] = 1.0 / d;
    }
}

// ---------------------------------------------------------------------------------
// aggregation: distance-2 maximal independent set
// ---------------------------------------------------------------------------------
// T[i] = state << 30 | hash30(i) ; state 3 root, 1 undecided, T = 0 out.  (Two nodes within
// distance 2 with equal hashes -- about one pair in 1e8 -- both become roots: harmless, the
// aggregates are a little smaller there.)

__device__ __forceinline__ uint32_t hash30(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x & 0x3fffffffu;
}

__global__ __launch_bounds__(TB) void mis_init(int64_t n, uint32_t *__restrict__ T) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        T[i] = (1u << 30) | hash30((uint32_t)i);
}

// max of v over the neighbours of row i: eight slots at a time, their columns requested together, then the eight
// gathers (a slot past the row's end re-reads the row's first one: the maximum does not mind) -- two dependent round
// trips per eight entries instead of per entry; the coarse levels' rows of 10-27 entries made these kernels 6-8 us
template <int NB>
__device__ __forceinline__ uint32_t nbr_max_batch(const Ell &A, const uint32_t *__restrict__ v, int64_t i, int32_t l, int32_t s0,
                                                  uint32_t best) {
    int32_t c[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) c[u] = A.col[(int64_t)(s0 + u < l ? s0 + u : 0) * A.ld + i];
    uint32_t t[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) t[u] = v[c[u]];
#pragma unroll
    for (int u = 0; u < NB; ++u) best = t[u] > best ? t[u] : best;
    return best;
}
__device__ __forceinline__ uint32_t nbr_max(const Ell &A, const uint32_t *__restrict__ v, int64_t i, int32_t l, uint32_t best) {
    if (l <= 5) return nbr_max_batch<5>(A, v, i, l, 0, best);  // (level 0 of a grid: bound by the memory, no surplus loads)
    for (int32_t s0 = 0; s0 < l; s0 += 8) best = nbr_max_batch<8>(A, v, i, l, s0, best);
    return best;
}

// m[i] = max of v over the closed neighbourhood of i
__global__ __launch_bounds__(TB) void mis_max(Ell A, const uint32_t *__restrict__ v, uint32_t *__restrict__ m) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB)
        m[i] = nbr_max(A, v, i, A.len[i], v[i]);
}

// second neighbour-max + state update: an undecided node that is the maximum of its distance-2
// neighbourhood becomes a root; one that sees a root there drops out.  (No count of the nodes left
// undecided: 15 000 wavefronts adding to one word cost 150 us, ten times the kernel.)
__global__ __launch_bounds__(TB) void mis_update(Ell A, uint32_t *__restrict__ T, const uint32_t *__restrict__ m1) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        const uint32_t own = T[i];
        if ((own >> 30) != 1) continue;
        const uint32_t best = nbr_max(A, m1, i, A.len[i], m1[i]);
        if (best == own) T[i] = own | (2u << 30);
        else if ((best >> 30) == 3) T[i] = 0;
    }
}

// tracing only: nodes still undecided
__global__ __launch_bounds__(TB) void mis_count(int64_t n, const uint32_t *__restrict__ T,
                                                unsigned long long *__restrict__ out) {
    uint32_t left = 0;
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        left += (T[i] >> 30) == 1 ? 1u : 0u;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) left += __shfl_down(left, off, 64);
    if ((threadIdx.x & 63) == 0 && left) atomicAdd(out, (unsigned long long)left);
}

// roots, plus the nodes still undecided after the last round that have no root next to them
__global__ __launch_bounds__(TB) void mis_flag_roots(Ell A, const uint32_t *__restrict__ T,
                                                     uint32_t *__restrict__ flag) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i <= A.n; i += (int64_t)gridDim.x * TB) {
        uint32_t f = 0;
        if (i < A.n) {
            const uint32_t own = T[i];
            if ((own >> 30) == 3) f = 1;
            else if ((own >> 30) == 1) {
                bool near = false;
                const int32_t l = A.len[i];
                for (int32_t s = 0; s < l; ++s) near = near || (T[A.col[(int64_t)s * A.ld + i]] >> 30) == 3;
                f = near ? 0u : 1u;
            }
        }
        flag[i] = f;
    }
}

// The whole independent-set computation of a SMALL level (init, every round's two neighbour-max passes,
// the root flags) in one workgroup: fourteen launches of 4.5 us each become one of ~20 us.
constexpr int64_t MIS_SMALL_MAX = 1024;  // (one row per thread; at 8.9 k rows nine rows per thread in sequence lose: 250 instead of 63 us)
__global__ __launch_bounds__(1024) void mis_small(Ell A, uint32_t *__restrict__ T, uint32_t *__restrict__ M,
                                                  uint32_t *__restrict__ flag, int rounds) {
    const int64_t n = A.n;
    for (int64_t i = threadIdx.x; i < n; i += 1024) T[i] = (1u << 30) | hash30((uint32_t)i);
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        for (int64_t i = threadIdx.x; i < n; i += 1024) {  // (mis_max)
            uint32_t best = T[i];
            const int32_t l = A.len[i];
            for (int32_t s = 0; s < l; ++s) {
                const uint32_t t = T[A.col[(int64_t)s * A.ld + i]];
                best = t > best ? t : best;
            }
            M[i] = best;
        }
        __syncthreads();
        for (int64_t i = threadIdx.x; i < n; i += 1024) {  // (mis_update)
            const uint32_t own = T[i];
            if ((own >> 30) != 1) continue;
            uint32_t best = M[i];
            const int32_t l = A.len[i];
            for (int32_t s = 0; s < l; ++s) {
                const uint32_t t = M[A.col[(int64_t)s * A.ld + i]];
                best = t > best ? t : best;
            }
            if (best == own) T[i] = own | (2u << 30);
            else if ((best >> 30) == 3) T[i] = 0;
        }
        __syncthreads();
    }
    for (int64_t i = threadIdx.x; i <= n; i += 1024) {  // (mis_flag_roots)
        uint32_t f = 0;
        if (i < n) {
            const uint32_t own = T[i];
            if ((own >> 30) == 3) f = 1;
            else if ((own >> 30) == 1) {
                bool near = false;
                const int32_t l = A.len[i];
                for (int32_t s = 0; s < l; ++s) near = near || (T[A.col[(int64_t)s * A.ld + i]] >> 30) == 3;
                f = near ? 0u : 1u;
            }
        }
        flag[i] = f;
    }
}

// roots and their neighbours (the root of highest priority if there are several)
__global__ __launch_bounds__(TB) void assign_near(Ell A, const uint32_t *__restrict__ T,
                                                  const uint32_t *__restrict__ flag,
                                                  const uint32_t *__restrict__ id, int32_t *__restrict__ agg1) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        int32_t a = -1;
        if (flag[i]) a = (int32_t)id[i];
        else {
            uint32_t best = 0;
            const int32_t l = A.len[i];
            // (slots in batches, their columns requested together, then what the columns name: see nbr_max; the
            // candidates are looked at in slot order, as the entry-by-entry loop did)
            auto batch = [&](auto NBc, int32_t s0) {
                constexpr int NB = decltype(NBc)::value;
                int32_t c[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) c[u] = A.col[(int64_t)(s0 + u < l ? s0 + u : 0) * A.ld + i];
                uint32_t fl[NB], tt[NB], idd[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    fl[u] = flag[c[u]];
                    tt[u] = T[c[u]];
                    idd[u] = id[c[u]];
                }
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (s0 + u >= l || c[u] == (int32_t)i || !fl[u]) continue;
                    const uint32_t t = tt[u] | (3u << 30);
                    if (t > best) { best = t; a = (int32_t)idd[u]; }
                }
            };
            if (l <= 5) batch(std::integral_constant<int, 5>{}, 0);
            else for (int32_t s0 = 0; s0 < l; s0 += 8) batch(std::integral_constant<int, 8>{}, s0);
        }
        agg1[i] = a;
    }
}

// the rest joins the aggregate of its strongest assigned neighbour
__global__ __launch_bounds__(TB) void assign_far(Ell A, const int32_t *__restrict__ agg1, int32_t *__restrict__ agg,
                                                 unsigned long long *__restrict__ stats) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        int32_t a = agg1[i];
        if (a < 0) {
            double bw = -1.0;
            const int32_t l = A.len[i];
            auto batch = [&](auto NBc, int32_t s0) {  // (as in assign_near)
                constexpr int NB = decltype(NBc)::value;
                int32_t c[NB];
                double w[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int64_t at = (int64_t)(s0 + u < l ? s0 + u : 0) * A.ld + i;
                    c[u] = A.col[at];
                    w[u] = fabs(A.val[at]);
                }
                int32_t aj[NB];
#pragma unroll
                for (int u = 0; u < NB; ++u) aj[u] = agg1[c[u]];
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    if (s0 + u >= l || c[u] == (int32_t)i || aj[u] < 0) continue;
                    if (w[u] > bw) { bw = w[u]; a = aj[u]; }
                }
            };
            if (l <= 5) batch(std::integral_constant<int, 5>{}, 0);
            else for (int32_t s0 = 0; s0 < l; s0 += 8) batch(std::integral_constant<int, 8>{}, s0);
            if (a < 0) stats[ST_UNASSIGNED] = 1;  // benign race
        }
        agg[i] = a;
    }
}

// ---------------------------------------------------------------------------------
// P = (I - w D^-1 A) P_tent, at most PW entries per row (the rest lumped into the own aggregate)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void build_P(Ell A, const double *__restrict__ dinv,
                                              const int32_t *__restrict__ agg, int32_t *__restrict__ pcol,
                                              double *__restrict__ pval, float *__restrict__ pvalf, double omega_p) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        int32_t c[PW];
        double v[PW];
#pragma unroll
        for (int k = 0; k < PW; ++k) { c[k] = -1; v[k] = 0.0; }
        const int32_t own = agg[i];
        c[0] = own;
        v[0] = 1.0;
        const double wd = omega_p * dinv[i];
        const int32_t l = A.len[i];
        for (int32_t s = 0; s < l; ++s) {
            const int32_t J = agg[A.col[(int64_t)s * A.ld + i]];
            const double t = -wd * A.val[(int64_t)s * A.ld + i];
            bool placed = false;
#pragma unroll
            for (int k = 0; k < PW; ++k) {
                if (!placed && c[k] == J) { v[k] += t; placed = true; }
            }
#pragma unroll
            for (int k = 1; k < PW; ++k) {
                if (!placed && c[k] < 0) { c[k] = J; v[k] = t; placed = true; }
            }
            if (!placed) v[0] += t;  // a fifth aggregate: lumped (row sum kept)
        }
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            pcol[(int64_t)k * A.ld + i] = c[k];
            pval[(int64_t)k * A.ld + i] = v[k];
            pvalf[(int64_t)k * A.ld + i] = (float)v[k];
        }
    }
}

// ---- R = P^T by coarse row: count, scan, fill, sort each row's (node << 2 | slot) keys ----
// (the counting atomic's return value is the entry's place inside its row -- any order will do, the rows
// are sorted afterwards --: kept per (slot, fine row), so that the fill needs no atomics of its own)
__global__ __launch_bounds__(TB) void r_count(int64_t n, int64_t ld, const int32_t *__restrict__ pcol,
                                              uint32_t *__restrict__ cnt, uint32_t *__restrict__ place) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int32_t J = pcol[(int64_t)k * ld + i];
            if (J >= 0) place[(int64_t)k * ld + i] = atomicAdd(&cnt[J], 1u);
        }
    }
}
__global__ __launch_bounds__(TB) void r_fill(int64_t n, int64_t ld, const int32_t *__restrict__ pcol,
                                             const uint32_t *__restrict__ rstart, const uint32_t *__restrict__ place,
                                             uint64_t *__restrict__ keys) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB) {
#pragma unroll
        for (int k = 0; k < PW; ++k) {
            const int32_t J = pcol[(int64_t)k * ld + i];
            if (J >= 0) keys[rstart[J] + place[(int64_t)k * ld + i]] = ((uint64_t)i << 2) | (uint64_t)k;
        }
    }
}
// sorted keys -> blocked R (blocks of 8 entries per coarse row, see k_restrict): eight lanes per row,
// so that the stores of a wavefront are contiguous
__global__ __launch_bounds__(TB) void r_to_ell(int64_t nc, int64_t rld, const uint32_t *__restrict__ rstart,
                                               const uint64_t *__restrict__ keys, int64_t ld,
                                               const double *__restrict__ pval, int32_t *__restrict__ rcol,
                                               double *__restrict__ rval, float *__restrict__ rvalf,
                                               int32_t *__restrict__ rlen,
                                               unsigned long long *__restrict__ stats, uint32_t *__restrict__ bstat,
                                               uint32_t rcap) {
    uint32_t mlen = 0;
    const int sub = threadIdx.x & 7;
    const int64_t rows_per_pass = (int64_t)gridDim.x * (TB / 8);
    for (int64_t I = (int64_t)xcd_block() * (TB / 8) + threadIdx.x / 8; I < nc; I += rows_per_pass) {
        const uint32_t s0 = rstart[I];
        uint32_t l = rstart[I + 1] - s0;
        mlen = l > mlen ? l : mlen;
        if (l > rcap) l = rcap;
        // zero padding: to the end of the last block of 8, and on small levels (the LDS tail copies
        // whole rows without looking at their length) up to the cap
        const uint32_t upto = nc <= 4096 ? rcap : ((l + 7u) & ~7u);
        for (uint32_t t = sub; t < upto; t += 8) {
            const int64_t at = ((int64_t)(t >> 3) * rld + I) * 8 + (t & 7);
            if (t < l) {
                const uint64_t key = keys[s0 + t];
                const int64_t i = (int64_t)(key >> 2);
                const double v = pval[(int64_t)(key & 3) * ld + i];
                rcol[at] = (int32_t)i;
                rval[at] = v;
                rvalf[at] = (float)v;
            } else {
                rcol[at] = 0;
                rval[at] = 0.0;
                rvalf[at] = 0.0f;
            }
        }
        if (sub == 0) rlen[I] = (int32_t)l;
    }
    if (mlen > rcap) atomicOr(&stats[ST_OVERFLOW], 2ull);  // (rare)
    __shared__ unsigned int s_max;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(mlen, off, 64);
        mlen = o > mlen ? o : mlen;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(&s_max, mlen);
    __syncthreads();
    if (threadIdx.x == 0) {
        bstat[blockIdx.x] = s_max;
        bstat[BSTAT_MAX + blockIdx.x] = 0;
    }
}

// The values of R again for a matrix of the SAME pattern (a value sweep, a repeated solve: the
// hierarchy's symbolic part -- aggregates, patterns of P, R and the coarse matrices -- is kept):
// entry t of coarse row I names fine row i; its value is the entry of P's row i whose column is I.
__global__ __launch_bounds__(TB) void r_refresh(int64_t nc, int64_t rld, const int32_t *__restrict__ rlen,
                                                const int32_t *__restrict__ rcol, int64_t ld,
                                                const int32_t *__restrict__ pcol, const double *__restrict__ pval,
                                                double *__restrict__ rval, float *__restrict__ rvalf) {
    const int sub = threadIdx.x & 7;
    const int64_t rows_per_pass = (int64_t)gridDim.x * (TB / 8);
    for (int64_t I = (int64_t)xcd_block() * (TB / 8) + threadIdx.x / 8; I < nc; I += rows_per_pass) {
        const int32_t l = rlen[I];
        for (int32_t t = sub; t < l; t += 8) {
            const int64_t at = ((int64_t)(t >> 3) * rld + I) * 8 + (t & 7);
            const int64_t i = rcol[at];
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < PW; ++k) {
                const double pv = pval[(int64_t)k * ld + i];
                v = pcol[(int64_t)k * ld + i] == (int32_t)I ? pv : v;
            }
            rval[at] = v;
            rvalf[at] = (float)v;
        }
    }
}

// per-row sort of the (node << 2 | slot) keys.  Up to 32 keys: one lane per row with a sorting
// network in registers (grp::sort_rows_short stops at 16, the rows of R average 17, and its
// workgroup-per-row LDS sort for the rest cost 80 us per 1e5 rows).
__global__ __launch_bounds__(TB) void r_sort_short(const uint32_t *__restrict__ rstart, uint64_t *__restrict__ keys,
                                                   int64_t nc) {
    for (int64_t I = (int64_t)xcd_block() * TB + threadIdx.x; I < nc; I += (int64_t)gridDim.x * TB) {
        const uint32_t s0 = rstart[I];
        const int len = (int)(rstart[I + 1] - s0);
        if (len < 2) continue;
        if (len <= 4) grp::sort_short_row<4>(keys + s0, len);
        else if (len <= 8) grp::sort_short_row<8>(keys + s0, len);
        else if (len <= 16) grp::sort_short_row<16>(keys + s0, len);
        else if (len <= 32) grp::sort_short_row<32>(keys + s0, len);
    }
}
// 33..512 keys (the coarser levels): one wavefront per row, bitonic network in LDS.  Longer rows are
// beyond every cap: left as they are, the setup declines.  (No work lists: ten thousand rows appending
// themselves to one list through one counter cost 100 us.)
__global__ __launch_bounds__(64) void r_sort_medium(const uint32_t *__restrict__ rstart, uint64_t *__restrict__ keys,
                                                    int64_t nc) {
    __shared__ uint64_t buf[RCAP_SMALL];
    const int lane = threadIdx.x;
    for (int64_t I = blockIdx.x; I < nc; I += gridDim.x) {
        const uint32_t s0 = rstart[I];
        const int len = (int)(rstart[I + 1] - s0);
        if (len <= 32 || len > RCAP_SMALL) continue;  // uniform over the wavefront
        const int P = len <= 64 ? 64 : (len <= 128 ? 128 : (len <= 256 ? 256 : 512));
        for (int e = lane; e < P; e += 64) buf[e] = e < len ? keys[s0 + e] : ~0ull;
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                // P / 2 compare-exchanges per step: pair (i, i ^ stride), i = the pair number with bit `stride` opened
                for (int p = lane; p < P / 2; p += 64) {
                    const int i = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                    const int j = i | stride;
                    const bool up = (i & size) == 0;
                    const uint64_t a = buf[i], b = buf[j];
                    if ((a > b) == up) { buf[i] = b; buf[j] = a; }
                }
                __syncthreads();
            }
        for (int e = lane; e < len; e += 64) keys[s0 + e] = buf[e];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------
// Galerkin product A_c = R A P: one wavefront (= one 64-thread workgroup) per coarse row
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int wave_excl_scan(int v, int *total) {
    const int lane = threadIdx.x & 63;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    *total = __shfl(incl, 63, 64);
    return incl - v;
}

// Step 1: AP = A P, one thread per fine row, at most APW distinct columns per row (the row of
// A_c an aggregate ends up with is the union of its members' AP rows, so APW = the coarse row
// cap is always enough; the classes below are picked from the fine row length).  Duplicates are
// merged in registers (compare chains, no dynamic indexing), in (A slot, P slot) order.
template <int APW>
__global__ __launch_bounds__(TB) void ap_rows(Ell A, const int32_t *__restrict__ pcol, const double *__restrict__ pval,
                                              int32_t *__restrict__ apcol, double *__restrict__ apval,
                                              int32_t *__restrict__ aplen, unsigned long long *__restrict__ stats) {
    bool overflow = false;
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < A.n; i += (int64_t)gridDim.x * TB) {
        int32_t c[APW];
        double v[APW];
#pragma unroll
        for (int q = 0; q < APW; ++q) { c[q] = -1; v[q] = 0.0; }
        int cnt = 0;
        const int32_t l = A.len[i];
        for (int32_t s = 0; s < l; ++s) {
            const int32_t k = A.col[(int64_t)s * A.ld + i];
            const double a = A.val[(int64_t)s * A.ld + i];
            int32_t J[PW];
            double pp[PW];
#pragma unroll
            for (int sp = 0; sp < PW; ++sp) {
                J[sp] = pcol[(int64_t)sp * A.ld + k];
                pp[sp] = pval[(int64_t)sp * A.ld + k];
            }
#pragma unroll
            for (int sp = 0; sp < PW; ++sp) {
                if (J[sp] < 0) continue;
                const double t = a * pp[sp];
                bool placed = false;
#pragma unroll
                for (int q = 0; q < APW; ++q)
                    if (!placed && c[q] == J[sp]) { v[q] += t; placed = true; }
#pragma unroll
                for (int q = 0; q < APW; ++q)
                    if (!placed && c[q] < 0) { c[q] = J[sp]; v[q] = t; placed = true; cnt = q + 1; }
                if (!placed) overflow = true;
            }
        }
#pragma unroll
        for (int q = 0; q < APW; ++q)
            if (q < cnt) {  // (the Galerkin kernel reads aplen[i] slots only)
                apcol[(int64_t)q * A.ld + i] = c[q];
                apval[(int64_t)q * A.ld + i] = v[q];
            }
        aplen[i] = cnt;
    }
    if (overflow) atomicOr(&stats[ST_OVERFLOW], 4ull);
}

// The same for the coarser levels, whose rows reach up to 64 aggregates: compare chains over 64
// register slots cost 128 predicated operations per product (400 us for 1.4e5 rows); here every
// lane keeps an open-addressing table of APW slots in LDS (slot-major: the lanes of a wavefront
// never share a bank), filled in (A slot, P slot) order and written out in slot order.
template <int APW>
__global__ __launch_bounds__(64) void ap_rows_lds(Ell A, const int32_t *__restrict__ pcol,
                                                  const double *__restrict__ pval, int32_t *__restrict__ apcol,
                                                  double *__restrict__ apval, int32_t *__restrict__ aplen,
                                                  unsigned long long *__restrict__ stats) {
    __shared__ int32_t tc[APW * 64];
    __shared__ double tv[APW * 64];
    const int lane = threadIdx.x;
    bool overflow = false;
    for (int64_t i0 = (int64_t)xcd_block() * 64; i0 < A.n; i0 += (int64_t)gridDim.x * 64) {
        const int64_t i = i0 + lane;
#pragma unroll 8
        for (int q = 0; q < APW; ++q) tc[q * 64 + lane] = -1;
        int cnt = 0;
        if (i < A.n) {
            const int32_t l = A.len[i];
            // Four entries of the row at a time: their columns and values, then the four P rows they name, are
            // requested together (unconditional loads: a slot past the row's end re-reads the row's first one and is
            // skipped below) -- two dependent round trips per four entries instead of per entry (a row of 27 entries on
            // a level of a few hundred rows was 54 dependent round trips: 33-54 us of pure latency per launch); the
            // products go into the table in the same (A slot, P slot) order: the same sums.
            constexpr int AB = 4;
            for (int32_t s0 = 0; s0 < l; s0 += AB) {
                int32_t kk[AB];
                double aa[AB];
#pragma unroll
                for (int u = 0; u < AB; ++u) {
                    const int32_t su = s0 + u < l ? s0 + u : 0;
                    kk[u] = A.col[(int64_t)su * A.ld + i];
                    aa[u] = A.val[(int64_t)su * A.ld + i];
                }
                int32_t J[AB][PW];
                double pp[AB][PW];
#pragma unroll
                for (int u = 0; u < AB; ++u)
#pragma unroll
                    for (int sp = 0; sp < PW; ++sp) {
                        J[u][sp] = pcol[(int64_t)sp * A.ld + kk[u]];
                        pp[u][sp] = pval[(int64_t)sp * A.ld + kk[u]];
                    }
#pragma unroll
                for (int u = 0; u < AB; ++u) {
                    if (s0 + u >= l) continue;
#pragma unroll
                    for (int sp = 0; sp < PW; ++sp) {
                        if (J[u][sp] < 0) continue;
                        const double t = aa[u] * pp[u][sp];
                        uint32_t slot = ((uint32_t)J[u][sp] * 2654435761u) >> 16 & (APW - 1);
                        bool placed = false;
                        for (int probe = 0; probe < APW && !placed; ++probe) {
                            const int32_t c = tc[slot * 64 + lane];
                            if (c == J[u][sp]) {
                                tv[slot * 64 + lane] += t;
                                placed = true;
                            } else if (c < 0) {
                                tc[slot * 64 + lane] = J[u][sp];
                                tv[slot * 64 + lane] = t;
                                ++cnt;
                                placed = true;
                            }
                            slot = (slot + 1) & (APW - 1);
                        }
                        if (!placed) overflow = true;
                    }
                }
            }
            int o = 0;
            for (int q = 0; q < APW; ++q) {
                const int32_t c = tc[q * 64 + lane];
                if (c >= 0) {
                    apcol[(int64_t)o * A.ld + i] = c;
                    apval[(int64_t)o * A.ld + i] = tv[q * 64 + lane];
                    ++o;
                }
            }
            aplen[i] = cnt;
        }
    }
    if (overflow) atomicOr(&stats[ST_OVERFLOW], 4ull);
}

// Step 2: A_c = R (AP), one wavefront (= one 64-thread workgroup) per coarse row.  The products
// w_t * AP[i_t, :] of a group of G entries of the R row go to an LDS list in (R entry, AP slot)
// order; pass 1 collects the row's column set in an LDS hash set, which is sorted with wave
// shuffles (lane l owns column l); pass 2 walks the list and every lane sums the products of
// its column in list order.  A row of one group (the usual case) keeps its products in LDS
// between the passes.  Three dependent round trips per group: R entry -> AP row length / slots.
__device__ __forceinline__ int galerkin_products(int64_t ld, const int32_t *__restrict__ apcol,
                                                 const double *__restrict__ apval, const int32_t *__restrict__ aplen,
                                                 int64_t rld, const int32_t *__restrict__ rcol,
                                                 const double *__restrict__ rval, int64_t I, int t0, int tn,
                                                 int32_t *lJ, double *lV) {
    const int lane = threadIdx.x;
    int32_t i = 0, len = 0;
    double w = 0.0;
    if (lane < tn) {
        const int t = t0 + lane;
        const int64_t rat = ((int64_t)(t >> 3) * rld + I) * 8 + (t & 7);
        i = rcol[rat];
        w = rval[rat];
        len = aplen[i];
    }
    int np = 0;
    int o = wave_excl_scan(len, &np);
    constexpr int INFL = 8;  // slots in flight (a row of A P has 9-13 entries on a grid: two dependent rounds instead of four)
    for (int32_t s0 = 0; s0 < len; s0 += INFL) {
        int32_t cj[INFL];
        double cv[INFL];
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
            const int32_t s = s0 + q < len ? s0 + q : s0;
            cj[q] = apcol[(int64_t)s * ld + i];
            cv[q] = apval[(int64_t)s * ld + i];
        }
#pragma unroll
        for (int q = 0; q < INFL; ++q)
            if (s0 + q < len) {
                lJ[o] = cj[q];
                lV[o] = w * cv[q];
                ++o;
            }
    }
    for (int p = lane + np; p < np + 72; p += 64) lJ[p] = -1;  // empty products behind the list (the accumulation
                                                              // loop reads whole segments of multiples of 8)
    __syncthreads();
    return np;
}

// G: R entries per group (G * APW products fit in the LDS list)
// NUMERIC: the row's sorted column set is already in ccol / clen (same pattern as at the symbolic
// setup): no hash set, no sort -- every lane takes its column and sums its products.
template <bool NUMERIC>
__global__ __launch_bounds__(64) void galerkin(int64_t ld, const int32_t *__restrict__ apcol,
                                               const double *__restrict__ apval, const int32_t *__restrict__ aplen,
                                               int64_t nc, int64_t rld, const int32_t *__restrict__ rcol,
                                               const double *__restrict__ rval, const int32_t *__restrict__ rlen,
                                               int64_t cld, int32_t *__restrict__ ccol, double *__restrict__ cval,
                                               float *__restrict__ cvalf,
                                               int32_t *__restrict__ clen, double *__restrict__ cdinv,
                                               unsigned long long *__restrict__ stats, uint32_t *__restrict__ bstat,
                                               int G, int pcap, int maxseg) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    int32_t *set = reinterpret_cast<int32_t *>(gsm);         // [256]
    double *lV = reinterpret_cast<double *>(gsm + 1024);     // [pcap]
    int32_t *lJ = reinterpret_cast<int32_t *>(lV + pcap);    // [pcap]
    const int lane = threadIdx.x;
    uint32_t my_max = 0, my_nnz = 0, my_flags = 0;
    for (int64_t I = blockIdx.x; I < nc; I += gridDim.x) {
        const int rl = rlen[I];
        const int ngroups = (rl + G - 1) / G;
        int np = 0;
        int total = 0;
        int32_t myJ = 0x7fffffff;
        if constexpr (NUMERIC) {
            total = clen[I];
            if (lane < total) myJ = ccol[(int64_t)lane * cld + I];
        } else {
        for (int k = lane; k < 256; k += 64) set[k] = -1;
        __syncthreads();
        // pass 1: the set of columns
        for (int g = 0; g < ngroups; ++g) {
            np = galerkin_products(ld, apcol, apval, aplen, rld, rcol, rval, I, g * G, min(G, rl - g * G), lJ, lV);
            for (int p = lane; p < np; p += 64) {
                const int32_t J = lJ[p];
                uint32_t hsh = ((uint32_t)J * 2654435761u) >> 24;
                for (int probe = 0; probe < 256; ++probe) {
                    const int32_t old = atomicCAS(&set[hsh], -1, J);
                    if (old == -1 || old == J) break;
                    hsh = (hsh + 1) & 255u;
                    if (probe == 255) my_flags |= 1u;
                }
            }
            __syncthreads();
        }
        // compact the set (each lane owns 4 slots), sort it with wave shuffles
        int32_t mine[4];
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mine[q] = set[lane * 4 + q];
            cnt += mine[q] >= 0 ? 1 : 0;
        }
        int off = wave_excl_scan(cnt, &total);
        __syncthreads();
        // (the compacted columns go through `set` itself: every lane has read its slots)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (mine[q] >= 0) set[off++] = mine[q];
        __syncthreads();
        if (total > ACAP) {
            my_flags |= 1u;
            total = ACAP;
        }
        myJ = lane < total ? set[lane] : 0x7fffffff;
        __syncthreads();
#pragma unroll
        for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
            for (int j = k >> 1; j > 0; j >>= 1) {
                const int32_t other = __shfl_xor(myJ, j, 64);
                const bool up = (lane & k) == 0, lower = (lane & j) == 0;
                const int32_t lo = myJ < other ? myJ : other, hi = myJ < other ? other : myJ;
                myJ = (lower == up) ? lo : hi;
            }
        }
        }  // !NUMERIC
        // pass 2: eight products per step, loaded unconditionally (a compare-then-load loop pays the
        // LDS latency twice per product); + 0.0 for the others leaves the sum, and its order, unchanged.
        // Every lane reading every product made the kernel LDS-bandwidth-bound (a wave-wide 8-byte read takes
        // 4 cycles whether it is a broadcast or not: 1500 cycles per coarse row of 250 products, 336 us at 1e6
        // nodes).  A row of at most 16 (32) columns is therefore summed by 4 (2) SEGMENTS of lanes: lane
        // (seg, c) sums the products of column c in segment seg of the list -- a quarter (half) of the reads,
        // four (two) addresses per read instruction -- and the partial sums are combined in a fixed order.
        const int nseg = min(maxseg, total <= 16 ? 4 : (total <= 32 ? 2 : 1)), segw = 64 / nseg;
        const int seg = lane / segw;
        const int32_t segJ = __shfl(myJ, lane % segw, 64);  // (lane c < total holds column c)
        double acc = 0.0;
        for (int g = 0; g < ngroups; ++g) {
            if (ngroups > 1 || NUMERIC)
                np = galerkin_products(ld, apcol, apval, aplen, rld, rcol, rval, I, g * G, min(G, rl - g * G), lJ, lV);
            int chunk = (((np + nseg - 1) / nseg) + 7) & ~7;
            if (nseg > 1 && ((chunk >> 3) & 1) == 0) chunk += 8;  // an odd number of 64-byte lines between the segments' reads: fewer bank conflicts
            // (nseg * chunk <= np + 64: behind the list there are empty products)
            for (int p = seg * chunk; p < (seg + 1) * chunk; p += 8) {
                int32_t j8[8];
                double v8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    j8[u] = lJ[p + u];
                    v8[u] = lV[p + u];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += j8[u] == segJ ? v8[u] : 0.0;
            }
            __syncthreads();
        }
        if (nseg == 4) {
            const double a1 = __shfl(acc, lane + 16, 64), a2 = __shfl(acc, lane + 32, 64), a3 = __shfl(acc, lane + 48, 64);
            acc = (acc + a1) + (a2 + a3);
        } else if (nseg == 2) {
            acc += __shfl(acc, lane + 32, 64);
        }
        if (lane >= total && lane < APAD) {  // zero padding: fixed-trip-count row loops (tail, unrolled kernels)
            ccol[(int64_t)lane * cld + I] = (int32_t)I;
            cval[(int64_t)lane * cld + I] = 0.0;
            cvalf[(int64_t)lane * cld + I] = 0.0f;
        }
        if (lane < total) {
            ccol[(int64_t)lane * cld + I] = myJ;
            cval[(int64_t)lane * cld + I] = acc;
            cvalf[(int64_t)lane * cld + I] = (float)acc;
            if (myJ == (int32_t)I) {
                if (!(acc > 0.0)) my_flags |= 4u;
                cdinv[I] = acc > 0.0 ? 1.0 / acc : 1.0;
            }
        }
        if (lane == 0) {
            clen[I] = total;
            my_max = (uint32_t)total > my_max ? (uint32_t)total : my_max;
            my_nnz += (uint32_t)total;
        }
        __syncthreads();
    }
    if (lane == 0) {
        bstat[blockIdx.x] = my_max;
        bstat[BSTAT_MAX + blockIdx.x] = my_nnz;
    }
    if (my_flags & 1u) atomicOr(&stats[ST_OVERFLOW], 1ull);  // (rare)
    if (my_flags & 4u) atomicOr(&stats[ST_BADDIAG], 1ull);
}

// dense inverse of the coarsest matrix (n <= COARSEST) by Gauss-Jordan in LDS; SPD, no pivoting
__global__ __launch_bounds__(256) void coarsest_inverse(Ell A, double *__restrict__ inv,
                                                        unsigned long long *__restrict__ stats) {
    __shared__ double M[COARSEST][2 * COARSEST + 1];
    const int n = (int)A.n;
    for (int t = threadIdx.x; t < n * 2 * n; t += 256) {
        const int r = t / (2 * n), c = t % (2 * n);
        M[r][c] = (c >= n && c - n == r) ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < n; r += 256)
        for (int s = 0; s < A.len[r]; ++s) M[r][A.col[(int64_t)s * A.ld + r]] = A.val[(int64_t)s * A.ld + r];
    __syncthreads();
    for (int k = 0; k < n; ++k) {
        const double pv = M[k][k];
        if (!(pv > 0.0)) {
            if (threadIdx.x == 0) atomicOr(&stats[ST_BADDIAG], 2ull);
            return;  // uniform: every thread reads the same pivot
        }
        const double rp = 1.0 / pv;
        __syncthreads();
        for (int c = threadIdx.x; c < 2 * n; c += 256) M[k][c] *= rp;
        __syncthreads();
        for (int t = threadIdx.x; t < n * 2 * n; t += 256) {
            const int r = t / (2 * n), c = t % (2 * n);
            if (r != k && c != k) M[r][c] = fma(-M[r][k], M[k][c], M[r][c]);
        }
        __syncthreads();
        for (int r = threadIdx.x; r < n; r += 256)
            if (r != k) M[r][k] = 0.0;
        __syncthreads();
    }
    for (int t = threadIdx.x; t < n * n; t += 256) inv[t] = M[t / n][n + t % n];
}

// ---- structural singularity: "touches ground" flags OR-ed up the hierarchy, components of the
//      last level by label propagation in one workgroup ------------------------------------
__global__ __launch_bounds__(TB) void flags_up(int64_t n, const int32_t *__restrict__ agg,
                                               const uint8_t *__restrict__ fine, uint8_t *__restrict__ coarse) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        if (fine[i]) coarse[agg[i]] = 1;  // benign race: every writer stores 1
}
__global__ __launch_bounds__(64) void last_level_floating(Ell A, const uint8_t *__restrict__ grounded,
                                                          uint32_t *__restrict__ floating) {
    __shared__ int label[COARSEST];
    __shared__ int changed;
    const int n = (int)A.n, i = threadIdx.x;
    if (i < n) label[i] = i;
    __syncthreads();
    for (int it = 0; it < COARSEST; ++it) {
        if (i == 0) changed = 0;
        __syncthreads();
        int best = i < n ? label[i] : 0;
        if (i < n)
            for (int s = 0; s < A.len[i]; ++s) {
                const int l = label[A.col[(int64_t)s * A.ld + i]];
                best = l < best ? l : best;
            }
        __syncthreads();
        if (i < n && best < label[i]) {
            label[i] = best;
            changed = 1;
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    __shared__ int ok[COARSEST];
    if (i < n) ok[i] = 0;
    __syncthreads();
    if (i < n && grounded[i]) ok[label[i]] = 1;
    __syncthreads();
    if (i < n && label[i] == i && !ok[i]) *floating = 1;
}

// diagonal last level: every node is a component of its own
__global__ __launch_bounds__(TB) void any_unflagged(int64_t n, const uint8_t *__restrict__ grounded,
                                                    uint32_t *__restrict__ floating) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        if (!grounded[i]) *floating = 1;  // benign race
}

}  // namespace

#include "sagg_cycle.h"

// ---------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------
namespace {

constexpr int SAGG_DECLINED = -3;  // not this path's kind of network: the caller takes amg.hip

// smallest unrolled width class >= maxlen (0: none)
int width_class(int maxlen) {
    static const int classes[] = {4, 5, 6, 8, 12, 16, 20, 24, 32};
    for (int c : classes)
        if (maxlen <= c) return c;
    return 0;
}
// padded rows pay when the padding is small, or when the level is so small that latency is all
int choose_wfix(int maxlen, int64_t n, int64_t nnz, int pad_limit) {
    const int c = width_class(maxlen);
    if (c == 0 || c > pad_limit) return 0;
    static const double slack = getenv("NODAL_SA_PADSLACK") ? atof(getenv("NODAL_SA_PADSLACK")) : 1.3;
    if ((double)c * (double)n <= slack * (double)nnz || n <= 32768) return c;
    return 0;
}

SHierarchy *hierarchy_of(nodal_ctx *h) {
    if (!h->sagg) h->sagg = new SHierarchy();
    return static_cast<SHierarchy *>(h->sagg);
}

struct SolveBufs {
    double *r, *p, *p2, *Ap, *part_rz, *part_zap, *part_rr, *part_pap, *sc;
    cyc_t *z, *x0;
    int g0;  // grid of the level-0 kernels that produce / consume dot partials
};

// A_c = R (A P) of level l into level l + 1: AP = A P (one thread per fine row), then one
// wavefront per coarse row.  NUMERIC: the pattern of A_c is already there (see SHierarchy::sym_valid).
// phase 1: A P only; phase 2: the sums only (after r_ready, the event behind R's construction on the hierarchy's other
// stream); 0: both
template <bool NUMERIC>
int galerkin_product(nodal_ctx *h, SHierarchy *H, int l, hipEvent_t r_ready = nullptr, int phase = 0) {
    hipStream_t st = h->stream;
    SLevel *L = H->level(l), *C = H->level(l + 1);
    const int64_t n = L->n, ld = L->ld, nc = C->n;
    Ell A = L->A();
    A.width = L->maxlen;
    unsigned long long *dstats = H->stats.as<unsigned long long>();
    // (an AP row reaches the aggregates within two steps of the node: up to ~10 on a 5-point grid)
    const int apw = A.width <= 6 ? 16 : 64;
    NODAL_HIP_TRY(h, H->apcol.reserve((size_t)apw * ld * 4 + 64));
    NODAL_HIP_TRY(h, H->apval.reserve((size_t)apw * ld * 8 + 64));
    NODAL_HIP_TRY(h, H->aplen.reserve((size_t)ld * 4 + 64));
    int32_t *apcol = H->apcol.as<int32_t>(), *aplen = H->aplen.as<int32_t>();
    double *apval = H->apval.as<double>();
    unsigned long long *lst = dstats + (size_t)l * ST_COUNT;
    const unsigned g = grid_for(n);
    if (phase != 2) {
        switch (apw) {
        case 16: ap_rows<16><<<g, TB, 0, st>>>(A, L->pcol.as<int32_t>(), L->pval.as<double>(), apcol, apval, aplen, lst); break;
        default: {
            const unsigned gl = (unsigned)((n + 63) / 64 < 4096 ? (n + 63) / 64 : 4096);
            ap_rows_lds<64><<<gl, 64, 0, st>>>(A, L->pcol.as<int32_t>(), L->pval.as<double>(), apcol, apval, aplen, lst);
        } break;
        }
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    if (phase == 1) return NODAL_OK;
    // R entries per group: G * apw products (12 bytes each) in the LDS list
    int G = 1024 / apw;
    G = (G < 8 ? 8 : (G > 64 ? 64 : G));
    // (level 0: 40 instead of 64 R entries per group -- 8.2 instead of 13.4 KB of LDS per wavefront, 19
    // instead of 11 wavefronts per CU; the few rows of R beyond 40 entries take a second group:
    // 8.80 -> 8.66 ms per fresh solve of the 1e6-node grid, 7.45 -> 7.22 ms with the patterns kept)
    static const int g_env = getenv("NODAL_SA_GG") ? atoi(getenv("NODAL_SA_GG")) : 40;
    if (g_env >= 8 && g_env <= 64 && apw == 16) G = g_env & ~7;
    const int pcap = G * apw + 72;  // (the list is padded with empty products up to the segments' common length)
    const size_t lds = 1024 + (size_t)pcap * 12;
    // (at most 16384 workgroups, each walking several rows: the one-workgroup fold of their statistics
    // reads 2 x 16384 words instead of 2 x 65536 -- 3 instead of 10 us)
    static const int64_t gcap = getenv("NODAL_SA_GCAP") ? atoll(getenv("NODAL_SA_GCAP")) : 16384;
    static const int maxseg = getenv("NODAL_SA_GSEG") ? atoi(getenv("NODAL_SA_GSEG")) : 4;
    const unsigned gg = (unsigned)(nc < gcap ? nc : (gcap < BSTAT_MAX ? gcap : BSTAT_MAX));
    if (r_ready) NODAL_HIP_TRY(h, hipStreamWaitEvent(st, r_ready, 0));  // (R was built on the hierarchy's other stream)
    galerkin<NUMERIC><<<gg, 64, lds, st>>>(
        ld, apcol, apval, aplen, nc, L->rld, L->rcol.as<int32_t>(), L->rval.as<double>(), L->rlen.as<int32_t>(),
        C->ld, C->acol.as<int32_t>(), C->aval.as<double>(), C->avalf.as<float>(), C->alen.as<int32_t>(),
        C->dinv.as<double>(), dstats + (size_t)(l + 1) * ST_COUNT, H->bstat.as<uint32_t>(), G, pcap, maxseg);
    if (!NUMERIC)
        reduce_bstat<<<1, 1024, 0, st>>>((int)gg, H->bstat.as<uint32_t>(), dstats + (size_t)(l + 1) * ST_COUNT,
                                         ST_MAXLEN, ST_NNZ);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// one aggregation + transfer operators + Galerkin product: level l -> l + 1.
// *stop: the level cannot be coarsened (every node is its own aggregate).
// (NODAL_SA_OMEGA_P="w0" or "w0,w1": the weight of level 0 and of the coarse levels)
double omega_p(int level) {
    static double w[2] = {-1.0, -1.0};
    if (w[0] < 0.0) {
        double a = OMEGA_P, b = OMEGA_P_COARSE;
        if (const char *e = getenv("NODAL_SA_OMEGA_P")) {
            a = atof(e);
            const char *c = strchr(e, ',');
            b = c ? atof(c + 1) : a;
        }
        w[1] = b > 0.0 && b < 1.2 ? b : OMEGA_P_COARSE;
        w[0] = a > 0.0 && a < 1.0 ? a : OMEGA_P;
    }
    return w[level > 0 ? 1 : 0];
}

int build_level(nodal_ctx *h, SHierarchy *H, int l, unsigned long long *hs, bool *declined, bool *stop) {
    hipStream_t st = h->stream;
    SLevel *L = H->level(l);
    const int64_t n = L->n, ld = L->ld;
    Ell A = L->A();
    A.width = L->maxlen;  // pair enumeration of the Galerkin kernel: slots actually in use
    unsigned long long *dstats = H->stats.as<unsigned long long>();

    NODAL_HIP_TRY(h, H->mis_t.reserve((size_t)n * 8 + 64));
    NODAL_HIP_TRY(h, H->mis_m.reserve((size_t)n * 8 + 64));
    NODAL_HIP_TRY(h, H->mis_flag.reserve((size_t)(n + 1) * 4 + 64));
    NODAL_HIP_TRY(h, H->mis_id.reserve((size_t)(n + 1) * 4 + 64 + scan_tmp_bytes(n + 1)));
    NODAL_HIP_TRY(h, H->agg1.reserve((size_t)n * 4 + 64));
    NODAL_HIP_TRY(h, L->agg.reserve((size_t)n * 4 + 64));
    NODAL_HIP_TRY(h, H->cursor.reserve((MIS_ROUNDS + 2) * 8 + 64));
    uint32_t *T = H->mis_t.as<uint32_t>(), *M = H->mis_m.as<uint32_t>();
    uint32_t *flag = H->mis_flag.as<uint32_t>(), *id = H->mis_id.as<uint32_t>();
    unsigned long long *cnt = H->cursor.as<unsigned long long>();
    void *scan_tmp = H->mis_id.as<char>() + (((size_t)(n + 1) * 4 + 63) & ~(size_t)63);

    const unsigned g = grid_for(n);
    static const bool trace_mis = getenv("NODAL_TRACE") != nullptr;
    if (trace_mis) NODAL_HIP_TRY(h, hipMemsetAsync(cnt, 0, (MIS_ROUNDS + 2) * 8, st));
    static const int mis_rounds = getenv("NODAL_SA_MIS") ? std::min(MIS_ROUNDS, std::max(1, atoi(getenv("NODAL_SA_MIS")))) : MIS_ROUNDS;
    if (n <= MIS_SMALL_MAX && !trace_mis) {
        mis_small<<<1, 1024, 0, st>>>(A, T, M, flag, mis_rounds);
    } else {
        mis_init<<<g, TB, 0, st>>>(n, T);
        for (int r = 0; r < mis_rounds; ++r) {
            mis_max<<<g, TB, 0, st>>>(A, T, M);
            mis_update<<<g, TB, 0, st>>>(A, T, M);
            if (trace_mis) mis_count<<<g, TB, 0, st>>>(n, T, cnt + r + 1);
        }
        mis_flag_roots<<<grid_for(n + 1), TB, 0, st>>>(A, T, flag);
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    // (the scan leaves its total -- flag[n] is 0 -- in the level's statistics block: ONE copy to pinned memory
    // brings everything; a second 4-byte copy to a stack variable was staged by the runtime and cost 18 us)
    NODAL_TRY(scan_exclusive_u32(h, flag, id, n + 1, reinterpret_cast<uint32_t *>(dstats + (size_t)l * ST_COUNT + ST_NC),
                                 scan_tmp));
    // the one round trip of this level: the number of aggregates, and what the Galerkin kernel
    // of the level above recorded about THIS level's matrix
    NODAL_HIP_TRY(h, hipMemcpyAsync(hs, dstats, (size_t)MAX_LEVELS * ST_COUNT * 8, hipMemcpyDeviceToHost, st));
    unsigned long long hcnt[MIS_ROUNDS + 1] = {0};
    if (trace_mis) NODAL_HIP_TRY(h, hipMemcpyAsync(hcnt, cnt, sizeof hcnt, hipMemcpyDeviceToHost, st));
    if (!H->ev_copy) NODAL_HIP_TRY(h, hipEventCreateWithFlags(&H->ev_copy, hipEventDisableTiming));
    NODAL_HIP_TRY(h, hipEventRecord(H->ev_copy, st));
    // (round 4) The aggregates' assignment and P depend on the level's size only, not on the count the host is
    // waiting for: they are enqueued BEFORE the wait, which then costs nothing -- the copy has long finished when
    // build_P has (the wait used to leave the GPU idle for 25-30 us per level).  If the level turns out to be the
    // last one, or the hierarchy is declined, three small kernels ran for nothing.
    NODAL_HIP_TRY(h, L->pcol.reserve((size_t)PW * ld * 4 + 64));
    NODAL_HIP_TRY(h, L->pval.reserve((size_t)PW * ld * 8 + 64));
    NODAL_HIP_TRY(h, L->pvalf.reserve((size_t)PW * ld * 4 + 64));
    assign_near<<<g, TB, 0, st>>>(A, T, flag, id, H->agg1.as<int32_t>());
    assign_far<<<g, TB, 0, st>>>(A, H->agg1.as<int32_t>(), L->agg.as<int32_t>(), dstats + (size_t)l * ST_COUNT);
    build_P<<<g, TB, 0, st>>>(A, L->dinv.as<double>(), L->agg.as<int32_t>(), L->pcol.as<int32_t>(),
                             L->pval.as<double>(), L->pvalf.as<float>(), omega_p(l));
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_WAIT_EVENT(h, H->ev_copy, st);  // (the copies, not the three kernels behind them)
    if (trace_mis) {
        fprintf(stderr, "[sagg] level %d: undecided after each MIS round:", l);
        for (int r = 1; r <= MIS_ROUNDS; ++r) fprintf(stderr, " %llu", hcnt[r]);
        fprintf(stderr, "\n");
    }
    const int64_t nc = (int64_t)(hs[(size_t)l * ST_COUNT + ST_NC] & 0xffffffffull);
    if (l > 0) {  // what the Galerkin kernel of the level above recorded about this level's matrix
        const unsigned long long *s = hs + (size_t)l * ST_COUNT;
        L->maxlen = (int32_t)s[ST_MAXLEN];
        L->nnz = (int64_t)s[ST_NNZ];
        A.width = L->maxlen;
        if (s[ST_OVERFLOW] || s[ST_BADDIAG] || hs[(size_t)(l - 1) * ST_COUNT + ST_OVERFLOW] ||
            hs[(size_t)(l - 1) * ST_COUNT + ST_UNASSIGNED]) {
            *declined = true;
            return NODAL_OK;
        }
    }
    if (nc >= n || L->maxlen <= 1) {  // nothing but isolated nodes (one per independent circuit): last level
        *stop = true;
        return NODAL_OK;
    }
    // coarsening too slow to pay: not this path's kind of graph.  (Small levels may shrink by less: the
    // last steps of a batch of small circuits go from one or two nodes per member to one.)
    if ((nc * 2 > n && n > 4096) || nc < 1) {
        *declined = true;
        return NODAL_OK;
    }
    L->nc = nc;

    // R = P^T by coarse row -- on the hierarchy's other stream when there is one (see SHierarchy::aux): every
    // launch of this section, the scans included, goes where h->stream points
    static const bool fork_allowed = !(getenv("NODAL_SA_FORK") && atoi(getenv("NODAL_SA_FORK")) == 0);
    if (H->aux && !nodal_extra_streams_ok(h) && nodal_calls_in_flight() <= 1) {  // (the option was taken back: so is the stream's hardware queue)
        NODAL_WAIT_STREAM(h, H->aux);
        (void)hipStreamDestroy(H->aux);
        H->aux = nullptr;
        H->aux_tried = false;
    }
    if (fork_allowed && !H->aux_tried && nodal_extra_streams_ok(h)) {
        H->aux_tried = true;
        if (hipStreamCreateWithFlags(&H->aux, hipStreamNonBlocking) != hipSuccess) H->aux = nullptr;
        if (H->aux && ((!H->ev_fork && hipEventCreateWithFlags(&H->ev_fork, hipEventDisableTiming) != hipSuccess) ||
                       (!H->ev_join && hipEventCreateWithFlags(&H->ev_join, hipEventDisableTiming) != hipSuccess) ||
                       (!H->ev_flags && hipEventCreateWithFlags(&H->ev_flags, hipEventDisableTiming) != hipSuccess))) {
            (void)hipStreamDestroy(H->aux);
            H->aux = nullptr;
        }
        (void)hipGetLastError();
    }
    const bool forked = fork_allowed && H->aux != nullptr && nodal_extra_streams_ok(h);  // (see api.hip)
    // every buffer the section writes is reserved BEFORE the fork: a buffer that grows is filled on the context's
    // own stream (DevBuf::reserve), which the other stream does not wait for after the fork
    const size_t a4 = (((size_t)(nc + 1) * 4) + 255) & ~(size_t)255;
    const size_t scan2 = (scan_tmp_bytes(nc + 1) + 255) & ~(size_t)255;
    NODAL_HIP_TRY(h, H->rstart.reserve(a4 + 256 + scan2 + (size_t)PW * ld * 4 + 64));
    NODAL_HIP_TRY(h, H->keys.reserve((size_t)PW * n * 8 + 64));
    SLevel *C = H->level(l + 1);
    C->n = nc;
    C->ld = pad64(nc);
    C->width = ACAP;
    C->nc = 0;
    L->rld = C->ld;
    const int rcap = rcap_for(nc);
    NODAL_HIP_TRY(h, L->rcol.reserve((size_t)rcap * C->ld * 4 + 64));
    NODAL_HIP_TRY(h, L->rval.reserve((size_t)rcap * C->ld * 8 + 64));
    NODAL_HIP_TRY(h, L->rvalf.reserve((size_t)rcap * C->ld * 4 + 64));
    NODAL_HIP_TRY(h, L->rlen.reserve((size_t)C->ld * 4 + 64));
    // (and the ones the Galerkin product fills on the main stream meanwhile)
    NODAL_HIP_TRY(h, C->acol.reserve((size_t)ACAP * C->ld * 4 + 64));
    NODAL_HIP_TRY(h, C->aval.reserve((size_t)ACAP * C->ld * 8 + 64));
    NODAL_HIP_TRY(h, C->avalf.reserve((size_t)ACAP * C->ld * 4 + 64));
    NODAL_HIP_TRY(h, C->alen.reserve((size_t)C->ld * 4 + 64));
    NODAL_HIP_TRY(h, C->dinv.reserve((size_t)C->ld * 8 + 64));
    if (forked && H->want_flags) {
        if (l == 0) NODAL_HIP_TRY(h, L->gflag.reserve((size_t)n + 64));
        NODAL_HIP_TRY(h, C->gflag.reserve((size_t)C->n + 64));
    }
    {
        const int apw = L->maxlen <= 6 ? 16 : 64;  // (as galerkin_product sizes them)
        NODAL_HIP_TRY(h, H->apcol.reserve((size_t)apw * ld * 4 + 64));
        NODAL_HIP_TRY(h, H->apval.reserve((size_t)apw * ld * 8 + 64));
        NODAL_HIP_TRY(h, H->aplen.reserve((size_t)ld * 4 + 64));
    }
    hipStream_t main_st = h->stream;
    struct StreamGuard {  // (whatever way this function is left, the context gets its stream back)
        nodal_ctx *h;
        hipStream_t keep;
        ~StreamGuard() { h->stream = keep; }
    } guard{h, main_st};
    if (forked) {
        NODAL_HIP_TRY(h, hipEventRecord(H->ev_fork, main_st));
        NODAL_HIP_TRY(h, hipStreamWaitEvent(H->aux, H->ev_fork, 0));
        H->aux_pending = true;
        NODAL_TRY(galerkin_product<false>(h, H, l, nullptr, 1));  // A P first: the long launch of the pair
        h->stream = H->aux;
        st = H->aux;
    }
    char *rs = H->rstart.as<char>();
    uint32_t *rstart = reinterpret_cast<uint32_t *>(rs);
    void *scan_tmp2 = rs + a4 + 256;
    uint32_t *place = reinterpret_cast<uint32_t *>(rs + a4 + 256 + scan2);
    NODAL_HIP_TRY(h, hipMemsetAsync(rs, 0, a4 + 256, st));
    r_count<<<g, TB, 0, st>>>(n, ld, L->pcol.as<int32_t>(), rstart, place);
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_TRY(scan_exclusive_u32(h, rstart, rstart, nc + 1, nullptr, scan_tmp2));
    uint64_t *keys = H->keys.as<uint64_t>();
    r_fill<<<g, TB, 0, st>>>(n, ld, L->pcol.as<int32_t>(), rstart, place, keys);
    r_sort_short<<<grid_for(nc), TB, 0, st>>>(rstart, keys, nc);
    r_sort_medium<<<(unsigned)(nc < 8192 ? nc : 8192), 64, 0, st>>>(rstart, keys, nc);
    NODAL_HIP_TRY(h, hipGetLastError());

    {
        const unsigned gr = grid_for(nc * 8);
        r_to_ell<<<gr, TB, 0, st>>>(nc, L->rld, rstart, keys, ld, L->pval.as<double>(), L->rcol.as<int32_t>(),
                                   L->rval.as<double>(), L->rvalf.as<float>(), L->rlen.as<int32_t>(),
                                   dstats + (size_t)l * ST_COUNT, H->bstat.as<uint32_t>(), (uint32_t)rcap);
        reduce_bstat<<<1, 1024, 0, st>>>((int)gr, H->bstat.as<uint32_t>(), dstats + (size_t)l * ST_COUNT, ST_MAXR, -1);
    }
    if (forked) {
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_HIP_TRY(h, hipEventRecord(H->ev_join, H->aux));
        if (H->want_flags && H->flags_levels == l) {  // behind R, while the main stream sums the Galerkin products
            if (l == 0) NODAL_TRY(grounded_flags(h, L->gflag.as<uint8_t>()));  // (on h->stream = the other stream)
            NODAL_HIP_TRY(h, hipMemsetAsync(C->gflag.p, 0, ((size_t)C->n + 63) & ~(size_t)63, st));
            flags_up<<<grid_for(n), TB, 0, st>>>(n, L->agg.as<int32_t>(), L->gflag.as<uint8_t>(), C->gflag.as<uint8_t>());
            NODAL_HIP_TRY(h, hipGetLastError());
            NODAL_HIP_TRY(h, hipEventRecord(H->ev_flags, H->aux));
            H->flags_levels = l + 1;
        }
        h->stream = main_st;
        st = main_st;
    }
    NODAL_TRY(galerkin_product<false>(h, H, l, forked ? H->ev_join : nullptr, forked ? 2 : 0));
    // (the main stream now waits for R; flags that went up behind R are joined by sagg_setup_csr, through ev_flags)
    if (forked && H->flags_levels != l + 1) H->aux_pending = false;
    return NODAL_OK;
}

// the tail kernel whose register slots cover ceil(width / lpr) of the first tail level
const void *tail_kernel(int slots) {
    return slots <= 8 ? reinterpret_cast<const void *>(k_tail<8>)
                      : slots <= 16 ? reinterpret_cast<const void *>(k_tail<16>)
                                    : reinterpret_cast<const void *>(k_tail<32>);
}

int build_tail(nodal_ctx *h, SHierarchy *H, const unsigned long long *hs) {
    H->tail = -1;
    const int last = H->nlev - 1;
    auto up16 = [](int x) { return (x + 15) & ~15; };
    for (int t = 1; t <= last; ++t) {
        if (H->pool[t]->n > TAIL_MAX_N || last - t + 1 > TAIL_LEVELS) continue;
        TailDesc d;
        memset(&d, 0, sizeof d);
        d.nlev = last - t + 1;
        d.nu = getenv("NODAL_SA_TAIL_NU") ? atoi(getenv("NODAL_SA_TAIL_NU")) : 3;
        if (d.nu < 1) d.nu = 1;
        int off = 0;
        for (int k = 0; k < d.nlev; ++k) {  // image part
            const SLevel *L = H->pool[t + k];
            TailLevelDesc &q = d.lv[k];
            q.n = (int)L->n; q.nc = (int)L->nc; q.ld = (int)L->ld; q.rld = (int)L->rld; q.width = L->maxlen;
            q.acol = L->acol.as<int32_t>(); q.aval = L->aval.as<double>();
            q.dinv = L->dinv.as<double>(); q.alen = L->alen.as<int32_t>();
            q.pcol = L->pcol.as<int32_t>(); q.pval = L->pval.as<double>();
            q.rcol = L->rcol.as<int32_t>(); q.rval = L->rval.as<double>(); q.rlen = L->rlen.as<int32_t>();
            if (k == d.nlev - 1) {  // dense inverse (n^2) or diagonal (n)
                q.o_aval = off; off = up16(off + (H->dense_coarsest ? q.n * q.n : q.n) * 8);
                continue;
            }
            // (slots past a row's end hold zeros up to APAD only: the pack kernel masks by the row length)
            const int maxr = (int)hs[(size_t)(t + k) * ST_COUNT + ST_MAXR];
            q.nq = (maxr + 7) / 8;
            q.lpr = 1;
            while (q.lpr < 8 && q.n * q.lpr * 2 <= 1024) q.lpr *= 2;
            q.o_aval = off; off = up16(off + q.n * q.width * 8);
            q.o_acol = off; off = up16(off + q.n * q.width * 2);
            q.o_dinv = off; off = up16(off + q.n * 8);
            q.o_pval = off; off = up16(off + q.n * PW * 8);
            q.o_pcol = off; off = up16(off + q.n * PW * 2);
            q.o_rval = off; off = up16(off + q.nq * q.nc * 8 * 8);
            q.o_rcol = off; off = up16(off + q.nq * q.nc * 8 * 2);
        }
        d.image_bytes = off;
        for (int k = 0; k < d.nlev; ++k) {  // vectors
            TailLevelDesc &q = d.lv[k];
            q.o_B = off; off = up16(off + q.n * 8);
            q.o_X = off; off = up16(off + q.n * 8);
            if (k == d.nlev - 1) continue;
            q.o_Y = off; off = up16(off + q.n * 8);
            q.o_R = off; off = up16(off + q.n * 8);
        }
        d.inv = H->dense_coarsest ? H->coarse_inv.as<double>() : nullptr;
        d.skip = d.nlev > 1 ? d.lv[0].o_dinv : 0;
        d.lds_bytes = off - d.skip;
        if (getenv("NODAL_TAIL_PROBE")) {
            NODAL_HIP_TRY(h, H->tail_stamps.reserve(64 * sizeof(long long)));
            d.stamps = H->tail_stamps.as<long long>();
        }
        if (d.lds_bytes > TAIL_LDS_BUDGET) continue;
        d.slots = d.nlev > 1 ? (d.lv[0].width + d.lv[0].lpr - 1) / d.lv[0].lpr : 1;
        if (d.slots > 32) continue;
        {   // the attribute belongs to the device's code object, not to this handle: raised ONCE per device
            // to the budget (a per-setup value could be lowered by another handle's smaller tail between
            // this setup and this hierarchy's launches)
            static std::atomic<bool> lds_allowed[64][3];
            const int dev = h->device >= 0 && h->device < 64 ? h->device : 0;
            const int which = d.slots <= 8 ? 0 : (d.slots <= 16 ? 1 : 2);
            if (!lds_allowed[dev][which].load(std::memory_order_acquire)) {
                NODAL_HIP_TRY(h, hipFuncSetAttribute(tail_kernel(d.slots), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                     TAIL_LDS_BUDGET));
                lds_allowed[dev][which].store(true, std::memory_order_release);
            }
        }
        NODAL_HIP_TRY(h, H->tail_image.reserve((size_t)d.image_bytes + 256));
        k_tail_pack<<<1, 1024, 0, h->stream>>>(d, H->tail_image.as<char>());
        NODAL_HIP_TRY(h, hipGetLastError());
        H->td = d;
        H->tail = t;
        break;
    }
    return NODAL_OK;
}

// Values of every level again for a level-0 matrix of the pattern the symbolic part was built for
// (SHierarchy::sym_valid): level-0 ELL values, then per level P (same aggregates), R (same entries),
// A P and the Galerkin sums into the known coarse patterns, the coarsest inverse, the tail's image.
// No aggregation, no sorting, no sizes to wait for: one read-back at the end for the value-dependent
// verdicts (graded links, a non-positive pivot).  *ok = false: those verdicts say the full setup has
// to look at this matrix (it will most likely decline it).
int sagg_refresh(nodal_ctx *h, SHierarchy *H, const int32_t *indptr0, const int32_t *indices0, const double *data0,
                 bool general, bool *ok) {
    *ok = false;
    hipStream_t st = h->stream;
    unsigned long long *dstats = H->stats.as<unsigned long long>();
    unsigned long long *hs = reinterpret_cast<unsigned long long *>(H->host_stats);
    NODAL_HIP_TRY(h, hipMemsetAsync(dstats, 0, (size_t)MAX_LEVELS * ST_COUNT * 8, st));
    SLevel *L0 = H->pool[0];
    const int64_t n0 = L0->n;
    {
        const unsigned gr = grid_for(n0);
        static const double spread = getenv("NODAL_SA_SPREAD") ? atof(getenv("NODAL_SA_SPREAD")) : 16.0;
        static const double share = getenv("NODAL_SA_SHARE") ? atof(getenv("NODAL_SA_SHARE")) : 0.9;
        row_stats<<<gr, TB, 0, st>>>(n0, indptr0, indices0, data0, share, spread, general, dstats, H->bstat.as<uint32_t>());
        reduce_bstat<<<1, 1024, 0, st>>>((int)gr, H->bstat.as<uint32_t>(), dstats, ST_MAXLEN, ST_GRADED);
    }
    csr_to_ell<<<grid_for(n0), TB, 0, st>>>(n0, L0->ld, indptr0, indices0, data0, L0->acol.as<int32_t>(),
                                           L0->aval.as<double>(), L0->avalf.as<float>(),
                                           L0->alen.as<int32_t>(), L0->dinv.as<double>(), L0->wfix,
                                           L0->d16 ? L0->adcol.as<int16_t>() : nullptr);
    NODAL_HIP_TRY(h, hipGetLastError());
    for (int l = 0; l + 1 < H->nlev; ++l) {
        SLevel *L = H->pool[l], *C = H->pool[l + 1];
        Ell A = L->A();
        A.width = L->maxlen;
        build_P<<<grid_for(L->n), TB, 0, st>>>(A, L->dinv.as<double>(), L->agg.as<int32_t>(), L->pcol.as<int32_t>(),
                                              L->pval.as<double>(), L->pvalf.as<float>(), omega_p(l));
        // (R's values on the hierarchy's other stream while this one computes A P: see build_level)
        const bool forked = H->aux != nullptr && nodal_extra_streams_ok(h) &&
                            !(getenv("NODAL_SA_FORK") && atoi(getenv("NODAL_SA_FORK")) == 0);
        hipStream_t rst = st;
        if (forked) {
            NODAL_HIP_TRY(h, hipEventRecord(H->ev_fork, st));
            NODAL_HIP_TRY(h, hipStreamWaitEvent(H->aux, H->ev_fork, 0));
            H->aux_pending = true;
            NODAL_TRY(galerkin_product<true>(h, H, l, nullptr, 1));
            rst = H->aux;
        }
        r_refresh<<<grid_for(C->n * 8), TB, 0, rst>>>(C->n, L->rld, L->rlen.as<int32_t>(), L->rcol.as<int32_t>(), L->ld,
                                                     L->pcol.as<int32_t>(), L->pval.as<double>(), L->rval.as<double>(),
                                                     L->rvalf.as<float>());
        NODAL_HIP_TRY(h, hipGetLastError());
        if (forked) NODAL_HIP_TRY(h, hipEventRecord(H->ev_join, H->aux));
        NODAL_TRY(galerkin_product<true>(h, H, l, forked ? H->ev_join : nullptr, forked ? 2 : 0));
        if (forked) H->aux_pending = false;
    }
    const int last = H->nlev - 1;
    if (H->dense_coarsest)
        coarsest_inverse<<<1, 256, 0, st>>>(H->pool[last]->A(), H->coarse_inv.as<double>(), dstats + (size_t)last * ST_COUNT);
    if (H->tail >= 0) k_tail_pack<<<1, 1024, 0, st>>>(H->td, H->tail_image.as<char>());
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipMemcpyAsync(hs, dstats, (size_t)MAX_LEVELS * ST_COUNT * 8, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    const int64_t bar = n0 / 100 < 32 ? (n0 / 100 > 0 ? n0 / 100 : 1) : 32;
    if (hs[ST_BADDIAG] || (int64_t)hs[ST_GRADED] >= bar) return NODAL_OK;
    for (int k = 1; k < H->nlev; ++k)
        if (hs[(size_t)k * ST_COUNT + ST_BADDIAG] || hs[(size_t)k * ST_COUNT + ST_OVERFLOW]) return NODAL_OK;
    *ok = true;
    return NODAL_OK;
}

}  // namespace

void sagg_destroy(nodal_ctx *h) {
    delete static_cast<SHierarchy *>(h->sagg);
    h->sagg = nullptr;
}

// Build the hierarchy for the context's CSR matrix.  *accepted = false: the network is not of
// the kind this path handles (graded links, hub rows, positive off-diagonals, a coarse row
// beyond the caps): nothing is kept, the caller uses amg.hip.  *floating: a connected
// component without a resistor to ground (structurally singular; the reference's spsolve
// returns NaNs).
// general: the matrix is the node block of a system with transconductance stamps (a few
// off-diagonals of either sign, not symmetric): the hierarchy serves as a preconditioner of
// FGMRES.  check_floating: the context's component table says which nodes a resistor joins to
// ground (grounded_flags); a connected component of the matrix pattern without such a node is
// reported in *floating.
static int sagg_setup_csr_body(nodal_ctx *h, int64_t n0, int64_t nnz0, const int32_t *indptr0, const int32_t *indices0,
                               const double *data0, bool general, bool check_floating, bool *accepted, int32_t *floating);

int sagg_setup_csr(nodal_ctx *h, int64_t n0, int64_t nnz0, const int32_t *indptr0, const int32_t *indices0,
                   const double *data0, bool general, bool check_floating, bool *accepted, int32_t *floating) {
    const int s = sagg_setup_csr_body(h, n0, nnz0, indptr0, indices0, data0, general, check_floating, accepted, floating);
    // Whatever way the setup ended -- accepted, declined at some level, a failed call -- nothing stays in flight on
    // the hierarchy's second stream behind the main stream's back (round 5: a declined hierarchy used to leave the
    // last level's "touches ground" kernels running there while the caller went on to its other paths).
    SHierarchy *H = static_cast<SHierarchy *>(h->sagg);
    if (H && H->aux && H->aux_pending && H->ev_flags) {
        if (hipEventRecord(H->ev_flags, H->aux) == hipSuccess) (void)hipStreamWaitEvent(h->stream, H->ev_flags, 0);
        H->aux_pending = false;
        (void)hipGetLastError();
    }
    return s;
}

static int sagg_setup_csr_body(nodal_ctx *h, int64_t n0, int64_t nnz0, const int32_t *indptr0, const int32_t *indices0,
                               const double *data0, bool general, bool check_floating, bool *accepted, int32_t *floating) {
    *accepted = false;
    *floating = 0;
    static const bool enabled = !(getenv("NODAL_SAGG") && atoi(getenv("NODAL_SAGG")) == 0);
    static const bool trace = getenv("NODAL_TRACE") != nullptr;
    if (!enabled) return NODAL_OK;
    SHierarchy *H = hierarchy_of(h);
    H->ready = false;
    hipStream_t st = h->stream;
    if (n0 >= (1ll << 30)) return NODAL_OK;
    static const bool reuse = !(getenv("NODAL_SA_REUSE") && atoi(getenv("NODAL_SA_REUSE")) == 0);
    if (reuse && H->sym_valid && H->sym_epoch == h->struct_epoch && H->sym_n == n0 && H->sym_nnz == nnz0 &&
        H->sym_general == general && H->sym_check == check_floating) {
        // same pattern as the matrix the hierarchy's symbolic part was built for: values only
        bool ok = false;
        NODAL_TRY(sagg_refresh(h, H, indptr0, indices0, data0, general, &ok));
        if (ok) {
            *floating = H->sym_floating;
            H->refreshed = true;
            H->ready = true;
            *accepted = true;
            if (trace) fprintf(stderr, "[sagg] symbolic setup kept (struct epoch %llu): values refreshed\n",
                               (unsigned long long)h->struct_epoch);
            return NODAL_OK;
        }
        if (trace) fprintf(stderr, "[sagg] the kept hierarchy does not take the new values: full setup\n");
    }
    H->sym_valid = false;
    H->refreshed = false;
    H->want_flags = check_floating;
    H->flags_levels = 0;
    H->last_iters = 0;
    H->drop_graph();
    NODAL_HIP_TRY(h, H->stats.reserve((size_t)MAX_LEVELS * ST_COUNT * 8 + 64));
    if (!H->host_stats)
        NODAL_HIP_TRY(h, hipHostMalloc(reinterpret_cast<void **>(&H->host_stats), (size_t)MAX_LEVELS * ST_COUNT * 8 + 64));
    unsigned long long *dstats = H->stats.as<unsigned long long>();
    unsigned long long *hs = reinterpret_cast<unsigned long long *>(H->host_stats);
    NODAL_HIP_TRY(h, hipMemsetAsync(dstats, 0, (size_t)MAX_LEVELS * ST_COUNT * 8, st));
    NODAL_HIP_TRY(h, H->bstat.reserve((size_t)2 * BSTAT_MAX * 4 + 64));
    {
        const unsigned gr = grid_for(n0);
        // graded = strongest link more than `spread` times the weakest, or one link above `share` of the
        // diagonal.  Measured on 300 x 300 grids: resistances over one decade / anisotropy 10 (ratios up to
        // 10) take this hierarchy at half the time of the contrast mode of amg.hip (5.4 vs 10.3 ms, 7.1 vs
        // 17.2 ms); from two decades on the contrast mode wins (three decades: 10.4 vs 16.8 ms, anisotropy
        // 1000: 17 vs 53 ms) -- hence 16, not amg.hip's own trigger of 8.
        static const double spread = getenv("NODAL_SA_SPREAD") ? atof(getenv("NODAL_SA_SPREAD")) : 16.0;
        static const double share = getenv("NODAL_SA_SHARE") ? atof(getenv("NODAL_SA_SHARE")) : 0.9;
        row_stats<<<gr, TB, 0, st>>>(n0, indptr0, indices0, data0, share, spread, general, dstats, H->bstat.as<uint32_t>());
        reduce_bstat<<<1, 1024, 0, st>>>((int)gr, H->bstat.as<uint32_t>(), dstats, ST_MAXLEN, ST_GRADED);
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_HIP_TRY(h, hipMemcpyAsync(hs, dstats, ST_COUNT * 8, hipMemcpyDeviceToHost, st));
    NODAL_WAIT_STREAM(h, st);
    const int64_t bar = n0 / 100 < 32 ? (n0 / 100 > 0 ? n0 / 100 : 1) : 32;  // amg.hip's contrast criterion
    if (hs[ST_MAXLEN] > (unsigned long long)W0_MAX || hs[ST_MAXLEN] == 0 || hs[ST_BADDIAG] ||
        (int64_t)hs[ST_GRADED] >= bar) {
        if (trace)
            fprintf(stderr, "[sagg] declined: longest row %llu, graded nodes %llu (bar %lld), M-matrix %s\n",
                    hs[ST_MAXLEN], hs[ST_GRADED], (long long)bar, hs[ST_BADDIAG] ? "no" : "yes");
        return NODAL_OK;
    }
    SLevel *L0 = H->level(0);
    L0->n = n0;
    L0->ld = pad64(n0);
    L0->maxlen = (int32_t)hs[ST_MAXLEN];
    L0->nnz = nnz0;
    L0->wfix = choose_wfix(L0->maxlen, n0, nnz0, 8);
    L0->width = L0->wfix ? L0->wfix : L0->maxlen;
    NODAL_HIP_TRY(h, L0->acol.reserve((size_t)L0->width * L0->ld * 4 + 64));
    NODAL_HIP_TRY(h, L0->aval.reserve((size_t)L0->width * L0->ld * 8 + 64));
    NODAL_HIP_TRY(h, L0->avalf.reserve((size_t)L0->width * L0->ld * 4 + 64));
    NODAL_HIP_TRY(h, L0->alen.reserve((size_t)L0->ld * 4 + 64));
    NODAL_HIP_TRY(h, L0->dinv.reserve((size_t)L0->ld * 8 + 64));
    // Columns as 16-bit offsets from the row (NODAL_SA_D16=0: off): the three level-0 row kernels of an iteration read 10
    // instead of 20 bytes of columns per row (W = 5); the same columns, hence the same sums, bit for bit.  Level 0 only
    // (the other levels are latency-bound), rows of fixed width or with their lengths (config 5's reduced system), and
    // only when every column is that close to its row.
    static const bool d16_env = !(getenv("NODAL_SA_D16") && atoi(getenv("NODAL_SA_D16")) == 0);
    L0->d16 = d16_env && hs[ST_FARCOL] == 0;
    if (L0->d16) NODAL_HIP_TRY(h, L0->adcol.reserve((size_t)L0->width * L0->ld * 2 + 64));
    csr_to_ell<<<grid_for(n0), TB, 0, st>>>(n0, L0->ld, indptr0, indices0, data0, L0->acol.as<int32_t>(),
                                           L0->aval.as<double>(), L0->avalf.as<float>(),
                                           L0->alen.as<int32_t>(), L0->dinv.as<double>(), L0->wfix,
                                           L0->d16 ? L0->adcol.as<int16_t>() : nullptr);
    NODAL_HIP_TRY(h, hipGetLastError());

    int l = 0;
    bool diag_last = false;
    for (;; ++l) {
        SLevel *L = H->level(l);
        if (L->n <= COARSEST && l > 0) break;
        if (l + 1 >= MAX_LEVELS) {
            if (trace) fprintf(stderr, "[sagg] declined: more than %d levels\n", MAX_LEVELS);
            return NODAL_OK;
        }
        bool declined = false, stop = false;
        NODAL_TRY(build_level(h, H, l, hs, &declined, &stop));
        if (declined) {
            if (trace)
                fprintf(stderr, "[sagg] declined at level %d (%lld rows): does not coarsen / over a cap / not positive "
                                "(this level: overflow %llx baddiag %llx; level above: overflow %llx unassigned %llu maxr %llu)\n",
                        l, (long long)L->n, hs[(size_t)l * ST_COUNT + ST_OVERFLOW], hs[(size_t)l * ST_COUNT + ST_BADDIAG],
                        l > 0 ? hs[(size_t)(l - 1) * ST_COUNT + ST_OVERFLOW] : 0ull,
                        l > 0 ? hs[(size_t)(l - 1) * ST_COUNT + ST_UNASSIGNED] : 0ull,
                        l > 0 ? hs[(size_t)(l - 1) * ST_COUNT + ST_MAXR] : 0ull);
            return NODAL_OK;
        }
        if (stop) {
            diag_last = true;
            break;
        }
    }
    H->nlev = l + 1;
    SLevel *last = H->level(l);
    if (H->nlev < 2) return NODAL_OK;  // too small for a hierarchy: the caller's other paths
    // statistics of the last level (its Galerkin kernel ran after the last round trip)
    NODAL_HIP_TRY(h, hipMemcpyAsync(hs, dstats, (size_t)MAX_LEVELS * ST_COUNT * 8, hipMemcpyDeviceToHost, st));
    if (!H->ev_copy) NODAL_HIP_TRY(h, hipEventCreateWithFlags(&H->ev_copy, hipEventDisableTiming));
    NODAL_HIP_TRY(h, hipEventRecord(H->ev_copy, st));
    if (check_floating) {
        // (round 4) the "touches ground" flags go up the aggregate maps while the host waits for that copy: they
        // need the aggregates of every level, which are all there, and nothing the copy brings
        // (the levels whose flags went up on the hierarchy's other stream behind their R are done: build_level)
        if (H->flags_levels == 0) {
            NODAL_HIP_TRY(h, H->level(0)->gflag.reserve((size_t)n0 + 64));
            NODAL_TRY(grounded_flags(h, H->level(0)->gflag.as<uint8_t>()));
        } else {
            NODAL_HIP_TRY(h, hipStreamWaitEvent(st, H->ev_flags, 0));
            H->aux_pending = false;
        }
        for (int k = H->flags_levels; k + 1 < H->nlev; ++k) {
            SLevel *L = H->pool[k], *C = H->pool[k + 1];
            NODAL_HIP_TRY(h, C->gflag.reserve((size_t)C->n + 64));
            NODAL_HIP_TRY(h, hipMemsetAsync(C->gflag.p, 0, ((size_t)C->n + 63) & ~(size_t)63, st));  // (one fill kernel)
            flags_up<<<grid_for(L->n), TB, 0, st>>>(L->n, L->agg.as<int32_t>(), L->gflag.as<uint8_t>(),
                                                   C->gflag.as<uint8_t>());
        }
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    NODAL_WAIT_EVENT(h, H->ev_copy, st);
    for (int k = 0; k < H->nlev; ++k) {
        const unsigned long long *s = hs + (size_t)k * ST_COUNT;
        if (k > 0) {
            H->pool[k]->maxlen = (int32_t)s[ST_MAXLEN];
            H->pool[k]->nnz = (int64_t)s[ST_NNZ];
            H->pool[k]->wfix = choose_wfix(H->pool[k]->maxlen, H->pool[k]->n, H->pool[k]->nnz, APAD);
        }
        if (s[ST_OVERFLOW] || (k > 0 && s[ST_BADDIAG]) || s[ST_UNASSIGNED]) {
            if (trace) fprintf(stderr, "[sagg] declined: level %d over a cap / not positive\n", k);
            return NODAL_OK;
        }
    }
    diag_last = last->maxlen <= 1;
    H->dense_coarsest = !diag_last;
    if (H->dense_coarsest) {
        if (last->n > COARSEST) return NODAL_OK;
        NODAL_HIP_TRY(h, H->coarse_inv.reserve((size_t)last->n * last->n * 8 + 64));
        coarsest_inverse<<<1, 256, 0, st>>>(last->A(), H->coarse_inv.as<double>(),
                                            dstats + (size_t)l * ST_COUNT);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    for (int k = 0; k < H->nlev; ++k) {
        SLevel *L = H->pool[k];
        NODAL_HIP_TRY(h, L->vec.reserve((size_t)V_COUNT * L->ld * 8 + 256));
        NODAL_HIP_TRY(h, L->part.reserve(5 * DOT_BLOCKS * 8 + 512));  // (+ s1, s2, the frozen triple, a ring of samples)
    }
    NODAL_TRY(build_tail(h, H, hs));

    // structural singularity: OR the "touches ground" flags up, look at the last level
    if (!check_floating) {
        NODAL_HIP_TRY(h, hipMemcpyAsync(hs, dstats, (size_t)MAX_LEVELS * ST_COUNT * 8, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        if (hs[(size_t)l * ST_COUNT + ST_BADDIAG] & 2ull) return NODAL_OK;  // coarsest pivot not positive
    } else {
        // (the flags went up the hierarchy in front of the statistics read-back above)
        // (the verdict's word: zero since this setup cleared the statistics block, nobody else writes it)
        uint32_t *fl = reinterpret_cast<uint32_t *>(dstats + (size_t)(MAX_LEVELS - 1) * ST_COUNT + ST_COUNT - 1);
        if (H->dense_coarsest) last_level_floating<<<1, 64, 0, st>>>(last->A(), last->gflag.as<uint8_t>(), fl);
        else any_unflagged<<<grid_for(last->n), TB, 0, st>>>(last->n, last->gflag.as<uint8_t>(), fl);
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_HIP_TRY(h, hipMemcpyAsync(hs, dstats, (size_t)MAX_LEVELS * ST_COUNT * 8, hipMemcpyDeviceToHost, st));
        NODAL_WAIT_STREAM(h, st);
        if (hs[(size_t)l * ST_COUNT + ST_BADDIAG] & 2ull) {  // coarsest pivot not positive
            if (trace) fprintf(stderr, "[sagg] declined: coarsest matrix not positive definite\n");
            return NODAL_OK;
        }
        *floating = hs[(size_t)(MAX_LEVELS - 1) * ST_COUNT + ST_COUNT - 1] != 0 ? 1 : 0;
    }
    // (The cycle is a preconditioner: its sweeps read f32 copies of A, P, R -- a third less traffic, the
    // iteration count does not move -- which csr_to_ell, build_P, r_to_ell and the Galerkin kernel wrote
    // beside the f64 values; so are its vectors (cyc_t, above); the outer iteration's vectors, the outer SpMV and
    // the residual recurrence stay f64.)
    if (trace) {
        fprintf(stderr, "[sagg] levels (rows/entries/longest row/padded width):");
        for (int k = 0; k < H->nlev; ++k)
            fprintf(stderr, " %lld/%lld/%d/%d", (long long)H->pool[k]->n, (long long)H->pool[k]->nnz,
                    H->pool[k]->maxlen, H->pool[k]->wfix);
        fprintf(stderr, "  tail %d (%d register slots)  coarsest %s\n", H->tail, H->tail >= 0 ? H->td.slots : 0,
                H->dense_coarsest ? "dense" : "diagonal");
    }
    H->ready = true;
    *accepted = true;
    H->sym_valid = true;
    H->sym_epoch = h->struct_epoch;
    H->sym_n = n0;
    H->sym_nnz = nnz0;
    H->sym_general = general;
    H->sym_check = check_floating;
    H->sym_floating = *floating;
    memcpy(H->sym_stats, hs, sizeof H->sym_stats);
    return NODAL_OK;
}

int sagg_setup(nodal_ctx *h, bool *accepted, int32_t *floating) {
    return sagg_setup_csr(h, h->n, h->nnz, h->indptr.as<int32_t>(), h->indices.as<int32_t>(), h->data.as<double>(),
                          false, true, accepted, floating);
}

namespace {

// out ~= A_l^-1 b.  At level 0 the post-smoothing kernel also leaves the partial dot products
// out.b and out.Ap of the outer iteration (sb != nullptr).
// the levels that end the recursion: the tail (one workgroup, sagg_cycle.h) or the coarsest level; f64 in and out
int last_level(nodal_ctx *h, SHierarchy *H, int l, const double *b, double *out) {
    hipStream_t st = h->stream;
    SLevel *L = H->pool[l];
    const int64_t n = L->n;
    if (l == H->tail) {
        if (H->td.slots <= 8) k_tail<8><<<1, 1024, (size_t)H->td.lds_bytes, st>>>(H->td, H->tail_image.as<char>(), b, out, 1);
        else if (H->td.slots <= 16) k_tail<16><<<1, 1024, (size_t)H->td.lds_bytes, st>>>(H->td, H->tail_image.as<char>(), b, out, 1);
        else k_tail<32><<<1, 1024, (size_t)H->td.lds_bytes, st>>>(H->td, H->tail_image.as<char>(), b, out, 1);
    } else {
        k_coarsest<<<grid_for(n), TB, 0, st>>>(n, H->dense_coarsest ? H->coarse_inv.as<double>() : nullptr,
                                               L->dinv.as<double>(), b, out);
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// K-cycle coefficients of outer iteration `it` of a flexible Krylov method (it == 0 starts a solve): adaptive in the
// first four iterations and in every fourth one after, frozen -- the mean of the last three samples; the first two
// iterations' are not taken: they run high -- in between.  The schedule depends on `it` alone: the same launches every
// time.  it < 0: an adaptive cycle that leaves no sample (every other caller).
void kcycle_schedule(SHierarchy *H, int it) {
    static const bool on = !(getenv("NODAL_SA_KFREEZE") && atoi(getenv("NODAL_SA_KFREEZE")) == 0);
    H->kfrozen = false;
    H->kslot = -1;
    if (!on || it < 0) return;
    if (it == 0) H->ksamples = 0;
    const bool adaptive = it < 4 || (it & 3) == 0;
    if (!adaptive && H->ksamples > 0) {
        H->kfrozen = true;
    } else if (it >= 2) {
        H->kslot = H->ksamples % 3;
        ++H->ksamples;
        H->kcount = H->ksamples < 3 ? H->ksamples : 3;
    }
}

// One cycle of level l (never a last level: the hierarchy has two levels at least and the tail starts at level 1
// or below).  TBV: type of the right-hand side (f64 from the outer iteration at level 0, cyc_t inside); TOUT: of the
// result (cyc_t; f64 when the general path's FGMRES takes it as a Krylov vector).
template <typename TBV, typename TOUT>
int cycle(nodal_ctx *h, SHierarchy *H, int l, const TBV *b, const cyc_t *x0, TOUT *out, const SolveBufs *sb) {
    hipStream_t st = h->stream;
    SLevel *L = H->pool[l];
    const int64_t n = L->n;
    SLevel *C = H->pool[l + 1];
    const int64_t nc = C->n;
    const Ell A = L->A();
    const double *dinv = L->dinv.as<double>();
    const cyc_t *x = x0;  // the pre-smoothed iterate w D^-1 b, from the producer of b
    cyc_t *r = L->v<cyc_t>(V_R), *xp = L->v<cyc_t>(V_XP);
    const unsigned g = sb ? (unsigned)sb->g0 : grid_for(n);
    const unsigned tb = L->wfix ? TB : TB * LPR_RAGGED;  // (ragged rows: LPR_RAGGED lanes each, same rows per workgroup)
    const int nu = H->nu[l < 2 ? l : 2];
    if (nu >= 2) {  // second pre-smoothing sweep: x1 = x0 + w D^-1 (b - A x0)
        cyc_t *x1 = L->v<cyc_t>(V_X1);
        SAGG_DISPATCH_W(L->wfix, (k_post<W, false, TBV, cyc_t><<<g, tb, 0, st>>>(A, dinv, b, x0, x1, nullptr, nullptr, nullptr)));
        x = x1;
    }
    if (nu >= 3) {  // third (NODAL_SA_NU experiments): into the slot the post-smoothing uses once x is no longer read
        cyc_t *x2 = L->v<cyc_t>(V_T);
        SAGG_DISPATCH_W(L->wfix, (k_post<W, false, TBV, cyc_t><<<g, tb, 0, st>>>(A, dinv, b, x, x2, nullptr, nullptr, nullptr)));
        x = x2;
    }
    SAGG_DISPATCH_W(L->wfix, (k_smooth_residual<W, TBV><<<g, tb, 0, st>>>(A, b, x, r)));
    const bool last = l + 1 == H->tail || l + 1 == H->nlev - 1;
    const double *coef = nullptr;
    if (last) {
        double *rc = C->v<double>(V_RC), *c1 = C->v<double>(V_C1);
        k_restrict<double><<<grid_for(nc * RL), TB, 0, st>>>(nc, L->rld, L->rcol.as<int32_t>(), L->rvalf.as<float>(),
                                                            L->rlen.as<int32_t>(), r, rc, C->dinv.as<double>(), nullptr);
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_TRY(last_level(h, H, l + 1, rc, c1));
        k_prolong<double><<<grid_for(n), TB, 0, st>>>(n, L->ld, L->pcol.as<int32_t>(), L->pvalf.as<float>(), x, c1,
                                                     (const double *)nullptr, nullptr, xp);
    } else {
        cyc_t *rc = C->v<cyc_t>(V_RC), *c1 = C->v<cyc_t>(V_C1), *c2 = C->v<cyc_t>(V_C2), *x0c = C->v<cyc_t>(V_X);
        k_restrict<cyc_t><<<grid_for(nc * RL), TB, 0, st>>>(nc, L->rld, L->rcol.as<int32_t>(), L->rvalf.as<float>(),
                                                           L->rlen.as<int32_t>(), r, rc, C->dinv.as<double>(), x0c);
        NODAL_HIP_TRY(h, hipGetLastError());
        int nparts = 0;
        // K-cycle (two flexible-CG steps on the coarse problem) at the first coarse level only, when
        // that level is large enough to be outside the tail; plain V hand-over everywhere else
        const bool kcycle = l < H->klevels && H->kcycle;
        if (kcycle) {
            cyc_t *v1 = C->v<cyc_t>(V_V1), *v2 = C->v<cyc_t>(V_V2), *r2 = C->v<cyc_t>(V_R2);
            double *part = C->part.as<double>();
            const Ell Ac = C->A();
            const unsigned gd = grid_for(nc, DOT_BLOCKS);  // (grid-stride beyond DOT_BLOCKS x TB rows)
            const unsigned tbc = C->wfix ? TB : TB * LPR_RAGGED;
            nparts = (int)gd;
            double *cf = part + 5 * DOT_BLOCKS;  // s1, s2 | frozen s1, s2, t | ring of samples (k_kcoef)
            if (H->kfrozen && l == 0) {
                // between two calibrations: the coefficients of the last adaptive cycles, no dot products -- the first
                // SpMV and the second residual in one launch, no second SpMV, no k_kcoef
                NODAL_TRY((cycle<cyc_t, cyc_t>(h, H, l + 1, rc, x0c, c1, nullptr)));
                SAGG_DISPATCH_W(C->wfix, (k_spmv_resid<W><<<gd, tbc, 0, st>>>(Ac, c1, rc, cf + 3, r2, C->dinv.as<double>(), x0c)));
                NODAL_HIP_TRY(h, hipGetLastError());
                NODAL_TRY((cycle<cyc_t, cyc_t>(h, H, l + 1, r2, x0c, c2, nullptr)));
                coef = cf + 3;
                nparts = 0;
            } else {
                NODAL_TRY((cycle<cyc_t, cyc_t>(h, H, l + 1, rc, x0c, c1, nullptr)));
                SAGG_DISPATCH_W(C->wfix, (k_spmv_dots<W><<<gd, tbc, 0, st>>>(Ac, c1, v1, rc, nullptr, part + 0 * DOT_BLOCKS,
                                                                            part + 1 * DOT_BLOCKS, nullptr)));
                k_second_residual<<<grid_for(nc), TB, 0, st>>>(nc, rc, v1, part, nparts, r2, C->dinv.as<double>(), x0c);
                NODAL_HIP_TRY(h, hipGetLastError());
                NODAL_TRY((cycle<cyc_t, cyc_t>(h, H, l + 1, r2, x0c, c2, nullptr)));
                SAGG_DISPATCH_W(C->wfix, (k_spmv_dots<W><<<gd, tbc, 0, st>>>(Ac, c2, v2, v1, r2, part + 3 * DOT_BLOCKS,
                                                                            part + 2 * DOT_BLOCKS, part + 4 * DOT_BLOCKS)));
                NODAL_HIP_TRY(h, hipGetLastError());
            }
        } else {
            NODAL_TRY((cycle<cyc_t, cyc_t>(h, H, l + 1, rc, x0c, c1, nullptr)));
        }
        if (nparts) {  // s1, s2 once, instead of five reductions in every workgroup of the prolongation
            double *cf = C->part.as<double>() + 5 * DOT_BLOCKS;
            const bool sample = l == 0 && H->kslot >= 0;
            k_kcoef<<<1, 320, 0, st>>>(C->part.as<double>(), nparts, cf, sample ? H->kslot : -1, sample ? H->kcount : 0);
            coef = cf;
        }
        k_prolong<cyc_t><<<grid_for(n), TB, 0, st>>>(n, L->ld, L->pcol.as<int32_t>(), L->pvalf.as<float>(), x, c1, c2, coef,
                                                    xp);
    }
    const cyc_t *cur = xp;
    if (nu >= 2) {  // first of two post-smoothing sweeps
        cyc_t *mid = L->v<cyc_t>(V_T);
        SAGG_DISPATCH_W(L->wfix, (k_post<W, false, TBV, cyc_t><<<g, tb, 0, st>>>(A, dinv, b, cur, mid, nullptr, nullptr, nullptr)));
        cur = mid;
    }
    if (nu >= 3) {  // second of three
        cyc_t *mid2 = L->v<cyc_t>(V_X1);
        SAGG_DISPATCH_W(L->wfix, (k_post<W, false, TBV, cyc_t><<<g, tb, 0, st>>>(A, dinv, b, cur, mid2, nullptr, nullptr, nullptr)));
        cur = mid2;
    }
    if (sb) {
        SAGG_DISPATCH_W(L->wfix, (k_post<W, true, TBV, TOUT><<<g, tb, 0, st>>>(A, dinv, b, cur, out, sb->Ap, sb->part_rz,
                                                                              sb->part_zap)));
    } else {
        SAGG_DISPATCH_W(L->wfix, (k_post<W, false, TBV, TOUT><<<g, tb, 0, st>>>(A, dinv, b, cur, out, nullptr, nullptr, nullptr)));
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

__global__ void f_set_scalars(double *__restrict__ sc, double tol2) {
    for (int k = 0; k < F_COUNT; ++k) sc[k] = 0.0;
    sc[F_TOL2] = tol2;
}

size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

// z ~= A^-1 r with one cycle of the hierarchy built by sagg_setup_csr (the preconditioner of the
// general path's FGMRES, sparse_general.hip)
__global__ __launch_bounds__(TB) void k_x0(int64_t n, const double *__restrict__ dinv, const double *__restrict__ r,
                                           cyc_t *__restrict__ x0) {
    for (int64_t i = (int64_t)xcd_block() * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        x0[i] = (cyc_t)(OMEGA * dinv[i] * r[i]);
}
bool sagg_ready(nodal_ctx *h, int64_t n) {
    SHierarchy *H = static_cast<SHierarchy *>(h->sagg);
    return H && H->ready && H->pool[0]->n == n;
}
void sagg_invalidate(nodal_ctx *h) {
    if (h->sagg) static_cast<SHierarchy *>(h->sagg)->ready = false;
}
int sagg_levels(nodal_ctx *h) { return h->sagg ? static_cast<SHierarchy *>(h->sagg)->nlev : 0; }
bool sagg_x0_slot(nodal_ctx *h, const double **dinv, nodal_cyc_t **x0, int64_t *n0, double *omega) {
    SHierarchy *H = static_cast<SHierarchy *>(h->sagg);
    if (!H || !H->ready) return false;
    SLevel *L0 = H->pool[0];
    *dinv = L0->dinv.as<double>();
    *x0 = L0->v<cyc_t>(V_X);
    *n0 = L0->n;
    *omega = OMEGA;
    return true;
}
// it >= 0: the cycle is the preconditioner of iteration `it` of a FLEXIBLE method (FGMRES: sparse_general.hip), which
// may calibrate and freeze the K-cycle's coefficients like sagg_fcg_solve does (kcycle_schedule)
int sagg_apply(nodal_ctx *h, const double *r, double *z, bool x0_ready, int it) {
    SHierarchy *H = static_cast<SHierarchy *>(h->sagg);
    if (!H || !H->ready) return nodal_fail(h, NODAL_E_INVALID, "sagg_setup_csr not called");
    SLevel *L0 = H->pool[0];
    cyc_t *x0 = L0->v<cyc_t>(V_X);
    if (!x0_ready) k_x0<<<grid_for(L0->n), TB, 0, h->stream>>>(L0->n, L0->dinv.as<double>(), r, x0);
    NODAL_HIP_TRY(h, hipGetLastError());
    kcycle_schedule(H, it);
    const int rc = cycle<double, double>(h, H, 0, r, x0, z, nullptr);
    H->kfrozen = false;
    H->kslot = -1;
    return rc;
}

// y = A x with the level-0 ELL copy of the matrix the hierarchy was built on
// (start / stop: optional events bound to the dispatch itself, as in sagg_fcg_solve)
int sagg_spmv(nodal_ctx *h, const double *x, double *y, hipEvent_t start, hipEvent_t stop) {
    SHierarchy *H = static_cast<SHierarchy *>(h->sagg);
    if (!H || !H->ready) return nodal_fail(h, NODAL_E_INVALID, "sagg_setup_csr not called");
    const SLevel *L0 = H->pool[0];
    const Ell A0 = L0->A();
    const unsigned tb0 = L0->wfix ? TB : TB * LPR_RAGGED;
    if (start && stop) {
        SAGG_DISPATCH_W(L0->wfix, (hipExtLaunchKernelGGL((k_ell_spmv<W>), dim3(grid_for(L0->n)), dim3(tb0), 0, h->stream,
                                                         start, stop, 0, A0, x, y)));
    } else {
        SAGG_DISPATCH_W(L0->wfix, (k_ell_spmv<W><<<grid_for(L0->n), tb0, 0, h->stream>>>(A0, x, y)));
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}

// Flexible CG preconditioned by the hierarchy.  Return: NODAL_OK (converged, *info = 0); -2 the
// network is structurally singular (caller fills NaNs); -1 breakdown / no convergence (caller
// falls back); SAGG_DECLINED (-3) the hierarchy does not take this matrix; > 0 a status.
int sagg_fcg_solve(nodal_ctx *h, const double *b, bool do_setup, int32_t *info, int32_t *iters, double *resid) {
    static const bool trace = getenv("NODAL_TRACE") != nullptr;
    const int64_t n = h->n;
    hipStream_t st = h->stream;
    SHierarchy *H = hierarchy_of(h);
    if (do_setup) {
        bool accepted = false;
        int32_t floating = 0;
        NODAL_TRY(sagg_setup(h, &accepted, &floating));
        if (!accepted) return SAGG_DECLINED;
        if (floating) {
            *info = 1;
            *iters = 0;
            *resid = 0.0;
            return -2;
        }
    }
    if (!H->ready || H->pool[0]->n != n) return SAGG_DECLINED;
    h->amg_levels = H->nlev;
    static const int kc = getenv("NODAL_SA_KCYCLE") ? atoi(getenv("NODAL_SA_KCYCLE")) : 1;
    H->kcycle = kc != 0;
    H->klevels = getenv("NODAL_SA_KLEVELS") ? atoi(getenv("NODAL_SA_KLEVELS")) : 1;
    if (const char *e = getenv("NODAL_SA_NU")) {  // e.g. "212": sweeps at level 0, level 1, deeper levels
        for (int k = 0; k < 3 && e[k] >= '1' && e[k] <= '3'; ++k) H->nu[k] = e[k] - '0';
    }

    const size_t vec = align_up((size_t)n * 8);
    NODAL_HIP_TRY(h, h->solver.reserve(6 * vec + 4 * MAX_PARTIALS * 8 + F_COUNT * 8 + 256));
    char *base = h->solver.as<char>();
    SolveBufs sb;
    sb.r = reinterpret_cast<double *>(base);
    sb.z = reinterpret_cast<cyc_t *>(base + vec);
    sb.p = reinterpret_cast<double *>(base + 2 * vec);
    sb.Ap = reinterpret_cast<double *>(base + 3 * vec);
    sb.x0 = reinterpret_cast<cyc_t *>(base + 4 * vec);
    sb.p2 = reinterpret_cast<double *>(base + 5 * vec);  // the direction of the odd iterations (fused f_dir_spmv)
    sb.part_rz = reinterpret_cast<double *>(base + 6 * vec);
    sb.part_zap = sb.part_rz + MAX_PARTIALS;
    sb.part_rr = sb.part_zap + MAX_PARTIALS;
    sb.part_pap = sb.part_rr + MAX_PARTIALS;
    sb.sc = sb.part_pap + MAX_PARTIALS;
    sb.g0 = (int)grid_for(n, MAX_PARTIALS);
    double *x = h->x.as<double>();
    const Ell A0 = H->pool[0]->A();

    const double tol = 1e-13;
    f_set_scalars<<<1, 1, 0, st>>>(sb.sc, tol * tol);
    const double *dinv0 = H->pool[0]->dinv.as<double>();
    f_init<<<sb.g0, TB, 0, st>>>(b, x, sb.r, sb.Ap, dinv0, sb.x0, sb.part_rr, n);
    NODAL_HIP_TRY(h, hipGetLastError());

    const int64_t maxit = getenv("NODAL_FCG_MAXIT") ? atoll(getenv("NODAL_FCG_MAXIT")) : 2000;
    hipEvent_t e0 = h->ev[2], e1 = h->ev[3];
    h->kern_ms = 0;
    h->kern_launches = 0;
    double hs[F_COUNT];
    int64_t enqueued = 0;
    int batch = 6;   // iterations enqueued before the first look at the residual
    // Same hierarchy as the last solve (values refreshed on the same pattern, or another right-hand side):
    // the iteration count will be about the same, so the first look comes when that many have run --
    // one or two polls instead of six or seven (each drains the queue: ~40 us).
    if (H->last_iters > 6 && (H->refreshed || !do_setup)) batch = H->last_iters + 1 > 32 ? 32 : H->last_iters + 1;
    else if (H->hint_iters > 8 && H->hint_n == n && H->hint_nnz == H->pool[0]->nnz) {
        batch = (3 * H->hint_iters) / 4;
        if (batch > 32) batch = 32;
    }
    int status = 0;  // 0 running, 1 converged, 2 breakdown, 3 maxit
    const bool poll_partials = !(getenv("NODAL_FCG_HOST_SUM") && atoi(getenv("NODAL_FCG_HOST_SUM")) == 0);
    double rr_prev = -1.0;
    int64_t it_prev = 0;
    int polls = 0;
    static const bool fuse_dir_env = !(getenv("NODAL_SA_FUSE_DIR") && atoi(getenv("NODAL_SA_FUSE_DIR")) == 0);
    const bool fuse_dir = fuse_dir_env && LPR_RAGGED == 1;
    // one outer iteration (the kernels take its parity only: see f_direction)
    // K-cycle coefficients calibrated, then frozen (kcycle_schedule; not under a replayed graph, whose launches are fixed)
    static const bool graphs_env = getenv("NODAL_SA_GRAPH") && atoi(getenv("NODAL_SA_GRAPH")) != 0;
    auto iteration = [&](int it, bool timed) -> int {
        kcycle_schedule(H, graphs_env ? -1 : it);
        struct KReset {  // (whatever way this iteration is left: the hierarchy's other users run adaptive cycles)
            SHierarchy *H;
            ~KReset() { H->kfrozen = false; H->kslot = -1; }
        } kreset{H};
        NODAL_TRY((cycle<double, cyc_t>(h, H, 0, sb.r, sb.x0, sb.z, &sb)));
        if (fuse_dir) {  // direction and SpMV in one launch; the direction buffers swap with the parity
            double *p_new = (it & 1) ? sb.p2 : sb.p;
            const double *p_old = (it & 1) ? sb.p : sb.p2;
            const unsigned tbf = TB;
            if (timed) {
                SAGG_DISPATCH_W(H->pool[0]->wfix, (hipExtLaunchKernelGGL((f_dir_spmv<W>), dim3(sb.g0), dim3(tbf), 0, st, e0, e1, 0,
                                                                         A0, (const cyc_t *)sb.z, p_old, p_new, sb.Ap,
                                                                         (const double *)sb.part_rz, (const double *)sb.part_zap,
                                                                         (const double *)sb.part_rr, sb.g0, sb.part_pap, sb.sc,
                                                                         it & 1)));
            } else {
                SAGG_DISPATCH_W(H->pool[0]->wfix, (f_dir_spmv<W><<<sb.g0, tbf, 0, st>>>(A0, sb.z, p_old, p_new, sb.Ap, sb.part_rz,
                                                                                       sb.part_zap, sb.part_rr, sb.g0,
                                                                                       sb.part_pap, sb.sc, it & 1)));
            }
            f_update<<<sb.g0, TB, 0, st>>>(x, sb.r, p_new, sb.Ap, sb.part_pap, sb.g0, dinv0, sb.x0, sb.part_rr, sb.sc, it & 1, n);
            NODAL_HIP_TRY(h, hipGetLastError());
            return NODAL_OK;
        }
        f_direction<<<sb.g0, TB, 0, st>>>(sb.z, sb.p, sb.part_rz, sb.part_zap, sb.part_rr, sb.g0, sb.sc, it & 1, n);
        // one launch per poll batch is timed: start / stop events tied to the dispatch itself
        // (hipExtLaunchKernelGGL), i.e. the kernel's own duration as rocprofv3 reports it -- a pair
        // of hipEventRecord calls around a 14-us kernel also measures ~3.5 us of launch
        const unsigned tb0 = H->pool[0]->wfix ? TB : TB * LPR_RAGGED;
        if (timed) {
            SAGG_DISPATCH_W(H->pool[0]->wfix, (hipExtLaunchKernelGGL((f_spmv<W>), dim3(sb.g0), dim3(tb0), 0, st, e0, e1, 0,
                                                                     A0, (const double *)sb.p, sb.Ap, sb.part_pap,
                                                                     (const double *)sb.sc, it & 1)));
        } else {
            SAGG_DISPATCH_W(H->pool[0]->wfix, (f_spmv<W><<<sb.g0, tb0, 0, st>>>(A0, sb.p, sb.Ap, sb.part_pap, sb.sc, it & 1)));
        }
        f_update<<<sb.g0, TB, 0, st>>>(x, sb.r, sb.p, sb.Ap, sb.part_pap, sb.g0, dinv0, sb.x0, sb.part_rr, sb.sc, it & 1, n);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    };
    // kept hierarchy (values refreshed on the same pattern, another right-hand side): replay a captured
    // pair of iterations.  The key covers every pointer and size the captured launches carry.
    // (NODAL_SA_GRAPH=1; off by default: measured without gain -- 7.53 instead of 7.28 ms for one stream,
    // and the same 180 / 211 / 182 / 210 circuits/s with 2 / 3 / 4 / 6 solves in flight: what bounds several
    // streams is not the host's launch rate)
    static const bool graphs = getenv("NODAL_SA_GRAPH") && atoi(getenv("NODAL_SA_GRAPH")) != 0;
    if (graphs && (H->refreshed || !do_setup)) {
        uint64_t key = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { key = (key ^ v) * 1099511628211ull; };
        mix((uint64_t)(uintptr_t)base); mix((uint64_t)(uintptr_t)x); mix((uint64_t)n); mix((uint64_t)sb.g0);
        mix((uint64_t)H->nlev); mix((uint64_t)(H->tail + 1)); mix((uint64_t)H->kcycle); mix((uint64_t)H->klevels);
        mix((uint64_t)(H->nu[0] * 100 + H->nu[1] * 10 + H->nu[2])); mix((uint64_t)H->td.lds_bytes);
        mix((uint64_t)(uintptr_t)H->tail_image.p); mix((uint64_t)(uintptr_t)H->coarse_inv.p);
        for (int k = 0; k < H->nlev; ++k) {
            const SLevel *L = H->pool[k];
            const void *ps[] = {L->acol.p, L->aval.p, L->avalf.p, L->alen.p, L->dinv.p, L->pcol.p, L->pvalf.p,
                                L->rcol.p, L->rvalf.p, L->rlen.p, L->vec.p, L->part.p};
            for (const void *q : ps) mix((uint64_t)(uintptr_t)q);
            mix((uint64_t)L->n); mix((uint64_t)L->ld); mix((uint64_t)L->wfix); mix((uint64_t)L->rld);
        }
        if (H->it_exec && H->it_key != key) H->drop_graph();
        if (!H->it_exec) {
            hipGraph_t g = nullptr;
            hipGraphExec_t ex = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                int cs = iteration(0, false);
                if (cs == NODAL_OK) cs = iteration(1, false);
                const hipError_t ce = hipStreamEndCapture(st, &g);
                if (cs == NODAL_OK && ce == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess) {
                    H->it_graph = g;
                    H->it_exec = ex;
                    H->it_key = key;
                } else {
                    if (g) (void)hipGraphDestroy(g);
                    (void)hipGetLastError();
                }
            } else {
                (void)hipGetLastError();
            }
        }
    } else if (H->it_exec && do_setup) {
        H->drop_graph();
    }
    while (status == 0) {
        for (int c = 0; c < batch;) {
            if (H->it_exec && c > 0 && c + 1 < batch && (enqueued & 1) == 0) {
                NODAL_HIP_TRY(h, hipGraphLaunch(H->it_exec, st));
                c += 2;
                enqueued += 2;
            } else {
                NODAL_TRY(iteration((int)enqueued, c == 0));
                ++c;
                ++enqueued;
            }
        }
        // The look brings the scalars AND the partial sums of |r|^2 that the last f_update left (they lie in front of
        // the scalars: one copy).  The device learns that iteration k has converged in f_direction of iteration k + 1,
        // behind that iteration's multigrid cycle -- 0.3 ms for nothing at the end of every solve; the host adds the
        // partials up itself (a fixed order) and stops the moment the residual is there.
        double rr_now = -1.0;
        {
            const size_t bytes = (size_t)(2 * MAX_PARTIALS + F_COUNT) * 8;  // part_rr | part_pap | sc
            double *stage = poll_partials ? static_cast<double *>(nodal_pinned_arena(h, bytes)) : nullptr;
            if (stage) {
                NODAL_HIP_TRY(h, hipMemcpyAsync(stage, sb.part_rr, bytes, hipMemcpyDeviceToHost, st));
                NODAL_WAIT_STREAM(h, st);
                memcpy(hs, stage + 2 * MAX_PARTIALS, F_COUNT * 8);
                double acc = 0.0;
                for (int k = 0; k < sb.g0; ++k) acc += stage[k];
                rr_now = acc;
            } else {
                NODAL_TRY(nodal_read_words(h, hs, sb.sc, F_COUNT * 8));
            }
        }
        ++polls;
        bool conv = hs[F_CONV + ((enqueued - 1) & 1)] != 0.0;
        bool host_conv = false;
        if (!conv && rr_now >= 0.0 && hs[F_FLAG] == 0.0 && hs[F_BB] > 0.0 && rr_now <= tol * tol * hs[F_BB]) {
            // (the residual after the iteration enqueued last; the device's own flag would follow one cycle later)
            host_conv = true;
            hs[F_RR] = rr_now;
            hs[F_ITERS] = (double)enqueued;
        } else if (!conv && rr_now >= 0.0) {
            hs[F_RR] = rr_now;  // (one iteration fresher than the device's word: the rate below is the better for it)
        }
        float ms = 0;
        // (the timed launch did its work unless the iteration had converged before it)
        if (!(conv && !host_conv && hs[F_ITERS] <= (double)(enqueued - batch)) && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
            h->kern_ms += ms;
            h->kern_launches += 1;
        }
        if (hs[F_FLAG] != 0.0 || !(hs[F_RR] == hs[F_RR])) status = 2;
        else if (conv || host_conv) status = 1;
        else if (enqueued >= maxit) status = 3;
        else {
            // remaining iterations from the observed reduction per iteration; three quarters of the
            // estimate are enqueued (the rate of the first iterations is not the asymptotic one): an
            // iteration past convergence still runs its multigrid cycle, ~250 us, a poll costs ~30 us
            int next = 2;
            const double target = tol * tol * hs[F_BB];
            const double rr_ref = rr_prev > 0.0 ? rr_prev : hs[F_BB];
            const int64_t it_ref = rr_prev > 0.0 ? it_prev : 0;
            if (hs[F_RR] > 0.0 && hs[F_RR] < rr_ref && enqueued > it_ref) {
                const double rate = log(hs[F_RR] / rr_ref) / (double)(enqueued - it_ref);  // < 0
                static const double look = getenv("NODAL_FCG_LOOK") ? atof(getenv("NODAL_FCG_LOOK")) : 0.75;
                next = (int)floor(look * log(target / hs[F_RR]) / rate);
            }
            if (next > enqueued) next = (int)enqueued;  // (at most doubling: early rates are pessimistic)
            batch = next < 1 ? 1 : (next > 32 ? 32 : next);
            rr_prev = hs[F_RR];
            it_prev = enqueued;
        }
    }
    const int64_t its = status == 1 ? (int64_t)hs[F_ITERS] : enqueued;
    *iters = (int32_t)its;
    *resid = hs[F_BB] > 0 ? sqrt(hs[F_RR] / hs[F_BB]) : 0.0;
    const SLevel *L0 = H->pool[0];
    h->kern_alg = 12.0 * (double)L0->nnz + 4.0 * (double)n + 16.0 * (double)n;
    if (fuse_dir) h->kern_alg += 12.0 * (double)n;  // the timed launch also reads z (4) and writes the direction (8)
    if (trace)
        fprintf(stderr, "[sagg] %d iterations (%lld enqueued, %d polls), relative residual %.2e, status %d\n", *iters,
                (long long)enqueued, polls, *resid, status);
    if (trace && H->tail >= 0 && H->td.stamps) {
        long long ts[64];
        if (hipMemcpy(ts, H->td.stamps, sizeof ts, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "[sagg] tail phases (us):");
            for (int k = 1; k < (int)ts[63] && k < 63; ++k) fprintf(stderr, " %.2f", (double)(ts[k] - ts[k - 1]) * 0.01);
            fprintf(stderr, "  total %.2f, image %d B, lds %d B\n", (double)(ts[ts[63] - 1] - ts[0]) * 0.01,
                    H->td.image_bytes, H->td.lds_bytes);
            for (int k = 0; k < H->td.nlev; ++k)
                fprintf(stderr, "[sagg]   tail level %d: n %d width %d lpr %d nq %d\n", k, H->td.lv[k].n,
                        H->td.lv[k].width, H->td.lv[k].lpr, H->td.lv[k].nq);
        }
    }
    if (status != 1) {
        H->last_iters = 0;
        return -1;
    }
    H->last_iters = (int)its;
    H->hint_iters = (int)its;
    H->hint_n = n;
    H->hint_nnz = H->pool[0]->nnz;
    *info = 0;
    return NODAL_OK;
}

#include "sagg_multi.h"
