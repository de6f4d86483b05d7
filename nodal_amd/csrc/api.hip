// extern "C" entry points of libnodal_hip.so (see include/nodal_hip.h).
#include <stdlib.h>
#include <string.h>

#include "ctx.h"
#include <algorithm>
#include <atomic>

int dense_prepare(nodal_ctx *h);  // sparse.hip
int pair_read_host(nodal_ctx *h, const double *x, int32_t ia, int32_t ib, double *out);  // sparse.hip
int nodal_upload_internal(nodal_ctx *h, int64_t ncomp, const uint8_t *type, const double *value,
                          const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *d,
                          const int32_t *drv, const int32_t *k, int32_t K, int32_t B);

// Host threads inside an API call right now, process-wide (nested calls of one thread count once).  A solve that
// has the device to itself spreads independent pieces over streams of its own (the multigrid setup's second stream,
// the direct route's lanes); with several solves in flight -- one context and host thread each -- the extra queues
// only get in each other's way (four contexts with two streams each: 232 -> 178 circuits/s), and the solves overlap
// anyway.
static std::atomic<int> g_calls_in_flight{0};
static thread_local int t_call_depth = 0;
int nodal_calls_in_flight() { return g_calls_in_flight.load(std::memory_order_relaxed); }
// Handles alive in the process.  (The runtime spreads a process's streams over a handful of hardware queues: a second
// stream per handle makes the main streams of four handles share queues -- four solves in flight, symbolic phases
// kept: 232 -> 181 circuits/s even with the extra streams idle, and a stream created while another handle's extra
// stream exists keeps its shared queue after that one is gone.  Hence extra streams are an OPTION of the handle.)
static std::atomic<int> g_live_handles{0};
int nodal_live_handles() { return g_live_handles.load(std::memory_order_relaxed); }
bool nodal_extra_streams_ok(const nodal_ctx *ctx) {  // the handle was told it is alone (NODAL_OPT_EXTRA_STREAMS) and no other call runs
    const nodal_ctx *h = ctx->stream_owner ? ctx->stream_owner : ctx;
    return h->extra_streams && nodal_calls_in_flight() <= 1;
}

namespace {

// selects the handle's device for the duration of an API call and gives the calling
// thread its own current device back afterwards
struct DeviceGuard {
    int prev = -1;
    FillStreamScope fill;
    explicit DeviceGuard(nodal_ctx *h) : fill(h->stream) {
        if (t_call_depth++ == 0) g_calls_in_flight.fetch_add(1, std::memory_order_relaxed);
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != h->device) (void)hipSetDevice(h->device);
        else prev = -1;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
        if (--t_call_depth == 0) g_calls_in_flight.fetch_sub(1, std::memory_order_relaxed);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

template <class T>
int upload(nodal_ctx *h, DevBuf &buf, const T *src, int64_t count) {
    NODAL_HIP_TRY(h, buf.reserve((size_t)count * sizeof(T) + 16));
    if (count > 0)
        NODAL_HIP_TRY(h, hipMemcpyAsync(buf.p, src, (size_t)count * sizeof(T),
                                        hipMemcpyHostToDevice, h->stream));
    return NODAL_OK;
}

// Range check of the uploaded table ON THE DEVICE (a kernel must never see an out-of-range node; the
// host loop that did this walked 2e6 rows x 8 columns on one thread: half of the upload time of the
// 1e6-node grid).  *bad = first offending row (~0 if none).
__global__ __launch_bounds__(256) void validate_table(int64_t ncomp, int32_t K, int32_t B,
                                                      const uint8_t *__restrict__ type, const int32_t *__restrict__ a,
                                                      const int32_t *__restrict__ b, const int32_t *__restrict__ c,
                                                      const int32_t *__restrict__ d, const int32_t *__restrict__ drv,
                                                      const int32_t *__restrict__ k, unsigned long long *__restrict__ bad) {
    unsigned sources = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < ncomp; i += (int64_t)gridDim.x * 256) {
        const uint8_t t = type[i];
        const bool branch = t >= NODAL_T_E && t <= NODAL_T_CCCS;
        const int32_t ci = c ? c[i] : -1, di = d ? d[i] : -1, ri = drv ? drv[i] : -1, ki = k ? k[i] : -1;
        // (a plain table -- no c / d / drv / k columns -- can only hold resistors and current sources: any other
        // row would stamp nothing, silently)
        const bool plain_ok = c != nullptr || t == NODAL_T_R || t == NODAL_T_A;
        const bool ok = t <= NODAL_T_GM && plain_ok && a[i] >= -1 && a[i] < K && b[i] >= -1 && b[i] < K && ci >= -1 &&
                        ci < K && di >= -1 && di < K && ri >= -1 && ri < ncomp && ki >= -1 && ki < B &&
                        (branch == (ki >= 0));
        if (!ok) atomicMin(bad, (unsigned long long)i);
        sources += (t == NODAL_T_A || t == NODAL_T_E) ? 1u : 0u;
    }
    // how many components stamp the right-hand side (bad[1]): lets the symbolic phase group a handful of rhs
    // stamps with two launches instead of thirteen.  One atomic per workgroup that has any.
    __shared__ unsigned wsum[4];
    for (int off = 32; off > 0; off >>= 1) sources += __shfl_down(sources, off, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sources;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned tot = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (tot) atomicAdd(bad + 1, (unsigned long long)tot);
    }
}

// rec: (row, c, d, drv, k) of the rows that read those columns (packed by the host: nodal_upload_internal)
__global__ __launch_bounds__(256) void place_dependent_rows(int64_t count, const int32_t *__restrict__ rec, int64_t ncomp,
                                                            int32_t *__restrict__ c, int32_t *__restrict__ d,
                                                            int32_t *__restrict__ drv, int32_t *__restrict__ k) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= count) return;
    const int32_t row = rec[5 * q];
    if (row < 0 || row >= ncomp) return;
    c[row] = rec[5 * q + 1];
    d[row] = rec[5 * q + 2];
    drv[row] = rec[5 * q + 3];
    k[row] = rec[5 * q + 4];
}

double elapsed(nodal_ctx *h, int a, int b) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev[a], h->ev[b]) != hipSuccess) return 0.0;
    return ms;
}

}  // namespace

extern "C" {

const char *nodal_version(void) { return "nodal_hip 0.1 gfx950"; }

int nodal_create(int device_id, nodal_handle *out) {
    if (!out) return NODAL_E_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count)
        return NODAL_E_HIP;
    nodal_ctx *h = new nodal_ctx();
    h->device = device_id;
    // main stream: highest priority -- it carries the latency-critical chains (LU panel
    // factorisation, multigrid cycles)
    int lo = 0, hi = 0;  // least / greatest priority
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (const char *e = getenv("NODAL_STREAM_PRIORITY"))  // "normal": the default priority instead of the highest
        if (e[0] == 'n') hi = 0;
    if (hipSetDevice(device_id) != hipSuccess ||
        hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, hi) != hipSuccess) {
        delete h;
        return NODAL_E_HIP;
    }
    for (auto &e : h->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete h;
            return NODAL_E_HIP;
        }
    if (const char *e = getenv("NODAL_DENSE_BLOCKINV")) h->dense_blockinv = atoi(e) != 0;
    if (const char *e = getenv("NODAL_GJ_SCALAR")) h->gj_scalar = atoi(e);
    if (const char *e = getenv("NODAL_GEPP_PANEL")) h->gepp_panel = atoi(e) != 0;
    if (const char *e = getenv("NODAL_PRESOLVE")) h->use_presolve = atoi(e) != 0;  // 0: branch equations stay in the system
    if (const char *e = getenv("NODAL_EXTRA_STREAMS")) h->extra_streams = atoi(e) != 0;
    g_live_handles.fetch_add(1, std::memory_order_relaxed);
    *out = h;
    return NODAL_OK;
}

}  // extern "C"


void *nodal_pinned(nodal_ctx *ctx) {
    nodal_ctx *h = ctx->stream_owner ? ctx->stream_owner : ctx;
    if (!h->pinned && hipHostMalloc(&h->pinned, NODAL_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) {
        h->pinned = nullptr;
        (void)hipGetLastError();
    }
    return h->pinned;
}

void *nodal_pinned_arena(nodal_ctx *ctx, size_t bytes) {
    nodal_ctx *h = ctx->stream_owner ? ctx->stream_owner : ctx;
    if (bytes <= h->arena_bytes) return h->arena;
    if (h->arena) (void)hipHostFree(h->arena);
    h->arena = nullptr;
    h->arena_bytes = 0;
    const size_t want = bytes + (bytes >> 2) + 4096;
    if (hipHostMalloc(&h->arena, want, hipHostMallocDefault) != hipSuccess) {
        h->arena = nullptr;
        (void)hipGetLastError();
        return nullptr;
    }
    h->arena_bytes = want;
    return h->arena;
}

int nodal_read_words(nodal_ctx *h, void *dst, const void *dev_src, size_t bytes) {
    void *pin = bytes <= NODAL_PINNED_BYTES ? nodal_pinned(h) : nullptr;
    NODAL_HIP_TRY(h, hipMemcpyAsync(pin ? pin : dst, dev_src, bytes, hipMemcpyDeviceToHost, h->stream));
    NODAL_WAIT_STREAM(h, h->stream);
    if (pin) memcpy(dst, pin, bytes);
    return NODAL_OK;
}

int nodal_ensure_aux_streams(nodal_ctx *ctx) {
    nodal_ctx *h = ctx->stream_owner ? ctx->stream_owner : ctx;
    if (!h->stream2 || !h->stream3) {
        int prev = -1;
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(h->device);
        int lo = 0, hi = 0;  // least / greatest priority
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        int status = NODAL_OK;
        // second stream for the dense LU's trailing updates (lookahead).  It is confined to
        // 224 of the 256 CUs: the panel chain on the main stream is a sequence of small
        // latency-bound kernels and must always find idle CUs, otherwise every one of its
        // ~1300 launches queues behind 58-us GEMM workgroups.  (If the CU mask is refused,
        // fall back to a low-priority stream.)
        if (!h->stream2) {
            uint32_t mask[8];
            int reserve = 32;  // CUs kept free for the main stream (NODAL_PANEL_CUS to tune)
            if (const char *e = getenv("NODAL_PANEL_CUS")) reserve = atoi(e);
            if (reserve < 0) reserve = 0;
            if (reserve > 224) reserve = 224;
            for (int i = 0; i < 8; ++i) mask[i] = 0xFFFFFFFFu;
            for (int cu = 0; cu < reserve; ++cu) mask[cu / 32] &= ~(1u << (cu % 32));
            if (hipExtStreamCreateWithCUMask(&h->stream2, 8, mask) != hipSuccess) {
                h->stream2 = nullptr;
                (void)hipGetLastError();
            }
            if (!h->stream2 && hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, lo) != hipSuccess) status = NODAL_E_HIP;
        }
        if (status == NODAL_OK && !h->stream3 &&
            hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, lo) != hipSuccess) status = NODAL_E_HIP;
        for (auto &e : h->ev_la)
            if (status == NODAL_OK && !e &&
                hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) status = NODAL_E_HIP;
        for (auto &e : h->ev_bi)
            if (status == NODAL_OK && !e &&
                hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice) != hipSuccess) status = NODAL_E_HIP;
        if (prev >= 0) (void)hipSetDevice(prev);
        if (status != NODAL_OK) return nodal_fail(ctx, status, "could not create the dense paths' streams");
    }
    if (ctx != h) {
        ctx->stream2 = h->stream2;
        ctx->stream3 = h->stream3;
        for (int i = 0; i < 2; ++i) ctx->ev_la[i] = h->ev_la[i];
        for (int i = 0; i < 6; ++i) ctx->ev_bi[i] = h->ev_bi[i];
    }
    return NODAL_OK;
}

void nodal_free_buffers(nodal_ctx *h) {
    amg_destroy(h);
    sagg_destroy(h);
    slu_destroy(h);
    presolve_free_plan(h);
    nodal_free_block_child(h);
    if (h->reduced) {
        nodal_free_buffers(h->reduced);
        delete h->reduced;
        h->reduced = nullptr;
    }
    if (h->lowdeg) {
        nodal_free_buffers(h->lowdeg);
        delete h->lowdeg;
        h->lowdeg = nullptr;
    }
    DevBuf *bufs[] = {&h->type, &h->value, &h->a, &h->b, &h->c, &h->d, &h->drv, &h->k,
                      &h->values_batch, &h->indptr, &h->indices, &h->rowidx, &h->cptr,
                      &h->contrib, &h->rhs_row, &h->rhs_cptr, &h->rhs_contrib, &h->diag_pos,
                      &h->data, &h->rhs, &h->status, &h->x, &h->dense, &h->piv, &h->work,
                      &h->work2, &h->work3, &h->solver, &h->krylov, &h->gn_indptr, &h->gn_indices,
                      &h->gn_rowidx, &h->gn_data, &h->gn_diag, &h->schur, &h->ps_buf, &h->ps_newidx,
                      &h->ps_hits, &h->ps_stage, &h->grounded, &h->ld_newidx, &h->ld_work, &h->batch_scale, &h->rhs_none};
    for (DevBuf *b : bufs) b->release();
    for (auto &e : h->evpool) (void)hipEventDestroy(e);
    h->evpool.clear();
}

// NODAL_POISON=2: every buffer that carries nothing from one solve to the next -- the solution, the dense
// panel, the scratch and Krylov areas -- becomes 0xFF bytes (NaNs / -1) in this context and its child
// contexts, in order on the handle's stream: the state a pooled handle is in after a singular solve of a
// larger system.  A kernel that reads a slot of these before writing it shows up as a NaN or a fault.
void nodal_poison_scratch(nodal_ctx *h) {
    if (nodal_poison_level() < 2 || !h) return;
    DevBuf *bufs[] = {&h->x, &h->dense, &h->piv, &h->work, &h->work2, &h->work3, &h->solver, &h->krylov,
                      &h->ld_work, &h->schur, &h->batch_x, &h->batch_scale, &h->rhs_none};
    for (DevBuf *b : bufs) b->poison(h->stream);
    slu_poison(h);
    nodal_poison_scratch(h->reduced);
    nodal_poison_scratch(h->lowdeg);
    nodal_poison_scratch(h->blocksys);
}

void nodal_nan_probe(nodal_ctx *h, const double *dev, int64_t n, const char *tag) {
    static const bool on = getenv("NODAL_NANCHECK") != nullptr;
    if (!on || !dev || n <= 0) return;
    std::vector<double> host((size_t)n);
    if (nodal_wait_stream(h, h->stream, NODAL_SITE) != NODAL_OK ||
        hipMemcpyAsync(host.data(), dev, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        nodal_wait_stream(h, h->stream, NODAL_SITE) != NODAL_OK) {
        fprintf(stderr, "[nancheck] %s: copy failed\n", tag);
        return;
    }
    int64_t bad = 0, first = -1;
    for (int64_t i = 0; i < n; ++i)
        if (!(host[i] - host[i] == 0.0)) {
            if (first < 0) first = i;
            ++bad;
        }
    fprintf(stderr, "[nancheck] %s: %lld of %lld not finite (first at %lld)\n", tag, (long long)bad, (long long)n,
            (long long)first);
}

extern "C" {

int nodal_destroy(nodal_handle h) {
    if (!h) return NODAL_OK;
    DeviceGuard g(h);
    // (bounded like every other wait: a handle whose device work never ends is leaked, not freed under running kernels)
    bool drained = !h->hung && nodal_wait_stream(h, h->stream, NODAL_SITE) == NODAL_OK;
    if (drained && h->stream2) drained = nodal_wait_stream(h, h->stream2, NODAL_SITE) == NODAL_OK;
    if (drained && h->stream3) drained = nodal_wait_stream(h, h->stream3, NODAL_SITE) == NODAL_OK;
    if (!drained) {
        fprintf(stderr, "[nodal] nodal_destroy: the handle's streams did not drain (%s): its device memory is leaked\n",
                h->err.c_str());
        g_live_handles.fetch_sub(1, std::memory_order_relaxed);
        return NODAL_E_HIP;
    }
    nodal_free_buffers(h);
    for (auto &e : h->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : h->ev_la)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : h->ev_bi)
        if (e) (void)hipEventDestroy(e);
    if (h->stream3) (void)hipStreamDestroy(h->stream3);
    if (h->stream2) (void)hipStreamDestroy(h->stream2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->pinned) (void)hipHostFree(h->pinned);
    if (h->arena) (void)hipHostFree(h->arena);
    delete h;
    g_live_handles.fetch_sub(1, std::memory_order_relaxed);
    return NODAL_OK;
}

const char *nodal_last_error(nodal_handle h) { return h ? h->err.c_str() : "null handle"; }

int nodal_upload_components(nodal_handle h, int64_t ncomp, const uint8_t *type,
                            const double *value, const int32_t *a, const int32_t *b,
                            const int32_t *c, const int32_t *d, const int32_t *drv,
                            const int32_t *k, int32_t K, int32_t B) {
    return nodal_upload_internal(h, ncomp, type, value, a, b, c, d, drv, k, K, B);
}

}  // extern "C"

int nodal_upload_internal(nodal_ctx *h, int64_t ncomp, const uint8_t *type, const double *value,
                          const int32_t *a, const int32_t *b, const int32_t *c, const int32_t *d,
                          const int32_t *drv, const int32_t *k, int32_t K, int32_t B) {
    if (!h || ncomp < 0 || K < 0 || B < 0) return NODAL_E_INVALID;
    // c, d, drv, k may be null TOGETHER when the table holds no dependent source and no branch (B == 0:
    // resistors and current sources read none of them): 16 of the 33 bytes per row stay on the host
    const bool plain = !c && !d && !drv && !k;
    if (ncomp > 0 && (!type || !value || !a || !b || (!plain && (!c || !d || !drv || !k)) || (plain && B != 0)))
        return nodal_fail(h, NODAL_E_INVALID, "null component column");
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    const int64_t n = (int64_t)K + B;
    h->have_table = h->have_symbolic = h->have_numeric = h->have_x = false;
    ++h->table_epoch;
    h->batch_count = 0;
    h->ncomp = ncomp;
    h->K = K;
    h->B = B;
    h->n = n;
    h->batch = 0;
    // (an early return must not leave DMA from the caller's columns in flight: the caller may free them)
    auto fail_synced = [&](int status) {
        (void)nodal_wait_stream(h, h->stream, NODAL_SITE);
        return status;
    };
#define UPLOAD_TRY(expr)                             \
    do {                                             \
        const int _s = (expr);                       \
        if (_s != NODAL_OK) return fail_synced(_s);  \
    } while (0)
    UPLOAD_TRY(upload(h, h->type, type, ncomp));
    UPLOAD_TRY(upload(h, h->value, value, ncomp));
    UPLOAD_TRY(upload(h, h->a, a, ncomp));
    UPLOAD_TRY(upload(h, h->b, b, ncomp));
    // c, d, drv, k: -1 everywhere, then (a table with branches) the entries of the rows that use them.  Those rows
    // are few -- 3e4 of config 5's 2e6 -- and 16 of the 33 bytes per row are these four columns: the host lists the
    // dependent rows while the first four columns are on their way (threads over chunks of the type column), packs
    // their entries into the page-locked arena, one copy and one scatter kernel place them (config 5: 67 -> 35 MB up).
    {
        DevBuf *cols[] = {&h->c, &h->d, &h->drv, &h->k};
        for (DevBuf *col : cols) {
            if (col->reserve((size_t)ncomp * 4 + 16) != hipSuccess ||
                (ncomp > 0 && hipMemsetAsync(col->p, 0xFF, (size_t)ncomp * 4, h->stream) != hipSuccess))  // -1
                return fail_synced(nodal_fail(h, NODAL_E_NOMEM, "upload: could not allocate a table column"));
        }
    }
    std::vector<int64_t> branch_rows;
    if (!plain && ncomp > 0) {
        // dependent rows by chunk of the type column, in file order
        constexpr int CHUNKS = 8;
        std::vector<int64_t> found[CHUNKS];
        {
            nodal_parallel_chunks(CHUNKS, 1, ncomp >= (1 << 18) ? CHUNKS : 1, [&](int64_t c0, int64_t c1) {
                for (int64_t q = c0; q < c1; ++q) {
                    std::vector<int64_t> &out = found[q];
                    for (int64_t i = ncomp * q / CHUNKS, e = ncomp * (q + 1) / CHUNKS; i < e; ++i)
                        if (type[i] >= NODAL_T_E) out.push_back(i);
                }
            });
            int64_t total = 0;
            for (int q = 0; q < CHUNKS; ++q) total += (int64_t)found[q].size();
            struct DepRow { int32_t row, c, d, drv, k; };
            DepRow *rec = total ? static_cast<DepRow *>(nodal_pinned_arena(h, (size_t)total * sizeof(DepRow))) : nullptr;
            if (total && !rec) return fail_synced(nodal_fail(h, NODAL_E_NOMEM, "upload: no page-locked staging memory"));
            int64_t at = 0;
            branch_rows.reserve((size_t)total);
            for (int q = 0; q < CHUNKS; ++q)
                for (const int64_t i : found[q]) {
                    rec[at++] = DepRow{(int32_t)i, c[i], d[i], drv[i], k[i]};
                    if (type[i] <= NODAL_T_CCCS) branch_rows.push_back(i);
                }
            if (total) {
                if (h->ps_stage.reserve((size_t)total * sizeof(DepRow) + 64) != hipSuccess)
                    return fail_synced(nodal_fail(h, NODAL_E_NOMEM, "upload: dependent rows"));
                if (hipMemcpyAsync(h->ps_stage.p, rec, (size_t)total * sizeof(DepRow), hipMemcpyHostToDevice, h->stream) != hipSuccess)
                    return fail_synced(nodal_fail(h, NODAL_E_HIP, "upload: dependent rows"));
                place_dependent_rows<<<(unsigned)((total + 255) / 256), 256, 0, h->stream>>>(
                    total, reinterpret_cast<const int32_t *>(h->ps_stage.p), ncomp, h->c.as<int32_t>(), h->d.as<int32_t>(),
                    h->drv.as<int32_t>(), h->k.as<int32_t>());
                if (hipGetLastError() != hipSuccess) return fail_synced(nodal_fail(h, NODAL_E_HIP, "upload: dependent rows launch failed"));
            }
        }
    }
    // the range check runs on the device, behind the copies; its verdict is the one word that comes back
    if (h->status.reserve(64) != hipSuccess) return fail_synced(nodal_fail(h, NODAL_E_NOMEM, "upload: status words"));
    unsigned long long *bad_dev = h->status.as<unsigned long long>() + 4;
    unsigned long long bad[2] = {~0ull, 0ull};
    if (hipMemsetAsync(bad_dev, 0xFF, 8, h->stream) != hipSuccess || hipMemsetAsync(bad_dev + 1, 0, 8, h->stream) != hipSuccess)
        return fail_synced(nodal_fail(h, NODAL_E_HIP, "upload: could not clear the status words"));
    if (ncomp > 0) {
        const int64_t blocks = (ncomp + 255) / 256;
        validate_table<<<(unsigned)(blocks > 4096 ? 4096 : blocks), 256, 0, h->stream>>>(
            ncomp, K, B, h->type.as<uint8_t>(), h->a.as<int32_t>(), h->b.as<int32_t>(),
            plain ? nullptr : h->c.as<int32_t>(), plain ? nullptr : h->d.as<int32_t>(),
            plain ? nullptr : h->drv.as<int32_t>(), plain ? nullptr : h->k.as<int32_t>(), bad_dev);
        // (the entries of the dependent rows only: a row number that is out of range was not placed and its k stays -1,
        // which the branch-type test above reports)
        if (hipGetLastError() != hipSuccess) return fail_synced(nodal_fail(h, NODAL_E_HIP, "upload: range check launch failed"));
    }
#undef UPLOAD_TRY
    // the host's view of the table (only systems with branch equations are presolved), made while the copies run:
    // the caller's columns in place when it has promised to keep them (NODAL_OPT_BORROW_TABLE), else copies (threads)
    if (h->keep_host_table && B > 0) {
        HostTable &t = h->host;
        if (h->borrow_table) {
            t.type.borrow(type, (size_t)ncomp);
            t.value.borrow(value, (size_t)ncomp);
            t.a.borrow(a, (size_t)ncomp);
            t.b.borrow(b, (size_t)ncomp);
            t.c.borrow(c, (size_t)ncomp);
            t.d.borrow(d, (size_t)ncomp);
            t.drv.borrow(drv, (size_t)ncomp);
            t.k.borrow(k, (size_t)ncomp);
        } else {
            t.type.resize((size_t)ncomp);
            t.value.resize((size_t)ncomp);
            HostCol<int32_t> *ic[] = {&t.a, &t.b, &t.c, &t.d, &t.drv, &t.k};
            const int32_t *src[] = {a, b, c, d, drv, k};
            for (HostCol<int32_t> *col : ic) col->resize((size_t)ncomp);
            nodal_parallel_chunks(ncomp, 1 << 16, 8, [&](int64_t lo, int64_t hi) {
                memcpy(t.type.own.data() + lo, type + lo, (size_t)(hi - lo));
                memcpy(t.value.own.data() + lo, value + lo, (size_t)(hi - lo) * 8);
                for (int q = 0; q < 6; ++q) memcpy(ic[q]->own.data() + lo, src[q] + lo, (size_t)(hi - lo) * 4);
            });
        }
        t.values_batch.clear();
        t.branch_rows.swap(branch_rows);
    } else {
        h->host = HostTable();
    }
    NODAL_TRY(nodal_read_words(h, bad, bad_dev, 16));
    h->rhs_items = (int64_t)bad[1];
    if (bad[0] != ~0ull) {
        h->ncomp = 0;  // (nothing may run on this table)
        h->host = HostTable();
        return nodal_fail(h, NODAL_E_INVALID, "component table row out of range");
    }
    h->have_table = true;
    return NODAL_OK;
}

extern "C" {

int nodal_upload_values(nodal_handle h, int32_t batch, const double *values) {
    if (!h || !h->have_table || batch < 1 || !values) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    NODAL_TRY(upload(h, h->values_batch, values, (int64_t)batch * h->ncomp));
    NODAL_WAIT_STREAM(h, h->stream);
    if (h->keep_host_table && h->B > 0)
        h->host.values_batch.assign(values, values + (size_t)batch * h->ncomp);
    h->batch = batch;
    h->have_numeric = h->have_x = false;
    return NODAL_OK;
}

int nodal_assemble_symbolic(nodal_handle h) {
    if (!h) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
    int s = stamp_symbolic(h);
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
    NODAL_WAIT_EVENT(h, h->ev[1], h->stream);
    h->ms[0] = elapsed(h, 0, 1);
    return s;
}

int nodal_assemble_numeric(nodal_handle h, int32_t member, int64_t *bad_component) {
    if (!h) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
    int s = stamp_numeric(h, member, bad_component);
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
    NODAL_WAIT_EVENT(h, h->ev[1], h->stream);
    h->ms[1] = elapsed(h, 0, 1);
    return s;
}

int nodal_get_sizes(nodal_handle h, int64_t *n, int64_t *nnz, int64_t *ncontrib) {
    if (!h || !h->have_symbolic) return NODAL_E_INVALID;
    if (n) *n = h->n;
    if (nnz) *nnz = h->nnz;
    if (ncontrib) *ncontrib = h->ncontrib;
    return NODAL_OK;
}

int nodal_export_csr(nodal_handle h, int32_t *indptr, int32_t *indices, double *data,
                     double *rhs) {
    if (!h || !h->have_symbolic) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    hipStream_t st = h->stream;
    if (indptr)
        NODAL_HIP_TRY(h, hipMemcpyAsync(indptr, h->indptr.p, (size_t)(h->n + 1) * 4,
                                        hipMemcpyDeviceToHost, st));
    if (indices && h->nnz)
        NODAL_HIP_TRY(h, hipMemcpyAsync(indices, h->indices.p, (size_t)h->nnz * 4,
                                        hipMemcpyDeviceToHost, st));
    if (data || rhs) {
        if (!h->have_numeric) return nodal_fail(h, NODAL_E_INVALID, "assemble_numeric not called");
        if (data && h->nnz)
            NODAL_HIP_TRY(h, hipMemcpyAsync(data, h->data.p, (size_t)h->nnz * 8,
                                            hipMemcpyDeviceToHost, st));
        if (rhs && h->n)
            NODAL_HIP_TRY(h, hipMemcpyAsync(rhs, h->rhs.p, (size_t)h->n * 8,
                                            hipMemcpyDeviceToHost, st));
    }
    NODAL_WAIT_STREAM(h, st);
    return NODAL_OK;
}

int nodal_export_dense(nodal_handle h, double *G, double *rhs) {
    if (!h || !h->have_numeric) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    const int64_t n = h->n;
    if (G && n) {
        NODAL_HIP_TRY(h, h->dense.reserve((size_t)dense_lda(n) * (size_t)(n + 1) * 8 + 64));
        NODAL_TRY(stamp_to_dense(h, h->dense.as<double>(), n, false));
        NODAL_HIP_TRY(h, hipMemcpyAsync(G, h->dense.p, (size_t)n * n * 8, hipMemcpyDeviceToHost,
                                        h->stream));
    }
    if (rhs && n)
        NODAL_HIP_TRY(h, hipMemcpyAsync(rhs, h->rhs.p, (size_t)n * 8, hipMemcpyDeviceToHost,
                                        h->stream));
    NODAL_WAIT_STREAM(h, h->stream);
    return NODAL_OK;
}

int nodal_download_x(nodal_handle h, double *x) {
    if (!h || !h->have_x || !x) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    if (h->n)
        NODAL_HIP_TRY(h, hipMemcpyAsync(x, h->x.p, (size_t)h->n * 8, hipMemcpyDeviceToHost,
                                        h->stream));
    NODAL_WAIT_STREAM(h, h->stream);
    return NODAL_OK;
}

int nodal_solve_dense(nodal_handle h, double *x, int32_t *info) {
    if (!h || !info) return NODAL_E_INVALID;
    if (!h->have_numeric) return nodal_fail(h, NODAL_E_INVALID, "assemble_numeric not called");
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    nodal_poison_scratch(h);
    *info = 0;
    h->have_x = false;
    h->last_batch_block = false;
    h->last_iterations = 0;
    h->amg_levels = 0;
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
    if (h->n > 0) {
        // Large systems with voltage-defined branches: eliminate those branches first
        // (presolve.hip).  If what remains is a passive network, the dense block elimination
        // solves it without pivoting; the answer is checked against the ORIGINAL equations.
        bool done = false;
        if (h->n > 512 && h->B > 0 && h->use_presolve && !h->force_pivoting && !h->passive_network) {
            int32_t it = 0;
            double rs = 0.0;
            NODAL_TRY(presolve_solve(h, &done, info, &it, &rs, true));
        }
        // Passive networks made of chains / ladders / trees: the exact elimination of nodes with
        // <= 2 neighbours (lowdeg.hip) shrinks the system before anything is formed densely
        // (ladder of 5000 sections: 1.9 ms instead of 5.7).  Grids have no such nodes: one probe
        // kernel, then the block elimination as before.
        if (!done && h->n > 1024 && h->B == 0 && h->passive_network && !h->force_pivoting) {
            int32_t it = 0;
            double rs = 0.0;
            NODAL_TRY(lowdeg_solve(h, 8, &done, info, &it, &rs));
            if (done && *info > 0) NODAL_TRY(dense_fill_nan(h, h->x.as<double>(), h->n));
        }
        if (!done) {
            *info = 0;
            NODAL_TRY(dense_prepare(h));
            NODAL_TRY(dense_factor_solve(h, info));
        }
    }
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
    NODAL_WAIT_EVENT(h, h->ev[1], h->stream);
    h->ms[2] = elapsed(h, 0, 1);
    if (*info > 0) return nodal_fail(h, NODAL_E_SINGULAR, "singular matrix: a zero pivot or a floating sub-network");
    h->have_x = true;
    if (x) return nodal_download_x(h, x);
    return NODAL_OK;
}

int nodal_solve_sparse(nodal_handle h, int32_t method, double *x, int32_t *info, int32_t *iters,
                       double *resid) {
    if (!h || !info) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    nodal_poison_scratch(h);
    int32_t it = 0;
    double rs = 0;
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
    h->amg_levels = 0;
    h->last_batch_block = false;
    int s = sparse_solve(h, method, info, &it, &rs);
    h->last_iterations = it;
    h->last_relres = rs;
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
    NODAL_WAIT_EVENT(h, h->ev[1], h->stream);
    h->ms[2] = elapsed(h, 0, 1);
    if (iters) *iters = it;
    if (resid) *resid = rs;
    if (s != NODAL_OK) return s;
    if (x) return nodal_download_x(h, x);
    return NODAL_OK;
}

int nodal_solve_pairs(nodal_handle h, int32_t dense, int32_t npairs, const int32_t *ia,
                      const int32_t *ib, double *resistance, int32_t *info) {
    if (!h || !info || npairs < 0 || (npairs > 0 && (!ia || !ib || !resistance)))
        return NODAL_E_INVALID;
    if (!h->have_numeric) return nodal_fail(h, NODAL_E_INVALID, "assemble_numeric not called");
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    nodal_poison_scratch(h);
    *info = 0;
    const int64_t n = h->n;
    for (int32_t q = 0; q < npairs; ++q)
        if (ia[q] < -1 || ia[q] >= h->K || ib[q] < -1 || ib[q] >= h->K)
            return nodal_fail(h, NODAL_E_INVALID, "pair index out of range");
    if (npairs == 0) return NODAL_OK;
    h->have_x = false;
    NODAL_HIP_TRY(h, h->schur.reserve((size_t)npairs * 8 + 64));
    double *res = h->schur.as<double>();
    if (n == 0) {
        for (int32_t q = 0; q < npairs; ++q) resistance[q] = 0.0;
        return NODAL_OK;
    }
    bool reduced = false;
    if (n > 1024 && h->B == 0 && h->passive_network && !h->force_pivoting) {
        // chains / ladders / trees: exact elimination of the low-degree nodes per pair (lowdeg.hip)
        NODAL_TRY(lowdeg_solve_pairs(h, npairs, ia, ib, res, &reduced, info));
        if (reduced && *info > 0) {
            if (dense) return nodal_fail(h, NODAL_E_SINGULAR, "singular matrix: a floating sub-network");
            for (int32_t q = 0; q < npairs; ++q) resistance[q] = __builtin_nan("");
            return NODAL_OK;
        }
    }
    if (reduced) {
        // results are in `res`
    } else if (dense || n <= 64 || (!(h->B == 0 && h->passive_network) && n <= 8192)) {
        // (a non-passive network above 8192 unknowns on the sparse switch: one sparse LU, sparse_solve_pairs)
        // one LU for up to CHUNK pairs: they are extra right-hand-side columns
        const int32_t CHUNK = 512;
        for (int32_t q0 = 0; q0 < npairs; q0 += CHUNK) {
            const int32_t m = npairs - q0 < CHUNK ? npairs - q0 : CHUNK;
            NODAL_TRY(dense_prepare_pairs(h, m, ia + q0, ib + q0));
            NODAL_HIP_TRY(h, h->solver.reserve((size_t)n * m * 8 + 64));
            NODAL_TRY(dense_factor_solve_multi(h, m, h->solver.as<double>(), n, info));
            if (*info > 0) {
                if (dense) return nodal_fail(h, NODAL_E_SINGULAR, "singular matrix: exact zero pivot");
                for (int32_t q = 0; q < npairs; ++q) resistance[q] = __builtin_nan("");
                return NODAL_OK;
            }
            for (int32_t q = 0; q < m; ++q)
                NODAL_TRY(pair_read_host(h, h->solver.as<double>() + (int64_t)q * n, ia[q0 + q],
                                         ib[q0 + q], res + q0 + q));
        }
    } else {
        NODAL_TRY(sparse_solve_pairs(h, npairs, ia, ib, res, info));
        if (*info > 0) {
            for (int32_t q = 0; q < npairs; ++q) resistance[q] = __builtin_nan("");
            return NODAL_OK;
        }
    }
    NODAL_HIP_TRY(h, hipMemcpyAsync(resistance, res, (size_t)npairs * 8, hipMemcpyDeviceToHost,
                                    h->stream));
    NODAL_WAIT_STREAM(h, h->stream);
    return NODAL_OK;
}

int nodal_residual(nodal_handle h, double *scaled_residual) {
    if (!h || !scaled_residual) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    if (h->last_batch_block && h->blocksys) {  // whole block-diagonal system of the last nodal_run_batch
        const int s = sparse_residual(h->blocksys, scaled_residual);
        if (s != NODAL_OK) h->err = h->blocksys->err;
        return s;
    }
    return sparse_residual(h, scaled_residual);
}

int nodal_run(nodal_handle h, int32_t dense, int32_t member, int32_t reuse_symbolic,
              int32_t *info) {
    if (!h || !info) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    // symbolic + numeric back to back: the symbolic phase's timer is read AFTER the numeric phase has waited for its
    // status words anyway (waiting for it in between left the GPU idle for ~30 us per circuit)
    hipEvent_t ev_s0 = h->ev[2], ev_s1 = h->ev[3];  // (free until the solve records them around its dominant kernel)
    const bool do_symbolic = !(reuse_symbolic && h->have_symbolic);
    if (do_symbolic) {
        NODAL_HIP_TRY(h, hipEventRecord(ev_s0, h->stream));
        NODAL_TRY(stamp_symbolic(h));
        NODAL_HIP_TRY(h, hipEventRecord(ev_s1, h->stream));
    }
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[0], h->stream));
    const int sn = stamp_numeric(h, member, nullptr);
    NODAL_HIP_TRY(h, hipEventRecord(h->ev[1], h->stream));
    NODAL_WAIT_EVENT(h, h->ev[1], h->stream);
    h->ms[1] = elapsed(h, 0, 1);
    h->ms[0] = 0.0;  // (kept: nothing ran)
    if (do_symbolic) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, ev_s0, ev_s1) == hipSuccess) h->ms[0] = ms;
    }
    NODAL_TRY(sn);
    if (dense) return nodal_solve_dense(h, nullptr, info);
    return nodal_solve_sparse(h, NODAL_SPARSE_AUTO, nullptr, info, nullptr, nullptr);
}

int nodal_last_timings(nodal_handle h, double *ms3) {
    if (!h || !ms3) return NODAL_E_INVALID;
    ms3[0] = h->ms[0];
    ms3[1] = h->ms[1];
    ms3[2] = h->ms[2];
    return NODAL_OK;
}

int nodal_last_kernel_stats(nodal_handle h, double *ms_total, int64_t *launches,
                            double *alg_bytes_or_flops) {
    if (!h) return NODAL_E_INVALID;
    if (ms_total) *ms_total = h->kern_ms;
    if (launches) *launches = h->kern_launches;
    if (alg_bytes_or_flops) *alg_bytes_or_flops = h->kern_alg;
    return NODAL_OK;
}

int nodal_set_option(nodal_handle h, int32_t option, int32_t value) {
    if (!h) return NODAL_E_INVALID;
    if (option == NODAL_OPT_FORCE_PIVOTING) {
        h->force_pivoting = value != 0;
        return NODAL_OK;
    }
    if (option == NODAL_OPT_GEPP_PANEL) {
        h->gepp_panel = value != 0;
        return NODAL_OK;
    }
    if (option == NODAL_OPT_EXTRA_STREAMS) {
        h->extra_streams = value != 0;
        return NODAL_OK;
    }
    if (option == NODAL_OPT_BORROW_TABLE) {
        h->borrow_table = value != 0;
        return NODAL_OK;
    }
    return nodal_fail(h, NODAL_E_INVALID, "unknown option");
}

int nodal_debug_gemm(nodal_handle h, int32_t M, int32_t N, int32_t K, const double *A,
                     const double *B, double *C) {
    if (!h || M < 1 || N < 1 || K < 1 || !A || !B || !C) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    const size_t sa = (size_t)M * K * 8, sb = (size_t)K * N * 8, sc = (size_t)M * N * 8;
    NODAL_HIP_TRY(h, h->work.reserve(sa + sb + sc + 768));
    char *w = h->work.as<char>();
    double *dA = reinterpret_cast<double *>(w);
    double *dB = reinterpret_cast<double *>(w + ((sa + 255) & ~(size_t)255));
    double *dC = reinterpret_cast<double *>(w + ((sa + 255) & ~(size_t)255) + ((sb + 255) & ~(size_t)255));
    NODAL_HIP_TRY(h, hipMemcpyAsync(dA, A, sa, hipMemcpyHostToDevice, h->stream));
    NODAL_HIP_TRY(h, hipMemcpyAsync(dB, B, sb, hipMemcpyHostToDevice, h->stream));
    NODAL_HIP_TRY(h, hipMemcpyAsync(dC, C, sc, hipMemcpyHostToDevice, h->stream));
    NODAL_TRY(gemm_sub_f64(h, h->stream, dC, M, dA, M, dB, K, M, N, K));
    NODAL_HIP_TRY(h, hipMemcpyAsync(C, dC, sc, hipMemcpyDeviceToHost, h->stream));
    NODAL_WAIT_STREAM(h, h->stream);
    return NODAL_OK;
}

int nodal_last_solve_info(nodal_handle h, int32_t *iterations, int32_t *amg_levels,
                          double *relative_residual) {
    if (!h) return NODAL_E_INVALID;
    if (iterations) *iterations = h->last_iterations;
    if (amg_levels) *amg_levels = h->amg_levels;
    if (relative_residual) *relative_residual = h->last_relres;
    return NODAL_OK;
}

int nodal_synchronize(nodal_handle h) {
    if (!h) return NODAL_E_INVALID;
    DeviceGuard g(h);
    if (h->hung) return NODAL_E_HIP;  // (a wait timed out earlier: nodal_last_error still says where)
    NODAL_WAIT_STREAM(h, h->stream);
    return NODAL_OK;
}

}  // extern "C"

// ---- pinned host memory for the component table (include/nodal_hip.h) ----
extern "C" {

int nodal_host_alloc(size_t bytes, void **out) {
    if (!out) return NODAL_E_INVALID;
    *out = nullptr;
    if (bytes == 0) return NODAL_OK;
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return e == hipErrorOutOfMemory ? NODAL_E_NOMEM : NODAL_E_HIP;
    }
    return NODAL_OK;
}

int nodal_host_free(void *p) {
    if (p && hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return NODAL_E_HIP;
    }
    return NODAL_OK;
}

}  // extern "C"
