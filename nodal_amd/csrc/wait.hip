// Bounded host waits and the launch log behind their diagnostics (round 5).
//
// Every place where the host waits for the device goes through nodal_wait_stream / nodal_wait_event: they poll
// hipStreamQuery / hipEventQuery against a wall-clock bound (NODAL_WAIT_TIMEOUT_S, default 60 s; 0 = no bound,
// the runtime's own blocking wait) instead of sitting in hipStreamSynchronize / hipEventSynchronize for ever.  A
// wait that runs into the bound returns NODAL_E_HIP, nodal_last_error names the wait site, how long it waited,
// and the last kernel this thread enqueued on that stream (and on any stream), and the handle is marked hung:
// every later call on it fails at once, nodal_destroy does not free memory that kernels may still be using.
//
// The launch log: the library is linked with --wrap=hipLaunchKernel / hipExtLaunchKernel (Makefile), so that
// every `kernel<<<...>>>` of every translation unit passes through the two functions below, which note
// (function, stream, grid) in a small per-thread ring before calling the runtime.  A name is looked up
// (hipKernelNameRefByPtr) only when a diagnostic is written.  Cost per launch: a handful of stores.
#include <sched.h>
#include <time.h>

#include "ctx.h"

namespace {

struct LaunchNote {
    const void *func = nullptr;
    hipStream_t stream = nullptr;
    unsigned gx = 0, gy = 0, bx = 0;
    unsigned long long seq = 0;
};
constexpr int RING = 16;
thread_local LaunchNote t_ring[RING];
thread_local unsigned long long t_seq = 0;

inline void note_launch(const void *f, dim3 g, dim3 b, hipStream_t st) {
    LaunchNote &n = t_ring[t_seq % RING];
    n.func = f;
    n.stream = st;
    n.gx = g.x;
    n.gy = g.y;
    n.bx = b.x;
    n.seq = ++t_seq;
}

double now_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

double wait_bound_s() {
    static const double bound = [] {
        const char *e = getenv("NODAL_WAIT_TIMEOUT_S");
        if (!e) return 60.0;
        const double v = atof(e);
        return v > 0.0 ? v : 0.0;
    }();
    return bound;
}

void describe(const LaunchNote *n, hipStream_t st, char *out, size_t cap) {
    if (!n) {
        snprintf(out, cap, "none in this thread's last %d launches", RING);
        return;
    }
    const char *name = hipKernelNameRefByPtr(n->func, st);
    (void)hipGetLastError();
    snprintf(out, cap, "%s (grid %u x %u, block %u, launch #%llu of this thread)", name ? name : "?", n->gx, n->gy, n->bx,
             n->seq);
}

int timed_out(nodal_ctx *h, hipStream_t st, const char *what, const char *site, double waited) {
    const LaunchNote *on_stream = nullptr, *any = nullptr;
    for (int k = 0; k < RING; ++k) {
        const LaunchNote &n = t_ring[k];
        if (!n.seq) continue;
        if (!any || n.seq > any->seq) any = &n;
        if (st && n.stream == st && (!on_stream || n.seq > on_stream->seq)) on_stream = &n;
    }
    char a[384], b[384], msg[1024];
    describe(on_stream, st, a, sizeof a);
    describe(any, any ? any->stream : st, b, sizeof b);
    snprintf(msg, sizeof msg,
             "wait for %s %p timed out after %.1f s at %s (NODAL_WAIT_TIMEOUT_S); last kernel enqueued on it: %s; last "
             "kernel enqueued by this thread: %s on stream %p",
             what, (void *)st, waited, site, a, b, any ? (void *)any->stream : nullptr);
    nodal_ctx *owner = h->stream_owner ? h->stream_owner : h;
    owner->hung = true;
    h->hung = true;
    h->err = msg;
    if (owner != h) owner->err = msg;
    fprintf(stderr, "[nodal] %s\n", msg);
    return NODAL_E_HIP;
}

// q: the query; returns hipSuccess / hipErrorNotReady / an error
template <class Query>
int bounded_wait(nodal_ctx *h, hipStream_t st, const char *what, const char *site, Query q) {
    const double bound = wait_bound_s();
    double t0 = 0.0;
    for (unsigned spins = 0;; ++spins) {
        const hipError_t e = q();
        if (e == hipSuccess) return NODAL_OK;
        if (e != hipErrorNotReady) {
            char buf[512];
            snprintf(buf, sizeof buf, "wait for %s failed at %s: %s", what, site, hipGetErrorString(e));
            h->err = buf;
            (void)hipGetLastError();
            return e == hipErrorOutOfMemory ? NODAL_E_NOMEM : NODAL_E_HIP;
        }
        (void)hipGetLastError();  // (hipErrorNotReady is sticky for hipGetLastError: later launch checks must not see it)
        if ((spins & 31) != 31) continue;
        const double t = now_s();
        if (t0 == 0.0) t0 = t;
        const double waited = t - t0;
        if (waited > bound) return timed_out(h, st, what, site, waited);
        if (waited > 2e-3) sched_yield();  // a long wait (a dense factorisation): leave the core to others between looks
    }
}

}  // namespace

int nodal_wait_stream(nodal_ctx *h, hipStream_t st, const char *site) {
    if (wait_bound_s() == 0.0) {
        const hipError_t e = hipStreamSynchronize(st);
        if (e == hipSuccess) return NODAL_OK;
        char buf[512];
        snprintf(buf, sizeof buf, "hipStreamSynchronize failed at %s: %s", site, hipGetErrorString(e));
        h->err = buf;
        return NODAL_E_HIP;
    }
    return bounded_wait(h, st, "stream", site, [st] { return hipStreamQuery(st); });
}

int nodal_wait_event(nodal_ctx *h, hipEvent_t ev, hipStream_t recorded_on, const char *site) {
    if (wait_bound_s() == 0.0) {
        const hipError_t e = hipEventSynchronize(ev);
        if (e == hipSuccess) return NODAL_OK;
        char buf[512];
        snprintf(buf, sizeof buf, "hipEventSynchronize failed at %s: %s", site, hipGetErrorString(e));
        h->err = buf;
        return NODAL_E_HIP;
    }
    return bounded_wait(h, recorded_on, "event on stream", site, [ev] { return hipEventQuery(ev); });
}

// ---- the launch log (see the head of this file; the linker routes every launch of the library here) ----
extern "C" {

hipError_t __real_hipLaunchKernel(const void *f, dim3 g, dim3 b, void **args, size_t shmem, hipStream_t st);
hipError_t __real_hipExtLaunchKernel(const void *f, dim3 g, dim3 b, void **args, size_t shmem, hipStream_t st,
                                     hipEvent_t e0, hipEvent_t e1, int flags);

hipError_t __wrap_hipLaunchKernel(const void *f, dim3 g, dim3 b, void **args, size_t shmem, hipStream_t st) {
    note_launch(f, g, b, st);
    return __real_hipLaunchKernel(f, g, b, args, shmem, st);
}

hipError_t __wrap_hipExtLaunchKernel(const void *f, dim3 g, dim3 b, void **args, size_t shmem, hipStream_t st,
                                     hipEvent_t e0, hipEvent_t e1, int flags) {
    note_launch(f, g, b, st);
    return __real_hipExtLaunchKernel(f, g, b, args, shmem, st, e0, e1, flags);
}

}  // extern "C"

unsigned long long nodal_launches_noted() { return t_seq; }
