// Probe: is the register panel (csrc/sparse_direct.hip, panel_factor_regs_body*) bound by instruction fetch?  One
// workgroup factors the same 16-column panel REPS times in one launch (the data restored from a copy each time)
// and stamps s_memtime around every pass: the first pass runs the fully unrolled code cold, the later ones from
// the instruction cache.   (build/panel_bodies.inc is cut out of sparse_direct.hip by tools/panel_probe.sh;
// -DPANEL_ONE_BARRIER adds the parked one-barrier / DPP body of tools/experiments/panel_one_barrier_dpp.diff.txt)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../build/panel_bodies.inc"

template <int RPT, int V>
__global__ __launch_bounds__(256) void probe(double *F, const double *F0, int dim, int s, int reps, int32_t *piv,
                                             unsigned long long *stats, long long *stamps) {
    for (int it = 0; it < reps; ++it) {
        for (int i = threadIdx.x; i < dim * PNB; i += 256) F[i] = F0[i];
        __threadfence_block();
        __syncthreads();
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (V == 0) panel_factor_regs_body<RPT>(F, dim, s, 0, PNB, piv, 1e-300, 1e-8, stats);
#ifdef PANEL_ONE_BARRIER  // (with tools/experiments/panel_one_barrier_dpp.diff.txt applied to sparse_direct.hip)
        else panel_factor_regs_body1<RPT>(F, dim, s, 0, PNB, piv, 1e-300, 1e-8, stats);
#endif
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        const long long t1 = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) stamps[it] = t1 - t0;
    }
}

template <int RPT, int V>
void run(int dim, const char *what) {
    const int s = dim / 3, reps = 6;
    std::vector<double> h((size_t)dim * PNB);
    srand(3);
    for (auto &v : h) v = rand() / (double)RAND_MAX - 0.5;
    double *F, *F0; int32_t *piv; unsigned long long *st; long long *stamps;
    hipMalloc(&F, h.size() * 8); hipMalloc(&F0, h.size() * 8); hipMalloc(&piv, 64); hipMalloc(&st, 8); hipMalloc(&stamps, 8 * reps);
    hipMemcpy(F0, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMemset(st, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<RPT, V><<<1, 256>>>(F, F0, dim, s, reps, piv, st, stamps);
    hipEventRecord(e1);
    if (hipDeviceSynchronize() != hipSuccess) { printf("failed\n"); exit(2); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[6]; hipMemcpy(c, stamps, 8 * reps, hipMemcpyDeviceToHost);
    std::vector<double> out(h.size()); hipMemcpy(out.data(), F, h.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long dig = 1469598103934665603ull;
    for (double v : out) { unsigned long long b; memcpy(&b, &v, 8); dig = (dig ^ b) * 1099511628211ull; }
    printf("%-28s dim %4d: s_memtime ticks (100 MHz: x 10 ns) per pass:", what, dim);
    for (int i = 0; i < reps; ++i) printf(" %lld", c[i]);
    printf("   launch %.1f us   digest %016llx\n", ms * 1e3, dig);
}

int main() {
    run<2, 0>(500, "two barriers, shuffles RPT 2");
    run<4, 0>(1000, "two barriers, shuffles RPT 4");
    run<6, 0>(1500, "two barriers, shuffles RPT 6");
#ifdef PANEL_ONE_BARRIER
    run<2, 1>(500, "one barrier, DPP      RPT 2");
    run<4, 1>(1000, "one barrier, DPP      RPT 4");
    run<6, 1>(1500, "one barrier, DPP      RPT 6");
#endif
    return 0;
}
