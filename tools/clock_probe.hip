// What a single small workgroup gets on an otherwise idle MI355X: shader clock (s_memtime cycles against
// the 100 MHz wall clock), dependent fp64 FMA latency, LDS round trip, DPP + readlane reduction, and a
// workgroup barrier -- the ingredients of the latency-bound single-workgroup kernels (gepp_panel,
// gj128_mfma16, k_tail).   hipcc --offload-arch=gfx950 -O3 -o tools/clock_probe tools/clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(double *out, long long *t, int iters) {
    __shared__ double sh[1024];
    const int tid = threadIdx.x;
    sh[tid] = tid;
    __syncthreads();
    double x = out[tid & 7] + 1.0;
    long long c0, c1, w0, w1;
    // 1. dependent FMAs
    w0 = wall_clock64(); c0 = clock64();
    for (int i = 0; i < iters; ++i) x = fma(x, 1.0000001, 0.5);
    c1 = clock64(); w1 = wall_clock64();
    if (tid == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
    // 2. dependent LDS round trips
    int idx = tid;
    c0 = clock64();
    for (int i = 0; i < iters; ++i) idx = (int)sh[idx & 1023] & 1023;
    c1 = clock64();
    if (tid == 0) t[2] = c1 - c0;
    // 3. DPP reduction + readlane
    unsigned v = (unsigned)idx + tid;
    c0 = clock64();
    for (int i = 0; i < iters; ++i) {
        unsigned r = v;
        r = max(r, (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x111, 0xf, 0xf, false));
        r = max(r, (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x112, 0xf, 0xf, false));
        r = max(r, (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x114, 0xf, 0xf, false));
        r = max(r, (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x118, 0xf, 0xf, false));
        r = max(r, (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x142, 0xa, 0xf, false));
        r = max(r, (unsigned)__builtin_amdgcn_update_dpp(0, (int)r, 0x143, 0xc, 0xf, false));
        v = (unsigned)__builtin_amdgcn_readlane((int)r, 63) + (v & 3);
    }
    c1 = clock64();
    if (tid == 0) t[3] = c1 - c0;
    // 4. barriers
    c0 = clock64();
    for (int i = 0; i < iters; ++i) { sh[tid] = x + i; __syncthreads(); x += sh[(tid + 64) & (blockDim.x - 1)]; __syncthreads(); }
    c1 = clock64();
    if (tid == 0) t[4] = c1 - c0;
    // 5. fp64 division
    c0 = clock64();
    for (int i = 0; i < iters; ++i) x = 1.0 / (x + 3.0);
    c1 = clock64();
    if (tid == 0) t[5] = c1 - c0;
    out[tid & 7] = x + v + idx;
}

int main() {
    double *out; long long *t;
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipMallocManaged(&t, 64);
    const int iters = 2000;
    for (int threads : {64, 256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            probe<<<1, threads>>>(out, t, iters);
            hipDeviceSynchronize();
        }
        const double mhz = (double)t[0] / ((double)t[1] / 100.0);  // cycles per us
        printf("%4d threads: shader clock %.0f MHz; per iteration: fma %.1f cycles, lds round trip %.1f, dpp max + readlane %.1f, "
               "2 barriers + lds %.1f, 1/x %.1f cycles\n", threads, mhz, (double)t[0] / iters, (double)t[2] / iters,
               (double)t[3] / iters, (double)t[4] / iters, (double)t[5] / iters);
    }
    return 0;
}
