// Probe: the in-wave inverse of a 16 x 16 pivot block (csrc/block_elim.hip, gj16_in_wave) with its lane
// exchanges as ds_bpermute (the shipped form up to round 5) against readlane / DPP row broadcasts.
// Same arithmetic in the same order: the probe compares BITS and prints wave clocks of both.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gj16_dpp_probe tools/gj16_dpp_probe.hip && /tmp/gj16_dpp_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__device__ __forceinline__ double rcp_f64(double p) {
    double ip = __builtin_amdgcn_rcp(p);
    ip = fma(fma(-p, ip, 1.0), ip, ip);
    return fma(fma(-p, ip, 1.0), ip, ip);
}

#include "../nodal_amd/csrc/gj16_wave.h"

__global__ __launch_bounds__(64) void probe_cycles(const double *__restrict__ P, double *__restrict__ out, int variant,
                                                   long long *__restrict__ clocks) {
    const int lane = threadIdx.x;
    const int r = lane & 15, g = lane >> 4;
    const double *src = P + (size_t)blockIdx.x * 256;
    double a[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a[t] = src[r * 16 + 4 * t + g];
    int32_t info = 0;
    __builtin_amdgcn_s_waitcnt(0);
    const long long t0 = clock64();
    if (variant == 0) gj16_in_wave_bperm(a, lane, &info, 0);
    else gj16_in_wave(a, lane, &info, 0);
    asm volatile("" ::"v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
    const long long t1 = clock64();
#pragma unroll
    for (int t = 0; t < 4; ++t) out[(size_t)blockIdx.x * 256 + r * 16 + 4 * t + g] = a[t];
    if (lane == 0) clocks[blockIdx.x] = t1 - t0;
}

int main() {
    const int nb = 512;
    std::vector<double> P((size_t)nb * 256);
    srand(7);
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double v = (rand() / (double)RAND_MAX - 0.5);
                if (b % 3 == 0 && i == j) v += 4.0;            // dominant diagonal
                if (b % 3 == 1) v = (i == j ? 4.0 : 0.0) - (abs(i - j) == 1 ? 1.0 : 0.0) + 1e-3 * v;  // grid-like
                P[(size_t)b * 256 + i * 16 + j] = v;
            }
    double *dP, *dO0, *dO1;
    long long *dC;
    hipMalloc(&dP, P.size() * 8);
    hipMalloc(&dO0, P.size() * 8);
    hipMalloc(&dO1, P.size() * 8);
    hipMalloc(&dC, nb * 8);
    hipMemcpy(dP, P.data(), P.size() * 8, hipMemcpyHostToDevice);
    std::vector<double> o0(P.size()), o1(P.size());
    std::vector<long long> c0(nb), c1(nb);
    for (int rep = 0; rep < 2; ++rep) {
        probe_cycles<<<nb, 64>>>(dP, dO0, 0, dC);
        hipMemcpy(c0.data(), dC, nb * 8, hipMemcpyDeviceToHost);
        probe_cycles<<<nb, 64>>>(dP, dO1, 1, dC);
        hipMemcpy(c1.data(), dC, nb * 8, hipMemcpyDeviceToHost);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    hipMemcpy(o0.data(), dO0, P.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), dO1, P.size() * 8, hipMemcpyDeviceToHost);
    const int same = memcmp(o0.data(), o1.data(), P.size() * 8) == 0;
    // a residual check of the inverse itself on the first blocks
    double worst = 0;
    for (int b = 0; b < 8; ++b)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0;
                for (int k = 0; k < 16; ++k) s += P[(size_t)b * 256 + i * 16 + k] * o1[(size_t)b * 256 + k * 16 + j];
                const double e = fabs(s - (i == j));
                if (e > worst) worst = e;
            }
    long long s0 = 0, s1 = 0;
    for (int b = 0; b < nb; ++b) { s0 += c0[b]; s1 += c1[b]; }
    printf("bits identical: %s; |P P^-1 - I|max (8 blocks) %.2e; cycles per inverse, one wave per CU slot: bpermute %.0f, readlane/dpp %.0f\n",
           same ? "yes" : "NO", worst, (double)s0 / nb, (double)s1 / nb);
    // one wave alone on the device (the owner's situation when the other waves wait at the barrier)
    probe_cycles<<<1, 64>>>(dP, dO0, 0, dC);
    hipMemcpy(c0.data(), dC, 8, hipMemcpyDeviceToHost);
    probe_cycles<<<1, 64>>>(dP, dO1, 1, dC);
    hipMemcpy(c1.data(), dC, 8, hipMemcpyDeviceToHost);
    printf("one wave alone: bpermute %lld cycles, readlane/dpp %lld cycles\n", c0[0], c1[0]);
    return same ? 0 : 1;
}
