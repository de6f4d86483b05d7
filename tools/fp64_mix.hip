// Microbenchmark: can fp64 MFMA and fp64 VALU FMA work be overlapped on gfx950?
//   mode 0: MFMA 16x16x4 only          mode 1: VALU FMA only
//   mode 2: both in the SAME wave (16 MFMA + R x 16 FMA per iteration, independent)
//   mode 3: wave-specialised (even waves MFMA, odd waves FMA), 8 waves per CU
//   mode 4: MFMA 4x4x4 (4 blocks) only
// Build: hipcc --offload-arch=gfx950 -O3 -w tools/fp64_mix.hip -o fp64_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int R>
__global__ __launch_bounds__(256) void both(double *out, int iters) {
    v4f64 acc[8];
    double f[16];
    for (int i = 0; i < 8; ++i) acc[i] = v4f64{0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) f[i] = i;
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-9 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 2 * R; ++r) f[(2 * R * i + r) & 15] = fma(f[(2 * R * i + r) & 15], b, a);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(512) void split(double *out, int iters, int fma_per_mfma) {
    const int wave = threadIdx.x >> 6;
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-9 + 1.0, s = 0;
    if (wave & 1) {
        double f[16];
        for (int i = 0; i < 16; ++i) f[i] = i;
        for (int it = 0; it < iters * fma_per_mfma; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = fma(f[i], b, a);
        }
        for (int i = 0; i < 16; ++i) s += f[i];
    } else {
        v4f64 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = v4f64{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void m444(double *out, int iters) {
    double acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = 0;
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-9 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double *out;
    (void)hipMalloc(&out, 512 * 8 * 2048 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
    const int iters = 10000;
    auto run = [&](const char *name, auto launch, double flops) {
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%-48s %8.2f ms  %6.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
    };
    for (int bpc = 1; bpc <= 2; ++bpc) {
        const int grid = 256 * bpc;
        char nm[96];
        snprintf(nm, sizeof nm, "same wave, 8 MFMA + 16 FMA / iter, %d WG/CU", bpc);
        run(nm, [&] { both<1><<<grid, 256>>>(out, iters); }, (double)grid * 4 * iters * (8 * 2048.0 + 16 * 128.0));
        snprintf(nm, sizeof nm, "same wave, 8 MFMA + 32 FMA / iter, %d WG/CU", bpc);
        run(nm, [&] { both<2><<<grid, 256>>>(out, iters); }, (double)grid * 4 * iters * (8 * 2048.0 + 32 * 128.0));
        snprintf(nm, sizeof nm, "same wave, 8 MFMA + 64 FMA / iter, %d WG/CU", bpc);
        run(nm, [&] { both<4><<<grid, 256>>>(out, iters); }, (double)grid * 4 * iters * (8 * 2048.0 + 64 * 128.0));
        snprintf(nm, sizeof nm, "same wave, 8 MFMA + 128 FMA / iter, %d WG/CU", bpc);
        run(nm, [&] { both<8><<<grid, 256>>>(out, iters); }, (double)grid * 4 * iters * (8 * 2048.0 + 128 * 128.0));
    }
    for (int fpm = 1; fpm <= 16; fpm *= 2) {
        char nm[96];
        snprintf(nm, sizeof nm, "split waves (4 MFMA + 4 FMA waves/CU), FMA x%d", fpm);
        run(nm, [&] { split<<<256, 512>>>(out, iters, fpm); },
            256.0 * 4 * iters * 16 * (2048.0 + fpm * 128.0));
    }
    run("mfma_f64_4x4x4 only, 1 WG/CU", [&] { m444<<<256, 256>>>(out, iters); }, 256.0 * 4 * iters * 16 * 512.0);
    run("mfma_f64_4x4x4 only, 2 WG/CU", [&] { m444<<<512, 256>>>(out, iters); }, 512.0 * 4 * iters * 16 * 512.0);
    return 0;
}
