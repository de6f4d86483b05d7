// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 (operands in registers,
// no memory) and of v_fma_f64, plus the shader clock actually held, to calibrate
// the fp64 roofline on this MI355X.  Build: hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void kmfma(double *out, int iters, unsigned long long *clk) {
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = v4f64{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-3 + 1.0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void kfma(double *out, int iters) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = threadIdx.x * 1e-3, b = blockIdx.x * 1e-9 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], b, a);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    double *out; unsigned long long *clk, hclk[2];
    (void)hipMalloc(&out, 256 * 8 * 256 * 8); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; ++rep)
    for (int bpc = 1; bpc <= 2; ++bpc) {
        int grid = 256 * bpc, iters = 20000;
        (void)hipEventRecord(e0); kmfma<16><<<grid, 256>>>(out, iters, clk); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
        double flops = (double)grid * 4 * iters * 16 * 2048.0;
        printf("mfma_f64_16x16x4 NACC=16 blocks/CU=%d: %.2f ms %.1f TFLOP/s; shader clock %.0f MHz; %.1f cycles/MFMA/wave\n",
               bpc, ms, flops / ms / 1e9, (double)hclk[0] / hclk[1] * 100.0, (double)hclk[0] / (iters * 16.0));
    }
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        int grid = 256 * bpc, iters = 20000;
        (void)hipEventRecord(e0); kfma<16><<<grid, 256>>>(out, iters); (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)grid * 256 * iters * 16 * 2.0;
        printf("v_fma_f64 blocks/CU=%d: %.2f ms %.1f TFLOP/s\n", bpc, ms, flops / ms / 1e9);
    }
    return 0;
}
