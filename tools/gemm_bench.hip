// Stand-alone timing of the fp64 trailing-update GEMM (nodal_amd/csrc/gemm_f64.hip).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Inodal_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include "../nodal_amd/csrc/gemm_f64.hip"
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char **argv) {
    const int64_t M = argc > 1 ? atoll(argv[1]) : 8192, N = argc > 2 ? atoll(argv[2]) : 8192,
                  K = argc > 3 ? atoll(argv[3]) : 256;
    const int mode = argc > 4 ? atoi(argv[4]) : GEMM_SUB;
    const int64_t ld = M + K + 32;
    nodal_ctx *h = new nodal_ctx();
    hipStream_t st;
    (void)hipStreamCreate(&st);
    double *buf;
    (void)hipMalloc(&buf, (size_t)ld * (N + K) * 8);
    std::vector<double> host((size_t)ld * (N + K));
    for (size_t i = 0; i < host.size(); ++i) host[i] = (double)((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    (void)hipMemcpy(buf, host.data(), host.size() * 8, hipMemcpyHostToDevice);
    // LU-like placement: C = trailing block, A = column panel left of it, B = row panel above
    double *C = buf + K * ld + K, *A = buf + K, *B = buf + K * ld;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0, st);
        const int iters = 10;
        // mode 3: the symmetric block elimination's bulk update, C -= At^T B on the upper block triangle (At: the K x M
        // panel above C, like B); the flops printed are the executed ones (the upper triangle, diagonal blocks whole)
        for (int i = 0; i < iters; ++i) {
            if (mode == 3) gemm_sub_tn_upper_f64(h, st, C, ld, B, ld, B, ld, M, N, K, 256);
            else gemm_f64(h, st, mode, C, ld, A, ld, B, ld, M, N, K);
        }
        (void)hipEventRecord(e1, st);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double area = mode == 3 ? ((double)M * N - 0.5 * (double)M * ((double)M - 256.0)) : (double)M * N;
        printf("M=%lld N=%lld K=%lld mode %d: %.1f us/launch  %.2f TFLOP/s\n", (long long)M, (long long)N, (long long)K, mode,
               ms * 1e3 / iters, 2.0 * area * K * iters / ms / 1e9);
    }
    // a digest of C after the 30 updates: the same for every kernel variant that keeps the order of the sums
    (void)hipMemcpy(host.data(), buf, host.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long hsh = 1469598103934665603ull;
    for (size_t i = 0; i < host.size(); ++i) {
        unsigned long long b;
        memcpy(&b, &host[i], 8);
        hsh = (hsh ^ b) * 1099511628211ull;
    }
    printf("digest %016llx\n", hsh);
    return 0;
}
