// Microbenchmark: cost of a device-wide barrier inside a persistent kernel on MI355X
// (hand-rolled sense-reversing counter vs cooperative_groups grid.sync()).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__device__ __forceinline__ void grid_barrier(unsigned *count, unsigned *gen, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
            __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
}
__global__ void k_manual(unsigned *count, unsigned *gen, int iters, double *data, int n) {
    for (int it = 0; it < iters; ++it) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) data[i] += 1.0;
        grid_barrier(count, gen, gridDim.x);
    }
}
__global__ void k_cg(int iters, double *data, int n) {
    cg::grid_group grid = cg::this_grid();
    for (int it = 0; it < iters; ++it) {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) data[i] += 1.0;
        grid.sync();
    }
}
int main() {
    unsigned *sync;
    double *data;
    (void)hipMalloc(&sync, 256);
    (void)hipMemset(sync, 0, 256);
    (void)hipMalloc(&data, 1 << 24);
    (void)hipMemset(data, 0, 1 << 24);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms;
    const int iters = 2000;
    for (int nb : {32, 64, 128, 256}) {
        for (int n : {16384, 131072}) {
            int it = iters, nn = n;
            unsigned *c = sync, *g = sync + 32;
            void *args[] = {&c, &g, &it, &data, &nn};
            (void)hipEventRecord(e0);
            (void)hipLaunchCooperativeKernel((void *)k_manual, dim3(nb), dim3(256), args, 0, 0);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("manual barrier  %3d WGs n=%6d: %.2f us / phase\n", nb, n, ms * 1e3 / iters);
            void *args2[] = {&it, &data, &nn};
            (void)hipEventRecord(e0);
            (void)hipLaunchCooperativeKernel((void *)k_cg, dim3(nb), dim3(256), args2, 0, 0);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("grid.sync()     %3d WGs n=%6d: %.2f us / phase\n", nb, n, ms * 1e3 / iters);
        }
    }
    return 0;
}
