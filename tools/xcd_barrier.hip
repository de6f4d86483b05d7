// Microbenchmark: a persistent kernel confined to ONE XCD of MI355X (workgroups b with b % 8 == 0 of a
// grid of 8 G: the dispatcher hands workgroups to the eight XCDs round-robin) exchanging data between
// its workgroups through that XCD's L2, with a barrier made of L2 atomics -- no L2 write-back /
// invalidate as a device-wide barrier (or a kernel boundary) needs.  Measures us per phase and
// verifies every exchanged word.  Every wait is bounded: a participant that waits too long raises
// `abort` and everybody leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct Ctl {
    unsigned count, gen, abort, errors;
    unsigned xcc[64];
};

__device__ __forceinline__ unsigned ld_l2(const unsigned *p) {  // bypasses the CU's vector L1
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__device__ __forceinline__ bool barrier(Ctl *c, unsigned nb, unsigned &gen_local) {
    if (MODE == 0) __threadfence();                       // agent scope: L2 write-back + invalidate
    else __builtin_amdgcn_s_waitcnt(0);                   // stores acknowledged by the L2 (vmcnt = 0 among others)
    __syncthreads();
    __shared__ unsigned ok;
    if (threadIdx.x == 0) {
        ok = 1;
        const unsigned target = gen_local + 1;
        unsigned prev;
        if (MODE == 0) prev = __hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        else prev = __hip_atomic_fetch_add(&c->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (prev == nb - 1) {
            if (MODE == 0) {
                __hip_atomic_store(&c->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&c->gen, target, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_exchange(&c->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_exchange(&c->gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            unsigned spins = 0;
            while (ld_l2(&c->gen) != target) {
                if (++spins > (1u << 22) || ld_l2(&c->abort)) {
                    __hip_atomic_store(&c->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        if (MODE == 0) __threadfence();
    }
    __syncthreads();
    ++gen_local;
    return ok != 0;
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_xcd(Ctl *c, unsigned *data, int iters, int G, int words) {
    if (blockIdx.x % 8 != 0) return;
    const unsigned rank = blockIdx.x / 8;
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        c->xcc[rank] = xcc & 15;
    }
    unsigned gen_local = 0;
    unsigned errors = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned *buf = data + (size_t)(it & 1) * G * words;
        for (int w = threadIdx.x; w < words; w += blockDim.x) buf[rank * words + w] = (unsigned)it * 65537u + rank * 8191u + w;
        if (!barrier<MODE>(c, G, gen_local)) return;
        const unsigned other = (rank + 1 + (unsigned)it) % (unsigned)G;
        for (int w = threadIdx.x; w < words; w += blockDim.x) {
            const unsigned v = MODE == 0 ? buf[other * words + w] : ld_l2(&buf[other * words + w]);
            errors += v != (unsigned)it * 65537u + other * 8191u + w;
        }
    }
    if (errors) atomicAdd(&c->errors, errors);
}

template <int MODE>
void run(const char *name, int G, int words, int iters) {
    Ctl *c;
    unsigned *data;
    (void)hipMalloc(&c, sizeof(Ctl));
    (void)hipMemset(c, 0, sizeof(Ctl));
    (void)hipMalloc(&data, (size_t)2 * G * words * 4);
    (void)hipMemset(data, 0, (size_t)2 * G * words * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(c, 0, 16);
        (void)hipEventRecord(e0);
        k_xcd<MODE><<<8 * G, 1024>>>(c, data, iters, G, words);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    Ctl h;
    (void)hipMemcpy(&h, c, sizeof h, hipMemcpyDeviceToHost);
    unsigned same = 1;
    for (int r = 1; r < G; ++r) same &= h.xcc[r] == h.xcc[0];
    printf("%-28s G=%2d words=%5d: %.2f us/phase, errors %u, abort %u, all on XCC %u: %s\n", name, G, words,
           ms * 1e3 / iters, h.errors, h.abort, h.xcc[0], same ? "yes" : "NO");
    (void)hipFree(c);
    (void)hipFree(data);
}

int main() {
    const int iters = 2000;
    for (int G : {2, 8, 16, 32}) {
        for (int words : {256, 8192}) {
            run<0>("agent fence + atomics", G, words, iters);
            run<1>("L2-local atomics, sc1 loads", G, words, iters);
        }
    }
    return 0;
}
