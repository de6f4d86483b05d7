// Round trip "the host learns a word the device just wrote", three ways (MI355X, one stream):
//  (a) hipMemcpyAsync D2H into pinned memory + hipStreamSynchronize          (nodal_read_words, rounds 1-4)
//  (b) the same + hipEventRecord / hipEventSynchronize
//  (c) a one-wavefront kernel copies the words into MAPPED pinned memory, fences, raises a sequence flag there;
//      the host spins on the flag
// each after a 5 us producer kernel, so the wait is a real one.
//   hipcc --offload-arch=gfx950 -O2 tools/sync_latency.hip -o /tmp/sync_latency && /tmp/sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void producer(unsigned long long *w, unsigned long long v, int spin) {
    unsigned long long t = 0;
    for (int i = 0; i < spin; ++i) t += __builtin_amdgcn_s_memtime() & 1;
    if (threadIdx.x == 0) w[0] = v + (t & 0);
}
__global__ void post(const unsigned long long *src, volatile unsigned long long *dst, int n,
                     volatile unsigned long long *flag, unsigned long long seq) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) *flag = seq;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned long long *dev, *pin, *mapped, *mapped_dev;
    CK(hipMalloc(&dev, 256));
    CK(hipHostMalloc(&pin, 256));
    CK(hipHostMalloc(&mapped, 512, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void **)&mapped_dev, mapped, 0));
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int reps = 2000, spin = 400;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    // producer alone (launch + wait), to subtract
    for (int mode = 0; mode < 4; ++mode) {
        double total = 0;
        for (int r = 0; r < reps + 100; ++r) {
            const unsigned long long v = 1000 + r;
            auto t0 = now();
            producer<<<1, 64, 0, st>>>(dev, v, spin);
            if (mode == 0) {
                CK(hipStreamSynchronize(st));
            } else if (mode == 1) {
                CK(hipMemcpyAsync(pin, dev, 24, hipMemcpyDeviceToHost, st));
                CK(hipStreamSynchronize(st));
                if (pin[0] != v) { printf("bad value\n"); return 1; }
            } else if (mode == 2) {
                CK(hipMemcpyAsync(pin, dev, 24, hipMemcpyDeviceToHost, st));
                CK(hipEventRecord(ev, st));
                CK(hipEventSynchronize(ev));
                if (pin[0] != v) { printf("bad value\n"); return 1; }
            } else {
                post<<<1, 64, 0, st>>>(dev, mapped_dev, 3, mapped_dev + 32, v);
                volatile unsigned long long *flag = mapped + 32;
                long spins = 0;
                while (*flag != v) {
                    if (++spins > 200000000) { printf("timeout\n"); return 1; }
                }
                if (((volatile unsigned long long *)mapped)[0] != v) { printf("bad value (mapped)\n"); return 1; }
            }
            auto t1 = now();
            if (r >= 100) total += us(t0, t1);
        }
        const char *names[] = {"producer + hipStreamSynchronize", "producer + memcpy D2H + hipStreamSynchronize",
                               "producer + memcpy D2H + event record/synchronize", "producer + post kernel + host spin on mapped flag"};
        printf("%-55s %7.2f us per round trip\n", names[mode], total / reps);
    }
    return 0;
}
