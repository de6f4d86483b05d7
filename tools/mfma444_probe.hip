// Probe the register layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: one-hot A (lane la) times
// one-hot B (lane lb) -> which lane of D lights up.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int *out) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) out[la * 64 + lb] = lane;
        }
}
int main() {
    int *out, h[4096];
    (void)hipMalloc(&out, 4096 * 4);
    (void)hipMemset(out, 0xff, 4096 * 4);
    probe<<<1, 64>>>(out);
    (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    for (int la = 0; la < 64; ++la) {
        printf("A lane %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h[la * 64 + lb] >= 0) printf(" B%d->D%d", lb, h[la * 64 + lb]);
        printf("\n");
    }
    return 0;
}
