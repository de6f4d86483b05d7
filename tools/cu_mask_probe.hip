// What does a hipExtStreamCreateWithCUMask mask select on MI355X (8 XCDs x 32 CUs)?
// Launch a long-resident kernel on masked streams and count the distinct (XCC, SE, CU) slots used.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>
__global__ void where(unsigned *out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hwid;
    }
    for (int i = 0; i < 300; ++i) __builtin_amdgcn_s_sleep(64);
}
static void run(const char *name, const uint32_t *mask) {
    hipStream_t st;
    if (hipExtStreamCreateWithCUMask(&st, 8, mask) != hipSuccess) { printf("%s: mask refused\n", name); return; }
    const int nb = 4096;
    unsigned *out;
    std::vector<unsigned> h(2 * nb);
    (void)hipMalloc(&out, 8 * nb);
    where<<<nb, 64, 0, st>>>(out);
    (void)hipMemcpyAsync(h.data(), out, 8 * nb, hipMemcpyDeviceToHost, st);
    (void)hipStreamSynchronize(st);
    std::set<unsigned> slots, xccs;
    for (int i = 0; i < nb; ++i) {
        const unsigned xcc = h[2 * i] & 15, hw = h[2 * i + 1];
        // HW_ID: [11:8] CU id, [7] SH id, [15:13] SE id (gfx9 layout)
        const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        slots.insert(xcc << 16 | se << 8 | sh << 4 | cu);
        xccs.insert(xcc);
    }
    printf("%-26s distinct CUs %3zu on %zu XCCs\n", name, slots.size(), xccs.size());
    (void)hipFree(out);
    (void)hipStreamDestroy(st);
}
int main() {
    uint32_t mask[8];
    char name[64];
    memset(mask, 0xff, sizeof mask);
    run("all 256 bits", mask);
    for (int nbits : {1, 2, 4, 8, 16, 32, 64, 128, 224, 248}) {
        memset(mask, 0, sizeof mask);
        for (int b = 0; b < nbits; ++b) mask[b / 32] |= 1u << (b % 32);
        snprintf(name, sizeof name, "first %d bits", nbits);
        run(name, mask);
    }
    memset(mask, 0, sizeof mask);
    mask[7] = 0xFFFFFFFFu;
    run("bits 224..255", mask);
    memset(mask, 0, sizeof mask);
    for (int b = 0; b < 256; b += 8) mask[b / 32] |= 1u << (b % 32);
    run("every 8th bit (32 bits)", mask);
    return 0;
}
