// Helpers shared by the dense factorisations (dense_lu.hip, block_elim.hip).
#pragma once
#include "ctx.h"

namespace {

inline unsigned blocks_for(int64_t work, int per_block) {
    int64_t b = (work + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (unsigned)b;
}

// HIP-event timing of the bulk GEMM launches (roofline.achieved of bench.py)
struct GemmTimer {  // the counters live in the context, so every translation unit sees one timer
    nodal_ctx *h;
    void reset() {
        h->gt_used = 0;
        h->gt_flops = 0.0;
    }
    int begin(hipStream_t st) {
        while (h->evpool.size() < 2 * (h->gt_used + 1)) {
            hipEvent_t e;
            NODAL_HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventReleaseToDevice));
            h->evpool.push_back(e);
        }
        NODAL_HIP_TRY(h, hipEventRecord(h->evpool[2 * h->gt_used], st));
        return NODAL_OK;
    }
    int end(hipStream_t st, double f) {
        NODAL_HIP_TRY(h, hipEventRecord(h->evpool[2 * h->gt_used + 1], st));
        ++h->gt_used;
        h->gt_flops += f;
        return NODAL_OK;
    }
    hipEvent_t last_end() const { return h->evpool[2 * h->gt_used - 1]; }
    void collect() {
        h->kern_ms = 0;
        for (size_t i = 0; i < h->gt_used; ++i) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->evpool[2 * i], h->evpool[2 * i + 1]) == hipSuccess)
                h->kern_ms += ms;
        }
        h->kern_launches = (int64_t)h->gt_used;
        h->kern_alg = h->gt_used ? h->gt_flops / (double)h->gt_used : 0.0;  // average flops per launch
    }
};

}  // namespace
