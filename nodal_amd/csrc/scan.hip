// Device-wide exclusive prefix sum (uint32), reduce-then-scan over 2048-item
// tiles.  Deterministic (integer adds), no inter-workgroup communication inside
// a launch: tile totals go through a second, recursive scan.
#include "ctx.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}

// exclusive scan of one value per thread across the 256-thread block;
// returns the exclusive prefix, *block_total gets the sum.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *block_total) {
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        uint32_t s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *block_total = tot;
    return base + incl - v;
}

// (`in` / `out` carry no __restrict__: the contract of scan_exclusive_u32 is that out may
// alias in, and every caller scans in place)
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_sums(const uint32_t *in, uint32_t *sums,
                                                               int64_t n) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) s += in[base + i];
    uint32_t tot;
    (void)block_excl_scan(s, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles(const uint32_t *in, uint32_t *out,
                                                           const uint32_t *tile_off, int64_t n,
                                                           uint32_t *total) {
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        s += v[i];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan(s, &tot) + (tile_off ? tile_off[blockIdx.x] : 0u);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) *total = run;
}

// The same with the tile's offset summed by the tile's own workgroup from the tile totals (at most
// SCAN_DIRECT_TILES of them: 32 KB that sit in the L2): no recursive scan of the totals, two launches per scan
// instead of three (up to 2048 tiles) or five.  Integer adds: any order gives the same offsets.
constexpr int64_t SCAN_DIRECT_TILES = 8192;
__global__ __launch_bounds__(SCAN_THREADS) void scan_tiles_direct(const uint32_t *in, uint32_t *out,
                                                                  const uint32_t *__restrict__ sums, int64_t n,
                                                                  uint32_t *total) {
    uint32_t before = 0;
    for (int64_t k = threadIdx.x; k < (int64_t)blockIdx.x; k += SCAN_THREADS) before += sums[k];
    uint32_t tile_off;
    (void)block_excl_scan(before, &tile_off);
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        v[i] = (base + i < n) ? in[base + i] : 0u;
        s += v[i];
    }
    uint32_t tot;
    uint32_t run = block_excl_scan(s, &tot) + tile_off;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) *total = run;
}

int64_t tiles_of(int64_t n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }

}  // namespace

size_t scan_tmp_bytes(int64_t n) {
    size_t total = 0;
    int64_t t = tiles_of(n);
    while (t > 1) {
        total += ((size_t)t * 4 + 255) & ~(size_t)255;
        t = tiles_of(t);
    }
    return total + 256;
}

int scan_exclusive_u32(nodal_ctx *h, const uint32_t *in, uint32_t *out, int64_t n,
                       uint32_t *total_dev, void *tmp) {
    if (n <= 0) {
        if (total_dev) NODAL_HIP_TRY(h, hipMemsetAsync(total_dev, 0, 4, h->stream));
        return NODAL_OK;
    }
    const int64_t t = tiles_of(n);
    if (t == 1) {
        scan_tiles<<<1, SCAN_THREADS, 0, h->stream>>>(in, out, nullptr, n, total_dev);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    }
    uint32_t *sums = reinterpret_cast<uint32_t *>(tmp);
    char *next = reinterpret_cast<char *>(tmp) + (((size_t)t * 4 + 255) & ~(size_t)255);
    scan_tile_sums<<<(unsigned)t, SCAN_THREADS, 0, h->stream>>>(in, sums, n);
    NODAL_HIP_TRY(h, hipGetLastError());
    if (t <= SCAN_DIRECT_TILES) {
        scan_tiles_direct<<<(unsigned)t, SCAN_THREADS, 0, h->stream>>>(in, out, sums, n, total_dev);
        NODAL_HIP_TRY(h, hipGetLastError());
        return NODAL_OK;
    }
    NODAL_TRY(scan_exclusive_u32(h, sums, sums, t, nullptr, next));
    scan_tiles<<<(unsigned)t, SCAN_THREADS, 0, h->stream>>>(in, out, sums, n, total_dev);
    NODAL_HIP_TRY(h, hipGetLastError());
    return NODAL_OK;
}
