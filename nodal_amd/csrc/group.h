// Deterministic grouping of (row, col, item) tuples into CSR entries -- the shared
// machinery of the stamping (stamp.hip: items = component stamps) and of the
// multigrid setup (amg.hip: items = fine-matrix entries mapped to aggregates).
//
//   1. an enumerator emits, per item, up to SLOTS tuples (row, col, slot);
//   2. tuples are bucketed by row (integer histogram + exclusive scan) and every
//      row bucket is sorted by key = col << 32 | item << 3 | slot, so the result
//      does not depend on the order in which the atomics filled the bucket;
//   3. runs of equal (row, col) become one entry: `cptr` delimits the run and
//      `contrib` lists item << 3 | slot in ascending (item, slot) order -- the order
//      in which a sequential program would have visited them.
//
// Enumerator concept:
//   struct E { static constexpr int SLOTS; int64_t nitems;
//              template <class F> __device__ void for_each(int64_t item, F f) const; }
//   with f(int slot, int row, int col) called for every tuple of `item`.
#pragma once
#include <type_traits>
#include "ctx.h"

namespace {
namespace grp {

constexpr int TB = 256;
constexpr unsigned MAX_GRID = 4096;

inline unsigned grid_for(int64_t n) {
    int64_t g = (n + TB - 1) / TB;
    if (g < 1) g = 1;
    return (unsigned)(g > MAX_GRID ? MAX_GRID : g);
}

// One atomic per DISTINCT row of an item, not per tuple: a resistor's four stamps fall in two rows,
// so half the atomics (and none of them colliding inside the lane).  `s` is a compile-time
// constant after the enumerator's unrolled loop is inlined: rows[] / cols[] stay in registers.
// `pos` (optional): the value the counting atomic returns IS the tuple's place inside its row's segment -- any
// order will do, the segments are sorted afterwards -- so it is kept (one word per leader slot, slot-major: coalesced) and
// emit_tuples needs no atomics of its own, nor a second array of fill counters.
template <class E>
__global__ __launch_bounds__(TB) void count_rows(E en, uint32_t *__restrict__ rowcount, uint32_t *__restrict__ pos) {
    constexpr int S = E::SLOTS;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < en.nitems;
         i += (int64_t)gridDim.x * TB) {
        int rows[S];
#pragma unroll
        for (int s = 0; s < S; ++s) rows[s] = -1;
        en.for_each(i, [&](int s, int row, int) { rows[s] = row; });
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (rows[s] < 0) continue;
            bool leader = true;
            unsigned cnt = 1;
#pragma unroll
            for (int t = 0; t < S; ++t) {
                if (t < s && rows[t] == rows[s]) leader = false;
                if (t > s && rows[t] == rows[s]) ++cnt;
            }
            if (leader) pos[(int64_t)s * en.nitems + i] = atomicAdd(&rowcount[rows[s]], cnt);
        }
    }
}

// The same count with the workgroup's rows aggregated in LDS first (round 5).  A device-scope returning atomic executes
// at the memory side, one 64-byte request per line a wave-instruction touches, and count_rows ran at 0.72 of the part's
// request rate (profiles/r05_pmc_stamping.txt): 119 requests per 256-item tile of a grid netlist, whose 257 DISTINCT rows
// -- every node is hit by the four resistors around it -- lie in two runs of consecutive rows.  Here the tile's
// (row, count) pairs meet in an LDS table that is direct-mapped by the row's low bits (consecutive rows stay in
// consecutive slots; linear probing on a clash), ONE returning atomic per distinct row goes out, issued slot by slot
// -- i.e. consecutive rows by consecutive lanes: a handful of requests --, and a tuple's place is its row's base + the
// rank the LDS atomic handed it.  A tile with more distinct rows than the table holds sends the overflow directly.
constexpr int COUNT_HT = 1024;
template <class E>
__global__ __launch_bounds__(TB) void count_rows_lds(E en, uint32_t *__restrict__ rowcount, uint32_t *__restrict__ pos) {
    constexpr int S = E::SLOTS;
    __shared__ int hrow[COUNT_HT];
    __shared__ uint32_t hcnt[COUNT_HT];
    __shared__ uint32_t hbase[COUNT_HT];
    for (int64_t i0 = (int64_t)blockIdx.x * TB; i0 < en.nitems; i0 += (int64_t)gridDim.x * TB) {
        for (int t = threadIdx.x; t < COUNT_HT; t += TB) {
            hrow[t] = -1;
            hcnt[t] = 0u;
        }
        __syncthreads();
        const int64_t i = i0 + threadIdx.x;
        int rows[S];
        int slot_of[S];       // >= 0: LDS slot; -1: not a leader / no tuple; -2: sent directly (table full)
        uint32_t rank[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            rows[s] = -1;
            slot_of[s] = -1;
            rank[s] = 0u;
        }
        if (i < en.nitems) en.for_each(i, [&](int s, int row, int) { rows[s] = row; });
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (rows[s] < 0) continue;
            bool leader = true;
            unsigned cnt = 1;
#pragma unroll
            for (int t = 0; t < S; ++t) {
                if (t < s && rows[t] == rows[s]) leader = false;
                if (t > s && rows[t] == rows[s]) ++cnt;
            }
            if (!leader) continue;
            unsigned slot = (unsigned)rows[s] & (COUNT_HT - 1);
            int probes = 0;
            for (; probes < 64; ++probes) {
                const int prev = atomicCAS(&hrow[slot], -1, rows[s]);
                if (prev == -1 || prev == rows[s]) break;
                slot = (slot + 1) & (COUNT_HT - 1);
            }
            if (probes < 64) {
                slot_of[s] = (int)slot;
                rank[s] = atomicAdd(&hcnt[slot], cnt);
            } else {
                slot_of[s] = -2;
                rank[s] = atomicAdd(&rowcount[rows[s]], cnt);
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < COUNT_HT; t += TB)
            if (hrow[t] >= 0) hbase[t] = atomicAdd(&rowcount[hrow[t]], hcnt[t]);
        __syncthreads();
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (slot_of[s] == -1) continue;
            pos[(int64_t)s * en.nitems + i] = slot_of[s] >= 0 ? hbase[slot_of[s]] + rank[s] : rank[s];
        }
        __syncthreads();  // (the next tile clears the table)
    }
}

template <class E>
__global__ __launch_bounds__(TB) void emit_tuples(E en, const uint32_t *__restrict__ rowstart,
                                                  const uint32_t *__restrict__ pos,
                                                  uint64_t *__restrict__ skey) {
    constexpr int S = E::SLOTS;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < en.nitems;
         i += (int64_t)gridDim.x * TB) {
        int rows[S], cols[S];
#pragma unroll
        for (int s = 0; s < S; ++s) rows[s] = -1;
        en.for_each(i, [&](int s, int row, int col) { rows[s] = row; cols[s] = col; });
        uint32_t at[S];  // position of tuple s inside its row's segment
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (rows[s] < 0) continue;
            int first = s;
            unsigned before = 0;
#pragma unroll
            for (int t = S - 1; t >= 0; --t)
                if (t < s && rows[t] == rows[s]) { first = t; ++before; }
            at[s] = first == s ? pos[(int64_t)s * en.nitems + i] : 0u;
#pragma unroll
            for (int t = 0; t < S; ++t)
                if (t == first && t < s) at[s] = at[t] + before;
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (rows[s] < 0) continue;
            const uint32_t p = rowstart[rows[s]] + at[s];
            skey[p] = ((uint64_t)(uint32_t)cols[s] << 32) | ((uint64_t)i << 3) | (uint64_t)s;
        }
    }
}

// ---- per-row sorts -------------------------------------------------------------

template <int N>
__device__ __forceinline__ void sort_network(uint64_t (&k)[N]) {
#pragma unroll
    for (int size = 2; size <= N; size <<= 1) {
#pragma unroll
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ stride;
                if (l > i) {
                    const bool up = (i & size) == 0;
                    const uint64_t lo = k[i] < k[l] ? k[i] : k[l];
                    const uint64_t hi = k[i] < k[l] ? k[l] : k[i];
                    k[i] = up ? lo : hi;
                    k[l] = up ? hi : lo;
                }
            }
        }
    }
}

constexpr int SHORT_MAX = 16;
constexpr int MEDIUM_MAX = 2048;

// A short row (at most 16 tuples) lives in one lane's registers: loaded and sorted by a network.
template <int N>
__device__ __forceinline__ void load_sorted(const uint64_t *__restrict__ seg, int len, uint64_t (&k)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) k[i] = i < len ? seg[i] : ~0ull;
    sort_network<N>(k);
}

// (in place: the multigrid setup sorts the rows of its restriction operator with it)
template <int N>
__device__ __forceinline__ void sort_short_row(uint64_t *seg, int len) {
    uint64_t k[N];
    load_sorted<N>(seg, len, k);
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < len) seg[i] = k[i];
}

template <int N>
__device__ __forceinline__ uint32_t short_row_heads(const uint64_t *__restrict__ seg, int len) {
    uint64_t k[N];
    load_sorted<N>(seg, len, k);
    uint32_t nh = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < len && (i == 0 || (k[i] >> 32) != (k[i - 1] >> 32))) ++nh;
    return nh;
}

// Pass 1 over the ROWS: one lane per row sorts its tuples in registers and counts the runs of equal columns
// -- the row's entries; the sorted keys are NOT written back (pass 2 sorts the row again: cheaper than 8 bytes
// per tuple out and in).  Rows longer than SHORT_MAX are appended to work lists; their sorts leave the count.
__global__ __launch_bounds__(TB) void row_heads_short(const uint32_t *__restrict__ rowstart,
                                                      const uint64_t *__restrict__ skey, int64_t nrows,
                                                      uint32_t *__restrict__ rowheads,
                                                      int32_t *__restrict__ medium_list,
                                                      int32_t *__restrict__ long_list,
                                                      uint32_t *__restrict__ list_counts) {
    // list_counts[5]: rows of two or three entries (a diagonal and one or two neighbours: what lowdeg.hip can
    // eliminate; rows handed to the longer sorts are counted too, their entries are not known here) -- an upper bound
    // the solver reads with the sizes, so that a network without such nodes does not pay a launch and a round trip to
    // learn that there is nothing to eliminate
    uint32_t low = 0;
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r <= nrows;
         r += (int64_t)gridDim.x * TB) {
        if (r == nrows) {  // (the scan runs over nrows + 1 slots: the last one receives the number of entries)
            rowheads[r] = 0;
            continue;
        }
        const uint32_t s = rowstart[r];
        const int len = (int)(rowstart[r + 1] - s);
        uint32_t heads = 2;
        if (len < 2) rowheads[r] = heads = (uint32_t)len;
        else if (len <= 4) rowheads[r] = heads = short_row_heads<4>(skey + s, len);
        else if (len <= 8) rowheads[r] = heads = short_row_heads<8>(skey + s, len);
        else if (len <= SHORT_MAX) rowheads[r] = heads = short_row_heads<16>(skey + s, len);
        else if (len <= MEDIUM_MAX) medium_list[atomicAdd(&list_counts[0], 1u)] = (int32_t)r;
        else long_list[atomicAdd(&list_counts[1], 1u)] = (int32_t)r;
        low += (heads == 2 || heads == 3) ? 1u : 0u;
    }
    __shared__ uint32_t low_total;
    if (threadIdx.x == 0) low_total = 0;
    __syncthreads();
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) low += __shfl_down(low, off, 64);
    if ((threadIdx.x & 63) == 0 && low) atomicAdd(&low_total, low);
    __syncthreads();
    if (threadIdx.x == 0 && low_total) atomicAdd(&list_counts[5], low_total);
}

// one workgroup per listed row, bitonic sort in LDS (len <= MEDIUM_MAX)
__global__ __launch_bounds__(TB) void sort_rows_medium(const uint32_t *__restrict__ rowstart,
                                                       uint64_t *__restrict__ skey,
                                                       const int32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ list_counts,
                                                       uint32_t *__restrict__ rowheads) {
    __shared__ uint64_t buf[MEDIUM_MAX];
    __shared__ uint32_t nheads;
    const uint32_t count = list_counts[0];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        const int32_t r = list[it];
        const uint32_t s = rowstart[r];
        const int len = (int)(rowstart[r + 1] - s);
        int P = 32;
        while (P < len) P <<= 1;
        for (int i = threadIdx.x; i < P; i += TB) buf[i] = i < len ? skey[s + i] : ~0ull;
        __syncthreads();
        for (int size = 2; size <= P; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = threadIdx.x; i < P; i += TB) {
                    const int l = i ^ stride;
                    if (l > i) {
                        const bool up = (i & size) == 0;
                        const uint64_t x = buf[i], y = buf[l];
                        if ((x > y) == up) { buf[i] = y; buf[l] = x; }
                    }
                }
                __syncthreads();
            }
        }
        uint32_t mine = 0;
        for (int i = threadIdx.x; i < len; i += TB) {
            skey[s + i] = buf[i];
            mine += (i == 0 || (buf[i] >> 32) != (buf[i - 1] >> 32)) ? 1u : 0u;
        }
        if (threadIdx.x == 0) nheads = 0;
        __syncthreads();
        if (mine) atomicAdd(&nheads, mine);  // (an integer count: the order of the additions does not matter)
        __syncthreads();
        if (threadIdx.x == 0) rowheads[r] = nheads;
        __syncthreads();
    }
}

// one workgroup per listed row, bitonic sort through a padded global scratch
// region [2*rowstart, 2*rowstart + P): regions of different rows are disjoint
// because P < 2*len.  Only hub nodes with > MEDIUM_MAX stamps get here.
__global__ __launch_bounds__(1024) void sort_rows_long(const uint32_t *__restrict__ rowstart,
                                                       uint64_t *__restrict__ skey,
                                                       uint64_t *__restrict__ scratch,
                                                       const int32_t *__restrict__ list,
                                                       const uint32_t *__restrict__ list_counts,
                                                       uint32_t *__restrict__ rowheads) {
    __shared__ uint32_t nheads;
    const uint32_t count = list_counts[1];
    for (uint32_t it = blockIdx.x; it < count; it += gridDim.x) {
        const int32_t r = list[it];
        const uint32_t s = rowstart[r];
        const int64_t len = (int64_t)rowstart[r + 1] - s;
        int64_t P = 1;
        while (P < len) P <<= 1;
        uint64_t *buf = scratch + 2 * (int64_t)s;
        for (int64_t i = threadIdx.x; i < P; i += 1024) buf[i] = i < len ? skey[s + i] : ~0ull;
        __syncthreads();
        for (int64_t size = 2; size <= P; size <<= 1) {
            for (int64_t stride = size >> 1; stride > 0; stride >>= 1) {
                for (int64_t i = threadIdx.x; i < P; i += 1024) {
                    const int64_t l = i ^ stride;
                    if (l > i) {
                        const bool up = (i & size) == 0;
                        const uint64_t x = buf[i], y = buf[l];
                        if ((x > y) == up) { buf[i] = y; buf[l] = x; }
                    }
                }
                __syncthreads();  // one workgroup: global writes are visible after the barrier
            }
        }
        uint32_t mine = 0;
        for (int64_t i = threadIdx.x; i < len; i += 1024) {
            skey[s + i] = buf[i];
            mine += (i == 0 || (buf[i] >> 32) != (buf[i - 1] >> 32)) ? 1u : 0u;
        }
        if (threadIdx.x == 0) nheads = 0;
        __syncthreads();
        if (mine) atomicAdd(&nheads, mine);
        __syncthreads();
        if (threadIdx.x == 0) rowheads[r] = nheads;
        __syncthreads();
    }
}

// ---- runs of equal (row, col) -> CSR entries -------------------------------------
// Pass 2 over the rows, behind the scan of the rows' entry counts (`eptr`, which IS the CSR row pointer): the
// lane sorts its row again and writes its entries (column, row, first contribution), the contributions in
// sorted order, and the position of the diagonal entry.
template <int N>
__device__ __forceinline__ void fill_short_row(const uint64_t *__restrict__ seg, int len, uint32_t s, int32_t r,
                                               uint32_t e, int32_t *__restrict__ indices,
                                               int32_t *__restrict__ rowidx, int32_t *__restrict__ cptr,
                                               uint32_t *__restrict__ contrib, int32_t &dp) {
    uint64_t k[N];
    load_sorted<N>(seg, len, k);
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (i < len) {
            contrib[s + i] = (uint32_t)k[i];
            if (i == 0 || (k[i] >> 32) != (k[i - 1] >> 32)) {
                const int32_t col = (int32_t)(k[i] >> 32);
                if (indices) indices[e] = col;
                rowidx[e] = r;
                cptr[e] = (int32_t)(s + i);
                if (col == r) dp = (int32_t)e;
                ++e;
            }
        }
}

__global__ __launch_bounds__(TB) void fill_rows_short(const uint32_t *__restrict__ rowstart,
                                                      const uint64_t *__restrict__ skey,
                                                      const uint32_t *__restrict__ eptr, int64_t nrows, int64_t C,
                                                      int32_t *__restrict__ indices, int32_t *__restrict__ rowidx,
                                                      int32_t *__restrict__ cptr, uint32_t *__restrict__ contrib,
                                                      int32_t *__restrict__ diag_pos) {
    for (int64_t r = (int64_t)blockIdx.x * TB + threadIdx.x; r < nrows;
         r += (int64_t)gridDim.x * TB) {
        if (r == 0) cptr[eptr[nrows]] = (int32_t)C;  // (the list of entries ends where the contributions end)
        const uint32_t s = rowstart[r];
        const int len = (int)(rowstart[r + 1] - s);
        if (len > SHORT_MAX) continue;  // (fill_rows_listed)
        const uint32_t e = eptr[r];
        int32_t dp = -1;
        if (len == 1) {
            const uint64_t key = skey[s];
            contrib[s] = (uint32_t)key;
            const int32_t col = (int32_t)(key >> 32);
            if (indices) indices[e] = col;
            rowidx[e] = (int32_t)r;
            cptr[e] = (int32_t)s;
            if (col == (int32_t)r) dp = (int32_t)e;
        } else if (len >= 2) {
            if (len <= 4) fill_short_row<4>(skey + s, len, s, (int32_t)r, e, indices, rowidx, cptr, contrib, dp);
            else if (len <= 8) fill_short_row<8>(skey + s, len, s, (int32_t)r, e, indices, rowidx, cptr, contrib, dp);
            else fill_short_row<16>(skey + s, len, s, (int32_t)r, e, indices, rowidx, cptr, contrib, dp);
        }
        if (diag_pos) diag_pos[r] = dp;
    }
}

// the same for the listed rows (sorted in place by their own kernels): one workgroup per row, TB tuples per
// step, entry numbers by a ballot scan
__global__ __launch_bounds__(TB) void fill_rows_listed(const uint32_t *__restrict__ rowstart,
                                                       const uint64_t *__restrict__ skey,
                                                       const uint32_t *__restrict__ eptr,
                                                       const int32_t *__restrict__ medium_list,
                                                       const int32_t *__restrict__ long_list,
                                                       const uint32_t *__restrict__ list_counts,
                                                       int32_t *__restrict__ indices, int32_t *__restrict__ rowidx,
                                                       int32_t *__restrict__ cptr, uint32_t *__restrict__ contrib,
                                                       int32_t *__restrict__ diag_pos) {
    __shared__ uint32_t wcount[TB / 64];
    __shared__ int32_t dpos;
    const uint32_t nm = list_counts[0], total = nm + list_counts[1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint32_t it = blockIdx.x; it < total; it += gridDim.x) {
        const int32_t r = it < nm ? medium_list[it] : long_list[it - nm];
        const uint32_t s = rowstart[r];
        const int64_t len = (int64_t)rowstart[r + 1] - s;
        uint32_t e0 = eptr[r];
        if (threadIdx.x == 0) dpos = -1;
        __syncthreads();
        for (int64_t base = 0; base < len; base += TB) {
            const int64_t i = base + threadIdx.x;
            const bool in = i < len;
            const uint64_t key = in ? skey[s + i] : 0ull;
            const bool head = in && (i == 0 || (key >> 32) != (skey[s + i - 1] >> 32));
            const unsigned long long hb = __ballot(head);
            if (lane == 0) wcount[wave] = (uint32_t)__popcll(hb);
            __syncthreads();
            uint32_t before = 0, all = 0;
#pragma unroll
            for (int w = 0; w < TB / 64; ++w) {
                if (w < wave) before += wcount[w];
                all += wcount[w];
            }
            if (in) contrib[s + i] = (uint32_t)key;
            if (head) {
                const uint32_t e = e0 + before + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
                const int32_t col = (int32_t)(key >> 32);
                if (indices) indices[e] = col;
                rowidx[e] = r;
                cptr[e] = (int32_t)(s + i);
                if (col == r) dpos = (int32_t)e;  // (one entry per column: one writer)
            }
            e0 += all;
            __syncthreads();
        }
        if (threadIdx.x == 0 && diag_pos) diag_pos[r] = dpos;
        __syncthreads();
    }
}

__global__ __launch_bounds__(TB) void fill_i32(int32_t *p, int32_t v, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < n; i += (int64_t)gridDim.x * TB)
        p[i] = v;
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

// Enumerators that emit exactly SLOTS tuples for every item declare
// `static constexpr bool EXACT = true`: the number of contributions is then known on the
// host and build_lists saves a device round trip (the multigrid setup calls it ~20 times
// per solve; each round trip idles the GPU for ~100 us).
template <class E, class = void>
struct emits_exactly : std::false_type {};
template <class E>
struct emits_exactly<E, std::void_t<decltype(E::EXACT)>> : std::bool_constant<E::EXACT> {};

// Group one family of stamps (matrix or rhs) into entries.  On return:
//   *nent entries, *ncon contributions; rowidx / cptr / contrib filled;
//   indices, indptr, diag_pos filled when non-null.
// `known_nent` / `known_C` >= 0: the caller knows the number of entries / contributions (one entry per
// aggregate; the same table grouped before): no round trip for them.
template <class E>
int build_lists(nodal_ctx *h, const E &en, int64_t nrows, int64_t *nent_out, int64_t *ncon_out,
                DevBuf &indices, DevBuf &rowidx, DevBuf &cptr, DevBuf &contrib, DevBuf *indptr,
                DevBuf *diag_pos, int64_t known_nent = -1, int64_t known_C = -1, int *long_rows = nullptr,
                int64_t *low_rows = nullptr) {
    // low_rows (optional, out): rows of two or three entries, an upper bound (row_heads_short); written only when this
    // grouping reads its sizes back
    // long_rows (optional): in, 0 = the caller knows that no row has more than SHORT_MAX tuples (the same
    // items grouped before): the two sorts of longer rows are not launched; out, what this grouping found
    // (only when it reads the number of entries back anyway), -1 otherwise.
    hipStream_t st = h->stream;
    if (en.nitems >= (1ll << 29) || nrows >= (1ll << 31) - 2)
        return nodal_fail(h, NODAL_E_UNSUPPORTED, "too many items for 32-bit grouping keys");

    // work layout: rowcount/rowstart [nrows+1] | counts[4] | scan tmp | pos [nitems x SLOTS] (leader slots only)
    const size_t off_start = 0;
    const size_t off_counts = align_up((size_t)(nrows + 1) * 4);
    const size_t off_scan = off_counts + 256;
    const size_t scan_bytes = scan_tmp_bytes(nrows + 1);
    const size_t off_pos = off_scan + align_up(scan_bytes);
    NODAL_HIP_TRY(h, h->work.reserve(off_pos + (size_t)en.nitems * E::SLOTS * 4 + 64));
    char *w = h->work.as<char>();
    uint32_t *rowstart = reinterpret_cast<uint32_t *>(w + off_start);
    uint32_t *counts = reinterpret_cast<uint32_t *>(w + off_counts);  // [0] medium [1] long [2] C [3] nent
    uint32_t *pos = reinterpret_cast<uint32_t *>(w + off_pos);
    NODAL_HIP_TRY(h, hipMemsetAsync(w, 0, off_scan, st));

    {   // (NODAL_COUNT_LDS=1: the rows of a tile aggregated in LDS first.  Measured at config 3, round 5: the symbolic phase
        // 0.216 -> 0.241 ms -- the table's clear, three barriers and the LDS atomics of every tile cost more than the
        // memory-side requests they save; off, kept as the cross-check and for hub-heavy tables)
        static const bool lds_count = getenv("NODAL_COUNT_LDS") && atoi(getenv("NODAL_COUNT_LDS")) != 0;
        if (lds_count) count_rows_lds<E><<<grid_for(en.nitems), TB, 0, st>>>(en, rowstart, pos);
        else count_rows<E><<<grid_for(en.nitems), TB, 0, st>>>(en, rowstart, pos);
    }
    NODAL_HIP_TRY(h, hipGetLastError());
    NODAL_TRY(scan_exclusive_u32(h, rowstart, rowstart, nrows + 1, &counts[2], w + off_scan));
    int64_t C;
    if (known_C >= 0) {  // (the caller grouped the very same items before: stamp_symbolic, same table_epoch)
        C = known_C;
    } else if constexpr (emits_exactly<E>::value) {
        C = en.nitems * E::SLOTS;
    } else {
        uint32_t C32 = 0;
        NODAL_TRY(nodal_read_words(h, &C32, &counts[2], 4));
        C = C32;
    }
    if (C > 0x7fffffffll)
        return nodal_fail(h, NODAL_E_UNSUPPORTED, "more than 2^31 stamps");
    *ncon_out = C;
    if (C == 0) {
        *nent_out = 0;
        NODAL_HIP_TRY(h, cptr.reserve(4));
        NODAL_HIP_TRY(h, hipMemsetAsync(cptr.p, 0, 4, st));
        if (indptr) {
            NODAL_HIP_TRY(h, indptr->reserve((size_t)(nrows + 1) * 4));
            NODAL_HIP_TRY(h, hipMemsetAsync(indptr->p, 0, (size_t)(nrows + 1) * 4, st));
        }
        if (diag_pos) {
            NODAL_HIP_TRY(h, diag_pos->reserve((size_t)nrows * 4 + 4));
            fill_i32<<<grid_for(nrows), TB, 0, st>>>(diag_pos->as<int32_t>(), -1, nrows);
        }
        return NODAL_OK;
    }

    // work2 layout: skey [C] u64 | rowheads / eptr [nrows+1] u32 | lists 2 x [nrows] | scan tmp
    const size_t o_key = 0;
    const size_t o_heads = o_key + align_up((size_t)C * 8);
    const size_t o_med = o_heads + align_up((size_t)(nrows + 1) * 4);
    const size_t o_long = o_med + align_up((size_t)nrows * 4);
    const size_t o_scan2 = o_long + align_up((size_t)nrows * 4);
    NODAL_HIP_TRY(h, h->work2.reserve(o_scan2 + scan_tmp_bytes(nrows + 1)));
    char *w2 = h->work2.as<char>();
    uint64_t *skey = reinterpret_cast<uint64_t *>(w2 + o_key);
    uint32_t *rowheads = reinterpret_cast<uint32_t *>(w2 + o_heads);
    int32_t *medium_list = reinterpret_cast<int32_t *>(w2 + o_med);
    int32_t *long_list = reinterpret_cast<int32_t *>(w2 + o_long);
    // the rows' entry counts are scanned into the CSR row pointer itself when the caller wants one
    if (indptr) NODAL_HIP_TRY(h, indptr->reserve((size_t)(nrows + 1) * 4));
    uint32_t *eptr = indptr ? indptr->as<uint32_t>() : rowheads;

    emit_tuples<E><<<grid_for(en.nitems), TB, 0, st>>>(en, rowstart, pos, skey);
    NODAL_HIP_TRY(h, hipGetLastError());
    row_heads_short<<<grid_for(nrows + 1), TB, 0, st>>>(rowstart, skey, nrows, rowheads, medium_list, long_list, counts);
    NODAL_HIP_TRY(h, hipGetLastError());
    // The medium / long row lists are counted on the device (counts[0], counts[1]) and the kernels loop over
    // them with a grid stride: launched unconditionally with a bounded grid (an empty list costs three idle
    // launches, a round trip to learn the counts ~100 us) unless the caller knows that there is no such row.
    const bool no_long_rows = long_rows && *long_rows == 0;
    if (long_rows) *long_rows = no_long_rows ? 0 : -1;
    const unsigned gm = (unsigned)(nrows < 1024 ? (nrows > 0 ? nrows : 1) : 1024);
    if (!no_long_rows) {
        sort_rows_medium<<<gm, TB, 0, st>>>(rowstart, skey, medium_list, counts, rowheads);
        NODAL_HIP_TRY(h, hipGetLastError());
        NODAL_HIP_TRY(h, h->work3.reserve((size_t)C * 16));  // padded scratch of the long sort
        sort_rows_long<<<gm < 128 ? gm : 128, 1024, 0, st>>>(rowstart, skey, h->work3.as<uint64_t>(),
                                                            long_list, counts, rowheads);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    // entries per row -> first entry of every row; slot nrows (zero) receives the number of entries
    NODAL_TRY(scan_exclusive_u32(h, rowheads, eptr, nrows + 1, &counts[3], w2 + o_scan2));
    int64_t nent = known_nent;
    if (nent < 0) {
        uint32_t back[6] = {0, 0, 0, 0, 0, 0};  // medium rows, long rows, contributions, entries, -, rows of 2-3 entries
        NODAL_TRY(nodal_read_words(h, back, counts, 24));
        nent = back[3];
        if (low_rows) *low_rows = back[5];
        if (long_rows && !no_long_rows) *long_rows = (back[0] | back[1]) ? 1 : 0;
    }
    *nent_out = nent;

    NODAL_HIP_TRY(h, indices.reserve((size_t)nent * 4 + 4));
    NODAL_HIP_TRY(h, rowidx.reserve((size_t)nent * 4 + 4));
    NODAL_HIP_TRY(h, cptr.reserve((size_t)(nent + 1) * 4));
    NODAL_HIP_TRY(h, contrib.reserve((size_t)C * 4));
    if (diag_pos) NODAL_HIP_TRY(h, diag_pos->reserve((size_t)nrows * 4 + 4));
    fill_rows_short<<<grid_for(nrows), TB, 0, st>>>(rowstart, skey, eptr, nrows, C, indices.as<int32_t>(),
                                                   rowidx.as<int32_t>(), cptr.as<int32_t>(), contrib.as<uint32_t>(),
                                                   diag_pos ? diag_pos->as<int32_t>() : nullptr);
    NODAL_HIP_TRY(h, hipGetLastError());
    if (!no_long_rows) {
        fill_rows_listed<<<gm, TB, 0, st>>>(rowstart, skey, eptr, medium_list, long_list, counts, indices.as<int32_t>(),
                                            rowidx.as<int32_t>(), cptr.as<int32_t>(), contrib.as<uint32_t>(),
                                            diag_pos ? diag_pos->as<int32_t>() : nullptr);
        NODAL_HIP_TRY(h, hipGetLastError());
    }
    return NODAL_OK;
}

// ---- a handful of tuples --------------------------------------------------------
// The right-hand side of a netlist with one current source is two stamps, and the pipeline above
// spends thirteen launches on them (every one a pass over the rows or a 5-us launch of nothing).  When an
// upper bound of at most FEW_MAX tuples is known beforehand -- the sizes of the same table's last grouping,
// or the number of source components counted at upload -- two launches do: every item's tuples go to a
// small buffer (in the order the atomics hand out, which the sort makes irrelevant), then ONE workgroup
// sorts them by (row, col, item, slot), marks the runs and writes entries and contributions.
constexpr int FEW_MAX = 1024;

template <class E>
__global__ __launch_bounds__(TB) void collect_tuples(E en, uint32_t *__restrict__ count, uint32_t *__restrict__ frow,
                                                     uint64_t *__restrict__ fkey) {
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < en.nitems; i += (int64_t)gridDim.x * TB)
        en.for_each(i, [&](int s, int row, int col) {
            const uint32_t p = atomicAdd(count, 1u);
            if (p < (uint32_t)FEW_MAX) {
                frow[p] = (uint32_t)row;
                fkey[p] = ((uint64_t)(uint32_t)col << 32) | ((uint64_t)i << 3) | (uint64_t)s;
            }
        });
}

// counts: [0] tuples collected (in), [2] contributions, [3] entries (out; [2] > FEW_MAX: the bound was wrong)
__global__ __launch_bounds__(FEW_MAX) void group_few(uint32_t *__restrict__ counts, const uint32_t *__restrict__ frow,
                                                     const uint64_t *__restrict__ fkey, int32_t *__restrict__ indices,
                                                     int32_t *__restrict__ rowidx, int32_t *__restrict__ cptr,
                                                     uint32_t *__restrict__ contrib) {
    __shared__ uint32_t srow_[FEW_MAX];
    __shared__ uint64_t skey_[FEW_MAX];
    __shared__ uint32_t wsum[FEW_MAX / 64];
    const int t = threadIdx.x;
    const uint32_t total = counts[0];
    const uint32_t C = total < (uint32_t)FEW_MAX ? total : (uint32_t)FEW_MAX;
    srow_[t] = (uint32_t)t < C ? frow[t] : 0xffffffffu;
    skey_[t] = (uint32_t)t < C ? fkey[t] : ~0ull;
    __syncthreads();
    int P = 64;  // (the padding sorts behind every tuple: a power of two that holds them all is enough)
    while (P < (int)C) P <<= 1;
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int l = t ^ stride;
            if (l > t && l < P) {
                const uint32_t ra = srow_[t], rb = srow_[l];
                const uint64_t ka = skey_[t], kb = skey_[l];
                const bool a_less = ra < rb || (ra == rb && ka < kb);
                const bool up = (t & size) == 0;
                if (a_less != up) {
                    srow_[t] = rb;
                    srow_[l] = ra;
                    skey_[t] = kb;
                    skey_[l] = ka;
                }
            }
            __syncthreads();
        }
    // runs of equal (row, col) are entries
    const bool valid = (uint32_t)t < C;
    const bool head = valid && (t == 0 || srow_[t] != srow_[t - 1] || (skey_[t] >> 32) != (skey_[t - 1] >> 32));
    const unsigned long long hb = __ballot(head);
    const int lane = t & 63, wave = t >> 6;
    if (lane == 0) wsum[wave] = (uint32_t)__popcll(hb);
    __syncthreads();
    uint32_t before = 0, nent = 0;
    for (int w = 0; w < FEW_MAX / 64; ++w) {
        if (w < wave) before += wsum[w];
        nent += wsum[w];
    }
    const uint32_t e = before + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull));
    if (valid) contrib[t] = (uint32_t)(skey_[t] & 0xffffffffull);
    if (head) {
        rowidx[e] = (int32_t)srow_[t];
        indices[e] = (int32_t)(skey_[t] >> 32);
        cptr[e] = t;
    }
    if (t == 0) {
        cptr[nent] = (int32_t)C;
        counts[2] = total;
        counts[3] = nent;
    }
}

// `bound`: an upper bound (<= FEW_MAX) of the number of tuples the items emit.  Matrix-free outputs only
// (no indptr / diagonal positions: the right-hand side's grouping).
template <class E>
int build_lists_few(nodal_ctx *h, const E &en, int64_t bound, int64_t *nent_out, int64_t *ncon_out, DevBuf &indices,
                    DevBuf &rowidx, DevBuf &cptr, DevBuf &contrib, int64_t known_nent, int64_t known_C) {
    hipStream_t st = h->stream;
    if (en.nitems >= (1ll << 29)) return nodal_fail(h, NODAL_E_UNSUPPORTED, "too many items for 32-bit grouping keys");
    NODAL_HIP_TRY(h, h->work.reserve(256 + (size_t)FEW_MAX * 12));
    uint32_t *counts = h->work.as<uint32_t>();
    uint64_t *fkey = reinterpret_cast<uint64_t *>(h->work.as<char>() + 256);
    uint32_t *frow = reinterpret_cast<uint32_t *>(h->work.as<char>() + 256 + (size_t)FEW_MAX * 8);
    NODAL_HIP_TRY(h, indices.reserve((size_t)(bound + 1) * 4 + 4));
    NODAL_HIP_TRY(h, rowidx.reserve((size_t)(bound + 1) * 4 + 4));
    NODAL_HIP_TRY(h, cptr.reserve((size_t)(bound + 2) * 4));
    NODAL_HIP_TRY(h, contrib.reserve((size_t)(bound + 1) * 4));
    NODAL_HIP_TRY(h, hipMemsetAsync(counts, 0, 64, st));
    collect_tuples<E><<<grid_for(en.nitems), TB, 0, st>>>(en, counts, frow, fkey);
    NODAL_HIP_TRY(h, hipGetLastError());
    group_few<<<1, FEW_MAX, 0, st>>>(counts, frow, fkey, indices.as<int32_t>(), rowidx.as<int32_t>(), cptr.as<int32_t>(),
                                     contrib.as<uint32_t>());
    NODAL_HIP_TRY(h, hipGetLastError());
    if (known_nent >= 0 && known_C >= 0) {
        *nent_out = known_nent;
        *ncon_out = known_C;
        return NODAL_OK;
    }
    uint32_t out[2] = {0, 0};
    NODAL_TRY(nodal_read_words(h, out, counts + 2, 8));
    if ((int64_t)out[0] > bound) return nodal_fail(h, NODAL_E_INVALID, "grouping: more stamps than the table's sources allow");
    *ncon_out = out[0];
    *nent_out = out[1];
    return NODAL_OK;
}

}  // namespace grp
}  // namespace
