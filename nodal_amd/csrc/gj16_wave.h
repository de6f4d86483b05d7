// In-wave inverse of a 16 x 16 pivot block (used by gj128_mfma16 in block_elim.hip; replaces the LAPACK
// getrf / getri pair behind np.linalg.solve, reference nodal/nodal.py:327, for one diagonal block).
//
// The block is held as a[t] = P[lane & 15][4 t + (lane >> 4)]: block Gauss-Jordan with 2 x 2 pivots (eight
// steps, one reciprocal -- of the 2 x 2 determinant -- per step).  With scalar pivots the sixteen dependent
// exchange -> reciprocal -> update chains took 3.4 us, more than everything else in an outer step together.
//
// Two forms with the SAME arithmetic in the same order (bit-identical; tools/gj16_dpp_probe.hip compares them):
//  * gj16_in_wave_bperm: every exchange a wave shuffle (ds_bpermute: 28 per step through the LDS crossbar);
//  * gj16_in_wave: the 2 x 2 pivot block read by v_readlane (its lanes are compile-time constants -> SGPRs),
//    the two pivot rows by DPP row broadcasts (row_newbcast: lane k of each row of 16 lanes to the whole row,
//    no LDS), only the lane's two pivot-column entries (they cross rows of lanes) still by ds_bpermute: 4 per step.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

__device__ __forceinline__ void gj16_in_wave_bperm(double (&a)[4], int lane, int32_t *__restrict__ dinfo, int first) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int k = 0; k < 16; k += 2) {
        const int kj = k >> 2, kl = k & 3;  // columns k, k + 1: register kj of lane groups kl, kl + 1 (compile-time)
        const double p00 = __shfl(a[kj], k + 16 * kl, 64), p01 = __shfl(a[kj], k + 16 * (kl + 1), 64);
        const double p10 = __shfl(a[kj], k + 1 + 16 * kl, 64), p11 = __shfl(a[kj], k + 1 + 16 * (kl + 1), 64);
        const double f0 = __shfl(a[kj], r + 16 * kl, 64), f1 = __shfl(a[kj], r + 16 * (kl + 1), 64);
        double r0[4], r1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            r0[t] = __shfl(a[t], k + 16 * g, 64);
            r1[t] = __shfl(a[t], k + 1 + 16 * g, 64);
        }
        const double det = fma(p00, p11, -p01 * p10);
        if (lane == 0 && !(det != 0.0 && det == det) && *dinfo == 0)
            *dinfo = first + k + ((p00 != 0.0 && p00 == p00) ? 2 : 1);  // (the column a scalar elimination stops at)
        const double id = rcp_f64(det);
        const double i00 = p11 * id, i01 = -p01 * id, i10 = -p10 * id, i11 = p00 * id;
        const double m0 = -fma(f0, i00, f1 * i10), m1 = -fma(f0, i01, f1 * i11);  // -[f0 f1] P2^-1
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool c0 = t == kj && g == kl, c1 = t == kj && g == kl + 1;  // my column is k / k + 1
            const double n0 = fma(i00, r0[t], i01 * r1[t]), n1 = fma(i10, r0[t], i11 * r1[t]);
            const double rowk = c0 ? i00 : (c1 ? i01 : n0), rowk1 = c0 ? i10 : (c1 ? i11 : n1);
            const double other = c0 ? m0 : (c1 ? m1 : fma(m0, r0[t], fma(m1, r1[t], a[t])));
            a[t] = r == k ? rowk : (r == k + 1 ? rowk1 : other);
        }
    }
}

// lane L (compile-time) of the wave, as a wave-uniform value
template <int L>
__device__ __forceinline__ double gj16_readlane(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), L);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), L);
    return __hiloint2double(hi, lo);
}
// lane K (compile-time, 0..15) of every row of 16 lanes, to the whole row: DPP row_newbcast (gfx90a and later)
template <int K>
__device__ __forceinline__ double gj16_rowbcast(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

template <int K>
__device__ __forceinline__ void gj16_step(double (&a)[4], int lane, int32_t *__restrict__ dinfo, int first) {
    constexpr int kj = K >> 2, kl = K & 3;  // columns K, K + 1: register kj of lane groups kl, kl + 1
    const int r = lane & 15, g = lane >> 4;
    // the lane's entries of the two pivot columns cross rows of lanes: the only exchanges left on the crossbar
    const double f0 = __shfl(a[kj], r + 16 * kl, 64), f1 = __shfl(a[kj], r + 16 * (kl + 1), 64);
    const double p00 = gj16_readlane<K + 16 * kl>(a[kj]), p01 = gj16_readlane<K + 16 * (kl + 1)>(a[kj]);
    const double p10 = gj16_readlane<K + 1 + 16 * kl>(a[kj]), p11 = gj16_readlane<K + 1 + 16 * (kl + 1)>(a[kj]);
    double r0[4], r1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        r0[t] = gj16_rowbcast<K>(a[t]);
        r1[t] = gj16_rowbcast<K + 1>(a[t]);
    }
    const double det = fma(p00, p11, -p01 * p10);
    if (lane == 0 && !(det != 0.0 && det == det) && *dinfo == 0)
        *dinfo = first + K + ((p00 != 0.0 && p00 == p00) ? 2 : 1);  // (the column a scalar elimination stops at)
    const double id = rcp_f64(det);
    const double i00 = p11 * id, i01 = -p01 * id, i10 = -p10 * id, i11 = p00 * id;
    const double m0 = -fma(f0, i00, f1 * i10), m1 = -fma(f0, i01, f1 * i11);  // -[f0 f1] P2^-1
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const bool c0 = t == kj && g == kl, c1 = t == kj && g == kl + 1;  // my column is K / K + 1
        const double n0 = fma(i00, r0[t], i01 * r1[t]), n1 = fma(i10, r0[t], i11 * r1[t]);
        const double rowk = c0 ? i00 : (c1 ? i01 : n0), rowk1 = c0 ? i10 : (c1 ? i11 : n1);
        const double other = c0 ? m0 : (c1 ? m1 : fma(m0, r0[t], fma(m1, r1[t], a[t])));
        a[t] = r == K ? rowk : (r == K + 1 ? rowk1 : other);
    }
}

__device__ __forceinline__ void gj16_in_wave(double (&a)[4], int lane, int32_t *__restrict__ dinfo, int first) {
    gj16_step<0>(a, lane, dinfo, first);
    gj16_step<2>(a, lane, dinfo, first);
    gj16_step<4>(a, lane, dinfo, first);
    gj16_step<6>(a, lane, dinfo, first);
    gj16_step<8>(a, lane, dinfo, first);
    gj16_step<10>(a, lane, dinfo, first);
    gj16_step<12>(a, lane, dinfo, first);
    gj16_step<14>(a, lane, dinfo, first);
}
