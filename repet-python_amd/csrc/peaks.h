// Shared by the two peak-picking kernels (peaks.hip: one workgroup per row; peaks_wave.hip: one wavefront per row).
#pragma once
#include "common.h"

namespace repet {

struct PeakArgs {
    const float* M; int64_t row0; int n; int64_t pitch; int mode; float min_value; int d; int number;
    int* idx; int idx_pitch; int* count; int dl; int groups; int peak_cap; int64_t shift;
    // near-tie refinement (see below): unit rows the similarities were computed from, or null
    const float* unit; int unit_pitch; float delta; double min_value64; unsigned int* stats;
    int64_t m_stride, idx_stride, cnt_stride, unit_stride;      // batch: blockIdx.y = clip
    // long rows: STAGE 1 workgroups handle one segment [seg * seg_len, (seg+1) * seg_len) of the row each (blockIdx.z)
    // and leave their candidates in cand_*[(row * n_seg + seg) * cand_cap ...]; STAGE 2 ranks a row's candidates
    int seg_len, n_seg, cand_cap;
    float* cand_val; int* cand_idx; int* cand_cnt;
};

constexpr int kAmbCap = 96;    // near-tied elements refined per row; a row with more keeps its fp32 decisions
constexpr int kRivalCap = 96;  // (near-tied element, rival) pairs per row, same fallback

// float64 cosine similarity of two fp32 rows (one wave, result in every lane). The rows are unit vectors up
// to fp32 rounding, so their float64 norms are divided out again: the value then depends on the fp32
// spectra alone, not on how the fp32 Gram kernel accumulated them.
__device__ __forceinline__ void exact_similarity2(const float* __restrict__ x, const float* __restrict__ y0,
                                                  const float* __restrict__ y1, int len4, int lane, double* e0, double* e1) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y04 = reinterpret_cast<const float4*>(y0);
    const float4* y14 = reinterpret_cast<const float4*>(y1);
    double xx = 0.0, xy0 = 0.0, yy0 = 0.0, xy1 = 0.0, yy1 = 0.0;
    for (int k0 = 0; k0 < len4; k0 += 320) {        // 15 loads in flight per lane: one round trip up to 1280 bins
        float4 p[5], q[5], r[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {                // clamped index: lanes past the row end are zeroed when consumed
            const int k = min(k0 + 64 * u + lane, len4 - 1);
            p[u] = x4[k];
            q[u] = y04[k];
            r[u] = y14[k];
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const double live = (k0 + 64 * u + lane < len4) ? 1.0 : 0.0;
            const double p0 = p[u].x * live, p1 = p[u].y * live, p2 = p[u].z * live, p3 = p[u].w * live;
            const double q0 = q[u].x, q1 = q[u].y, q2 = q[u].z, q3 = q[u].w;
            const double r0 = r[u].x, r1 = r[u].y, r2 = r[u].z, r3 = r[u].w;
            xx += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
            xy0 += p0 * q0 + p1 * q1 + p2 * q2 + p3 * q3;
            yy0 += live * (q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
            xy1 += p0 * r0 + p1 * r1 + p2 * r2 + p3 * r3;
            yy1 += live * (r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xx += __shfl_xor(xx, o);
        xy0 += __shfl_xor(xy0, o);
        yy0 += __shfl_xor(yy0, o);
        xy1 += __shfl_xor(xy1, o);
        yy1 += __shfl_xor(yy1, o);
    }
    *e0 = xy0 / sqrt(xx * yy0);
    *e1 = xy1 / sqrt(xx * yy1);
}

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float nan_to_inf(float v) { return (v != v) ? INFINITY : v; }

// the wavefront-per-row kernel (peaks_wave.hip); hipErrorNotSupported when the shape is outside its range
hipError_t launch_local_maxima_wave(const PeakArgs& a, int64_t n_rows, int n_batch, hipStream_t s);

}  // namespace repet
