// Shared by the two peak-picking kernels (peaks.hip: one workgroup per row; peaks_wave.hip: one wavefront per row).
#pragma once
#include "common.h"

namespace repet {

// mode: 0 = rows of a full matrix; 1 = simonline on band[t][l] = sim(t, t + l): element i of row j is frame j - ((j - i) mod n),
// a walk down a diagonal; 2 = the same elements in the look-back band[j][l] = sim(j, j - l): row j is contiguous
struct PeakArgs {
    const float* M; int64_t row0; int n; int64_t pitch; int mode; float min_value; int d; int number;
    int* idx; int idx_pitch; int* count; int dl; int groups; int peak_cap; int64_t shift;
    // near-tie refinement (see below): unit rows the similarities were computed from, or null
    const float* unit; int unit_pitch; float delta; double min_value64; unsigned int* stats;
    const double* unit_norm;       // (nullable) float64 norm of every unit row (same row index as `unit`, batch stride unit_stride / unit_pitch)
    int64_t m_stride, idx_stride, cnt_stride, unit_stride;      // batch: blockIdx.y = clip
    // long rows: STAGE 1 workgroups handle one segment [seg * seg_len, (seg+1) * seg_len) of the row each (blockIdx.z)
    // and leave their candidates in cand_*[(row * n_seg + seg) * cand_cap ...]; STAGE 2 ranks a row's candidates
    int seg_len, n_seg, cand_cap;
    float* cand_val; int* cand_idx; int* cand_cnt;
    // Second level: rows with a float64 verdict on the fp32 spectra closer than delta2, and flat rows, are decided again from
    // float64 spectra. General path (peaks_exact.hip): (row, clip) pairs appended to redo_list (count in stats[4];
    // redo_flag[clip * flag_stride + row] == gen marks a listed row). Fast path of the wavefront kernel (peaks_wave.hip):
    // the row's lists go into records[slot * record_bytes ..], (row, clip) into lite_list (count in stats[12], lite_flag as
    // above), the frames whose float64 unit rows will be needed into frame_list as clip * frame_clip_stride + frame
    // (count in stats[10], frame_flag[...] == gen marks a queued frame). redo_list == nullptr: no second level.
    double delta2; int* redo_list; unsigned int* redo_flag; unsigned int gen; int64_t flag_stride;
    unsigned char* records; int record_bytes; int* lite_list; unsigned int* lite_flag;
    int* frame_list; unsigned int* frame_flag; int64_t frame_clip_stride;
    // (mode 0, peaks_wave.hip) segment records of every row: planes m1 | m2 | arg of seg_pitch entries each, row r at
    // seg + r * 3 * seg_pitch -- the largest value of every aligned run of kSegWidth columns, the largest of its OTHER
    // elements, the offset of the largest (NaN counted as +inf). With them the first pass looks at raw elements of the row
    // only where a window's edge cuts a segment whose maximum lies outside it, and around near-ties.
    const float* seg; int seg_pitch;
};

// diagnostics counters (common.h: kStatShards): the copy of this workgroup
__device__ __forceinline__ void stat_add(unsigned int* stats, int k, unsigned int v) {
    atomicAdd(&stats[kRefineStats * (1 + (int)(blockIdx.x & (kStatShards - 1))) + k], v);
}
__device__ __forceinline__ void stat_max(unsigned int* stats, int k, unsigned int v) {
    atomicMax(&stats[kRefineStats * (1 + (int)(blockIdx.x & (kStatShards - 1))) + k], v);
}
__device__ __forceinline__ void flag_row_for_exact(const PeakArgs& a, int64_t r, int clip) {
    if (!a.redo_list) return;
    if (atomicExch(a.redo_flag + (int64_t)clip * a.flag_stride + r, a.gen) != a.gen) {
        const unsigned int s = atomicAdd(&a.stats[4], 1u);
        a.redo_list[2 * s] = (int)r;
        a.redo_list[2 * s + 1] = clip;
    }
}
// a record slot for the row (fast path); -1: the row has one already, -2: no fast path in this launch
__device__ __forceinline__ int claim_lite_slot(const PeakArgs& a, int64_t r, int clip) {
    if (!a.lite_list) return -2;
    if (atomicExch(a.lite_flag + (int64_t)clip * a.flag_stride + r, a.gen) == a.gen) return -1;
    const unsigned int s = atomicAdd(&a.stats[12], 1u);
    a.lite_list[2 * s] = (int)r;
    a.lite_list[2 * s + 1] = clip;
    return (int)s;
}
__device__ __forceinline__ void enqueue_frame_for_exact(const PeakArgs& a, int clip, int64_t frame) {
    if (!a.frame_list) return;
    const int64_t lin = (int64_t)clip * a.frame_clip_stride + frame;
    if (atomicExch(a.frame_flag + lin, a.gen) != a.gen) a.frame_list[atomicAdd(&a.stats[10], 1u)] = (int)lin;
}

constexpr int kAmbCap = 96;    // near-tied elements refined per row; a row with more keeps its fp32 decisions
constexpr int kRivalCap = 96;  // (near-tied element, rival) pairs per row, same fallback

// Sum of a double over the wave, the same value in every lane, without the LDS: four DPP steps inside the rows of 16
// lanes, then gfx950's v_permlane16_swap / v_permlane32_swap across them. (Five quantities reduced by __shfl_xor were
// 60 ds_bpermute round trips per pair of similarities -- the refinement of a row with forty near-ties took 100 us.)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    int lo = (int)b, hi = (int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += dpp_mov_f64<0xB1>(v);       // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);       // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);      // row_half_mirror
    v += dpp_mov_f64<0x140>(v);      // row_mirror: every lane of a row holds the row's sum
    {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
        const auto l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);     // {rows 0 0 2 2, rows 1 1 3 3}
        const auto h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v = __builtin_bit_cast(double, ((long long)(unsigned)h[0] << 32) | (unsigned)l[0]) +
            __builtin_bit_cast(double, ((long long)(unsigned)h[1] << 32) | (unsigned)l[1]);
    }
    {
        const long long b = __builtin_bit_cast(long long, v);
        const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
        const auto l = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);     // {low half twice, high half twice}
        const auto h = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v = __builtin_bit_cast(double, ((long long)(unsigned)h[0] << 32) | (unsigned)l[0]) +
            __builtin_bit_cast(double, ((long long)(unsigned)h[1] << 32) | (unsigned)l[1]);
    }
    return v;
}

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
    xx = wave_sum_f64(xx);
    xy0 = wave_sum_f64(xy0);
    yy0 = wave_sum_f64(yy0);
    xy1 = wave_sum_f64(xy1);
    yy1 = wave_sum_f64(yy1);
    *e0 = xy0 / sqrt(xx * yy0);
    *e1 = xy1 / sqrt(xx * yy1);
}

// The same similarities for a LIST of rows against one row x (rows of up to 1 280 bins: x stays in registers, the next
// item's row is fetched while the current one is reduced). Values equal exact_similarity2's bit for bit: same per-lane
// sums in the same order, same wave reduction. row_of(i) gives item i's row, store(i, e) takes its similarity (every lane
// calls it with the same e).
template <class RowOf, class Store>
__device__ __forceinline__ void exact_similarity_list(const float* __restrict__ x, int len4, int lane, int n_items, RowOf row_of, Store store) {
    if (len4 > 320) {                                   // longer rows: pairwise, re-reading x
        for (int it = 0; it < n_items; it += 2) {
            const bool two = it + 1 < n_items;
            double e0, e1;
            exact_similarity2(x, row_of(it), row_of(two ? it + 1 : it), len4, lane, &e0, &e1);
            store(it, e0);
            if (two) store(it + 1, e1);
        }
        return;
    }
    // lanes past the row end read a clamped index and are zeroed (x * 0 and y * 0 are what exact_similarity2 adds there)
    auto fetch_row = [&](const float* row, float4 (&dst)[5]) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
        for (int u = 0; u < 5; ++u) dst[u] = r4[min(64 * u + lane, len4 - 1)];
    };
    auto zero_tail = [&](float4 (&v)[5]) {
#pragma unroll
        for (int u = 0; u < 5; ++u)
            if (64 * u + lane >= len4) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // The item rows sit in the Infinity Cache or HBM (a different row of the unit spectra each): one dword per 128-byte
    // line of eight rows at a time pulls them into this XCD's L2 behind ONE round trip, so the row fetches below hit L2
    // instead of costing a far round trip per item (3 000 cycles each, the whole of the slow rows' refinement time).
    {
        float sink = 0.f;
        const int at = min(32 * lane, 4 * len4 - 1);
        for (int it0 = 0; it0 < n_items; it0 += 8) {
            float t[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) t[b] = row_of(min(it0 + b, n_items - 1))[at];
#pragma unroll
            for (int b = 0; b < 8; ++b) sink += t[b];
        }
        asm volatile("" ::"v"(sink));
    }
    float4 p[5], q[5];
    fetch_row(x, p);
    if (n_items > 0) fetch_row(row_of(0), q);
    zero_tail(p);
    double xx = 0.0;
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const double p0 = p[u].x, p1 = p[u].y, p2 = p[u].z, p3 = p[u].w;
        xx += p0 * p0 + p1 * p1 + p2 * p2 + p3 * p3;
    }
    xx = wave_sum_f64(xx);
    for (int it = 0; it < n_items; ++it) {
        float4 r[5];
        fetch_row(row_of(it + 1 < n_items ? it + 1 : it), r);      // in flight during the sums below
        zero_tail(q);
        double xy = 0.0, yy = 0.0;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const double p0 = p[u].x, p1 = p[u].y, p2 = p[u].z, p3 = p[u].w;
            const double q0 = q[u].x, q1 = q[u].y, q2 = q[u].z, q3 = q[u].w;
            xy += p0 * q0 + p1 * q1 + p2 * q2 + p3 * q3;
            yy += q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3;
        }
        xy = wave_sum_f64(xy);
        yy = wave_sum_f64(yy);
        store(it, xy / sqrt(xx * yy));
#pragma unroll
        for (int u = 0; u < 5; ++u) q[u] = r[u];
    }
}

// The same list with the float64 norms of the rows on a table (PeakArgs::unit_norm; round 6): cos = x.y / (|x| |y|) is ONE dot
// product per item -- twenty float64 multiply-adds and one wave reduction where the form above takes forty, two reductions and
// a square root (all at the float64 rate, by one wave: 66 k of the 145 k cycles of the slowest rows of the first pass were
// these similarities). The values differ from the form above in the last bits (the norm is summed in another order): they
// are level-1 values, compared with a tolerance of delta2 = 2.5e-7 before anything is decided from them.
template <class RowOf, class NormOf, class Store>
__device__ __forceinline__ void exact_similarity_list_normed(const float* __restrict__ x, double norm_x, int len4, int lane, int n_items,
                                                             RowOf row_of, NormOf norm_of, Store store) {
    auto fetch_row = [&](const float* row, float4 (&dst)[5]) {
        const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
        for (int u = 0; u < 5; ++u) dst[u] = r4[min(64 * u + lane, len4 - 1)];
    };
    {   // (the rows into this XCD's L2 behind one round trip: see exact_similarity_list)
        float sink = 0.f;
        const int at = min(32 * lane, 4 * len4 - 1);
        for (int it0 = 0; it0 < n_items; it0 += 8) {
            float t[8];
#pragma unroll
            for (int b = 0; b < 8; ++b) t[b] = row_of(min(it0 + b, n_items - 1))[at];
#pragma unroll
            for (int b = 0; b < 8; ++b) sink += t[b];
        }
        asm volatile("" ::"v"(sink));
    }
    float4 p[5], q[5];
    fetch_row(x, p);
    if (n_items > 0) fetch_row(row_of(0), q);
#pragma unroll
    for (int u = 0; u < 5; ++u)
        if (64 * u + lane >= len4) p[u] = make_float4(0.f, 0.f, 0.f, 0.f);      // (x * 0: the clamped tail of y adds nothing)
    double ny = n_items > 0 ? norm_of(0) : 1.0;
    for (int it = 0; it < n_items; ++it) {
        float4 r[5];
        const int nx = it + 1 < n_items ? it + 1 : it;
        fetch_row(row_of(nx), r);                                      // in flight during the sums below
        const double ny_next = norm_of(nx);
        double xy = 0.0;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const double p0 = p[u].x, p1 = p[u].y, p2 = p[u].z, p3 = p[u].w;
            const double q0 = q[u].x, q1 = q[u].y, q2 = q[u].z, q3 = q[u].w;
            xy += p0 * q0 + p1 * q1 + p2 * q2 + p3 * q3;
        }
        xy = wave_sum_f64(xy);
        store(it, xy / (norm_x * ny));
        ny = ny_next;
#pragma unroll
        for (int u = 0; u < 5; ++u) q[u] = r[u];
    }
}

// float64 dot product of two float64 unit rows (one wave, result in every lane; len a multiple of 2)
__device__ __forceinline__ double dot_rows_f64(const double* __restrict__ x, const double* __restrict__ y, int len, int lane) {
    double acc = 0.0;
    const double2* x2 = reinterpret_cast<const double2*>(x);
    const double2* y2 = reinterpret_cast<const double2*>(y);
    const int len2 = len >> 1;
    for (int k0 = 0; k0 < len2; k0 += 512) {                                       // sixteen loads in flight per lane
        double2 p[8], q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = min(k0 + 64 * u + lane, len2 - 1); p[u] = x2[k]; q[u] = y2[k]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double live = (k0 + 64 * u + lane < len2) ? 1.0 : 0.0; acc += live * (p[u].x * q[u].x + p[u].y * q[u].y); }
    }
    return wave_sum_f64(acc);
}

// Level-2 values of a LIST of frames against one frame: float64 dot products of float64 unit rows (peaks_exact.hip), the row
// of `self_frame` kept in registers, the next item's row in flight while the current one is reduced (rows of up to 1 152
// components; longer ones item by item). All stamps are checked first, in one batch: false when a row is not there.
// frame_of(i): frame row of item i; store(i, e): every lane calls it with the same e.
template <class FrameOf, class Store>
__device__ __forceinline__ bool level2_similarity_list(const double* __restrict__ base, const unsigned int* __restrict__ gens, unsigned int gen,
                                                       int64_t self_frame, int FS, int lane, int n_items, FrameOf frame_of, Store store) {
    bool ok = gens[self_frame] == gen;
    for (int it = lane; it < n_items; it += 64) ok = ok && gens[frame_of(it)] == gen;
    if (!__all(ok)) return false;
    const int len2 = FS >> 1;
    const double* self = base + self_frame * (int64_t)FS;
    if (len2 > 576) {
        for (int it = 0; it < n_items; ++it) store(it, dot_rows_f64(self, base + frame_of(it) * (int64_t)FS, FS, lane));
        return true;
    }
    auto fetch = [&](const double* row, double2 (&dst)[9]) {
        const double2* r2 = reinterpret_cast<const double2*>(row);
#pragma unroll
        for (int u = 0; u < 9; ++u) dst[u] = r2[min(64 * u + lane, len2 - 1)];
    };
    double2 p[9], q[9];
    fetch(self, p);
    if (n_items > 0) fetch(base + frame_of(0) * (int64_t)FS, q);
#pragma unroll
    for (int u = 0; u < 9; ++u) if (64 * u + lane >= len2) p[u] = make_double2(0.0, 0.0);
    for (int it = 0; it < n_items; ++it) {
        double2 r[9];
        fetch(base + frame_of(it + 1 < n_items ? it + 1 : it) * (int64_t)FS, r);      // in flight during the sums below
        double acc = 0.0;
#pragma unroll
        for (int u = 0; u < 9; ++u) acc += p[u].x * q[u].x + p[u].y * q[u].y;
        store(it, wave_sum_f64(acc));
#pragma unroll
        for (int u = 0; u < 9; ++u) q[u] = r[u];
    }
    return true;
}

__device__ __forceinline__ float4 max4(float4 a, float4 b) {
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float nan_to_inf(float v) { return (v != v) ? INFINITY : v; }

// the wavefront-per-row kernel (peaks_wave.hip); hipErrorNotSupported when the shape is outside its range
hipError_t launch_local_maxima_wave(const PeakArgs& a, int64_t n_rows, int n_batch, hipStream_t s);
// does launch_local_maxima_wave take this shape (then the fast second level applies), and its record size
bool local_maxima_wave_supported(int n_cols, int d, int* record_bytes);

}  // namespace repet
