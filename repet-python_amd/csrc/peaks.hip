// K4 / K4b: peak picking of REPET-SIM on gfx950 -- replaces _localmaxima / _indices
// (repet.py:1294-1383), 54 % of the reference's run time as a pure-Python loop.
//
// One 256-thread workgroup per row of the similarity matrix. Element i survives iff
//   v[i] >= min_value,  v[i] > every v[max(i-d,0) : i],  v[i] > every v[i+1 : min(i+d+1,n)]
// (strict, window clipped at the ends, no wrap; a NaN never wins and blocks its neighbours).
//
// The two window maxima come from a doubling ("sparse table") pass instead of per-element scans:
// the row is staged once in LDS between two runs of d "-inf" pads (NaN stored as +inf), then
// log2(w) in-place passes M[p] = max(M[p], M[p+step]) leave M[p] = max(v[p .. p+w-1]) with
// w = 2^floor(log2 d) > d/2, so each clipped window is the union of two overlapping w-windows:
//   left  = max(M[p-d], M[p-w]),  right = max(M[p+1], M[p+d-w+1]).
// Everything moves in float4: 16-byte global loads and ds_read/write_b128, the first two doubling steps
// are done in registers from a thread's own group and its right neighbour, and the later steps are
// aligned float4 shifts. Every thread keeps its own elements in registers, so the final test costs
// 4 LDS reads per element and there is no data-dependent loop.
// Survivors are compacted with one LDS atomic per wave and ranked by counting (value descending,
// higher index first on exact ties) so the top `number` land in idx[row][0..count) already sorted.
//
// Near-tie refinement (PeakRefine): the rows are fp32 similarities whose rounding error (~1e-6) is larger than
// the margins repeating music produces, so a decision within `delta` of a tie -- against the window maximum,
// the threshold or the top-`number` cut -- is not taken from the fp32 value. The few elements involved are
// recomputed as float64 dot products of the fp32 unit rows (one wavefront per pair of elements) and decided
// from those; the index lists then equal the float64 reference's (tools/refine_probe.py).
#include "peaks.h"

#include <cstdlib>
#include <type_traits>

namespace repet {

#ifdef REPET_PEAK_STAMPS
__device__ unsigned long long g_peak_stamps[8 * 64];
#define STAMP(k) if (threadIdx.x == 0 && (blockIdx.x % 997) == 5 && blockIdx.x / 997 < 8) g_peak_stamps[(blockIdx.x / 997) * 8 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define STAMP(k)
#endif

__device__ __forceinline__ int wave_prefix_slot(bool flag, int* counter, int lane) {
    // returns the list slot of this lane if flag, using one LDS atomic per wave
    const unsigned long long ballot = __ballot(flag);
    int base = 0;
    if (lane == 0 && ballot) base = atomicAdd(counter, __popcll(ballot));
    base = __shfl(base, 0);
    return base + __popcll(ballot & ((1ull << lane) - 1ull));
}

// The padded row lives in LDS as `groups` float4: dl = round_up(d,4) "-inf" pads, the n values, then
// "-inf" up to the end (at least d + 4 of them). Thread `tid` owns groups tid + 256*q, q < QMAX, and
// keeps their original values in registers.
// STAGE 0: the whole row in one workgroup. STAGE 1 / 2: rows too long for that (one row of a 10-minute clip is 100 KB
// of LDS, i.e. one workgroup per CU and 11 ms for the matrix) are cut into segments with a halo of d elements on both
// sides: stage 1 finds (and refines) the strict maxima of one segment per workgroup, stage 2 ranks them per row.
template <int QMAX, int STAGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QMAX <= 8 ? 4 : (QMAX <= 16 ? 2 : 1), QMAX <= 16 ? 8 : 1))) void local_maxima_kernel(PeakArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4* M4 = reinterpret_cast<float4*>(smem);              // groups float4 (row, then window maxima)
    float* pval = smem + 4 * a.groups;                         // peak_cap (multiple of 4)
    int* pidx = reinterpret_cast<int*>(pval + a.peak_cap);     // peak_cap
    __shared__ int n_peak, n_amb, n_riv, n_unl, n_close;
    __shared__ float cutv[2];
    __shared__ double amb_exact[kAmbCap], riv_exact[kRivalCap];
    __shared__ unsigned long long amb_best[kAmbCap];       // best rival of a near-tied element (bit pattern of a double >= 0)
    __shared__ int amb_idx[kAmbCap], riv_idx[kRivalCap];
    __shared__ float amb_val[kAmbCap];
    __shared__ short riv_owner[kRivalCap], riv_ref[kRivalCap], unl_list[kRivalCap];
    __shared__ unsigned char amb_ok[kAmbCap], amb_lose[kAmbCap];

    const int tid = threadIdx.x, lane = tid & 63;
    const int n = a.n, d = a.d, dl = a.dl;
    const int groups = (STAGE == 2) ? 0 : a.groups;      // stage 2 has no row: every row loop below is empty
    a.M += blockIdx.y * a.m_stride;
    a.idx += blockIdx.y * a.idx_stride;
    a.count += blockIdx.y * a.cnt_stride;
    if (a.unit) a.unit += blockIdx.y * a.unit_stride;
    const int64_t r = blockIdx.x;           // row within this launch
    const int64_t j = a.row0 + r;           // absolute row (mode 1: current frame)
    if (tid == 0) { n_peak = 0; n_amb = 0; n_riv = 0; n_unl = 0; n_close = 0; }
    float dlt = a.delta;                    // 0: no refinement
    const int seg = (STAGE == 1) ? (int)blockIdx.z : 0;
    const int seg_lo = (STAGE == 1) ? seg * a.seg_len : 0;                                  // elements tested here:
    const int seg_hi = (STAGE == 1) ? (seg_lo + a.seg_len < n ? seg_lo + a.seg_len : n) : n;   // [seg_lo, seg_hi)
    STAMP(0)

    auto fetch = [&](int i) -> float {      // element i of the row, -inf outside [0, n)
        if (i < 0 || i >= n) return -INFINITY;
        if (a.mode == 0) return nan_to_inf(a.M[j * a.pitch + i]);
        // circular-buffer order of the online variant: column c holds frame j - ((j - c) mod B)
        int l = (int)(j - i) % n;
        if (l < 0) l += n;
        return nan_to_inf(a.M[(a.mode == 2 ? j - a.shift : j - l - a.shift) * a.pitch + l]);     // mode 2: the look-back band
    };

    const bool vec_ok = (a.mode == 0) && ((a.pitch & 3) == 0);
    const float* src = a.M + j * a.pitch;
    float4 own[QMAX];
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int g = tid + 256 * q;
        if (g < groups) {
            const int i0 = seg_lo + 4 * g - dl;       // LDS group g holds elements i0 .. i0+3 (the halo included)
            float4 v;
            if (vec_ok && i0 >= 0 && i0 + 3 < n) {
                v = *reinterpret_cast<const float4*>(src + i0);
                v = make_float4(nan_to_inf(v.x), nan_to_inf(v.y), nan_to_inf(v.z), nan_to_inf(v.w));
            } else {
                v = make_float4(fetch(i0), fetch(i0 + 1), fetch(i0 + 2), fetch(i0 + 3));
            }
            own[q] = v;
            M4[g] = v;
        }
    }
    __syncthreads();
    STAMP(1)

    int w = 1;
    while (2 * w <= d) w *= 2;              // w = 2^floor(log2 d) (1 when d <= 1)
    const float4 ninf = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (w >= 2) {
        // steps 1 (and 2) in registers: windows of min(w,4) starting at each of the thread's elements
        float4 tmp[QMAX];
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            const int g = tid + 256 * q;
            if (g < groups) {
                const float4 x = own[q];
                const float4 y = (g + 1 < groups) ? M4[g + 1] : ninf;
                const float p01 = fmaxf(x.x, x.y), p12 = fmaxf(x.y, x.z), p23 = fmaxf(x.z, x.w), p34 = fmaxf(x.w, y.x);
                if (w == 2) tmp[q] = make_float4(p01, p12, p23, p34);
                else {
                    const float p45 = fmaxf(y.x, y.y), p56 = fmaxf(y.y, y.z);
                    tmp[q] = make_float4(fmaxf(p01, p23), fmaxf(p12, p34), fmaxf(p23, p45), fmaxf(p34, p56));
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < QMAX; ++q) {
            const int g = tid + 256 * q;
            if (g < groups) M4[g] = tmp[q];
        }
        __syncthreads();
        for (int gs = 1; 4 * gs < w; gs *= 2) {     // steps 4, 8, ...: aligned float4 shifts
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int g = tid + 256 * q;
                if (g < groups) tmp[q] = (g + gs < groups) ? max4(M4[g], M4[g + gs]) : M4[g];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < QMAX; ++q) {
                const int g = tid + 256 * q;
                if (g < groups) M4[g] = tmp[q];
            }
            __syncthreads();
        }
    }

    STAMP(2)
    // strict local-maximum test; survivors are compacted with one LDS atomic per wave and element slot.
    // The four window maxima of a thread's 4 consecutive elements are fetched as aligned float4 (two per
    // unaligned offset, shifted in registers): ds_read_b32 at a 16-byte lane stride would be 4-way conflicted.
    auto read4 = [&](int g, int off) -> float4 {              // M[4g+off .. 4g+off+3]
        const int q = off >> 2, r = off & 3;                  // floor division, wave-uniform remainder
        const float4 x = M4[g + q];
        if (r == 0) return x;
        const float4 y = M4[g + q + 1];
        if (r == 1) return make_float4(x.y, x.z, x.w, y.x);
        if (r == 2) return make_float4(x.z, x.w, y.x, y.y);
        return make_float4(x.w, y.x, y.y, y.z);
    };
    for (;;) {
#pragma unroll
    for (int q = 0; q < QMAX; ++q) {
        const int g = tid + 256 * q;
        if (256 * q < groups) {             // wave-uniform guard
            const int i0 = seg_lo + 4 * g - dl;   // dl, seg_lo are multiples of 4: a group is all halo/pad or starts on a tested element
            const bool real = (g < groups) && i0 >= seg_lo && i0 < seg_hi;
            float4 left = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY), right = left;
            if (real && d > 0) {
                left = max4(read4(g, -d), read4(g, -w));
                right = max4(read4(g, 1), read4(g, d - w + 1));
            }
            const float vals[4] = {own[q].x, own[q].y, own[q].z, own[q].w};
            const float lefts[4] = {left.x, left.y, left.z, left.w};
            const float rights[4] = {right.x, right.y, right.z, right.w};
            bool oks[4], nears[4], was_ok[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = vals[e];
                const bool valid = real && i0 + e < seg_hi && (v < INFINITY);
                oks[e] = valid && (v >= a.min_value) && (v > lefts[e]) && (v > rights[e]);
                nears[e] = false;
                was_ok[e] = oks[e];
                if (dlt > 0.0f) {
                    // decisions that an fp32 rounding error of the Gram kernel could flip are taken out of the
                    // fp32 path here and settled in float64 below
                    const float m = fmaxf(lefts[e], rights[e]);
                    const bool sure_yes = (v >= a.min_value + dlt) && (v > m + dlt);
                    const bool sure_no = (v < a.min_value - dlt) || (v < m - dlt);
                    nears[e] = valid && !sure_yes && !sure_no;
                    oks[e] = oks[e] && !nears[e];
                }
            }
            if (dlt > 0.0f && __any(nears[0] || nears[1] || nears[2] || nears[3])) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (__any(nears[e])) {
                        const int slot = wave_prefix_slot(nears[e], &n_amb, lane);
                        if (nears[e] && slot < kAmbCap) { amb_idx[slot] = i0 + e; amb_val[slot] = vals[e]; amb_ok[slot] = was_ok[e]; }
                    }
                }
            }
            if (d >= 3) {
                // peaks are more than d >= 3 apart: at most one of the 4 consecutive elements survives,
                // so one compaction per group (8 per row and thread) instead of one per element
                const bool ok = oks[0] || oks[1] || oks[2] || oks[3];
                const int e = oks[0] ? 0 : (oks[1] ? 1 : (oks[2] ? 2 : 3));
                if (__any(ok)) {
                    const int slot = wave_prefix_slot(ok, &n_peak, lane);
                    const float vsel = oks[0] ? vals[0] : (oks[1] ? vals[1] : (oks[2] ? vals[2] : vals[3]));
                    if (ok && slot < a.peak_cap) { pval[slot] = vsel; pidx[slot] = i0 + e; }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (__any(oks[e])) {
                        const int slot = wave_prefix_slot(oks[e], &n_peak, lane);
                        if (oks[e] && slot < a.peak_cap) { pval[slot] = vals[e]; pidx[slot] = i0 + e; }
                    }
                }
            }
        }
    }
    __syncthreads();
        bool redo = false;
        if (dlt > 0.0f && n_amb > 0) {
            if (n_amb > kAmbCap) redo = true;       // a flat row: more near-ties than the list holds
            else {
                // Rivals of the near-tied elements, found from the values every thread still holds in registers:
                // element k is a rival of near-tied i when it lies in i's window and within delta below it
                // (nothing in the window is more than delta above i, or i would have been a safe "no").
                const int n_near = n_amb;
                for (int k = tid; k < n_near; k += 256) amb_lose[k] = 0;
                for (int s = 0; s < n_near; ++s) {
                    const int i = amb_idx[s];
                    // this thread's groups inside the window: one when the window spans fewer than 256 groups
                    const int g_lo = (i - seg_lo - d + dl) >> 2, g_hi = (i - seg_lo + d + dl) >> 2;
                    const float lim = amb_val[s] - dlt;
                    for (int g = g_lo + ((tid - g_lo) & 255); g <= g_hi && g < groups; g += 256) {
                        const int qsel = g >> 8;
                        float4 v = own[0];
#pragma unroll
                        for (int q = 1; q < QMAX; ++q) if (qsel == q) v = own[q];
                        const float vals[4] = {v.x, v.y, v.z, v.w};
                        const int i0 = seg_lo + 4 * g - dl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = i0 + e;
                            if (k != i && k >= i - d && k <= i + d && k >= 0 && k < n && vals[e] >= lim) {
                                const int entry = atomicAdd(&n_riv, 1);
                                if (entry < kRivalCap) {
                                    int ref = -1;
                                    for (int t = 0; t < n_near; ++t) if (amb_idx[t] == k) ref = t;
                                    riv_owner[entry] = (short)s; riv_ref[entry] = (short)ref; riv_idx[entry] = k;
                                    if (ref < 0) unl_list[atomicAdd(&n_unl, 1)] = (short)entry;
                                }
                            }
                        }
                    }
                }
                __syncthreads();
                if (n_riv > kRivalCap) redo = true;
            }
        }
        if (!redo) break;
        // redo the test with the plain fp32 decisions
        __syncthreads();
        if (tid == 0) {
            n_peak = 0; n_amb = 0; n_riv = 0; n_unl = 0;
            if (a.stats) { stat_add(a.stats, 3, 1u); flag_row_for_exact(a, r, blockIdx.y); }    // (the second level decides the row again)
        }
        dlt = 0.0f;
        __syncthreads();
    }

    STAMP(5)
    auto elem_row = [&](int i) -> const float* {               // unit row of the frame behind element i
        int64_t fr = i;
        if (a.mode != 0) {
            int l = (int)(j - i) % n;
            if (l < 0) l += n;
            fr = j - l - a.shift;
        }
        return a.unit + fr * (int64_t)a.unit_pitch;
    };
    if (dlt > 0.0f && n_amb > 0) {
        // Near-tie refinement. An element within delta of its window maximum (or of the threshold) is decided
        // from float64 similarities of the same fp32 spectra: it survives iff its value is >= the threshold
        // and strictly above every rival (anything lower cannot win, anything higher would have made the
        // fp32 decision safe).
        const int n_near = n_amb, n_rival = n_riv, n_items = n_near + n_unl;
        const int len4 = a.unit_pitch >> 2;
        const float* self_row = a.unit + (j - a.shift) * (int64_t)a.unit_pitch;
        auto item_row = [&](int it) -> const float* {          // unit row of the frame behind work item `it`
            return elem_row(it < n_near ? amb_idx[it] : riv_idx[unl_list[it - n_near]]);
        };
        // phase 1: float64 values, two work items per wave and memory round trip
        for (int it = 2 * (tid >> 6); it < n_items; it += 8) {
            const bool two = it + 1 < n_items;
            double e0, e1;
            exact_similarity2(self_row, item_row(it), item_row(two ? it + 1 : it), len4, lane, &e0, &e1);
            if (lane == 0) {
                if (it < n_near) amb_exact[it] = e0; else riv_exact[unl_list[it - n_near]] = e0;
                if (two) { if (it + 1 < n_near) amb_exact[it + 1] = e1; else riv_exact[unl_list[it + 1 - n_near]] = e1; }
            }
        }
        __syncthreads();
        // phase 2: every rival pair, then the verdicts
        for (int k = tid; k < n_near; k += 256) amb_best[k] = 0ull;
        __syncthreads();
        for (int e = tid; e < n_rival; e += 256) {
            const int s = riv_owner[e], ref = riv_ref[e];
            const double er = ref >= 0 ? amb_exact[ref] : riv_exact[e];
            if (!(amb_exact[s] > er)) amb_lose[s] = 1;
            atomicMax(&amb_best[s], (unsigned long long)__double_as_longlong(er));
        }
        __syncthreads();
        int changed = 0;
        for (int k = tid; k < n_near; k += 256) {
            const double ek = amb_exact[k];
            // a verdict the fp32 spectra cannot settle (peaks_exact.hip): the element against its best rival, or against
            // the threshold, closer than delta2
            if (fabs(ek - __longlong_as_double((long long)amb_best[k])) < a.delta2 || fabs(ek - a.min_value64) < a.delta2) n_close = 1;
            const bool win = !amb_lose[k] && ek >= a.min_value64;
            if (win) {
                const int slot = atomicAdd(&n_peak, 1);
                if (slot < a.peak_cap) { pval[slot] = (float)ek; pidx[slot] = amb_idx[k]; }
            }
            changed += (win != (amb_ok[k] != 0));
        }
        if (a.stats) {
            if (tid == 0) { stat_add(a.stats, 0, 1u); stat_add(a.stats, 1, (unsigned)n_near); }
            if (changed) stat_add(a.stats, 2, (unsigned)changed);
        }
        __syncthreads();
    }

    STAMP(3)
    if constexpr (STAGE == 1) {
        // leave this segment's candidates (refined ones included) for the ranking kernel
        const int cnt = n_peak < a.peak_cap ? n_peak : a.peak_cap;
        const int64_t slot = r * a.n_seg + seg;
        for (int k = tid; k < cnt; k += 256) {
            a.cand_val[slot * a.cand_cap + k] = pval[k];
            a.cand_idx[slot * a.cand_cap + k] = pidx[k];
        }
        if (tid == 0) { a.cand_cnt[slot] = cnt; if (n_close) flag_row_for_exact(a, r, blockIdx.y); }
        return;
    }
    if constexpr (STAGE == 2) {
        __shared__ int seg_base[65];
        if (tid == 0) {
            int acc = 0;
            for (int k = 0; k < a.n_seg; ++k) { seg_base[k] = acc; acc += a.cand_cnt[r * a.n_seg + k]; }
            seg_base[a.n_seg] = acc;
            n_peak = acc;
        }
        __syncthreads();
        for (int k = 0; k < a.n_seg; ++k) {
            const int cnt = seg_base[k + 1] - seg_base[k];
            const int64_t slot = r * a.n_seg + k;
            for (int e = tid; e < cnt; e += 256) {
                const int to = seg_base[k] + e;
                if (to < a.peak_cap) { pval[to] = a.cand_val[slot * a.cand_cap + e]; pidx[to] = a.cand_idx[slot * a.cand_cap + e]; }
            }
        }
        __syncthreads();
    }
    // rank by counting: value descending, higher index first on ties (np.argsort(...)[::-1])
    int np_ = n_peak;
    if (np_ > a.peak_cap) np_ = a.peak_cap;
    const int kept = np_ < a.number ? np_ : a.number;
    for (int k = np_ + tid; k < ((np_ + 3) & ~3); k += 256) pval[k] = -INFINITY;   // pad to a float4 boundary
    __syncthreads();
    int* out = a.idx + r * (int64_t)a.idx_pitch;
    const float4* pv4 = reinterpret_cast<const float4*>(pval);
    auto out_index = [&](int i) -> int {                       // what the list holds for element i
        if (a.mode == 0) return i;
        int l = (int)(j - i) % n;
        if (l < 0) l += n;
        return (int)(j - l - a.shift);
    };
    const bool cut_check = dlt > 0.0f && np_ > a.number;
    int* prank = reinterpret_cast<int*>(smem);                 // the window maxima are no longer needed
    for (int p = tid; p < np_; p += 256) {
        const float v = pval[p];
        const int i = pidx[p];
        int rank = 0;
#pragma unroll 4
        for (int q4 = 0; 4 * q4 < np_; ++q4) {
            const float4 u = pv4[q4];
            rank += (u.x > v) + (u.y > v) + (u.z > v) + (u.w > v);
            if (u.x == v || u.y == v || u.z == v || u.w == v) {     // exact ties: rare
                rank += (u.x == v && pidx[4 * q4] > i) + (u.y == v && pidx[4 * q4 + 1] > i) +
                        (u.z == v && pidx[4 * q4 + 2] > i) + (u.w == v && pidx[4 * q4 + 3] > i);
            }
        }
        if (rank < a.number) out[rank] = out_index(i);
        if (cut_check) prank[p] = rank;
    }
    if (cut_check) {
        // Top-`number` cut with more candidates than slots: if the last value kept and the first one dropped are
        // within delta, every candidate within delta of that boundary is re-ranked by float64 similarity. The
        // candidates above that band keep their places (they stay in the top `number` whatever the band's true
        // order is), the band's members fill the remaining slots in float64 order.
        __syncthreads();
        if (tid == 0) { n_amb = 0; n_riv = 0; }
        for (int p = tid; p < np_; p += 256) {
            if (prank[p] == a.number - 1) cutv[0] = pval[p];
            if (prank[p] == a.number) cutv[1] = pval[p];
        }
        __syncthreads();
        const float c_in = cutv[0], c_out = cutv[1];
        if (c_in - c_out <= dlt) {
            const float lo = c_out - dlt, hi = c_in + dlt;
            for (int p = tid; p < np_; p += 256) {
                const float v = pval[p];
                if (v > hi) atomicAdd(&n_riv, 1);                          // candidates safely above the band
                else if (v >= lo) {
                    const int slot = atomicAdd(&n_amb, 1);
                    if (slot < kAmbCap) { amb_idx[slot] = pidx[p]; amb_ok[slot] = prank[p] < a.number; }
                }
            }
            __syncthreads();
            const int n_band = n_amb, n_above = n_riv;
            if (n_band <= kAmbCap) {
                const int len4 = a.unit_pitch >> 2;
                const float* self_row = a.unit + (j - a.shift) * (int64_t)a.unit_pitch;
                for (int it = 2 * (tid >> 6); it < n_band; it += 8) {
                    const bool two = it + 1 < n_band;
                    double e0, e1;
                    exact_similarity2(self_row, elem_row(amb_idx[it]), elem_row(amb_idx[two ? it + 1 : it]), len4, lane, &e0, &e1);
                    if (lane == 0) { amb_exact[it] = e0; if (two) amb_exact[it + 1] = e1; }
                }
                __syncthreads();
                int changed = 0;
                for (int k = tid; k < n_band; k += 256) {
                    const double e = amb_exact[k];
                    const int i = amb_idx[k];
                    int crank = 0;
                    for (int t = 0; t < n_band; ++t) crank += (amb_exact[t] > e) || (amb_exact[t] == e && amb_idx[t] > i);
                    const bool keep = n_above + crank < a.number;
                    // the cut separates the lowest value kept from the highest one dropped: closer than delta2, the fp32
                    // spectra cannot say on which side they belong (peaks_exact.hip). Kept k against every dropped t:
                    if (keep)
                        for (int t = 0; t < n_band; ++t)
                            if (e - amb_exact[t] < a.delta2 && ((amb_exact[t] < e) || (amb_exact[t] == e && amb_idx[t] < i))) {
                                int rank_t = 0;
                                for (int u = 0; u < n_band; ++u) rank_t += (amb_exact[u] > amb_exact[t]) || (amb_exact[u] == amb_exact[t] && amb_idx[u] > amb_idx[t]);
                                if (n_above + rank_t >= a.number) n_close = 1;
                            }
                    if (keep) out[n_above + crank] = out_index(i);
                    changed += (keep != (amb_ok[k] != 0));
                }
                if (a.stats) {
                    if (tid == 0) stat_add(a.stats, 1, (unsigned)n_band);
                    if (changed) stat_add(a.stats, 2, (unsigned)changed);
                }
            } else if (a.stats && tid == 0) { stat_add(a.stats, 3, 1u); n_close = 1; }
        }
    }
    __syncthreads();
    if (tid == 0 && n_close) flag_row_for_exact(a, r, blockIdx.y);
    STAMP(4)
    for (int k = kept + tid; k < a.number; k += 256) out[k] = -1;
    if (tid == 0) a.count[r] = kept;
}

template <int QMAX, int STAGE>
static hipError_t launch_one(const PeakArgs& a, dim3 grid, size_t bytes, hipStream_t s) {
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void*>(&local_maxima_kernel<QMAX, STAGE>), (int)bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((local_maxima_kernel<QMAX, STAGE>), grid, dim3(256), bytes, s, a);
    return hipGetLastError();
}

template <int STAGE>
static hipError_t launch_by_size(const PeakArgs& a, dim3 grid, size_t bytes, hipStream_t s) {
    const int per_thread = (int)ceil_div(a.groups, 256);
    if (bytes > 160 * 1024 - 4096 || per_thread > 32) return hipErrorInvalidValue;
    if (per_thread <= 1) return launch_one<1, STAGE>(a, grid, bytes, s);
    if (per_thread <= 2) return launch_one<2, STAGE>(a, grid, bytes, s);
    if (per_thread <= 4) return launch_one<4, STAGE>(a, grid, bytes, s);
    if (per_thread <= 8) return launch_one<8, STAGE>(a, grid, bytes, s);
    if constexpr (STAGE == 0) {
        if (per_thread <= 16) return launch_one<16, 0>(a, grid, bytes, s);
        return launch_one<32, 0>(a, grid, bytes, s);
    }
    return hipErrorInvalidValue;      // segments are sized for at most 8 groups per thread
}

// Elements per segment of a long row: what 8 float4 groups per thread hold beside the two halos of d elements
// (0: the window is too wide to segment, the whole-row kernel is used).
static int peak_segment_length(int d) {
    const int seg = ((8 * 256 - 3) * 4 - (int)round_up(d, 4) - d) & ~3;
    return seg >= 2048 ? seg : 0;
}

size_t local_maxima_scratch_bytes(int64_t n_rows, int32_t n_cols, int32_t d) {
    if (d > n_cols) d = n_cols;
    const int seg = peak_segment_length(d);
    if (seg == 0 || n_cols <= seg + 512) return 0;      // fits one workgroup at 8 groups per thread (or nearly)
    const int64_t n_seg = ceil_div(n_cols, seg);
    const int64_t cap = round_up(d > 0 ? seg / (d + 1) + 2 : seg + 1, 4);
    return (size_t)(n_rows * n_seg * cap * 8 + n_rows * n_seg * 4 + 256);
}

hipError_t launch_local_maxima_lite(const PeakArgs& a, const ExactSource& src, hipStream_t s);      // peaks_wave.hip

hipError_t launch_local_maxima(const float* M, int64_t n_rows, int64_t row0, int32_t n_cols, int64_t pitch,
                               int32_t mode, float min_value, int32_t d, int32_t number, int32_t* idx,
                               int32_t idx_pitch, int32_t* count, hipStream_t s, int64_t shift, const PeakRefine* refine,
                               const PeakBatch* batch, void* scratch, const ExactSource* lite_src, const float* seg,
                               int32_t seg_pitch) {
    if (n_rows <= 0 && !lite_src) return hipSuccess;
    if (d > n_cols) d = n_cols;                                       // a wider window changes nothing
    PeakArgs a{};
    a.M = M; a.row0 = row0; a.n = n_cols; a.pitch = pitch; a.mode = mode; a.min_value = min_value; a.d = d;
    a.number = number; a.idx = idx; a.idx_pitch = idx_pitch; a.count = count; a.shift = shift;
    a.min_value64 = min_value;
    a.seg = seg; a.seg_pitch = seg_pitch;
    if (refine && refine->unit_rows && refine->delta > 0.0f && (refine->pitch & 3) == 0) {
        a.unit = refine->unit_rows; a.unit_pitch = refine->pitch; a.delta = refine->delta;
        a.min_value64 = refine->min_value; a.stats = refine->stats;
        a.unit_norm = refine->unit_norms;
        if (refine->redo_list && refine->stats) {
            a.delta2 = refine->delta2; a.redo_list = refine->redo_list; a.redo_flag = refine->redo_flag; a.gen = refine->gen;
            a.flag_stride = refine->flag_stride;
            a.records = refine->records; a.record_bytes = refine->record_bytes; a.lite_list = refine->lite_list;
            a.lite_flag = refine->lite_flag; a.frame_list = refine->frame_list; a.frame_flag = refine->frame_flag;
            a.frame_clip_stride = refine->frame_clip_stride;
        }
    }
    int n_batch = 1;
    if (batch && batch->n_batch > 0) {
        n_batch = batch->n_batch;
        a.m_stride = batch->m_stride; a.idx_stride = batch->idx_stride; a.cnt_stride = batch->cnt_stride;
        a.unit_stride = batch->unit_stride;
    }
    if (lite_src) return launch_local_maxima_lite(a, *lite_src, s);      // (the rows the first pass left records of)
    {   // one wavefront per row where the shape allows it (peaks_wave.hip); this kernel is the general fallback
        const hipError_t ew = launch_local_maxima_wave(a, n_rows, n_batch, s);
        if (ew != hipErrorNotSupported) return ew;
    }
    a.dl = (int)round_up(d, 4);
    const int total_cap = (int)round_up(d > 0 ? n_cols / (d + 1) + 2 : n_cols + 1, 4);   // peaks are more than d apart
    const size_t scratch_bytes = local_maxima_scratch_bytes(n_rows, n_cols, d);
    if (scratch && scratch_bytes > 0 && mode == 0 && n_batch == 1) {
        // long rows: one workgroup per (row, segment) finds the maxima, a second kernel ranks them per row
        const int seg_len = peak_segment_length(d);
        const int n_seg = (int)ceil_div(n_cols, seg_len);
        if (n_seg > 64) return hipErrorInvalidValue;
        const int cap = (int)round_up(d > 0 ? seg_len / (d + 1) + 2 : seg_len + 1, 4);
        a.seg_len = seg_len; a.n_seg = n_seg; a.cand_cap = cap;
        a.cand_val = reinterpret_cast<float*>(scratch);
        a.cand_idx = reinterpret_cast<int*>(a.cand_val + n_rows * n_seg * cap);
        a.cand_cnt = a.cand_idx + n_rows * n_seg * cap;
        PeakArgs a1 = a;
        a1.groups = (int)(round_up(a.dl + seg_len + d, 4) / 4 + 3);
        a1.peak_cap = cap;
        const size_t bytes1 = (size_t)(4 * a1.groups + 2 * a1.peak_cap) * 4;
        hipError_t e = launch_by_size<1>(a1, dim3((unsigned)n_rows, 1, (unsigned)n_seg), bytes1, s);
        if (e != hipSuccess) return e;
        PeakArgs a2 = a;
        a2.peak_cap = (int)round_up(total_cap + n_seg, 4);
        a2.groups = a2.peak_cap / 4 + 1;             // room for the ranks that reuse the row area
        const size_t bytes2 = (size_t)(4 * a2.groups + 2 * a2.peak_cap) * 4;
        return launch_one<1, 2>(a2, dim3((unsigned)n_rows), bytes2, s);
    }
    a.groups = (int)(round_up(a.dl + n_cols + d, 4) / 4 + 3);      // slack for the aligned window reads past the end
    a.peak_cap = total_cap;
    const size_t bytes = (size_t)(4 * a.groups + 2 * a.peak_cap) * 4;
    return launch_by_size<0>(a, dim3((unsigned)n_rows, (unsigned)n_batch), bytes, s);
}

#ifdef REPET_PEAK_STAMPS
extern "C" int repet_debug_peak_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_peak_stamps), sizeof(unsigned long long) * 64);
}
#endif

}  // namespace repet
