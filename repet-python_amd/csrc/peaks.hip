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
// Every thread keeps its own elements in registers, so the test costs 4 LDS reads per element and
// the whole row about 20 LDS accesses per element, with no data-dependent loop.
// Survivors are compacted with one LDS atomic per wave and ranked by counting (value descending,
// higher index first on exact ties) so the top `number` land in idx[row][0..count) already sorted.
#include "common.h"

#include <type_traits>

namespace repet {

__device__ __forceinline__ int wave_prefix_slot(bool flag, int* counter, int lane) {
    // returns the list slot of this lane if flag, using one LDS atomic per wave
    const unsigned long long ballot = __ballot(flag);
    int base = 0;
    if (lane == 0 && ballot) base = atomicAdd(counter, __popcll(ballot));
    base = __shfl(base, 0);
    return base + __popcll(ballot & ((1ull << lane) - 1ull));
}

struct PeakArgs {
    const float* M; int64_t row0; int n; int64_t pitch; int mode; float min_value; int d; int number;
    int* idx; int idx_pitch; int* count; int lp; int peak_cap;
};

// JMAX: padded elements per thread (lp = n + 2d <= 256*JMAX). KEEP: originals stay in registers.
template <int JMAX, bool KEEP>
__global__ __launch_bounds__(256) void local_maxima_kernel(PeakArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* M = smem;                                           // lp floats (padded row, then window maxima)
    float* pval = smem + a.lp;                                 // peak_cap
    int* pidx = reinterpret_cast<int*>(pval + a.peak_cap);     // peak_cap
    __shared__ int n_peak;

    const int tid = threadIdx.x, lane = tid & 63;
    const int n = a.n, d = a.d, lp = a.lp;
    const int64_t r = blockIdx.x;           // row within this launch
    const int64_t j = a.row0 + r;           // absolute row (mode 1: current frame)
    if (tid == 0) n_peak = 0;

    // element q of this thread sits at padded position p = tid + 256*q; real index i = p - d
    auto fetch = [&](int i) -> float {
        float v;
        if (a.mode == 0) {
            v = a.M[j * a.pitch + i];
        } else {   // circular-buffer order of the online variant: column c holds frame j - ((j - c) mod B)
            int l = (int)((j - i) % n);
            if (l < 0) l += n;
            v = a.M[(j - l) * a.pitch + l];
        }
        return (v != v) ? INFINITY : v;
    };

    float own[KEEP ? JMAX : 1];
#pragma unroll
    for (int q = 0; q < JMAX; ++q) {
        const int p = tid + 256 * q;
        if (p < lp) {
            const int i = p - d;
            const float v = (i >= 0 && i < n) ? fetch(i) : -INFINITY;
            if constexpr (KEEP) own[q] = v;
            M[p] = v;
        }
    }
    __syncthreads();

    int w = 1;
    while (2 * w <= d) w *= 2;              // w = 2^floor(log2 d) (1 when d <= 1)
    for (int step = 1; step < w; step *= 2) {
        float tmp[JMAX];
#pragma unroll
        for (int q = 0; q < JMAX; ++q) {
            const int p = tid + 256 * q;
            if (p < lp) {
                const float x = M[p];
                tmp[q] = (p + step < lp) ? fmaxf(x, M[p + step]) : x;
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < JMAX; ++q) {
            const int p = tid + 256 * q;
            if (p < lp) M[p] = tmp[q];
        }
        __syncthreads();
    }

    // strict local-maximum test and compaction of the survivors
#pragma unroll
    for (int q = 0; q < JMAX; ++q) {
        const int p = tid + 256 * q;
        const int i = p - d;
        bool ok = false;
        float v = 0.f;
        if (256 * q < lp) {                 // wave-uniform guard: whole rounds past the row are skipped
            if (p < lp && i >= 0 && i < n) {
                if constexpr (KEEP) v = own[q]; else v = fetch(i);
                ok = (v >= a.min_value) && (v < INFINITY);
                if (d > 0) {
                    const float left = fmaxf(M[p - d], M[p - w]);
                    const float right = fmaxf(M[p + 1], M[p + d - w + 1]);
                    ok = ok && (v > left) && (v > right);
                }
            }
            const int slot = wave_prefix_slot(ok, &n_peak, lane);
            if (ok && slot < a.peak_cap) { pval[slot] = v; pidx[slot] = i; }
        }
    }
    __syncthreads();

    // rank by counting: value descending, higher index first on ties (np.argsort(...)[::-1])
    int np_ = n_peak;
    if (np_ > a.peak_cap) np_ = a.peak_cap;
    const int kept = np_ < a.number ? np_ : a.number;
    int* out = a.idx + r * (int64_t)a.idx_pitch;
    for (int p = tid; p < np_; p += 256) {
        const float v = pval[p];
        const int i = pidx[p];
        int rank = 0;
        for (int q = 0; q < np_; ++q) {
            const float u = pval[q];
            rank += (u > v) || (u == v && pidx[q] > i);
        }
        if (rank < a.number) {
            int o = i;
            if (a.mode == 1) {
                int l = (int)((j - i) % n);
                if (l < 0) l += n;
                o = (int)(j - l);
            }
            out[rank] = o;
        }
    }
    for (int k = kept + tid; k < a.number; k += 256) out[k] = -1;
    if (tid == 0) a.count[r] = kept;
}

template <int JMAX, bool KEEP>
static hipError_t launch_one(const PeakArgs& a, int64_t n_rows, size_t bytes, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&local_maxima_kernel<JMAX, KEEP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((local_maxima_kernel<JMAX, KEEP>), dim3((unsigned)n_rows), dim3(256), bytes, s, a);
    return hipGetLastError();
}

hipError_t launch_local_maxima(const float* M, int64_t n_rows, int64_t row0, int32_t n_cols, int64_t pitch,
                               int32_t mode, float min_value, int32_t d, int32_t number, int32_t* idx,
                               int32_t idx_pitch, int32_t* count, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    if (d > n_cols) d = n_cols;                                       // a wider window changes nothing
    PeakArgs a{};
    a.M = M; a.row0 = row0; a.n = n_cols; a.pitch = pitch; a.mode = mode; a.min_value = min_value; a.d = d;
    a.number = number; a.idx = idx; a.idx_pitch = idx_pitch; a.count = count;
    a.lp = n_cols + 2 * d;
    a.peak_cap = (d > 0 ? n_cols / (d + 1) + 2 : n_cols + 1);         // peaks are more than d apart
    const size_t bytes = (size_t)(a.lp + 2 * a.peak_cap) * 4;
    const int per_thread = (int)ceil_div(a.lp, 256);
    if (bytes > 160 * 1024 - 64 || per_thread > 128) return hipErrorInvalidValue;
    if (per_thread <= 2) return launch_one<2, true>(a, n_rows, bytes, s);
    if (per_thread <= 4) return launch_one<4, true>(a, n_rows, bytes, s);
    if (per_thread <= 8) return launch_one<8, true>(a, n_rows, bytes, s);
    if (per_thread <= 16) return launch_one<16, true>(a, n_rows, bytes, s);
    if (per_thread <= 32) return launch_one<32, true>(a, n_rows, bytes, s);
    if (per_thread <= 64) return launch_one<64, true>(a, n_rows, bytes, s);
    return launch_one<128, false>(a, n_rows, bytes, s);
}

}  // namespace repet
