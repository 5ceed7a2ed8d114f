// K4 / K4b: peak picking of REPET-SIM on gfx950 -- replaces _localmaxima / _indices
// (repet.py:1294-1383), 54 % of the reference's run time as a pure-Python loop.
//
// One 256-thread workgroup per row of the similarity matrix. The row is staged once in LDS
// (coalesced 16-byte loads; NaN is stored as +inf so that it can never win and always blocks its
// neighbours, exactly like the reference's `all(v[i] > window)` tests). Element i survives iff
//   v[i] >= min_value,  v[i] > every v[max(i-d,0) : i],  v[i] > every v[i+1 : min(i+d+1,n)]
// (strict, window clipped at the ends, no wrap). The kernel first compacts the 1-neighbour peaks
// into an LDS list (about a third of the row at worst), then only those scan the full +-d window, and
// the survivors are ranked by counting (value descending, higher index first on exact ties) so the top
// `number` land in idx[row][0..count) already sorted -- no atomics on global memory, no sort pass.
#include "common.h"

namespace repet {

__device__ __forceinline__ int wave_prefix_slot(bool flag, int* counter, int lane) {
    // returns the list slot of this lane if flag, using one LDS atomic per wave
    const unsigned long long ballot = __ballot(flag);
    int base = 0;
    if (lane == 0 && ballot) base = atomicAdd(counter, __popcll(ballot));
    base = __shfl(base, 0);
    return base + __popcll(ballot & ((1ull << lane) - 1ull));
}

__global__ __launch_bounds__(256) void local_maxima_kernel(const float* __restrict__ M, int64_t row0, int n,
                                                           int64_t pitch, int mode, float min_value, int d,
                                                           int number, int* __restrict__ idx, int idx_pitch,
                                                           int* __restrict__ count, int n_pad, int cand_cap,
                                                           int peak_cap) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* row = smem;                                         // n_pad floats
    int* cand = reinterpret_cast<int*>(smem + n_pad);          // cand_cap
    float* pval = smem + n_pad + cand_cap;                     // peak_cap
    int* pidx = reinterpret_cast<int*>(pval + peak_cap);       // peak_cap
    __shared__ int n_cand, n_peak;

    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t r = blockIdx.x;           // row within this launch
    const int64_t j = row0 + r;             // absolute row (mode 1: current frame)
    if (tid == 0) { n_cand = 0; n_peak = 0; }

    if (mode == 0) {
        const float* src = M + j * pitch;
        if ((pitch & 3) == 0) {
            for (int i = tid * 4; i < n; i += 1024) {
                if (i + 3 < n) {
                    float4 v = *reinterpret_cast<const float4*>(src + i);
                    row[i] = (v.x != v.x) ? INFINITY : v.x;
                    row[i + 1] = (v.y != v.y) ? INFINITY : v.y;
                    row[i + 2] = (v.z != v.z) ? INFINITY : v.z;
                    row[i + 3] = (v.w != v.w) ? INFINITY : v.w;
                } else {
                    for (int q = i; q < n; ++q) { const float v = src[q]; row[q] = (v != v) ? INFINITY : v; }
                }
            }
        } else {
            for (int i = tid; i < n; i += 256) { const float v = src[i]; row[i] = (v != v) ? INFINITY : v; }
        }
    } else {
        // circular-buffer order of the online variant: column c holds frame j - ((j - c) mod B)
        for (int c = tid; c < n; c += 256) {
            int l = (int)((j - c) % n);
            if (l < 0) l += n;
            const float v = M[(j - l) * pitch + l];
            row[c] = (v != v) ? INFINITY : v;
        }
    }
    __syncthreads();

    // pass A: 1-neighbour peaks above the threshold
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        bool ok = false;
        if (i < n) {
            const float v = row[i];
            ok = (v >= min_value) && (v < INFINITY);
            if (d > 0) {
                if (i > 0) ok = ok && (v > row[i - 1]);
                if (i + 1 < n) ok = ok && (v > row[i + 1]);
            }
        }
        const int slot = wave_prefix_slot(ok, &n_cand, lane);
        if (ok) cand[slot] = i;
    }
    __syncthreads();

    // pass B: full +-d scan for the candidates only
    const int nc = n_cand;
    for (int q0 = 0; q0 < nc; q0 += 256) {
        const int q = q0 + tid;
        bool ok = q < nc;
        int i = 0;
        float v = 0.f;
        if (ok) {
            i = cand[q];
            v = row[i];
            const int lo = (i - d > 0) ? i - d : 0;
            const int hi = (i + d < n - 1) ? i + d : n - 1;
            for (int k = i - 2; k >= lo && ok; --k) ok = v > row[k];
            for (int k = i + 2; k <= hi && ok; ++k) ok = v > row[k];
        }
        const int slot = wave_prefix_slot(ok, &n_peak, lane);
        if (ok && slot < peak_cap) { pval[slot] = v; pidx[slot] = i; }
    }
    __syncthreads();

    // rank by counting: value descending, higher index first on ties (np.argsort(...)[::-1])
    int np_ = n_peak;
    if (np_ > peak_cap) np_ = peak_cap;
    const int kept = np_ < number ? np_ : number;
    int* out = idx + r * (int64_t)idx_pitch;
    for (int p = tid; p < np_; p += 256) {
        const float v = pval[p];
        const int i = pidx[p];
        int rank = 0;
        for (int q = 0; q < np_; ++q) {
            const float u = pval[q];
            rank += (u > v) || (u == v && pidx[q] > i);
        }
        if (rank < number) {
            int o = i;
            if (mode == 1) {
                int l = (int)((j - i) % n);
                if (l < 0) l += n;
                o = (int)(j - l);
            }
            out[rank] = o;
        }
    }
    for (int k = kept + tid; k < number; k += 256) out[k] = -1;
    if (tid == 0) count[r] = kept;
}

hipError_t launch_local_maxima(const float* M, int64_t n_rows, int64_t row0, int32_t n_cols, int64_t pitch,
                               int32_t mode, float min_value, int32_t d, int32_t number, int32_t* idx,
                               int32_t idx_pitch, int32_t* count, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    const int n_pad = (int)round_up(n_cols, 4);
    const int cand_cap = (d > 0 ? (n_cols + 1) / 2 : n_cols) + 4;    // strict 1-neighbour peaks cannot be adjacent
    const int peak_cap = (d > 0 ? n_cols / (d + 1) + 2 : n_cols + 1); // peaks are more than d apart
    const size_t bytes = (size_t)(n_pad + cand_cap + 2 * peak_cap) * 4;
    if (bytes > 160 * 1024 - 64) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&local_maxima_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(local_maxima_kernel, dim3((unsigned)n_rows), dim3(256), bytes, s, M, row0, n_cols, pitch,
                       mode, min_value, d, number, idx, idx_pitch, count, n_pad, cand_cap, peak_cap);
    return hipGetLastError();
}

}  // namespace repet
