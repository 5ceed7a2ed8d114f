// K5 / K8 / K8b: repeating-model median + soft mask for gfx950 -- replaces _simmask, _mask and
// _adaptivemask (repet.py:1386-1545) and the per-frame median of the online variant (repet.py:872-887).
//
// Spectrograms are frame-major V[c][t][FS], so "the same 64 frequency bins of frame idx[k]" is one
// coalesced 256-byte read per wave. A wave owns 64 consecutive bins of one frame; every lane gathers
// its <= 128 values into registers (the index list is wave-uniform, so the row bases sit in SGPRs),
// runs a pruned odd-even merge-sort network (median_networks.inc) and reads the two middle order
// statistics; np.median's even/odd rule and its NaN for an empty list are reproduced. Lists longer
// than 128 fall back to a 31-step bisection on the float bit patterns that re-reads the values.
// The soft mask (min(V,model)+eps)/(V+eps) (repet.py:1441-1448), the high-pass override
// mask[1..cutoff] = 1 (repet.py:185) and the multiplication into the STFT happen in the same pass.
#include "common.h"

namespace repet {

#include "median_networks.inc"

// Median of n gathered values with the N-wire network; `load(k)` returns the k-th value (k < n).
template <int N, class Load>
__device__ __forceinline__ float median_network(int n, Load load) {
    float a[N];
    const int low_pads = (N - n) >> 1;        // -1 pads below, +inf above: medians stay at N/2-1, N/2
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (k < n) a[k] = load(k);
        else a[k] = (k - n < low_pads) ? -1.0f : INFINITY;
    }
    MedianNet<N>::run(a);
    return (n & 1) ? a[N / 2 - 1] : 0.5f * (a[N / 2 - 1] + a[N / 2]);
}

// Order statistics by bisection over the (non-negative) float bit patterns.
template <class Load>
__device__ __forceinline__ float median_bisect(int n, Load load) {
    const int k = (n - 1) >> 1;               // lower median rank
    unsigned lo = 0u, hi = 0x7f800000u;       // answer in [lo, hi]
    while (lo < hi) {
        const unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for (int q = 0; q < n; ++q) c += (__float_as_uint(load(q)) <= mid);
        if (c >= k + 1) hi = mid; else lo = mid + 1;
    }
    const float lower = __uint_as_float(lo);
    if (n & 1) return lower;
    int c_le = 0;
    float next = INFINITY;
    for (int q = 0; q < n; ++q) {
        const float v = load(q);
        c_le += (v <= lower);
        if (v > lower) next = fminf(next, v);
    }
    const float upper = (c_le >= k + 2) ? lower : next;
    return 0.5f * (lower + upper);
}

template <class Load>
__device__ __forceinline__ float median_select(int n, Load load) {
    if (n <= 0) return __uint_as_float(0x7fc00000u);     // np.median of an empty slice
    if (n <= 2) return median_network<2>(n, load);
    if (n <= 4) return median_network<4>(n, load);
    if (n <= 8) return median_network<8>(n, load);
    if (n <= 16) return median_network<16>(n, load);
    if (n <= 24) return median_network<24>(n, load);
    if (n <= 32) return median_network<32>(n, load);
    if (n <= 48) return median_network<48>(n, load);
    if (n <= 64) return median_network<64>(n, load);
    if (n <= 80) return median_network<80>(n, load);
    if (n <= 100) return median_network<100>(n, load);
    if (n <= 128) return median_network<128>(n, load);
    return median_bisect(n, load);
}

__device__ __forceinline__ float soft_mask(float v, float model, int f, int cutoff) {
    const float m = (fminf(v, model) + kMaskEps) / (v + kMaskEps);
    // fminf drops a NaN model; np.minimum propagates it (empty similarity list -> NaN frame)
    const float mm = (model != model) ? model : m;
    return (f >= 1 && f <= cutoff) ? 1.0f : mm;
}

__device__ __forceinline__ void emit(const MaskArgs& a, int c, int64_t t, int f, float m) {
    const int64_t o = c * a.chan_stride + t * a.FS + f;
    if (a.mask) a.mask[o] = m;
    if (a.X) { float2 x = a.X[o]; x.x *= m; x.y *= m; a.X[o] = x; }
}

// ---- REPET-SIM / online: list of similar frames per frame ----------------------------------------
__global__ __launch_bounds__(256) void mask_sim_kernel(MaskArgs a, const int* __restrict__ idx, int idx_pitch,
                                                       const int* __restrict__ count, int64_t first_frame) {
    const int64_t t = blockIdx.x;
    const int c = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nfb = (a.F + 63) >> 6;
    const float* Vc = a.V + c * a.chan_stride;
    if (t < first_frame) {           // online warm-up frames contribute nothing (repet.py:834)
        for (int f = threadIdx.x; f < a.F; f += 256) emit(a, c, t, f, 0.f);
        return;
    }
    const int64_t r = t - first_frame;
    const int n = count[r];
    const int* list = idx + r * (int64_t)idx_pitch;
    for (int fb = wave; fb < nfb; fb += 4) {
        const int f = fb * 64 + lane;
        const bool active = f < a.F;
        const int fc = active ? f : a.F - 1;
        const float med = median_select(n, [&](int k) { return Vc[(int64_t)list[k] * a.FS + fc]; });
        if (active) emit(a, c, t, f, soft_mask(Vc[t * a.FS + fc], med, f, a.cutoff));
    }
}

hipError_t launch_mask_sim(const MaskArgs& m, const int32_t* idx, int32_t idx_pitch, const int32_t* count,
                           int64_t first_frame, hipStream_t s) {
    if (m.T <= 0) return hipSuccess;
    hipLaunchKernelGGL(mask_sim_kernel, dim3((unsigned)m.T, (unsigned)m.n_channels), dim3(256), 0, s, m, idx,
                       idx_pitch, count, first_frame);
    return hipGetLastError();
}

// ---- adaptive: taps at i + {..}*period[i] (repet.py:1478-1498) ------------------------------------
__global__ __launch_bounds__(256) void mask_adaptive_kernel(MaskArgs a, const int* __restrict__ periods, int order) {
    const int64_t t = blockIdx.x;
    const int c = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nfb = (a.F + 63) >> 6;
    const float* Vc = a.V + c * a.chan_stride;
    const int64_t per = periods[t];
    // center_indices = arange(1, order+1) - ceil(order/2); the in-range taps form a contiguous run
    const int first_tap = 1 - ((order + 1) >> 1);
    int k_lo = 0, n = 0;
    for (int k = 0; k < order; ++k) {
        const int64_t j = t + (int64_t)(first_tap + k) * per;
        if (j >= 0 && j < a.T) { if (n == 0) k_lo = k; ++n; }
    }
    const int64_t base = t + (int64_t)(first_tap + k_lo) * per;
    for (int fb = wave; fb < nfb; fb += 4) {
        const int f = fb * 64 + lane;
        const bool active = f < a.F;
        const int fc = active ? f : a.F - 1;
        const float med = median_select(n, [&](int k) { return Vc[(base + (int64_t)k * per) * a.FS + fc]; });
        if (active) emit(a, c, t, f, soft_mask(Vc[t * a.FS + fc], med, f, a.cutoff));
    }
}

hipError_t launch_mask_adaptive(const MaskArgs& m, const int32_t* periods, int32_t order, hipStream_t s) {
    if (m.T <= 0) return hipSuccess;
    hipLaunchKernelGGL(mask_adaptive_kernel, dim3((unsigned)m.T, (unsigned)m.n_channels), dim3(256), 0, s, m,
                       periods, order);
    return hipGetLastError();
}

// ---- original / extended: one median per position q inside the period (repet.py:1401-1446) ---------
// grid (period_max, C); workgroups with q >= period exit. The model of position q is the median over
// the segments that really contain frame s*p+q (all S for q < T-(S-1)p, else the first S-1); it is
// computed once and applied to every segment.
__global__ __launch_bounds__(256) void mask_period_kernel(MaskArgs a, const int* __restrict__ period_dev,
                                                          int period_host) {
    const int p = period_dev ? period_dev[0] : period_host;
    const int q = blockIdx.x;
    if (q >= p) return;
    const int c = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nfb = (a.F + 63) >> 6;
    const float* Vc = a.V + c * a.chan_stride;
    const int S = (int)((a.T + p - 1) / p);
    const int n = (q < a.T - (int64_t)(S - 1) * p) ? S : S - 1;     // segments holding real data at q
    for (int fb = wave; fb < nfb; fb += 4) {
        const int f = fb * 64 + lane;
        const bool active = f < a.F;
        const int fc = active ? f : a.F - 1;
        const float med = median_select(n, [&](int k) { return Vc[((int64_t)k * p + q) * a.FS + fc]; });
        if (active)
            for (int s = 0; s < n; ++s) {
                const int64_t t = (int64_t)s * p + q;
                emit(a, c, t, f, soft_mask(Vc[t * a.FS + fc], med, f, a.cutoff));
            }
    }
}

hipError_t launch_mask_period(const MaskArgs& m, const int32_t* period_dev, int32_t period_host, hipStream_t s) {
    if (m.T <= 0) return hipSuccess;
    // the period lives on the device when it was just estimated there; the grid covers the largest
    // admissible period (a third of the frames, repet.py:1266) and surplus workgroups exit
    unsigned gx = period_dev ? (unsigned)(m.T / 3 + 2) : (unsigned)period_host;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(mask_period_kernel, dim3(gx, (unsigned)m.n_channels), dim3(256), 0, s, m, period_dev,
                       period_host);
    return hipGetLastError();
}

}  // namespace repet
