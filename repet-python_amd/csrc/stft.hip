// K1 / K9: Hamming-windowed STFT and overlap-add inverse for gfx950, written around an in-LDS
// Stockham FFT (radix 4 with a closing radix-2 stage). A real frame of W samples is transformed
// as ONE complex FFT of W/2 points (even samples -> re, odd -> im) plus a split/merge pass, so a
// frame never leaves LDS between windowing and the magnitude / channel-mean epilogue.
//
// Replaces repet.py:1001-1060 (_stft), :1063-1105 (_istft), the magnitude + channel mean of
// :158,:162,:667 and the column normalisation of :1220.
#include "common.h"

namespace repet {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cconj(float2 a) { return make_float2(a.x, -a.y); }

// In-LDS Stockham autosort FFT of N complex points by all threads of the block.
// in: `a` filled and synchronised. Returns the buffer holding the result (synchronised).
// tw[m] = exp(-2 pi i m / TW) for m < TW, TW a multiple of N. INVERSE conjugates (no 1/N scale).
template <int N, bool INVERSE>
__device__ __forceinline__ float2* fft_lds(float2* a, float2* b, const float2* __restrict__ tw, int TW) {
    const int tid = threadIdx.x, nth = blockDim.x;
    for (int p = 1; p < N;) {
        if (N / p >= 4) {
            const int tstep = TW / (p * 4);
            for (int i = tid; i < N / 4; i += nth) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 2) + k;
                float2 u0 = a[i], u1 = a[i + N / 4], u2 = a[i + N / 2], u3 = a[i + 3 * N / 4];
                if (p > 1) {
                    float2 w1 = tw[k * tstep], w2 = tw[2 * k * tstep], w3 = tw[3 * k * tstep];
                    if (INVERSE) { w1 = cconj(w1); w2 = cconj(w2); w3 = cconj(w3); }
                    u1 = cmul(u1, w1); u2 = cmul(u2, w2); u3 = cmul(u3, w3);
                }
                const float2 t0 = cadd(u0, u2), t1 = csub(u0, u2), t2 = cadd(u1, u3);
                const float2 d = csub(u1, u3);
                const float2 t3 = INVERSE ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);
                b[j] = cadd(t0, t2);
                b[j + p] = cadd(t1, t3);
                b[j + 2 * p] = csub(t0, t2);
                b[j + 3 * p] = csub(t1, t3);
            }
            p *= 4;
        } else {
            const int tstep = TW / (p * 2);
            for (int i = tid; i < N / 2; i += nth) {
                const int k = i & (p - 1);
                const int j = ((i - k) << 1) + k;
                float2 u0 = a[i], u1 = a[i + N / 2];
                float2 w1 = tw[k * tstep];
                if (INVERSE) w1 = cconj(w1);
                u1 = cmul(u1, w1);
                b[j] = cadd(u0, u1);
                b[j + p] = csub(u0, u1);
            }
            p *= 2;
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    return a;
}

constexpr int kFftThreads = 256;

template <int W>
__global__ __launch_bounds__(kFftThreads) void stft_kernel(StftArgs a) {
    constexpr int N = W / 2;                       // complex FFT length; also the Nyquist bin index
    constexpr int SLOTS = N / kFftThreads + 1;     // bins k = tid + 256*i, k <= N
    __shared__ float2 buf0[N];
    __shared__ float2 buf1[N];
    __shared__ float red[kFftThreads / kWave];

    const int tid = threadIdx.x;
    const int64_t t = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int C = a.n_channels;
    const int64_t start = t * a.H - (a.centred ? W / 2 : 0);
    const int64_t row = t * a.FS;
    a.sample_offset += b * a.batch_sample_stride;
    a.X += b * a.batch_spec_stride;
    a.V += b * a.batch_spec_stride;
    if (a.Vm) a.Vm += b * a.batch_mean_stride;
    if (a.Vn) a.Vn += b * a.batch_mean_stride;
    if (a.P) a.P += b * a.batch_mean_stride;

    float acc[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) acc[i] = 0.f;

    for (int c = 0; c < C; ++c) {
        for (int n = tid; n < N; n += kFftThreads) {
            const int64_t s0 = start + 2 * n, s1 = s0 + 1;
            float x0 = 0.f, x1 = 0.f;
            if (s0 >= 0 && s0 < a.n_samples) x0 = a.audio[(a.sample_offset + s0) * C + c];
            if (s1 >= 0 && s1 < a.n_samples) x1 = a.audio[(a.sample_offset + s1) * C + c];
            buf0[n] = make_float2(x0 * a.window[2 * n], x1 * a.window[2 * n + 1]);
        }
        __syncthreads();
        const float2* Z = fft_lds<N, false>(buf0, buf1, a.twiddle, W);
        float2* Xrow = a.X + c * a.chan_stride + row;
        float* Vrow = a.V + c * a.chan_stride + row;
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int k = tid + kFftThreads * i;
            if (k <= N) {
                const float2 zk = Z[k & (N - 1)];
                const float2 zc = cconj(Z[(N - k) & (N - 1)]);
                const float2 e = make_float2(0.5f * (zk.x + zc.x), 0.5f * (zk.y + zc.y));
                const float2 d = csub(zk, zc);
                const float2 o = make_float2(0.5f * d.y, -0.5f * d.x);   // (zk - zc) / (2i)
                const float2 x = cadd(e, cmul(a.twiddle[k], o));
                const float mag = sqrtf(x.x * x.x + x.y * x.y);
                Xrow[k] = x;
                Vrow[k] = mag;
                acc[i] += mag;
            }
        }
        if (tid < a.FS - (N + 1)) {      // zero the pad bins [F, FS)
            Xrow[N + 1 + tid] = make_float2(0.f, 0.f);
            Vrow[N + 1 + tid] = 0.f;
        }
        __syncthreads();                 // Z (buf0/buf1) is re-filled by the next channel
    }

    if (a.Vm == nullptr && a.Vn == nullptr && a.P == nullptr) return;

    // channel mean (repet.py:162,:667 np.mean(axis=2)) and its squared L2 norm over frequency
    const float inv_c = 1.0f / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int k = tid + kFftThreads * i;
        acc[i] = (C == 1) ? acc[i] : acc[i] * inv_c;
        if (k <= N) ss += acc[i] * acc[i];
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) ss += __shfl_down(ss, off);
    if ((tid & (kWave - 1)) == 0) red[tid / kWave] = ss;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int w = 0; w < kFftThreads / kWave; ++w) total += red[w];
    const float norm = sqrtf(total);     // 0 for a silent frame: 0/0 = NaN like repet.py:1220
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int k = tid + kFftThreads * i;
        if (k <= N) {
            if (a.Vm) a.Vm[row + k] = acc[i];
            if (a.Vn) a.Vn[row + k] = acc[i] / norm;
            if (a.P) a.P[row + k] = acc[i] * acc[i];
        }
    }
    if (tid < a.FS - (N + 1)) {
        if (a.Vm) a.Vm[row + N + 1 + tid] = 0.f;
        if (a.Vn) a.Vn[row + N + 1 + tid] = 0.f;
        if (a.P) a.P[row + N + 1 + tid] = 0.f;
    }
}

template <int W>
__global__ __launch_bounds__(kFftThreads) void istft_frames_kernel(IstftArgs a) {
    constexpr int N = W / 2;
    __shared__ float2 buf0[N];
    __shared__ float2 buf1[N];
    const int tid = threadIdx.x;
    const int64_t t = blockIdx.x;
    const int c = blockIdx.y;
    const float2* Y = a.Y + c * a.chan_stride + t * a.FS;
    for (int k = tid; k < N; k += kFftThreads) {
        const float2 xk = Y[k];
        const float2 xc = cconj(Y[N - k]);
        const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
        const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
        const float2 o = cmul(d, cconj(a.twiddle[k]));       // * exp(+2 pi i k / W)
        buf0[k] = make_float2(e.x - o.y, e.y + o.x);         // E + i O
    }
    __syncthreads();
    const float2* z = fft_lds<N, true>(buf0, buf1, a.twiddle, W);
    const float scale = 1.0f / (float)N;
    float2* out = reinterpret_cast<float2*>(a.frames + (c * a.T + t) * (int64_t)W);
    for (int n = tid; n < N; n += kFftThreads) out[n] = make_float2(z[n].x * scale, z[n].y * scale);
}

__global__ __launch_bounds__(256) void overlap_add_kernel(OlaArgs a) {
    const int C = a.n_channels;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_out * C) return;
    const int64_t n = e / C;
    const int c = (int)(e - n * C);
    const int64_t m = n + a.trim;
    int64_t j_lo = (m - a.W + a.H) / a.H;        // ceil((m - W + 1) / H) for m - W + 1 > 0
    if (m - a.W + 1 <= 0) j_lo = 0;
    int64_t j_hi = m / a.H;
    if (j_hi > a.T - 1) j_hi = a.T - 1;
    float sum = 0.f;
    for (int64_t j = j_lo; j <= j_hi; ++j) sum += a.frames[(c * a.T + j) * (int64_t)a.W + (m - j * a.H)];
    sum *= a.scale;
    float* dst = a.out + (a.out_offset + n) * C + c;
    if (a.accumulate_weighted) {
        float w = 1.f;
        if (n < a.fade_in) w = (float)(2 * n + 1) / (float)(2 * a.fade_in);
        else if (n >= a.n_out - a.fade_out && a.fade_out > 0) {
            const int64_t q = a.n_out - 1 - n;   // mirrored index into the rising half
            w = (float)(2 * q + 1) / (float)(2 * a.fade_out);
        }
        *dst += w * sum;
    } else {
        *dst = sum;
    }
}

// K9 fused: masked spectrum -> inverse FFT -> overlap-add -> interleaved output, hop H = W/2.
// One workgroup produces RUN consecutive hops of every channel: hop h = second half of frame h-1 + first
// half of frame h, so it inverts frames h0-1 .. h0+RUN-1 and carries each channel's tail in LDS. The
// time-domain frames never touch HBM and every output sample is written exactly once, coalesced over
// the interleaved channels.
constexpr int kOlaRun = 8;
template <int W>
__device__ __forceinline__ const float2* inverse_frame(const float2* __restrict__ Y, const float2* __restrict__ tw,
                                                       float2* buf0, float2* buf1) {
    constexpr int N = W / 2;
    for (int k = threadIdx.x; k < N; k += kFftThreads) {
        const float2 xk = Y[k];
        const float2 xc = cconj(Y[N - k]);
        const float2 e = make_float2(0.5f * (xk.x + xc.x), 0.5f * (xk.y + xc.y));
        const float2 d = make_float2(0.5f * (xk.x - xc.x), 0.5f * (xk.y - xc.y));
        const float2 o = cmul(d, cconj(tw[k]));
        buf0[k] = make_float2(e.x - o.y, e.y + o.x);
    }
    __syncthreads();
    return fft_lds<N, true>(buf0, buf1, tw, W);
}

template <int W>
__global__ __launch_bounds__(kFftThreads) void istft_ola_kernel(IstftOlaArgs a) {
    constexpr int N = W / 2;          // complex points per frame = samples per hop (H = W/2)
    constexpr int HP = N / 2;         // float2 per half frame
    __shared__ float2 buf0[N];
    __shared__ float2 buf1[N];
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const int C = a.n_channels;
    float2* tails = reinterpret_cast<float2*>(dyn);            // [C][HP] second half of the previous frame
    float* stage = dyn + (size_t)C * N;                        // [N samples][C] one hop, interleaved
    const int tid = threadIdx.x;
    const int64_t h0 = a.first_hop + (int64_t)blockIdx.x * kOlaRun;
    const float inv_n = 1.0f / (float)N;
    if (a.n_batch > 0) {
        const int j = a.batch_first + (int)blockIdx.y * a.batch_step;
        a.Y += (int64_t)(a.batch_local0 + (int)blockIdx.y * a.batch_step) * a.batch_spec_stride;
        a.out_offset += (int64_t)j * a.batch_out_stride;
        a.fade_in = j > 0 ? a.overlap : 0;
        a.fade_out = j < a.batch_total - 1 ? a.overlap : 0;
    }

    for (int c = 0; c < C; ++c) {                              // tails of frame h0-1
        const int64_t t = h0 - 1;
        if (t >= 0 && t < a.T) {
            const float2* z = inverse_frame<W>(a.Y + c * a.chan_stride + t * a.FS, a.twiddle, buf0, buf1);
            for (int m = tid; m < HP; m += kFftThreads) tails[c * HP + m] = z[HP + m];
        } else {
            for (int m = tid; m < HP; m += kFftThreads) tails[c * HP + m] = make_float2(0.f, 0.f);
        }
        __syncthreads();
    }
    for (int r = 0; r < kOlaRun; ++r) {
        const int64_t h = h0 + r;
        if (h > a.last_hop) break;
        for (int c = 0; c < C; ++c) {
            if (h < a.T) {
                const float2* z = inverse_frame<W>(a.Y + c * a.chan_stride + h * a.FS, a.twiddle, buf0, buf1);
                for (int m = tid; m < HP; m += kFftThreads) {
                    const float2 head = z[m], tail = tails[c * HP + m];
                    stage[(2 * m) * C + c] = (head.x + tail.x) * inv_n;
                    stage[(2 * m + 1) * C + c] = (head.y + tail.y) * inv_n;
                    tails[c * HP + m] = z[HP + m];
                }
            } else {                                           // past the last frame: only the tail remains
                for (int m = tid; m < HP; m += kFftThreads) {
                    const float2 tail = tails[c * HP + m];
                    stage[(2 * m) * C + c] = tail.x * inv_n;
                    stage[(2 * m + 1) * C + c] = tail.y * inv_n;
                    tails[c * HP + m] = make_float2(0.f, 0.f);
                }
            }
            __syncthreads();
        }
        // hop h covers padded samples [h*N, (h+1)*N); output sample n = padded - trim
        const int64_t n_base = h * N - a.trim;
        for (int i = tid; i < N * C; i += kFftThreads) {
            const int64_t n = n_base + i / C;
            if (n < 0 || n >= a.n_out) continue;
            float v = stage[i] * a.scale;
            float* dst = a.out + (a.out_offset + n) * C + (i % C);
            if (a.accumulate_weighted) {
                float w = 1.f;
                if (n < a.fade_in) w = (float)(2 * n + 1) / (float)(2 * a.fade_in);
                else if (a.fade_out > 0 && n >= a.n_out - a.fade_out) w = (float)(2 * (a.n_out - 1 - n) + 1) / (float)(2 * a.fade_out);
                *dst += w * v;
            } else {
                *dst = v;
            }
        }
        __syncthreads();
    }
}

template <typename Fn>
static hipError_t dispatch_window(int W, Fn&& fn) {
    switch (W) {
        case 64: fn(std::integral_constant<int, 64>{}); break;
        case 128: fn(std::integral_constant<int, 128>{}); break;
        case 256: fn(std::integral_constant<int, 256>{}); break;
        case 512: fn(std::integral_constant<int, 512>{}); break;
        case 1024: fn(std::integral_constant<int, 1024>{}); break;
        case 2048: fn(std::integral_constant<int, 2048>{}); break;
        case 4096: fn(std::integral_constant<int, 4096>{}); break;
        case 8192: fn(std::integral_constant<int, 8192>{}); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_stft(const StftArgs& a, hipStream_t s) {
    if (a.T <= 0) return hipSuccess;
    return dispatch_window(a.W, [&](auto w) {
        hipLaunchKernelGGL(stft_kernel<decltype(w)::value>, dim3((unsigned)a.T, (unsigned)(a.n_batch > 0 ? a.n_batch : 1)),
                           dim3(kFftThreads), 0, s, a);
    });
}

hipError_t launch_istft_frames(const IstftArgs& a, hipStream_t s) {
    if (a.T <= 0) return hipSuccess;
    return dispatch_window(a.W, [&](auto w) {
        hipLaunchKernelGGL(istft_frames_kernel<decltype(w)::value>, dim3((unsigned)a.T, (unsigned)a.n_channels),
                           dim3(kFftThreads), 0, s, a);
    });
}

hipError_t launch_istft_ola(const IstftOlaArgs& a0, hipStream_t s) {
    IstftOlaArgs a = a0;
    if (a.T <= 0 || a.n_out <= 0) return hipSuccess;
    // hops that intersect [trim, trim + n_out): first = floor(trim / N), last = floor((trim + n_out - 1) / N)
    const int N = a.W / 2;
    a.first_hop = a.trim / N;
    a.last_hop = (a.trim + a.n_out - 1) / N;
    if (a.last_hop > a.T) a.last_hop = a.T;                  // hop T holds the last frame's tail, later hops are empty
    const int64_t hops = a.last_hop - a.first_hop + 1;
    if (hops <= 0) return hipSuccess;
    const size_t dyn = (size_t)a.n_channels * N * sizeof(float) * 2;   // tails [C][N/2] float2 + stage [N][C]
    if (dyn > 96 * 1024) return hipErrorInvalidValue;
    return dispatch_window(a.W, [&](auto w) {
        constexpr int Wc = decltype(w)::value;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&istft_ola_kernel<Wc>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        hipLaunchKernelGGL(istft_ola_kernel<Wc>, dim3((unsigned)ceil_div(hops, kOlaRun), (unsigned)(a.n_batch > 0 ? a.n_batch : 1)),
                           dim3(kFftThreads), dyn, s, a);
    });
}

hipError_t launch_overlap_add(const OlaArgs& a, hipStream_t s) {
    const int64_t total = a.n_out * a.n_channels;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(overlap_add_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ---- elementwise helpers --------------------------------------------------------------------------
template <typename T>
__global__ void convert_in_kernel(const T* src, float* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (float)src[i];
}
__global__ void convert_out_kernel(const float* src, double* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (double)src[i];
}
__global__ void square_kernel(const float* src, float* dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i] * src[i];
}
// dst[t][0..FS) = src[t][0..F) / ||src[t]||, pad bins zero (repet.py:1220 for a caller-supplied matrix)
__global__ __launch_bounds__(256) void unit_rows_kernel(const float* src, float* dst, int F, int FS) {
    __shared__ float red[4];
    const int64_t t = blockIdx.x;
    const int tid = threadIdx.x;
    float ss = 0.f;
    for (int k = tid; k < F; k += 256) { const float v = src[t * F + k]; ss += v * v; }
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    const float norm = sqrtf(red[0] + red[1] + red[2] + red[3]);
    for (int k = tid; k < FS; k += 256) dst[t * FS + k] = (k < F) ? src[t * F + k] / norm : 0.f;
}

static unsigned stream_grid(int64_t n) {
    int64_t g = ceil_div(n, 256);
    return (unsigned)(g > 2048 ? 2048 : (g < 1 ? 1 : g));
}

hipError_t launch_convert_in(const void* src, int dtype, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    switch (dtype) {
        case 0: hipLaunchKernelGGL(convert_in_kernel<float>, dim3(stream_grid(n)), dim3(256), 0, s, (const float*)src, dst, n); break;
        case 1: hipLaunchKernelGGL(convert_in_kernel<double>, dim3(stream_grid(n)), dim3(256), 0, s, (const double*)src, dst, n); break;
        case 2: hipLaunchKernelGGL(convert_in_kernel<int16_t>, dim3(stream_grid(n)), dim3(256), 0, s, (const int16_t*)src, dst, n); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_convert_out(const float* src, double* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(convert_out_kernel, dim3(stream_grid(n)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_square(const float* src, float* dst, int64_t n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(square_kernel, dim3(stream_grid(n)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}
hipError_t launch_unit_rows(const float* src, float* dst, int64_t T, int32_t F, int32_t FS, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(unit_rows_kernel, dim3((unsigned)T), dim3(256), 0, s, src, dst, F, FS);
    return hipGetLastError();
}

}  // namespace repet
