// 1024-point complex FFT of one wavefront in float64 registers: the 16 x 16 x 4 decomposition of stft_reg.hip (which see
// for the index algebra) with double2 elements -- the float64 spectra of the peak picking's second level (peaks_exact.hip)
// for the 2048-sample window. v[n1] = x[64 n1 + lane] in, v[s] = X[lane + 64 s] out; two transposes through a private LDS
// region of kExPitch double2 per wave; stage twiddles a[k1][lane] = W_1024^(lane k1), b[m2][j1] = W_64^(m2 j1) in LDS.
#pragma once
#include <hip/hip_runtime.h>

namespace repet {
namespace f64fft {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

__device__ __forceinline__ void wave_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void dft4(double2& a, double2& b, double2& c, double2& d) {
    const double2 t0 = cadd(a, c), t1 = csub(a, c), t2 = cadd(b, d), e = csub(b, d);
    const double2 t3 = make_double2(e.y, -e.x);          // -i e
    a = cadd(t0, t2); b = cadd(t1, t3); c = csub(t0, t2); d = csub(t1, t3);
}

// 16-point DFT in place: input v[n] natural (n = 4p + q); output X[k] in v[tr16(k)].
__host__ __device__ constexpr int tr16(int k) { return (k >> 2) + 4 * (k & 3); }

__device__ __forceinline__ void dft16(double2 (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) dft4(v[q], v[q + 4], v[q + 8], v[q + 12]);
    constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
    auto mulw = [](double2 x, double wr, double wi) { return make_double2(x.x * wr + x.y * wi, x.y * wr - x.x * wi); };   // x (wr - i wi)
    v[1 + 4] = mulw(v[1 + 4], c1, s1);
    v[1 + 8] = mulw(v[1 + 8], h, h);
    v[1 + 12] = mulw(v[1 + 12], s1, c1);
    v[2 + 4] = mulw(v[2 + 4], h, h);
    v[2 + 8] = mulw(v[2 + 8], 0.0, 1.0);
    v[2 + 12] = mulw(v[2 + 12], -h, h);
    v[3 + 4] = mulw(v[3 + 4], s1, c1);
    v[3 + 8] = mulw(v[3 + 8], -h, h);
    v[3 + 12] = mulw(v[3 + 12], -c1, -s1);
#pragma unroll
    for (int r = 0; r < 4; ++r) dft4(v[4 * r], v[4 * r + 1], v[4 * r + 2], v[4 * r + 3]);
}

constexpr int kRegN = 1024;          // complex FFT length
constexpr int kExPitch = 1280;       // double2 per wave: max(16 * 68, 16 * 80)
constexpr int kTwCount = 16 * 64 + 64;
struct Twiddles { const double2* a; const double2* b; };

// tw2048[m] = exp(-2 pi i m / 2048), m < 2048
__device__ __forceinline__ Twiddles load_twiddles(double2* lds, const double2* __restrict__ tw2048, int tid, int n_threads) {
    for (int i = tid; i < 16 * 64; i += n_threads) lds[i] = tw2048[2 * (i & 63) * (i >> 6)];      // exp(-2 pi i l k1 / 1024)
    if (tid < 64) lds[16 * 64 + tid] = tw2048[32 * (tid >> 4) * (tid & 15)];                       // exp(-2 pi i m2 j1 / 64)
    return Twiddles{lds, lds + 16 * 64};
}

__device__ __forceinline__ void wave_fft1024(double2 (&v)[16], double2* ex, const Twiddles& t, int lane) {
    dft16(v);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const double2 x = v[tr16(k1)];
        ex[68 * k1 + lane] = (k1 == 0) ? x : cmul(x, t.a[64 * k1 + lane]);
    }
    wave_fence();
    {
        const double2* src = ex + 68 * (lane >> 2) + (lane & 3);
#pragma unroll
        for (int m1 = 0; m1 < 16; ++m1) v[m1] = src[4 * m1];
    }
    wave_fence();
    dft16(v);
    {
        double2* dst = ex + (lane >> 2) + 20 * (lane & 3);
#pragma unroll
        for (int j1 = 0; j1 < 16; ++j1) {
            const double2 x = v[tr16(j1)];
            dst[80 * j1] = (j1 == 0) ? x : cmul(x, t.b[16 * (lane & 3) + j1]);
        }
    }
    wave_fence();
    {
        const double2* src = ex + 80 * (lane >> 4) + (lane & 15);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double2 c0 = src[320 * i], c1 = src[320 * i + 20], c2 = src[320 * i + 40], c3 = src[320 * i + 60];
            dft4(c0, c1, c2, c3);
            v[i] = c0; v[i + 4] = c1; v[i + 8] = c2; v[i + 12] = c3;       // slot s = i + 4 j2
        }
    }
    wave_fence();
}

}  // namespace f64fft
}  // namespace repet
