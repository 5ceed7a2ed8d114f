// K3 / K3b / K6: Gram matrix A A^T of a frame-major spectrogram A[Tpad][FS] in exact fp32 on the
// matrix cores (v_mfma_f32_32x32x2_f32), for
//   * the cosine self-similarity matrix of REPET-SIM        (repet.py:1223, np.matmul(Vn.T, Vn))
//   * the banded similarity of the online variant            (repet.py:1244, one column per frame)
//   * the beat spectrum as diagonal sums of the Gram of V^2  (repet.py:1108-1158, autocorrelation)
//
// Tile 128x128 per 256-thread workgroup (2x2 waves, each 2x2 MFMA tiles of 32x32), BK = 32, register
// prefetch two K-tiles ahead and a double-buffered padded LDS image (pitch 36 floats: the 16-lane
// groups of ds_read_b128 then cover all 64 banks). Both operands are row panels of the same matrix,
// so A-tile and B-tile loads are identical and coalesce on 128-byte row chunks.
// Only tiles on or above the diagonal are computed; the full variant mirrors them through LDS.
#include "common.h"

#include <vector>

namespace repet {

typedef float floatx16 __attribute__((ext_vector_type(16)));

#ifdef REPET_GRAM_TRACE
__device__ unsigned long long g_gram_trace[4096 * 3];
#endif

constexpr int BK = 32;
constexpr int LDP = BK + 4;                     // LDS row pitch in floats
constexpr int TILE_FLOATS = kTile * LDP;        // one operand tile in LDS
constexpr int kGramLds = 4 * TILE_FLOATS * 4;   // 2 operands x 2 buffers, bytes (73,728)

enum GramMode { GRAM_FULL = 0, GRAM_BAND = 1, GRAM_CROSS = 2 };   // CROSS: out = A B^T, two different matrices

// tiles[blockIdx.x] = (bi, bj) with bj >= bi, or (-1,-1) for a filler slot. The host orders the list so
// that the blocks resident together on one XCD (ids congruent mod 8 under round-robin dispatch) walk one
// 8x8 super-block of tiles at a time: 64 tiles share 8 + 8 row panels through that XCD's L2.
template <int MODE>
__global__ __launch_bounds__(256) void gram_kernel(const float* __restrict__ A, int64_t T, int FS,
                                                      float* __restrict__ out, int64_t pitch, int n_lags,
                                                      const int2* __restrict__ tiles, int64_t a_batch_stride,
                                                      int64_t out_batch_stride, const float* __restrict__ B,
                                                      int64_t TB) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    A += blockIdx.y * a_batch_stride;           // batch of equal-shape matrices (segments of `extended`)
    out += blockIdx.y * out_batch_stride;
    const int2 tile = tiles[blockIdx.x];
    const int bi = tile.x, bj = tile.y;
    if (bi < 0) return;
#ifdef REPET_GRAM_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_gram_trace[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        g_gram_trace[3 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
    }
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    const float* Ag = A + (int64_t)bi * kTile * FS;
    const float* Bg = (MODE == GRAM_CROSS ? B : A) + (int64_t)bj * kTile * FS;

    floatx16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // staging: 128 rows x 8 float4 per operand tile = 1024 float4, 4 per thread and operand
    // (wave-uniform 64-bit base in SGPRs + one 32-bit per-lane offset shared by both operands)
    const unsigned g0 = (unsigned)(((tid + 0) >> 3) * FS + (tid & 7) * 4);
    const unsigned g1 = (unsigned)(((tid + 256) >> 3) * FS + (tid & 7) * 4);
    const unsigned g2 = (unsigned)(((tid + 512) >> 3) * FS + (tid & 7) * 4);
    const unsigned g3 = (unsigned)(((tid + 768) >> 3) * FS + (tid & 7) * 4);
    const int l0 = ((tid + 0) >> 3) * LDP + (tid & 7) * 4;      // LDS float offsets of the same four chunks
    const int l1 = l0 + 32 * LDP, l2 = l0 + 64 * LDP, l3 = l0 + 96 * LDP;
    // two staging register sets: the loads of K-tile kt+2 are issued while tile kt is multiplied and have a
    // whole iteration to land before they are written to LDS (HBM/L2 latency ~ one K-tile of MFMAs)
    float4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;      // set P
    float4 qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;      // set Q
#define REPET_LOAD_TILE(S, kt)                                                  \
    {                                                                           \
        const float* Ak = Ag + (kt) * BK;                                       \
        const float* Bk = Bg + (kt) * BK;                                       \
        S##a0 = *reinterpret_cast<const float4*>(Ak + g0);                      \
        S##b0 = *reinterpret_cast<const float4*>(Bk + g0);                      \
        S##a1 = *reinterpret_cast<const float4*>(Ak + g1);                      \
        S##b1 = *reinterpret_cast<const float4*>(Bk + g1);                      \
        S##a2 = *reinterpret_cast<const float4*>(Ak + g2);                      \
        S##b2 = *reinterpret_cast<const float4*>(Bk + g2);                      \
        S##a3 = *reinterpret_cast<const float4*>(Ak + g3);                      \
        S##b3 = *reinterpret_cast<const float4*>(Bk + g3);                      \
    }
#define REPET_STORE_TILE(S, buf)                                                \
    {                                                                           \
        float* As_ = lds + (buf) * 2 * TILE_FLOATS;                             \
        float* Bs_ = As_ + TILE_FLOATS;                                         \
        *reinterpret_cast<float4*>(As_ + l0) = S##a0;                           \
        *reinterpret_cast<float4*>(Bs_ + l0) = S##b0;                           \
        *reinterpret_cast<float4*>(As_ + l1) = S##a1;                           \
        *reinterpret_cast<float4*>(Bs_ + l1) = S##b1;                           \
        *reinterpret_cast<float4*>(As_ + l2) = S##a2;                           \
        *reinterpret_cast<float4*>(Bs_ + l2) = S##b2;                           \
        *reinterpret_cast<float4*>(As_ + l3) = S##a3;                           \
        *reinterpret_cast<float4*>(Bs_ + l3) = S##b3;                           \
    }
#define REPET_MFMA4(AX, BX)                                                                       \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.AX, b0.BX, acc[0][0], 0, 0, 0);           \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.AX, b1.BX, acc[0][1], 0, 0, 0);           \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.AX, b0.BX, acc[1][0], 0, 0, 0);           \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.AX, b1.BX, acc[1][1], 0, 0, 0);
#define REPET_COMPUTE(buf)                                                                              \
    {                                                                                                   \
        const float* As = lds + (buf) * 2 * TILE_FLOATS + (wr * 64 + lr) * LDP + 4 * lh;                \
        const float* Bs = lds + (buf) * 2 * TILE_FLOATS + TILE_FLOATS + (wc * 64 + lr) * LDP + 4 * lh;  \
        _Pragma("unroll") for (int ks = 0; ks < BK / 8; ++ks) {                                         \
            const float4 a0 = *reinterpret_cast<const float4*>(As + ks * 8);                            \
            const float4 a1 = *reinterpret_cast<const float4*>(As + 32 * LDP + ks * 8);                 \
            const float4 b0 = *reinterpret_cast<const float4*>(Bs + ks * 8);                            \
            const float4 b1 = *reinterpret_cast<const float4*>(Bs + 32 * LDP + ks * 8);                 \
            REPET_MFMA4(x, x) REPET_MFMA4(y, y) REPET_MFMA4(z, z) REPET_MFMA4(w, w)                     \
        }                                                                                               \
    }

    const int nk = FS / BK;                 // FS is a multiple of 32; nk >= 2 for every supported window
    REPET_LOAD_TILE(p, 0)
    REPET_STORE_TILE(p, 0)
    if (nk > 1) REPET_LOAD_TILE(q, 1)
    __syncthreads();
    // iteration kt: LDS[kt&1] holds tile kt, one register set holds tile kt+1, the other receives kt+2.
    // The MFMA blocks are unconditional inside the loop (an odd last tile is peeled) so the accumulators
    // stay in AGPRs across iterations instead of being copied around the control flow.
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        if (kt + 2 < nk) REPET_LOAD_TILE(p, kt + 2)
        REPET_COMPUTE(0)
        REPET_STORE_TILE(q, 1)
        __syncthreads();
        if (kt + 3 < nk) REPET_LOAD_TILE(q, kt + 3)
        REPET_COMPUTE(1)
        if (kt + 2 < nk) REPET_STORE_TILE(p, 0)
        __syncthreads();
    }
    if (kt < nk) REPET_COMPUTE(0)
#undef REPET_LOAD_TILE
#undef REPET_STORE_TILE
#undef REPET_MFMA4
#undef REPET_COMPUTE

    // ---- epilogue. acc[m][n][r]: i = wr*64 + m*32 + (r&3) + 8*(r>>2) + 4*lh ; j = wc*64 + n*32 + lr
    const int64_t gi0 = (int64_t)bi * kTile + wr * 64;
    const int64_t gj0 = (int64_t)bj * kTile + wc * 64;
    if (MODE == GRAM_FULL || MODE == GRAM_CROSS) {
        const int64_t TJ = (MODE == GRAM_CROSS) ? TB : T;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t gi = gi0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int64_t gj = gj0 + n * 32 + lr;
                    if (gi < T && gj < TJ) out[gi * pitch + gj] = acc[m][n][r];
                }
        if (MODE == GRAM_FULL && bi != bj) {
            // mirror: transpose this wave's 64x64 block through a private LDS patch (pitch 65). The patches alias the
            // tile buffers, which a slower wave may still be reading in its last K-tile: wait for the whole block.
            __syncthreads();
            float* patch = lds + wave * (64 * 65);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int j = n * 32 + lr;
                        patch[j * 65 + i] = acc[m][n][r];
                    }
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the patch is wave-private
            __builtin_amdgcn_wave_barrier();
            for (int j = 0; j < 64; ++j) {
                const int64_t gj = gj0 + j, gi = gi0 + lane;
                if (gj < T && gi < T) out[gj * pitch + gi] = patch[j * 65 + lane];
            }
        }
    } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t gi = gi0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int64_t gj = gj0 + n * 32 + lr;
                    const int64_t lag = gj - gi;
                    if (gi < T && gj < T && lag >= 0 && lag < n_lags) out[gi * pitch + lag] = acc[m][n][r];
                }
    }
#ifdef REPET_GRAM_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_gram_trace[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
}

static hipError_t set_lds(const void* fn) {
    return ensure_dynamic_lds(fn, kGramLds);
}

// Upper-triangle tile list (bj - bi < ndiag) in XCD-aware order, see gram_kernel.
int gram_tile_list(int nb, int ndiag, std::vector<int2>* out) {
    std::vector<int2> seq;
    const int SB = 8;                                       // super-block edge in tiles
    const int nsb = (nb + SB - 1) / SB;
    for (int I = 0; I < nsb; ++I)
        for (int J = I; J < nsb; ++J)
            for (int i = I * SB; i < (I + 1) * SB && i < nb; ++i)
                for (int j = (J * SB > i ? J * SB : i); j < (J + 1) * SB && j < nb; ++j)
                    if (j - i < ndiag) seq.push_back(make_int2(i, j));
    const int n = (int)seq.size();
    const int per = (n + 7) / 8;                            // contiguous share of each XCD
    out->assign((size_t)per * 8, make_int2(-1, -1));
    for (int x = 0; x < 8; ++x)
        for (int k = 0; k < per; ++k) {
            const int src = x * per + k;
            if (src < n) (*out)[(size_t)k * 8 + x] = seq[src];
        }
    return per * 8;
}

hipError_t launch_gram_full(const float* A, int64_t T, int32_t FS, float* S, int64_t TS, const int2* tiles,
                            int32_t n_tiles, hipStream_t s) {
    if (T <= 0 || n_tiles <= 0) return hipSuccess;
    hipError_t attr = set_lds(reinterpret_cast<const void*>(&gram_kernel<GRAM_FULL>));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(gram_kernel<GRAM_FULL>, dim3((unsigned)n_tiles), dim3(256), kGramLds, s, A, T, FS, S, TS, 0, tiles,
                       (int64_t)0, (int64_t)0, (const float*)nullptr, (int64_t)0);
    return hipGetLastError();
}

hipError_t launch_gram_band(const float* A, int64_t T, int32_t FS, float* band, int32_t n_lags, int32_t LP,
                            const int2* tiles, int32_t n_tiles, int32_t n_batch, int64_t a_batch_stride,
                            int64_t band_batch_stride, hipStream_t s) {
    if (T <= 0 || n_lags <= 0 || n_tiles <= 0) return hipSuccess;
    hipError_t attr = set_lds(reinterpret_cast<const void*>(&gram_kernel<GRAM_BAND>));
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(gram_kernel<GRAM_BAND>, dim3((unsigned)n_tiles, (unsigned)(n_batch > 0 ? n_batch : 1)), dim3(256),
                       kGramLds, s, A, T, FS, band, (int64_t)LP, n_lags, tiles, a_batch_stride, band_batch_stride,
                       (const float*)nullptr, (int64_t)0);
    return hipGetLastError();
}

// out[TA][pitch] = A B^T for A[TApad][FS], B[TBpad][FS] (rows beyond TA / TB zero): every tile, no mirror.
hipError_t launch_matmul_nt(const float* A, int64_t TA, const float* B, int64_t TB, int32_t FS, float* out,
                            int64_t pitch, hipStream_t s) {
    if (TA <= 0 || TB <= 0) return hipSuccess;
    hipError_t attr = set_lds(reinterpret_cast<const void*>(&gram_kernel<GRAM_CROSS>));
    if (attr != hipSuccess) return attr;
    const int na = (int)ceil_div(TA, kTile), nbt = (int)ceil_div(TB, kTile);
    std::vector<int2> host;
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nbt; ++j) host.push_back(make_int2(i, j));
    int2* tiles = nullptr;
    hipError_t e = hipMalloc(&tiles, host.size() * sizeof(int2));
    if (e != hipSuccess) return e;
    e = hipMemcpyAsync(tiles, host.data(), host.size() * sizeof(int2), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(gram_kernel<GRAM_CROSS>, dim3((unsigned)host.size()), dim3(256), kGramLds, s, A, TA, FS, out, pitch, 0,
                           tiles, (int64_t)0, (int64_t)0, B, TB);
        e = hipGetLastError();
    }
    (void)hipStreamSynchronize(s);      // stage export only: the temporary tile list is freed here
    (void)hipFree(tiles);
    return e;
}

// _acorr (repet.py:1108-1139) as its own stage: ac[l][c] = sum_t x[t][c] x[t+l][c] / (R - l), x[R][pitch].
// (The pipelines never form it: the beat spectrum takes the mean over c first, as diagonal sums of a Gram band.)
__global__ __launch_bounds__(256) void acorr_kernel(const float* __restrict__ x, int R, int n_cols, int pitch,
                                                    float* __restrict__ ac) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int l = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (c >= n_cols || l >= R) return;
    float sum = 0.f;
    for (int t = 0; t + l < R; ++t) sum += x[(int64_t)t * pitch + c] * x[(int64_t)(t + l) * pitch + c];
    ac[(int64_t)l * pitch + c] = sum / (float)(R - l);
}

hipError_t launch_acorr(const float* x, int32_t R, int32_t n_cols, int32_t pitch, float* ac, hipStream_t s) {
    if (R <= 0 || n_cols <= 0) return hipSuccess;
    hipLaunchKernelGGL(acorr_kernel, dim3((unsigned)ceil_div(n_cols, 64), (unsigned)ceil_div(R, 4)), dim3(256), 0, s, x, R,
                       n_cols, pitch, ac);
    return hipGetLastError();
}

// ---- windowed diagonal sums of the band (beat spectrum / beat spectrogram) -------------------------
// beat[w][l] = sum over the window's rows of band[t][l], unbiased and averaged over F. The rows of a window are cut
// into chunks of kBeatChunk (fixed, so the summation order -- and the result -- never depends on the launch shape):
// pass 1 sums each chunk with one workgroup per (64-lag block, chunk, window x batch), four waves taking a quarter
// of the chunk each; pass 2 adds the chunk sums in order. A 180-s clip's single window is 7 753 rows: one workgroup
// per lag block took 0.7 ms for it, 31 chunks take a few tens of microseconds.
constexpr int kBeatChunk = 256;

__global__ __launch_bounds__(256) void band_chunk_sum_kernel(const float* __restrict__ band, int64_t T, int LP, int n_lags,
                                                             int64_t start0, int64_t step, int64_t len, int n_windows,
                                                             int n_chunks, float* __restrict__ partial,
                                                             int64_t band_batch_stride) {
    __shared__ float part[4][64];
    const int w = blockIdx.z % n_windows, batch = blockIdx.z / n_windows;
    const int chunk = blockIdx.y;
    band += batch * band_batch_stride;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = blockIdx.x * 64 + lane;
    const int64_t a = start0 + (int64_t)w * step;    // first frame of the window (may be < 0)
    const int64_t lo = a < 0 ? 0 : a;
    int64_t hi = a + len - 1 - l;                    // last t with t + l inside the window
    if (hi > T - 1 - l) hi = T - 1 - l;
    float sum = 0.f;
    if (l < n_lags && l < len) {
        const int64_t c0 = lo + (int64_t)chunk * kBeatChunk;          // this chunk: rows [c0, c0 + kBeatChunk) of the window
        const int64_t t0 = c0 + wave * (kBeatChunk / 4);
        int64_t t1 = t0 + kBeatChunk / 4;
        if (t1 > hi + 1) t1 = hi + 1;
        // sixteen rows in flight, added in row order (the order is part of the result): one load -> add chain per row was
        // a memory round trip per row, 64 in a row (cfg 4: 22 us for 14 MB)
        const float* p = band + t0 * LP + l;
        int64_t t = t0;
        for (; t + 16 <= t1; t += 16, p += 16 * (int64_t)LP) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = p[u * (int64_t)LP];
#pragma unroll
            for (int u = 0; u < 16; ++u) sum += v[u];
        }
        for (; t < t1; ++t, p += LP) sum += *p;
    }
    part[wave][lane] = sum;
    __syncthreads();
    if (wave == 0 && l < n_lags)
        partial[((int64_t)blockIdx.z * n_chunks + chunk) * LP + l] = ((part[0][lane] + part[1][lane]) + part[2][lane]) + part[3][lane];
}

__global__ __launch_bounds__(64) void band_chunk_reduce_kernel(const float* __restrict__ partial, int LP, int n_lags, int n_freq,
                                                               int64_t len, int n_windows, int n_chunks, float* beat,
                                                               int beat_pitch, int64_t beat_batch_stride) {
    const int w = blockIdx.y % n_windows, batch = blockIdx.y / n_windows;
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= n_lags) return;
    float total = 0.f;
    for (int k = 0; k < n_chunks; ++k) total += partial[((int64_t)blockIdx.y * n_chunks + k) * LP + l];
    // unbiased by the window length minus the lag (repet.py:1135-1137), mean over F (repet.py:1156)
    beat[batch * beat_batch_stride + (int64_t)w * beat_pitch + l] = (l < len) ? total / ((float)(len - l) * (float)n_freq) : 0.f;
}

hipError_t launch_band_window_sum(const float* band, int64_t T, int32_t LP, int32_t n_lags, int32_t n_freq,
                                  int64_t start0, int64_t step, int64_t len, int32_t n_windows, float* beat,
                                  int32_t beat_pitch, int32_t n_batch, int64_t band_batch_stride,
                                  int64_t beat_batch_stride, float* partial, hipStream_t s) {
    if (n_windows <= 0 || n_lags <= 0) return hipSuccess;
    const int nb = n_batch > 0 ? n_batch : 1;
    const int n_chunks = band_window_chunks(T, len);
    hipLaunchKernelGGL(band_chunk_sum_kernel, dim3((unsigned)ceil_div(n_lags, 64), (unsigned)n_chunks, (unsigned)(n_windows * nb)),
                       dim3(256), 0, s, band, T, LP, n_lags, start0, step, len, n_windows, n_chunks, partial, band_batch_stride);
    hipLaunchKernelGGL(band_chunk_reduce_kernel, dim3((unsigned)ceil_div(n_lags, 64), (unsigned)(n_windows * nb)), dim3(64), 0, s,
                       partial, LP, n_lags, n_freq, len, n_windows, n_chunks, beat, beat_pitch, beat_batch_stride);
    return hipGetLastError();
}

int band_window_chunks(int64_t T, int64_t len) {
    const int64_t rows = len < T ? len : T;
    return (int)ceil_div(rows > 0 ? rows : 1, kBeatChunk);
}

// ---- K7: period = argmax(beat[lo:hi]) + 1 + lo, first maximum wins (repet.py:1263-1289) ------------
__global__ __launch_bounds__(64) void periods_kernel(const float* __restrict__ beat, int pitch, int lo, int hi,
                                                     int* period) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* b = beat + (int64_t)c * pitch;
    float best = -INFINITY;
    int arg = 0x7fffffff;
    int nan_at = 0x7fffffff;
    for (int l = lo + lane; l < hi; l += 64) {
        const float v = b[l];
        // a value that is not finite: repet.py's autocorrelation goes through an FFT over time (:1108-1139), so ONE such power
        // spectrum makes every lag NaN and np.argmax returns index 0 -- here the sums of the lags that pair the frame are inf or
        // NaN, the others finite (none of those inside the searched range): any non-finite value means "first lag"
        if (!(fabsf(v) <= 3.4028234664e38f)) { nan_at = lo; continue; }
        if (v > best) { best = v; arg = l; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_down(best, off);
        const int oa = __shfl_down(arg, off);
        const int on = __shfl_down(nan_at, off);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
        if (on < nan_at) nan_at = on;
    }
    if (lane == 0) {
        // np.argmax returns the first NaN if any is present
        if (nan_at != 0x7fffffff) arg = nan_at;
        if (arg == 0x7fffffff) arg = lo;      // all -inf cannot happen; keep defined
        period[c] = (arg - lo) + 1 + lo;
    }
}

hipError_t launch_periods(const float* beat, int32_t n_cols, int32_t pitch, int32_t n_lags, int32_t lo,
                          int32_t hi, int32_t* period, hipStream_t s) {
    if (n_cols <= 0) return hipSuccess;
    int h = hi < n_lags / 3 ? hi : n_lags / 3;
    hipLaunchKernelGGL(periods_kernel, dim3((unsigned)n_cols), dim3(64), 0, s, beat, pitch, lo, h, period);
    return hipGetLastError();
}

__global__ void expand_periods_kernel(const int* win_period, int n_windows, int step, int64_t T, int lo,
                                      int* frame_period) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int w = (int)(t / step);
    const int64_t i = (int64_t)w * step;
    // columns i .. min(i+step-1, T)-1 copy window w; column i+step-1 keeps zeros -> argmax 0 -> lo+1
    const int64_t end = (i + step - 1 < T) ? i + step - 1 : T;
    int p = (t < end || t == i) ? win_period[w] : lo + 1;
    frame_period[t] = p;
}

hipError_t launch_expand_periods(const int32_t* win_period, int32_t n_windows, int32_t step, int64_t T,
                                 int32_t lo, int32_t* frame_period, hipStream_t s) {
    if (T <= 0) return hipSuccess;
    hipLaunchKernelGGL(expand_periods_kernel, dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, s, win_period,
                       n_windows, step, T, lo, frame_period);
    return hipGetLastError();
}

#ifdef REPET_GRAM_TRACE
extern "C" int repet_debug_gram_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gram_trace), sizeof(unsigned long long) * 4096 * 3);
}
#endif

}  // namespace repet
