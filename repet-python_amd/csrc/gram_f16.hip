// K3h: the cosine self-similarity matrix of REPET-SIM (repet.py:1223, np.matmul(Vn.T, Vn)) on the f16 matrix
// cores of gfx950 with fp32-class accuracy.
//
// v_mfma_f32_32x32x2_f32 runs at 157 TFLOP/s, v_mfma_f32_32x32x16_f16 at 2.5 PFLOP/s. Every fp32 operand x (a
// component of a unit row, 0 <= x <= 1) is split once into two halves of its significand,
//     x * 2^7 = hi + lo + r,   hi = f16(x 2^7),  lo = f16(x 2^7 - hi),  |r| <= 2^-22 |x 2^7|,
// and a product is taken as hi hi' + hi lo' + lo hi' (three MFMAs instead of one; lo lo' <= 2^-22 of the product
// is dropped). The f16 MFMA multiplies exactly and accumulates in fp32, so an entry keeps ~22 significant bits
// per product -- measured against float64: the same error as the exact-fp32 kernel, whose own fp32 accumulation
// over 1056 terms dominates both -- for 3/16 of the matrix-core time. The scale 2^7 keeps hi and lo in the normal
// f16 range (lo of a typical component is ~1e-3); it is removed exactly (2^-14) in the epilogue. NaN rows (silent
// frames) stay NaN. Decisions that hinge on the last bits are re-taken in float64 anyway (peaks.hip).
//
// Same tiling as gram.hip: 128x128 tile per 256-thread workgroup (2x2 waves of 2x2 MFMA blocks), BK = 32, double-
// buffered LDS, XCD-ordered upper-triangle tile list, mirror through LDS. With the MFMA share down to a fifth the
// kernel lives on its staging path, so: (1) the global image interleaves hi and lo per 32 components -- one 128-byte
// line per (row, K-tile) instead of two half-used ones; (2) three register sets carry K-tiles kt+1 .. kt+3 and a
// pipeline step has no branch in it, so hipcc's vmcnt accounting stays exact (waits for the oldest set only);
// (3) the LDS image of a plane is [128 rows][4 chunks of 8 halves] with chunk c of row r stored at
// c ^ ((r >> 2) & 3) and the lo plane 64 bytes off a bank-row boundary: the 16-lane groups of ds_read_b128 (rows
// {0-3,12-15,20-27}, ...) and the 8-lane groups of ds_write_b128 (one row's 4 hi + 4 lo chunks) touch every bank once.
// Measured at cfg 2 (T = 7753, FS = 1056): 0.227 ms against 0.61 ms for the fp32 kernel; without the K loop 0.06 ms
// (split + epilogue), without the MFMAs 0.20 ms -- the staging path, not the matrix cores, is what remains.
#include "common.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>

namespace repet {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int HBK = 32;                        // K elements per K-tile
constexpr int kPlaneHalves = kTile * HBK;      // one plane tile in LDS: 128 rows x 32 halves = 8 KB
constexpr int kPlanePitch = kPlaneHalves + 32; // + 64 bytes: the hi and lo chunks one 8-lane ds_write group stores land on different banks
constexpr int kOperandHalves = 2 * kPlanePitch;
constexpr int kGramF16Lds = 2 /*buffers*/ * 2 /*A, B*/ * kOperandHalves * 2;   // 66,048 bytes
constexpr float kSplitScale = 128.0f;          // 2^7

// planes[row][kb][hi | lo][32]: the hi and the lo halves of the 32 components kb*32 .. kb*32+31 of a row sit side by
// side, 64 + 64 bytes = one 128-byte cache line per (row, K-tile). With two separate planes a K-tile touched only
// half of every line it pulled into L1 and the other half was gone again by the next K-tile.
// General matrices (power spectra for the beat spectrum: any dynamic range) are scaled ROW BY ROW: row t by the power of
// two that brings its largest magnitude to [2^13, 2^14), well inside the f16 range; exact to apply and to remove (the
// epilogue multiplies entry (i, j) by inv[i] * inv[j]). One scale for the whole matrix would flush a passage 60-100 dB
// below the loudest one to f16 zeros -- for a power spectrum that is 30-50 dB of level -- and hand its frames an all-zero
// beat spectrum. Unit rows (row_inv == null) use the fixed 2^7.
__device__ __forceinline__ float row_scale(float m) { return f16_row_scale(m); }

__device__ __forceinline__ void split4(const float4 x, float sc, _Float16* __restrict__ planes, int64_t i) {
    const float v[4] = {x.x * sc, x.y * sc, x.z * sc, x.w * sc};
    _Float16 h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = (_Float16)v[k];
        l[k] = (_Float16)(v[k] - (float)h[k]);
    }
    const int64_t blk = i >> 5, kk = i & 31;                                    // 32-component block, offset inside it
    _Float16* dst = planes + blk * 64 + kk;
    *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(h);
    *reinterpret_cast<uint2*>(dst + 32) = *reinterpret_cast<const uint2*>(l);
}

__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ planes, int64_t count) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;     // FS is a multiple of 32: never straddles
    if (i >= count) return;
    split4(*reinterpret_cast<const float4*>(src + i), kSplitScale, planes, i);
}

// one wavefront per row: largest magnitude, scale, split (the second read of the row hits the cache)
__global__ __launch_bounds__(256) void split_f16_rows_kernel(const float* __restrict__ src, _Float16* __restrict__ planes,
                                                             int64_t n_rows, int FS, float* __restrict__ row_inv) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= n_rows) return;
    const float4* r4 = reinterpret_cast<const float4*>(src + row * FS);
    const int n4 = FS >> 2;
    float m = 0.f;
    for (int k = lane; k < n4; k += 64) {
        const float4 x = r4[k];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(x.x), fabsf(x.y))), fmaxf(fabsf(x.z), fabsf(x.w)));   // fmaxf drops NaN
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    const float sc = row_scale(m);
    if (lane == 0) row_inv[row] = 1.0f / sc;
    for (int k = lane; k < n4; k += 64) split4(r4[k], sc, planes, row * FS + 4 * k);
}

// BAND: out[t][l] = row t . row t+l for 0 <= l < n_lags (pitch = band pitch), only the tiles that touch those lags;
// blockIdx.y = clip of a batch (strides in halves / floats).
template <bool BAND>
__global__ __launch_bounds__(256) void gram_f16_kernel(const _Float16* __restrict__ planes, int64_t T, int FS,
                                                       float* __restrict__ out, int64_t pitch,
                                                       const int2* __restrict__ tiles, int n_lags,
                                                       int64_t plane_batch_stride, int64_t out_batch_stride,
                                                       const float* __restrict__ row_inv, int64_t inv_batch_stride, int lookback) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    planes += blockIdx.y * plane_batch_stride;
    out += blockIdx.y * out_batch_stride;
    // The list is dealt to the XCDs in groups of eight entries (entry 8 k + x is XCD x's k-th tile: workgroup id modulo 8) and
    // padded to a multiple of eight: in a batch every clip would put its padding on the same XCDs (a segment of `extended`
    // has seven tiles + one: XCD 7 idle, 0.203 -> 0.189 ms at cfg 3 without that), so clip b takes the group rotated by b.
    const int2 tile = tiles[(blockIdx.x & ~7u) | ((blockIdx.x + blockIdx.y) & 7u)];
    const int bi = tile.x, bj = tile.y;
    if (bi < 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    const int64_t a_row0 = (int64_t)bi * kTile, b_row0 = (int64_t)bj * kTile;      // first row of the A / B panel

    floatx16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    {
    // Round 3: the K loop of gram_f16_big.hip's rescheduled kernel on this tile (the register-staged loop of rounds 1-2 -- three
    // staging register sets, ds_write_b128 -- is in the history). The K-tiles go global -> LDS directly
    // (global_load_lds_dwordx4, 1-KB pieces of 16 rows x 64 bytes of one plane, the XOR swizzle applied on the per-lane
    // SOURCE chunk): no staging registers (three sets of 32 were a third of the register file) and no ds_write_b128
    // traffic (32 KB per K-tile at 79 B/clk was more LDS time than the fragment reads). One barrier per K-tile, in its
    // middle; the DMA pieces of tile k+2 and the fragment reads of the next half tile ride between the MFMAs, the
    // fragments into a second register set. Same products in the same order as every other f16-split kernel.
    const unsigned grow = (unsigned)(2 * FS);
    const int prow = lane >> 2;
    const _Float16* src_lane[8];
    int dst_piece[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pc = __builtin_amdgcn_readfirstlane(wave) * 8 + j;                    // 0 .. 31
        const int operand = pc >> 4, plane = (pc >> 3) & 1, rb = pc & 7;                // 8 blocks of 16 rows
        const int r = rb * 16 + prow;
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        src_lane[j] = planes + ((operand ? b_row0 : a_row0) + r) * grow + plane * 32 + chunk * 8;
        dst_piece[j] = operand * kOperandHalves + plane * kPlanePitch + rb * 16 * HBK;  // halves, wave-uniform
    }
    auto issue_piece = [&](int kt, int j) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_lane[j] + kt * 64),
                                         (__attribute__((address_space(3))) void*)(ldsh + (kt & 1) * 2 * kOperandHalves + dst_piece[j]), 16, 0, 0);
    };
    struct Frags { halfx8 ah[2], al[2], bh[2], bl[2]; };
    auto frag = [&](const _Float16* plane_ptr, int w, int blk, int ks) -> halfx8 {
        return *reinterpret_cast<const halfx8*>(plane_ptr + (w * 64 + blk * 32 + lr) * HBK + (((2 * ks + lh) ^ ((lr >> 2) & 3)) << 3));
    };
    // fragment q of 8, in the order the next half uses them
    auto load_frag = [&](int kt, int ks, Frags& f, int q) {
        const _Float16* base = ldsh + (kt & 1) * 2 * kOperandHalves;
        if (q < 2) f.al[q] = frag(base + kPlanePitch, wr, q, ks);
        else if (q < 4) f.bh[q - 2] = frag(base + kOperandHalves, wc, q - 2, ks);
        else if (q < 6) f.ah[q - 4] = frag(base, wr, q - 4, ks);
        else f.bl[q - 6] = frag(base + kOperandHalves + kPlanePitch, wc, q - 6, ks);
    };
    auto half_tile = [&](const Frags& f, Frags& g, int g_kt, int g_ks, int dma_kt, bool dma) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int grp = i >> 2, m = (i >> 1) & 1, n = i & 1;          // lo hi', hi lo', hi hi'
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(grp == 0 ? f.al[m] : f.ah[m], grp == 1 ? f.bl[n] : f.bh[n],
                                                                acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i < 8) load_frag(g_kt, g_ks, g, i);
            if (dma && (i % 3) != 0) issue_piece(dma_kt, (i / 3) * 2 + (i % 3) - 1);     // (wave-uniform) pieces 0 .. 7 behind MFMAs 1, 2, 4, 5, 7, 8, 10, 11
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const int nk = FS / HBK;
#pragma unroll
    for (int j = 0; j < 8; ++j) issue_piece(0, j);
    if (nk > 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) issue_piece(1, j);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F78);     // vmcnt(8): tile 0 has landed, tile 1 may be in flight
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frags fa, fb;
#pragma unroll
    for (int q = 0; q < 8; ++q) load_frag(0, 0, fa, q);
    for (int kt = 0; kt < nk; ++kt) {
        half_tile(fa, fb, kt, 1, 0, false);
        // every wave has read tile kt out of its buffer (its second half's fragments are in registers: lgkmcnt(0)); tile
        // kt+1 has landed (vmcnt(0)); tile kt+2 goes into the buffer tile kt has left
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        half_tile(fb, fa, kt + 1 < nk ? kt + 1 : kt, 0, kt + 2, kt + 2 < nk);
    }
    __syncthreads();
    }

    // ---- epilogue. acc[m][n][r]: i = wr*64 + m*32 + (r&3) + 8*(r>>2) + 4*lh ; j = wc*64 + n*32 + lr
    const int64_t gi0 = a_row0 + wr * 64;
    const int64_t gj0 = b_row0 + wc * 64;
    if (row_inv) {                                     // per-row scales: entry (i, j) carries scale[i] * scale[j]
        const float* inv = row_inv + blockIdx.y * inv_batch_stride;
        float cj[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) cj[n] = inv[gj0 + n * 32 + lr];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ci = inv[gi0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh];
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n][r] *= ci * cj[n];       // powers of two: exact
            }
    } else {
        constexpr float unscale = 1.0f / (kSplitScale * kSplitScale);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] *= unscale;
    }
    if (BAND && lookback) {
        // LOOK-BACK layout (simonline): out[j][l] = row j . row j - l, the similarities of frame j with the frames BEFORE it
        // side by side -- the peak picking of frame j reads one contiguous row instead of walking a diagonal of
        // out[t][l] = row t . row t + l one cache line per element (cfg 5: 55 000 rows of 431 elements, a 64-byte sector
        // each). The transposed block goes through this wave's LDS patch so that a store instruction covers 64 consecutive
        // lags of one row.
        __syncthreads();                                  // the patches alias the tile buffers
        float* patch = reinterpret_cast<float*>(ldsh) + wave * (64 * 65);
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
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the patch is wave-private
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < 64; ++j) {
            const int64_t gj = gj0 + j, gi = gi0 + lane;
            const int64_t lag = gj - gi;
            if (gj < T && gi < T && lag >= 0 && lag < n_lags) out[gj * pitch + lag] = patch[j * 65 + lane];
        }
        return;
    }
    if (BAND || bi != bj) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t gi = gi0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    const int64_t gj = gj0 + n * 32 + lr;
                    if (BAND) {
                        const int64_t lag = gj - gi;
                        if (gi < T && gj < T && lag >= 0 && lag < n_lags) out[gi * pitch + lag] = acc[m][n][r];
                    } else {
                        if (gi < T && gj < T) out[gi * pitch + gj] = acc[m][n][r];
                    }
                }
    }
    if (BAND) return;
    // Every wave's 64x64 block goes through an LDS patch (pitch 65 floats = 16,640 B per wave, 66,560 B in all: the
    // launch asks for that much). The patches alias the tile buffers: wait until every wave is done with them.
    __syncthreads();
    float* patches = reinterpret_cast<float*>(ldsh);
    float* patch = patches + wave * (64 * 65);
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
    if (bi != bj) {
        // mirror of an off-diagonal tile: the transposed block, straight from this wave's own patch
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the patch is wave-private
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < 64; ++j) {
            const int64_t gj = gj0 + j, gi = gi0 + lane;
            if (gj < T && gi < T) out[gj * pitch + gi] = patch[j * 65 + lane];
        }
    } else {
        // diagonal tile: hi hi' + hi lo' + lo hi' is accumulated in a different order for (i, j) and (j, i), so the
        // two differ in the last bit. S is stored exactly symmetric: both take the value computed for i <= j.
        __syncthreads();
        const float* upper = patches + (wc * 2 + wr) * (64 * 65);     // block (wc, wr): holds T(J, I) for this block's (I, J)
        for (int i = 0; i < 64; ++i) {
            const int I = wr * 64 + i, J = wc * 64 + lane;
            const float v = (I <= J) ? patch[lane * 65 + i] : upper[i * 65 + lane];
            const int64_t gi = gi0 + i, gj = gj0 + lane;
            if (gi < T && gj < T) out[gi * pitch + gj] = v;
        }
    }
}

constexpr int kGramF16LdsAsk = 4 * 64 * 65 * 4 > kGramF16Lds ? 4 * 64 * 65 * 4 : kGramF16Lds;   // 66,560 bytes

}  // namespace

hipError_t launch_split_f16(const float* src, void* planes, int64_t count, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)ceil_div(count, 1024)), dim3(256), 0, s, src,
                       reinterpret_cast<_Float16*>(planes), count);
    return hipGetLastError();
}

hipError_t launch_split_f16_rows(const float* src, void* planes, int64_t n_rows, int32_t FS, float* row_inv, hipStream_t s) {
    if (n_rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_f16_rows_kernel, dim3((unsigned)ceil_div(n_rows, 4)), dim3(256), 0, s, src,
                       reinterpret_cast<_Float16*>(planes), n_rows, FS, row_inv);
    return hipGetLastError();
}

hipError_t launch_gram_full_f16(const void* planes, int64_t T, int32_t FS, float* S, int64_t TS,
                                const int2* tiles, int32_t n_tiles, hipStream_t s) {
    if (T <= 0 || n_tiles <= 0) return hipSuccess;
    hipError_t attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_f16_kernel<false>), kGramF16LdsAsk);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((gram_f16_kernel<false>), dim3((unsigned)n_tiles), dim3(256), kGramF16LdsAsk, s,
                       reinterpret_cast<const _Float16*>(planes), T, FS, S, TS, tiles, 0, (int64_t)0, (int64_t)0,
                       (const float*)nullptr, (int64_t)0, 0);
    return hipGetLastError();
}

hipError_t launch_gram_band_f16(const void* planes, int64_t T, int32_t FS, float* band, int32_t n_lags, int32_t LP,
                                const int2* tiles, int32_t n_tiles, int32_t n_batch, int64_t plane_batch_stride,
                                int64_t band_batch_stride, hipStream_t s, const float* row_inv, int64_t inv_batch_stride, bool lookback) {
    if (T <= 0 || n_lags <= 0 || n_tiles <= 0) return hipSuccess;
    hipError_t attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_f16_kernel<true>), kGramF16LdsAsk);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((gram_f16_kernel<true>), dim3((unsigned)n_tiles, (unsigned)(n_batch > 0 ? n_batch : 1)), dim3(256),
                       kGramF16LdsAsk, s, reinterpret_cast<const _Float16*>(planes), T, FS, band, (int64_t)LP, tiles, n_lags,
                       plane_batch_stride, band_batch_stride, row_inv, inv_batch_stride, lookback ? 1 : 0);
    return hipGetLastError();
}

}  // namespace repet
