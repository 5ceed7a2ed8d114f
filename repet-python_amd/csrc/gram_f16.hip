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
// Same tiling as gram.hip: 128x128 tile per 256-thread workgroup (2x2 waves of 2x2 MFMA blocks), BK = 32, register
// prefetch two K-tiles ahead, double-buffered LDS, XCD-ordered upper-triangle tile list, mirror through LDS. The
// LDS image of a plane is [128 rows][4 chunks of 8 halves] with chunk c of row r stored at c ^ ((r >> 2) & 3): the
// 16-lane groups of ds_read_b128 (rows {0-3,12-15,20-27}, ...) and the 8-lane groups of ds_write_b128 then touch
// every bank once.
#include "common.h"

#include <hip/hip_fp16.h>

namespace repet {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int HBK = 32;                        // K elements per K-tile
constexpr int kPlaneHalves = kTile * HBK;      // one plane tile in LDS: 128 rows x 32 halves = 8 KB
constexpr int kGramF16Lds = 2 /*buffers*/ * 4 /*A hi, A lo, B hi, B lo*/ * kPlaneHalves * 2;   // 65,536 bytes
constexpr float kSplitScale = 128.0f;          // 2^7
constexpr float kUnscale = 1.0f / (kSplitScale * kSplitScale);

__global__ __launch_bounds__(256) void split_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ hi,
                                                        _Float16* __restrict__ lo, int64_t count) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= count) return;
    const float4 x = *reinterpret_cast<const float4*>(src + i);
    const float v[4] = {x.x * kSplitScale, x.y * kSplitScale, x.z * kSplitScale, x.w * kSplitScale};
    _Float16 h[4], l[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        h[k] = (_Float16)v[k];
        l[k] = (_Float16)(v[k] - (float)h[k]);
    }
    *reinterpret_cast<uint2*>(hi + i) = *reinterpret_cast<const uint2*>(h);
    *reinterpret_cast<uint2*>(lo + i) = *reinterpret_cast<const uint2*>(l);
}

__global__ __launch_bounds__(256) void gram_f16_kernel(const _Float16* __restrict__ Ah, const _Float16* __restrict__ Al,
                                                       int64_t T, int FS, float* __restrict__ out, int64_t pitch,
                                                       const int2* __restrict__ tiles) {
    extern __shared__ __attribute__((aligned(16))) _Float16 ldsh[];
    const int2 tile = tiles[blockIdx.x];
    const int bi = tile.x, bj = tile.y;
    if (bi < 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;

    const int64_t a_row0 = (int64_t)bi * kTile, b_row0 = (int64_t)bj * kTile;

    floatx16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // staging: a plane tile is 128 rows x 4 chunks of 16 bytes = 512 chunks, two per thread (rows r and r + 64)
    const int srow = tid >> 2, schunk = tid & 3;
    const unsigned g0 = (unsigned)(srow * FS + schunk * 8);               // halves, relative to the tile's first row
    const unsigned g1 = g0 + (unsigned)(64 * FS);
    const int l0 = srow * HBK + ((schunk ^ ((srow >> 2) & 3)) << 3);      // halves inside a plane tile
    const int l1 = l0 + 64 * HBK;                                          // (row + 64) has the same swizzle key
    float4 pah0, pah1, pal0, pal1, pbh0, pbh1, pbl0, pbl1;                // register set P
    float4 qah0, qah1, qal0, qal1, qbh0, qbh1, qbl0, qbl1;                // register set Q
#define F16_LOAD_TILE(S, kt)                                                          \
    {                                                                                 \
        const _Float16* ah = Ah + a_row0 * FS + (kt) * HBK;                           \
        const _Float16* al = Al + a_row0 * FS + (kt) * HBK;                           \
        const _Float16* bh = Ah + b_row0 * FS + (kt) * HBK;                           \
        const _Float16* bl = Al + b_row0 * FS + (kt) * HBK;                           \
        S##ah0 = *reinterpret_cast<const float4*>(ah + g0);                           \
        S##bh0 = *reinterpret_cast<const float4*>(bh + g0);                           \
        S##al0 = *reinterpret_cast<const float4*>(al + g0);                           \
        S##bl0 = *reinterpret_cast<const float4*>(bl + g0);                           \
        S##ah1 = *reinterpret_cast<const float4*>(ah + g1);                           \
        S##bh1 = *reinterpret_cast<const float4*>(bh + g1);                           \
        S##al1 = *reinterpret_cast<const float4*>(al + g1);                           \
        S##bl1 = *reinterpret_cast<const float4*>(bl + g1);                           \
    }
#define F16_STORE_TILE(S, buf)                                                        \
    {                                                                                 \
        _Float16* base = ldsh + (buf) * 4 * kPlaneHalves;                             \
        *reinterpret_cast<float4*>(base + 0 * kPlaneHalves + l0) = S##ah0;            \
        *reinterpret_cast<float4*>(base + 1 * kPlaneHalves + l0) = S##al0;            \
        *reinterpret_cast<float4*>(base + 2 * kPlaneHalves + l0) = S##bh0;            \
        *reinterpret_cast<float4*>(base + 3 * kPlaneHalves + l0) = S##bl0;            \
        *reinterpret_cast<float4*>(base + 0 * kPlaneHalves + l1) = S##ah1;            \
        *reinterpret_cast<float4*>(base + 1 * kPlaneHalves + l1) = S##al1;            \
        *reinterpret_cast<float4*>(base + 2 * kPlaneHalves + l1) = S##bh1;            \
        *reinterpret_cast<float4*>(base + 3 * kPlaneHalves + l1) = S##bl1;            \
    }
    // fragment of lane (lr, lh) for MFMA block row/col `blk` (0/1) of this wave and K-step `ks` (0/1): 8 halves
    // k = 16 ks + 8 lh .. +7 of tile row  w*64 + blk*32 + lr  -> chunk 2 ks + lh, swizzled by the row
#define F16_FRAG(plane_ptr, w, blk, ks)                                                                        \
    (*reinterpret_cast<const halfx8*>((plane_ptr) + ((w) * 64 + (blk) * 32 + lr) * HBK +                       \
                                      (((2 * (ks) + lh) ^ (((lr) >> 2) & 3)) << 3)))
#define F16_COMPUTE(buf)                                                                                        \
    {                                                                                                           \
        const _Float16* base = ldsh + (buf) * 4 * kPlaneHalves;                                                 \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                      \
            const halfx8 ah0 = F16_FRAG(base, wr, 0, ks), ah1 = F16_FRAG(base, wr, 1, ks);                      \
            const halfx8 al0 = F16_FRAG(base + kPlaneHalves, wr, 0, ks), al1 = F16_FRAG(base + kPlaneHalves, wr, 1, ks); \
            const halfx8 bh0 = F16_FRAG(base + 2 * kPlaneHalves, wc, 0, ks), bh1 = F16_FRAG(base + 2 * kPlaneHalves, wc, 1, ks); \
            const halfx8 bl0 = F16_FRAG(base + 3 * kPlaneHalves, wc, 0, ks), bl1 = F16_FRAG(base + 3 * kPlaneHalves, wc, 1, ks); \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh0, acc[0][0], 0, 0, 0);                    \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh1, acc[0][1], 0, 0, 0);                    \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh0, acc[1][0], 0, 0, 0);                    \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1, acc[1][1], 0, 0, 0);                    \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl0, acc[0][0], 0, 0, 0);                    \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl1, acc[0][1], 0, 0, 0);                    \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl0, acc[1][0], 0, 0, 0);                    \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1, acc[1][1], 0, 0, 0);                    \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh0, acc[0][0], 0, 0, 0);                    \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh1, acc[0][1], 0, 0, 0);                    \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh0, acc[1][0], 0, 0, 0);                    \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1, acc[1][1], 0, 0, 0);                    \
        }                                                                                                       \
    }

    const int nk = FS / HBK;                // FS is a multiple of 32; nk >= 2 for every supported window
    F16_LOAD_TILE(p, 0)
    F16_STORE_TILE(p, 0)
    if (nk > 1) F16_LOAD_TILE(q, 1)
    __syncthreads();
    int kt = 0;
    for (; kt + 1 < nk; kt += 2) {
        if (kt + 2 < nk) F16_LOAD_TILE(p, kt + 2)
        F16_COMPUTE(0)
        F16_STORE_TILE(q, 1)
        __syncthreads();
        if (kt + 3 < nk) F16_LOAD_TILE(q, kt + 3)
        F16_COMPUTE(1)
        if (kt + 2 < nk) F16_STORE_TILE(p, 0)
        __syncthreads();
    }
    if (kt < nk) F16_COMPUTE(0)
#undef F16_LOAD_TILE
#undef F16_STORE_TILE
#undef F16_FRAG
#undef F16_COMPUTE

    // ---- epilogue. acc[m][n][r]: i = wr*64 + m*32 + (r&3) + 8*(r>>2) + 4*lh ; j = wc*64 + n*32 + lr
    const int64_t gi0 = a_row0 + wr * 64;
    const int64_t gj0 = b_row0 + wc * 64;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[m][n][r] *= kUnscale;
                const int64_t gi = gi0 + m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int64_t gj = gj0 + n * 32 + lr;
                if (gi < T && gj < T) out[gi * pitch + gj] = acc[m][n][r];
            }
    if (bi != bj) {
        // mirror: transpose this wave's 64x64 block through a private LDS patch (pitch 65 floats = 16,640 B per
        // wave, 66,560 B in all: the launch asks for that much). Every wave must be done with the tile buffers.
        __syncthreads();
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
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the patch is wave-private
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < 64; ++j) {
            const int64_t gj = gj0 + j, gi = gi0 + lane;
            if (gj < T && gi < T) out[gj * pitch + gi] = patch[j * 65 + lane];
        }
    }
}

constexpr int kGramF16LdsAsk = 4 * 64 * 65 * 4 > kGramF16Lds ? 4 * 64 * 65 * 4 : kGramF16Lds;   // 66,560 bytes

}  // namespace

hipError_t launch_split_f16(const float* src, void* hi, void* lo, int64_t count, hipStream_t s) {
    if (count <= 0) return hipSuccess;
    hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)ceil_div(count, 1024)), dim3(256), 0, s, src,
                       reinterpret_cast<_Float16*>(hi), reinterpret_cast<_Float16*>(lo), count);
    return hipGetLastError();
}

hipError_t launch_gram_full_f16(const void* hi, const void* lo, int64_t T, int32_t FS, float* S, int64_t TS,
                                const int2* tiles, int32_t n_tiles, hipStream_t s) {
    if (T <= 0 || n_tiles <= 0) return hipSuccess;
    hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&gram_f16_kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, kGramF16LdsAsk);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(gram_f16_kernel, dim3((unsigned)n_tiles), dim3(256), kGramF16LdsAsk, s,
                       reinterpret_cast<const _Float16*>(hi), reinterpret_cast<const _Float16*>(lo), T, FS, S, TS, tiles);
    return hipGetLastError();
}

}  // namespace repet
