// K3h, large-tile form: the cosine self-similarity matrix of REPET-SIM (repet.py:1223) on the f16 matrix cores with a
// 256 x 256 tile per 512-thread workgroup and LDS-DMA staging.
//
// Same arithmetic as gram_f16.hip (every fp32 unit-row component as hi + lo f16 halves, hi hi' + hi lo' + lo hi' on
// v_mfma_f32_32x32x16_f16, fp32 accumulators, the three products of a K-step in the same order), so the matrix is
// bit-identical to that kernel's. What changes is what the 128 x 128 kernel was bound by -- the LDS, not the matrix cores:
//   per K-tile of 32 components it wrote 32 KB to LDS (415 cycles of the CU's 79 B/clk ds_write_b128 path) and its four
//   waves read 64 KB of fragments back (256 cycles) for 768 cycles of MFMA per SIMD; two workgroups per CU made that
//   1 340 LDS cycles against 768 MFMA cycles.
// A 256 x 256 tile (8 waves as 2 x 4, 128 x 64 of output per wave = 4 x 2 MFMA blocks) stages 64 KB per K-tile for FOUR
// times the flops and reads 24 KB of fragments per wave for 48 MFMAs: 1 600 LDS cycles against 3 072 MFMA cycles per
// SIMD. The staging itself goes global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write);
// the XOR-swizzled LDS image of gram_f16.hip is kept by permuting the per-lane SOURCE addresses (the LDS side of an
// LDS-DMA is lane-linear). Two LDS buffers, one barrier per K-tile: the DMA of tile k+1 flies during the 3 072-cycle
// compute of tile k (moving that barrier to the middle of the K-tile, with the next tile's first fragments fetched behind it
// while the second half is multiplied, was measured and is slower: K loop 141 k -> 155 k cycles -- the loop is held by
// the LDS, 192 KB of fragment reads + 64 KB of DMA per 3 072 MFMA cycles, not by the wait after the barrier).
// Upper-triangle tiles only, mirrored through per-wave LDS patches; diagonal tiles store the values
// computed for i <= j on both sides, so S is exactly symmetric.
//
// Round 3 -- gram_f16_big_pipe_kernel, the default: the same tiles, buffers, DMA pieces and MFMA order (still bit-identical),
// with the K loop rescheduled. What held the loop was neither the wait for the DMA to land (without the wait: the same
// 137 k cycles) nor the barrier (without it: slower, the waves drift), but instruction ISSUE in front of the MFMAs: the
// eight LDS-DMA instructions of a wave issued in one block behind the barrier (64 1-KB requests per workgroup backing up
// the address path, 22 k cycles) and the fragment reads in front of every half tile (7 k). Both now ride between the
// MFMAs, one instruction at a time, the fragments of the next half tile into a second register set: K loop 137 k ->
// 114.5 k cycles (MFMA alone 101 k, with the barriers 108 k). The epilogue classes every 64 x 64 patch by its own position
// instead of the tile's (diagonal and edge workgroups took 100 us against 80 and ended each round). cfg 2: 197 -> 170 us.
#include "common.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace repet {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

#ifdef REPET_GRAM_STAMPS
__device__ unsigned long long g_gram_stamps[4 * 8];
__device__ unsigned long long g_gram_span[2 * 4096];        // (start, end) of every workgroup, s_memrealtime (100 MHz, chip-wide)
#define GSPAN(k) if (tid == 0 && blockIdx.x < 4096) g_gram_span[2 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime();
#define GSTAMP(k) if (tid == 0 && (blockIdx.x % 61) == 7 && blockIdx.x / 61 < 8) g_gram_stamps[(blockIdx.x / 61) * 4 + (k)] = __builtin_amdgcn_s_memtime();
#else
#define GSTAMP(k)
#define GSPAN(k)
#endif

namespace {

constexpr int BT = 256;                       // tile edge
constexpr int HBK = 32;                       // K elements per K-tile
constexpr int kPlane = BT * HBK;              // halves of one plane tile in LDS: 256 rows x 32 halves = 16 KB
constexpr int kBuffer = 4 * kPlane;           // A hi, A lo, B hi, B lo = 64 KB
constexpr int kLoopLds = 2 * kBuffer * 2;     // bytes: two buffers = 131 072
constexpr int kPatchPitch = 68;               // floats per patch row: 16-byte aligned rows for float4 reads
constexpr int kPatchLds = 8 * 64 * kPatchPitch * 4;   // bytes: one 64 x 68 float patch per wave = 139 264
constexpr int kBigLds = kPatchLds > kLoopLds ? kPatchLds : kLoopLds;

// The rescheduled form (see the head of the file). Two waves per SIMD at 250 VGPRs: accumulators 128, two fragment sets 96.
// (The forms with the DMA pieces or the fragment reads issued in blocks -- 130 k / 115 k / 137 k cycles of K loop against
// 114.5 k -- and the round-2 kernel with one barrier per K-tile are in the history: round 3, `REPET_GRAM_PIPE`.)
// seg (nullable): the segment records of S's rows for the peak picking (peaks.h: PeakArgs::seg), written from the same LDS
// patches the stores read: lane l takes row l of the 64 x 64 patch, i.e. two segments of 32 columns.
__global__ __launch_bounds__(512) void gram_f16_big_pipe_kernel(const _Float16* __restrict__ planes, int64_t T, int FS,
                                                           float* __restrict__ out, int64_t pitch,
                                                           const int2* __restrict__ tiles, float* __restrict__ seg, int seg_pitch,
                                                           int stagger_ticks, int stagger_groups) {
    extern __shared__ __attribute__((aligned(16))) _Float16 lds_big[];
    const int2 tile = tiles[blockIdx.x];
    const int bi = tile.x, bj = tile.y;
    if (bi < 0) return;
    // Experiment of round 6 (REPET_GRAM_STAGGER_US = d, REPET_GRAM_STAGGER_GROUPS = g; default off): the workgroups of the FIRST
    // round start in g groups, group q held back by q d microseconds, so that the store bursts at the end of a round (all 256
    // workgroups leave their K loops together: 128 MB) arrive in g pieces under the other groups' MFMAs.
    if (stagger_ticks > 0 && blockIdx.x < 256u) {
        const int q = (int)((blockIdx.x >> 3) % (unsigned)stagger_groups);      // (the eight XCDs alike)
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(q * stagger_ticks);
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(32);
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                      // 2 x 4 waves: rows wr*128, columns wc*64
    const int lr = lane & 31, lh = lane >> 5;

    const int64_t a_row0 = (int64_t)bi * BT, b_row0 = (int64_t)bj * BT;
    const unsigned grow = (unsigned)(2 * FS);                     // halves per row of the interleaved global image

    floatx16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // The tiles and their LDS-DMA are those of gram_f16_big_kernel above (32 components, 64-byte row segments, two 64-KB
    // buffers; 32-byte segments in four buffers measured 218 us against 197: twice the requests per byte at the L2). What
    // changes is WHEN the fragments are read: those of a tile's second half behind the first eight MFMAs of its first
    // half, those of the NEXT tile's first half behind the first eight MFMAs of its second half -- into a second register
    // set, so no MFMA waits for an LDS read it was issued in front of (141 k cycles of K loop for 101 k of MFMA before).
    // A tile's buffer is read out completely during its first half: the one barrier per tile sits between the halves,
    // and the DMA of tile t+2 goes into that buffer right behind it (lead: one tile period, as before).
    const int prow = lane >> 2;                                                  // row inside the piece
    const _Float16* src_lane[8];
    int dst_piece[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int pc = wave * 8 + j;
        const int operand = pc >> 5, plane = (pc >> 4) & 1, rb = pc & 15;
        const int r = rb * 16 + prow;
        const int chunk = (lane & 3) ^ ((r >> 2) & 3);
        src_lane[j] = planes + ((operand ? b_row0 : a_row0) + r) * grow + plane * 32 + chunk * 8;
        dst_piece[j] = (operand * 2 + plane) * kPlane + rb * 16 * HBK;           // halves, wave-uniform
    }
    auto issue_piece = [&](int kt, int j) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_lane[j] + kt * 64),
                                         (__attribute__((address_space(3))) void*)(lds_big + (kt & 1) * kBuffer + dst_piece[j]), 16, 0, 0);
    };
    auto issue_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 8; ++j) issue_piece(kt, j);
    };
    auto frag = [&](const _Float16* plane_ptr, int row, int ks) -> halfx8 {
        return *reinterpret_cast<const halfx8*>(plane_ptr + row * HBK + (((2 * ks + lh) ^ ((row >> 2) & 3)) << 3));
    };
    struct Frags { halfx8 ah[4], al[4], bh[2], bl[2]; };
    // fragment q of 12, in the order the next half uses them: lo rows, hi columns, hi rows, lo columns
    auto load_frag = [&](int kt, int ks, Frags& f, int q) {
        const _Float16* base = lds_big + (kt & 1) * kBuffer;
        if (q < 4) f.al[q] = frag(base + kPlane, wr * 128 + q * 32 + lr, ks);
        else if (q < 6) f.bh[q - 4] = frag(base + 2 * kPlane, wc * 64 + (q - 4) * 32 + lr, ks);
        else if (q < 10) f.ah[q - 6] = frag(base, wr * 128 + (q - 6) * 32 + lr, ks);
        else f.bl[q - 10] = frag(base + 3 * kPlane, wc * 64 + (q - 10) * 32 + lr, ks);
    };
    // One half tile: the 24 MFMAs on `f` in the order of gram_f16.hip (lo hi', hi lo', hi hi': the same sums, bit for
    // bit), with the twelve fragment reads of the NEXT half (into `g`) behind MFMAs 0..11 and the eight LDS-DMA
    // instructions of a later tile behind every third. Issued in one block behind the barrier the DMA cost 22 k of the
    // loop's 137 k cycles and the reads 7 k (builds without the one, the other, both: 115 k / 130 k / 108 k; now 114.5 k):
    // sixty-four 1-KB requests per workgroup back up the address path while both waves of every SIMD stand in front of
    // their MFMAs. sched_barrier pins the interleave.
    auto half_tile = [&](const Frags& f, Frags& g, int g_kt, int g_ks, int dma_kt, bool dma) {
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            const int grp = i >> 3, m = (i >> 1) & 3, n = i & 1;
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(grp == 0 ? f.al[m] : f.ah[m], grp == 1 ? f.bl[n] : f.bh[n],
                                                                acc[m][n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (i < 12) load_frag(g_kt, g_ks, g, i);
            if (i % 3 == 1 && dma) issue_piece(dma_kt, i / 3);           // (wave-uniform)
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto tile_barrier = [&] {
        // every wave has read tile kt out of its buffer (the reads of its second half were issued in its first half and
        // are in registers: lgkmcnt(0)); tile kt+1, whose DMA was spread over the half tile before, has landed
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0x0070);             // vmcnt(0) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    const int nk = FS / HBK;
    GSTAMP(0)
    GSPAN(0)
    issue_tile(0);
    if (nk > 1) issue_tile(1);
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) __builtin_amdgcn_s_waitcnt(0x0F78);     // vmcnt(8): tile 0 has landed, tile 1 may be in flight
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frags fa, fb;
#pragma unroll
    for (int q = 0; q < 12; ++q) load_frag(0, 0, fa, q);
    for (int kt = 0; kt < nk; ++kt) {
        half_tile(fa, fb, kt, 1, 0, false);
        tile_barrier();
        // tile kt+2 goes into the buffer tile kt has left; behind the last tile the reads fetch fragments nobody uses
        half_tile(fb, fa, kt + 1 < nk ? kt + 1 : kt, 0, kt + 2, kt + 2 < nk);
    }
    __syncthreads();

    GSTAMP(1)
    // ---- epilogue. acc[m][n][r]: i = wr*128 + m*32 + (r&3) + 8*(r>>2) + 4*lh ; j = wc*64 + n*32 + lr.
    // Both copies of the block leave through the wave's LDS patch (64 rows x 68 floats) so that every global store is
    // 16 bytes per lane, four 256-byte rows per instruction (the 4-byte stores of the fragment layout were 256 store
    // instructions per lane -- the epilogue was store-issue-bound): first the 64 x 64 half as it is, then transposed.
    // The patches alias the tile buffers, which every wave left at the loop's last barrier. (Measured against it at cfg 2,
    // 0.197 ms: the natural copy straight from the accumulators as 4-byte stores, 128 contiguous bytes per half-wave --
    // 0.198; the mirror straight from the accumulators as 16-byte stores, registers 4q..4q+3 being four consecutive i of
    // row j, 32 bytes per row and instruction -- 0.211. With a quarter of the tiles in flight a workgroup's epilogue is
    // 15 us instead of 31: half of it is the chip-wide burst of 134 MB at the end of a round, not the store sequence.)
    const int64_t gi0 = a_row0 + wr * 128;
    const int64_t gj0 = b_row0 + wc * 64;
    constexpr float unscale = 1.0f / (128.0f * 128.0f);          // the 2^7 scale of both operands, exact
    float* patch = reinterpret_cast<float*>(lds_big) + wave * (64 * kPatchPitch);
    const int srow = lane >> 4, scol = (lane & 15) * 4;          // store role: row srow + 4 k of the patch, floats scol .. scol+3
    // Every 64 x 64 patch is classed by its own position (wave-uniform): whole inside the matrix and on the kept side of the
    // diagonal -> float4 stores without a test; whole outside or on the other side -> nothing to do, the transposition
    // through LDS included; only the patches the diagonal or the matrix's edge runs through take the tested path (with the
    // tests in the loop the compiler splits every float4 into a dword and a dwordx3 store under exec masks). Classed per
    // TILE, the 62 diagonal and edge workgroups of cfg 2 took 100 us against 80 and ended both rounds.
    // natural copy: S[gr][gc] kept where gr <= gc; mirror: where gr > gc.
    auto patch_class = [&](int64_t row0, int64_t col0, bool transposed) -> int {          // 0 skip, 1 plain, 2 tested
        if (row0 >= T || col0 >= T) return 0;
        const bool all = transposed ? row0 > col0 + 63 : row0 + 63 <= col0;
        const bool none = transposed ? row0 + 63 <= col0 : row0 > col0 + 63;
        if (none) return 0;
        return (all && row0 + 64 <= T && col0 + 64 <= T) ? 1 : 2;
    };
    auto store_rows = [&](int64_t row0, int64_t col0, bool transposed, int cls) {
        // patch row p, float q is S[row0 + p][col0 + q]
        if (cls == 1) {
            float* dst0 = out + (row0 + srow) * pitch + col0 + scol;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                // non-temporal: 240 MB that the peak picking reads a few lines of -- stored the ordinary way they push the segment
                // records and the spectrogram (which the column sort reads next) out of the caches (peaks + sort 0.255 -> 0.243 ms)
                typedef float f4 __attribute__((ext_vector_type(4)));
                const float4 v = *reinterpret_cast<const float4*>(patch + (srow + 4 * k) * kPatchPitch + scol);
                f4 y; y.x = v.x; y.y = v.y; y.z = v.z; y.w = v.w;
                __builtin_nontemporal_store(y, reinterpret_cast<f4*>(dst0 + (int64_t)(4 * k) * pitch));
            }
            return;
        }
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int p = srow + 4 * k;
            const float4 v = *reinterpret_cast<const float4*>(patch + p * kPatchPitch + scol);
            const int64_t gr = row0 + p, gc = col0 + scol;
            if (gr >= T) continue;
            float* dst = out + gr * pitch + gc;
            const float vals[4] = {v.x, v.y, v.z, v.w};
            const bool all = gc + 3 < T && (transposed ? gr > gc + 3 : gr <= gc);
            if (all) *reinterpret_cast<float4*>(dst) = v;
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool keep = gc + e < T && (transposed ? gr > gc + e : gr <= gc + e);
                    if (keep) dst[e] = vals[e];
                }
            }
        }
    };
    // Segment records of the patch's rows (patch row p, float q = S[row0 + p][col0 + q]): lane l owns row l, columns 0 .. 31
    // are one aligned segment and 32 .. 63 the next. Rows at pitch 68 floats: the sixteen lanes of a ds_read_b128 group
    // fall on sixteen different bank quads. NaN counts as +inf (v_min_f32 returns the other operand for a quiet NaN).
    // The patch the diagonal runs through holds one half of a symmetric block in each copy: its records are written from
    // the natural copy, element (p, q) read at (min, max).
    struct Top { float m1, m2; int at; };
    auto top_insert = [](Top& t, float raw, int at) {
        float val;
        asm("v_min_f32 %0, 0x7f800000, %1" : "=v"(val) : "v"(raw));
        t.m2 = __builtin_amdgcn_fmed3f(t.m1, t.m2, val);
        t.at = val > t.m1 ? at : t.at;
        asm("v_max_f32 %0, %1, %2" : "=v"(t.m1) : "v"(t.m1), "v"(val));     // (fmaxf: a canonicalising v_max_f32 val, val in front)
    };
    auto emit_segments = [&](int64_t row0, int64_t col0, bool transposed, int cls) {
        if (!seg) return;
        const bool on_diag = row0 == col0;
        if (on_diag && transposed) return;
        Top t0{-INFINITY, -INFINITY, 0}, t1{-INFINITY, -INFINITY, 0};       // (two named triples: an array indexed by the loop
        if (cls == 1) {                                                      // variable of the slow path went to scratch)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 x = *reinterpret_cast<const float4*>(patch + lane * kPatchPitch + 4 * q);
                top_insert(t0, x.x, 4 * q); top_insert(t0, x.y, 4 * q + 1); top_insert(t0, x.z, 4 * q + 2); top_insert(t0, x.w, 4 * q + 3);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 x = *reinterpret_cast<const float4*>(patch + lane * kPatchPitch + 32 + 4 * q);
                top_insert(t1, x.x, 4 * q); top_insert(t1, x.y, 4 * q + 1); top_insert(t1, x.z, 4 * q + 2); top_insert(t1, x.w, 4 * q + 3);
            }
        } else {
            auto element = [&](int q) {                                      // (p, q) of the symmetric block at (min, max)
                const int pr = on_diag ? min(lane, q) : lane, pc = on_diag ? max(lane, q) : q;
                return patch[pr * kPatchPitch + pc];
            };
#pragma unroll 1
            for (int q = 0; q < 32 && col0 + q < T; ++q) top_insert(t0, element(q), q);
#pragma unroll 1
            for (int q = 32; q < 64 && col0 + q < T; ++q) top_insert(t1, element(q), q - 32);
        }
        if (row0 + lane < T) {
            float* rec = seg + (row0 + lane) * 3 * (int64_t)seg_pitch + (col0 >> 5);
            *reinterpret_cast<float2*>(rec) = make_float2(t0.m1, t1.m1);
            *reinterpret_cast<float2*>(rec + seg_pitch) = make_float2(t0.m2, t1.m2);
            *reinterpret_cast<int2*>(rec + 2 * seg_pitch) = make_int2(t0.at, t1.at);
        }
    };
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        // natural: patch[i][j]
        const int cn = patch_class(gi0 + mh * 64, gj0, false);
        if (cn) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = mm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        patch[i * kPatchPitch + n * 32 + lr] = acc[2 * mh + mm][n][r] * unscale;
                    }
            __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): the patch is wave-private
            __builtin_amdgcn_wave_barrier();
            store_rows(gi0 + mh * 64, gj0, false, cn);
            emit_segments(gi0 + mh * 64, gj0, false, cn);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
        // mirror: patch[j][i]
        const int cm = patch_class(gj0, gi0 + mh * 64, true);
        if (cm) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int i = mm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        patch[(n * 32 + lr) * kPatchPitch + i] = acc[2 * mh + mm][n][r] * unscale;
                    }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            store_rows(gj0, gi0 + mh * 64, true, cm);
            emit_segments(gj0, gi0 + mh * 64, true, cm);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    }
    GSTAMP(2)
    GSTAMP(3)
    GSPAN(1)
}

}  // namespace

#ifdef REPET_GRAM_STAMPS
extern "C" int repet_debug_gram_spans(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(repet::g_gram_span), sizeof(unsigned long long) * 2 * n);
}
extern "C" int repet_debug_gram_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gram_stamps), sizeof(unsigned long long) * 32);
}
#endif

int gram_big_tile() { return BT; }

hipError_t launch_gram_full_f16_big(const void* planes, int64_t T, int32_t FS, float* S, int64_t TS,
                                    const int2* tiles, int32_t n_tiles, hipStream_t s, float* seg, int32_t seg_pitch) {
    if (T <= 0 || n_tiles <= 0) return hipSuccess;
    hipError_t attr = ensure_dynamic_lds(reinterpret_cast<const void*>(&gram_f16_big_pipe_kernel), kBigLds);
    if (attr != hipSuccess) return attr;
    static const int stagger_ticks = [] { const char* e = getenv("REPET_GRAM_STAGGER_US"); return e ? (int)(100.0 * atof(e)) : 0; }();
    static const int stagger_groups = [] { const char* e = getenv("REPET_GRAM_STAGGER_GROUPS"); const int g = e ? atoi(e) : 2; return g < 2 ? 2 : g; }();
    hipLaunchKernelGGL(gram_f16_big_pipe_kernel, dim3((unsigned)n_tiles), dim3(512), kBigLds, s,
                       reinterpret_cast<const _Float16*>(planes), T, FS, S, TS, tiles, seg, seg_pitch, stagger_ticks, stagger_groups);
    return hipGetLastError();
}

}  // namespace repet
