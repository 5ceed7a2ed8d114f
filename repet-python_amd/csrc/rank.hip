// Rank transform of the magnitude spectrogram for the rank-domain median of `sim` (mask.hip, mask_sim_rank_kernel).
//
// np.median over the similar frames of a bin (_simmask, repet.py:1511-1545) is a SELECTION: it only needs the order
// of the values. So every column (one frequency bin over all T frames of a channel) is sorted once per clip:
//   R[c][t][f]  = 0x0400 + number of frames whose magnitude in bin f is smaller than frame t's   (16-bit rank code)
//   Vs[c][f][r] = the r-th smallest magnitude of bin f                                            (rank -> value)
// The mask kernel then gathers 2-byte codes instead of 4-byte floats, runs the selection network on TWO bins per
// register with packed 16-bit min/max (half the instructions per bin -- the kernel is VALU-issue-bound), and turns the
// one or two middle codes back into magnitudes with a lookup in Vs. Equal magnitudes share a code (the count of
// strictly smaller values), so the result is bit-identical to selecting on the floats. Codes start at 0x0400 so that
// they are positive NORMAL f16 bit patterns: unsigned-integer order and f16 order agree and the three-input
// v_pk_minimum3_f16 / v_pk_maximum3_f16 can be mixed with v_pk_min_u16 / v_pk_max_u16 (pads: 0 and 0x7C00 = +inf).
//
// One workgroup sorts one column: N = 2^LOG2N >= T keys (pads +inf), 32 keys per thread, bitonic network. A thread
// first sorts its 32 contiguous keys in registers; every later merge phase runs in "trips" through LDS: a thread
// fetches 32 keys whose indices differ in bits {0,1} and three higher bits (8 float4 loads), does up to three stages
// in registers and stores them back; the closing trip of a phase takes bits 0..4 (five stages). The first stage of a
// phase is the "flip" form (partner = index with all lower bits inverted), realised as an address inversion of the
// loads, so that every comparator is ascending and no direction flags exist. LDS rows of 32 keys are padded by 4.
#include "common.h"

namespace repet {

namespace {

__device__ __forceinline__ void cex(unsigned& a, unsigned& b) {
    const unsigned lo = a < b ? a : b, hi = a < b ? b : a;
    a = lo; b = hi;
}
// physical LDS index of key i: one float4 of padding behind every 32 keys (conflict-free for both trip shapes)
__device__ __forceinline__ int phys(int i) { return i + ((i >> 5) << 2); }

template <int QHI, int QLO, int HB>
__device__ __forceinline__ void stages_high(unsigned (&r)[8][4]) {
#pragma unroll
    for (int q = QHI; q >= QLO; --q) {
        const int bit = 1 << (q - HB);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (!(u & bit)) {
#pragma unroll
                for (int w = 0; w < 4; ++w) cex(r[u][w], r[u | bit][w]);
            }
    }
}

// One trip over the stages QHI..QLO (all >= 5) of merge phase P. FLIP: QHI is the first stage of the phase.
template <int LOG2N, int P, int QHI, int QLO, bool FLIP>
__device__ __forceinline__ void trip_high(unsigned* s, int tid) {
    constexpr int HB = QLO < LOG2N - 3 ? QLO : LOG2N - 3;      // the thread's three high bits are [HB, HB+3)
    static_assert(HB >= 5 && QHI < HB + 3 && QLO >= HB, "stage bits must lie inside the thread's bit group");
    const int low = tid & ((1 << (HB - 2)) - 1), high = tid >> (HB - 2);
    const int base = (low << 2) | (high << (HB + 3));
    unsigned r[8][4];
    __syncthreads();                                           // the previous trip's stores
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int i0 = base | (u << HB);
        if (FLIP && ((u >> (P - 1 - HB)) & 1)) {               // upper half of a 2^P block: mirrored below bit P-1
            const int src = (i0 ^ ((1 << (P - 1)) - 1)) & ~3;
            const uint4 v = *reinterpret_cast<const uint4*>(s + phys(src));
            r[u][0] = v.w; r[u][1] = v.z; r[u][2] = v.y; r[u][3] = v.x;
        } else {
            const uint4 v = *reinterpret_cast<const uint4*>(s + phys(i0));
            r[u][0] = v.x; r[u][1] = v.y; r[u][2] = v.z; r[u][3] = v.w;
        }
    }
    if (FLIP) __syncthreads();                                 // mirrored reads touch other threads' keys: all read first
    stages_high<QHI, QLO, HB>(r);
#pragma unroll
    for (int u = 0; u < 8; ++u)
        *reinterpret_cast<uint4*>(s + phys(base | (u << HB))) = make_uint4(r[u][0], r[u][1], r[u][2], r[u][3]);
}

template <int LOG2N, int P, int QHI>
struct HighTrips {
    static __device__ __forceinline__ void run(unsigned* s, int tid) {
        constexpr int QLO = QHI - 2 > 5 ? QHI - 2 : 5;
        trip_high<LOG2N, P, QHI, QLO, QHI == P - 1>(s, tid);
        if constexpr (QLO > 5) HighTrips<LOG2N, P, QLO - 1>::run(s, tid);
    }
};

// Round 6: trips of up to FIVE stages. The trip above takes its 32 keys as 8 x 4 (three stage bits and the two lowest index
// bits, so that every access is 16 bytes): three stages per trip through LDS, 23 trips for 2^13 keys. A thread may as well hold
// 16 x 2 (four stage bits, 8-byte accesses) or 32 x 1 (five stage bits, 4-byte accesses): the LDS moves the same bytes per
// trip whatever the access width, and the merge phases 9 .. 13 then need one trip less each (19 trips). Built, validated
// (the index algebra of both forms in NumPy for 2^11 .. 2^15 keys, the rank tests on the GPU) and measured: no faster (see
// launch_rank_n) -- kept behind REPET_RANK_TRIPS=5.
// NB = stage bits of the trip (3, 4, 5), VW = 32 >> NB consecutive keys per access.
template <int LOG2N, int P, int QHI, int QLO, int NB, bool FLIP>
__device__ __forceinline__ void trip_high_n(unsigned* s, int tid) {
    constexpr int VW = 32 >> NB, LV = NB == 3 ? 2 : (NB == 4 ? 1 : 0), NU = 1 << NB;
    constexpr int HB = QLO < LOG2N - NB ? QLO : LOG2N - NB;      // the thread's NB high bits are [HB, HB + NB)
    static_assert(NB >= 3 && NB <= 5 && HB >= 5 && QHI < HB + NB && QLO >= HB, "stage bits must lie inside the thread's bit group");
    const int low = tid & ((1 << (HB - LV)) - 1), high = tid >> (HB - LV);
    const int base = (low << LV) | (high << (HB + NB));
    unsigned r[NU][VW];
    __syncthreads();                                           // the previous trip's stores
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i0 = base | (u << HB);
        const bool mirrored = FLIP && ((u >> (P - 1 - HB)) & 1);   // upper half of a 2^P block: read mirrored below bit P - 1
        const int src = mirrored ? ((i0 ^ ((1 << (P - 1)) - 1)) & ~(VW - 1)) : i0;
        if constexpr (VW == 4) {
            const uint4 v = *reinterpret_cast<const uint4*>(s + phys(src));
            if (mirrored) { r[u][0] = v.w; r[u][1] = v.z; r[u][2] = v.y; r[u][3] = v.x; }
            else { r[u][0] = v.x; r[u][1] = v.y; r[u][2] = v.z; r[u][3] = v.w; }
        } else if constexpr (VW == 2) {
            const uint2 v = *reinterpret_cast<const uint2*>(s + phys(src));
            if (mirrored) { r[u][0] = v.y; r[u][1] = v.x; } else { r[u][0] = v.x; r[u][1] = v.y; }
        } else r[u][0] = s[phys(src)];
    }
    if (FLIP) __syncthreads();                                 // mirrored reads touch other threads' keys: all read first
#pragma unroll
    for (int q = QHI; q >= QLO; --q) {
        const int bit = 1 << (q - HB);
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (!(u & bit)) {
#pragma unroll
                for (int w = 0; w < VW; ++w) cex(r[u][w], r[u | bit][w]);
            }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        unsigned* dst = s + phys(base | (u << HB));
        if constexpr (VW == 4) *reinterpret_cast<uint4*>(dst) = make_uint4(r[u][0], r[u][1], r[u][2], r[u][3]);
        else if constexpr (VW == 2) *reinterpret_cast<uint2*>(dst) = make_uint2(r[u][0], r[u][1]);
        else dst[0] = r[u][0];
    }
}
// how many of the n high stages (n >= 1) the next trip takes: all of them up to five, else about half (6: 3 + 3, 7: 4 + 3, 8: 4 + 4, 9: 5 + 4, 10: 5 + 5)
constexpr int wide_trip_stages(int n) { return n <= 5 ? n : ((n + 1) / 2 > 5 ? 5 : (n + 1) / 2); }
template <int LOG2N, int P, int QHI>
struct WideTrips {
    static __device__ __forceinline__ void run(unsigned* s, int tid) {
        constexpr int n = QHI - 4;                             // stages QHI .. 5
        constexpr int take = wide_trip_stages(n);
        constexpr int QLO = QHI - take + 1;
        constexpr int NB = take <= 3 ? 3 : take;
        trip_high_n<LOG2N, P, QHI, QLO, NB, QHI == P - 1>(s, tid);
        if constexpr (QLO > 5) WideTrips<LOG2N, P, QLO - 1>::run(s, tid);
    }
};

// stages 4..0 on 32 contiguous keys in registers
__device__ __forceinline__ void stages_low(unsigned (&k)[32]) {
#pragma unroll
    for (int q = 4; q >= 0; --q) {
#pragma unroll
        for (int i = 0; i < 32; ++i)
            if (!(i & (1 << q))) cex(k[i], k[i | (1 << q)]);
    }
}

__device__ __forceinline__ void load_run(const unsigned* s, int tid, unsigned (&k)[32]) {
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const uint4 x = *reinterpret_cast<const uint4*>(s + phys(tid * 32 + v * 4));
        k[4 * v] = x.x; k[4 * v + 1] = x.y; k[4 * v + 2] = x.z; k[4 * v + 3] = x.w;
    }
}
__device__ __forceinline__ void store_run(unsigned* s, int tid, const unsigned (&k)[32]) {
#pragma unroll
    for (int v = 0; v < 8; ++v)
        *reinterpret_cast<uint4*>(s + phys(tid * 32 + v * 4)) = make_uint4(k[4 * v], k[4 * v + 1], k[4 * v + 2], k[4 * v + 3]);
}

template <int LOG2N, int P, bool WIDE>
struct Phases {
    static __device__ __forceinline__ void run(unsigned* s, int tid, unsigned (&k)[32]) {
        if constexpr (WIDE) WideTrips<LOG2N, P, P - 1>::run(s, tid);
        else HighTrips<LOG2N, P, P - 1>::run(s, tid);
        __syncthreads();
        load_run(s, tid, k);
        stages_low(k);
        store_run(s, tid, k);
        if constexpr (P < LOG2N) Phases<LOG2N, P + 1, WIDE>::run(s, tid, k);
    }
};

}  // namespace

// Tiled transposes either side of the column sort: the spectrogram is frame-major (a column is one float every FS),
// the sort wants whole columns. in[c][t][FS] -> out[c * n_cols + f][pitch] (fp32, 64 x 64 tiles through LDS).
// TF = frames per tile. 64 (the round-4 form): 16.6 KB of LDS per workgroup, eight workgroups per CU hold 133 KB. 32 (round 6):
// half of that for the same 256 threads -- the stage this kernel opens is bound by the chip's LDS, and its
// workgroups have to find room beside the first pass of the peak picking. A column's 32 frames still leave as one 128-byte line.
template <int TF>
__global__ __launch_bounds__(256) void columns_from_rows_kernel(RankArgs a) {
    static_assert(TF == 64 || TF == 32, "tile height");
    constexpr int kRounds = TF / 16;                              // frame rows per thread on the way in
    constexpr int kOutLanes = TF / 4;                             // lanes that cover a column's frames on the way out
    constexpr int kOutRounds = 64 * kOutLanes / 256;              // columns per thread on the way out
    __shared__ float tile[TF][65];
    const int c = blockIdx.z, f0 = blockIdx.y * 64;
    const int64_t t0 = (int64_t)blockIdx.x * TF;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;       // 16 bytes per lane both ways: 4 bins in, 4 frames out
    const float* in = a.V + c * a.chan_stride;
    // all loads first (clamped row: a load behind "t < T ?" sits in a branch and is waited for at its join), the choice
    // afterwards
    float4 r[kRounds];
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        const int64_t t = t0 + ty + 16 * k;
        r[k] = *reinterpret_cast<const float4*>(in + (t < a.T ? t : a.T - 1) * a.FS + f0 + 4 * tx);
    }
#pragma unroll
    for (int k = 0; k < kRounds; ++k) {
        const bool live = t0 + ty + 16 * k < a.T;
        float* row = tile[ty + 16 * k] + 4 * tx;                   // (pitch 65: the four words go one by one, conflict-free both ways)
        row[0] = live ? r[k].x : 0.f; row[1] = live ? r[k].y : 0.f; row[2] = live ? r[k].z : 0.f; row[3] = live ? r[k].w : 0.f;
    }
    __syncthreads();
    float* out = a.Vs + ((int64_t)c * a.n_cols + f0) * a.vs_pitch;
    // thread -> (4 frames ox, column): a wave's reads of the tile must fall on 64 different banks (row pitch 65: bank = 4 ox + column)
    const int ox = threadIdx.x % kOutLanes, oy = threadIdx.x / kOutLanes;
    const int col0 = TF == 64 ? oy : (oy & 3) + 32 * ((oy >> 2) & 1) + 4 * (threadIdx.x >> 6);
    if (t0 + 4 * ox < a.vs_pitch) {
#pragma unroll
        for (int k = 0; k < kOutRounds; ++k) {
            const int f = col0 + 16 * k;
            *reinterpret_cast<float4*>(out + (int64_t)f * a.vs_pitch + t0 + 4 * ox) =
                make_float4(tile[4 * ox][f], tile[4 * ox + 1][f], tile[4 * ox + 2][f], tile[4 * ox + 3][f]);
        }
    }
}

// codes[c * n_cols + f][pitch] (u16, column-major) -> R[c][t][FS]. Work unit: a 2 x 2 block of codes = one dword in
// (frames t, t+1 of bin f) and one dword out (bins f, f+1 of frame t); 64 x 64 dword tiles through LDS.
__global__ __launch_bounds__(256) void rows_from_code_columns_kernel(RankArgs a) {
    __shared__ unsigned tile[2][64][65];                // [bin parity][bin pair][frame pair]: conflict-free both ways
    const int c = blockIdx.z, f0 = blockIdx.y * 128;
    const int64_t t0 = (int64_t)blockIdx.x * 128;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const unsigned* in = reinterpret_cast<const unsigned*>(a.codes + ((int64_t)c * a.n_cols + f0) * a.vs_pitch + t0);
    const bool in_range = t0 + 2 * tx < a.vs_pitch;      // vs_pitch is even; codes of frames >= T are never stored
    unsigned r[32];                                     // (loads first, choice afterwards: as above)
    const int txc = in_range ? tx : 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) r[k] = in[(int64_t)(ty + 4 * k) * (a.vs_pitch / 2) + txc];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int fl = ty + 4 * k;
        tile[fl & 1][fl >> 1][tx] = in_range ? r[k] : 0u;
    }
    __syncthreads();
    unsigned* out = reinterpret_cast<unsigned*>(a.R + c * a.r_chan_stride + f0);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int tp = ty + 4 * k;                       // frame pair of the tile; lane tx = bin pair (2 tx, 2 tx + 1)
        const unsigned lo = tile[0][tx][tp], hi = tile[1][tx][tp];
        const int64_t t = t0 + 2 * tp;
        if (t < a.T) out[t * (a.FS / 2) + tx] = (lo & 0xffffu) | (hi << 16);
        if (t + 1 < a.T) out[(t + 1) * (a.FS / 2) + tx] = (lo >> 16) | (hi & 0xffff0000u);
    }
}

// The plane words of the bit-sliced selection (MaskArgs::P) come out of a 16 x 16 bit transpose: with codes b and b + 16 in the
// two halves of register b, transposing both halves at once gives, in register p, bit p of all 32 codes -- bit i of half h of
// register p = bit p of code 16 h + i.
template <int J>
__device__ __forceinline__ void plane_transpose_stage(unsigned (&x)[16]) {
    constexpr unsigned m = J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (!(k & J)) {
            const unsigned t = ((x[k] >> J) ^ x[k + J]) & m;
            x[k + J] ^= t;
            x[k] ^= t << J;
        }
}
// codes[c * n_cols + f][pitch] (column-major, as the sort leaves them) -> P[t][plane][64] without the detour over R (the
// bit-sliced selection reads the planes only): a workgroup of 1 024 threads takes 64 frames and the 32 cells 64 b + l of 32
// values of l -- 1 024 columns, 128 bytes of each, through LDS (column pitch 66 codes: lanes on adjacent l read adjacent
// banks) -- and a thread builds the plane words of two (frame, l) with the transpose above, lanes adjacent in l so that a
// wave writes two 128-byte halves of plane rows.
// PF = frames per workgroup: 64 (round 4: 135 KB of LDS, one workgroup per CU -- it cannot start on a CU before the column sort
// has left it altogether) or 32 (round 6: 70 KB, two per CU, a workgroup fits as soon as two of a CU's four sort workgroups
// are gone; a plane row's words are still written 128 bytes at a time).
template <int PF>
__global__ __launch_bounds__(1024) void code_planes_from_columns_kernel(RankArgs a) {
    constexpr int kPlaneFrames = PF, kPlaneColPitch = PF + 2, kPieces = PF / 8;
    extern __shared__ unsigned short plane_lds[];                 // [1024 columns][PF + 2]
    const int64_t t0 = (int64_t)blockIdx.x * kPlaneFrames;
    const int l0 = blockIdx.y * 32;
    const int bpc = a.n_cols >> 6, n_bits = a.n_channels * bpc;
    // column q of the tile: bit b = q / 32, l = l0 + q % 32 -> cell 64 b + l = channel b / bpc, bin 64 (b % bpc) + l.
    // 16-byte pieces (8 frames of one column), eight per thread, all loads first
    uint4 v[kPieces];
#pragma unroll
    for (int k = 0; k < kPieces; ++k) {
        const int i = threadIdx.x + 1024 * k, q = i / kPieces, piece = i % kPieces;
        const int b = q >> 5, l = l0 + (q & 31);
        const int64_t t = t0 + piece * 8;
        const bool in = b < n_bits && t < a.vs_pitch;
        const int c = in ? b / bpc : 0, f = in ? (b - c * bpc) * 64 + l : 0;
        v[k] = *reinterpret_cast<const uint4*>(a.codes + ((int64_t)c * a.n_cols + f) * a.vs_pitch + (in ? t : 0));
        if (!in) v[k] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int k = 0; k < kPieces; ++k) {
        const int i = threadIdx.x + 1024 * k, q = i / kPieces, piece = i % kPieces;
        unsigned* dst = reinterpret_cast<unsigned*>(plane_lds + q * kPlaneColPitch + piece * 8);      // 4-byte aligned (even pitch)
        dst[0] = v[k].x; dst[1] = v[k].y; dst[2] = v[k].z; dst[3] = v[k].w;
    }
    __syncthreads();
    const int l = threadIdx.x & 31;
#pragma unroll
    for (int k = 0; k < PF / 32; ++k) {
        const int tl = (threadIdx.x >> 5) + 32 * k;
        const int64_t t = t0 + tl;
        if (t >= a.T) continue;
        unsigned x[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const unsigned lo = b < n_bits ? plane_lds[(b * 32 + l) * kPlaneColPitch + tl] - kRankCodeBase : 0u;
            const unsigned hi = b + 16 < n_bits ? plane_lds[((b + 16) * 32 + l) * kPlaneColPitch + tl] - kRankCodeBase : 0u;
            x[b] = (lo & 0xffffu) | (hi << 16);
        }
        plane_transpose_stage<8>(x); plane_transpose_stage<4>(x); plane_transpose_stage<2>(x); plane_transpose_stage<1>(x);
        unsigned* out = a.P + t * (int64_t)a.n_planes * 64 + l0 + l;
#pragma unroll
        for (int p = 0; p < 15; ++p)
            if (p < a.n_planes) out[p * 64] = x[p];
    }
}

#ifndef REPET_RANK_LEAF_BITS
#define REPET_RANK_LEAF_BITS 4
#endif
// the tree ends at leaves of 2^bits keys (<= 5: inside a padded row); the 1 024-thread workgroup of N = 2^15 has 128 registers
// per thread and keeps the one-node-per-thread form
constexpr int rank_leaf_bits(int log2n) { return log2n >= 15 ? 5 : REPET_RANK_LEAF_BITS; }

template <int LOG2N, bool WIDE>
__global__ __launch_bounds__((1 << LOG2N) / 32) void rank_columns_kernel(RankArgs a) {
    constexpr int N = 1 << LOG2N, THREADS = N / 32;
    extern __shared__ uint4 rank_lds[];
    unsigned* s = reinterpret_cast<unsigned*>(rank_lds);
    const int64_t col = blockIdx.x;
    const int tid = threadIdx.x;
    // the column (T floats, contiguous after columns_from_rows_kernel) is replaced by its sorted self at the end.
    // Thread tid takes keys 4 tid + 4 THREADS j + {0..3}, j < 8 (any 32 do for the first five merge phases): whole
    // float4 loads, coalesced across the workgroup. Frames >= T lie outside the resource (0) and become +inf keys.
    float* column = a.Vs + col * a.vs_pitch;
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(column, 0, (int)(a.T * 4), 0x00020000);
    unsigned k[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = 4 * tid + 4 * THREADS * j;
        const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, t * 4, 0, 0));
        k[4 * j] = t < a.T ? v.x : 0x7f800000u;          // magnitudes are >= 0: unsigned order == float order
        k[4 * j + 1] = t + 1 < a.T ? v.y : 0x7f800000u;
        k[4 * j + 2] = t + 2 < a.T ? v.z : 0x7f800000u;
        k[4 * j + 3] = t + 3 < a.T ? v.w : 0x7f800000u;
    }
    // merge phases 1..5 inside the thread's 32 keys
#pragma unroll
    for (int p = 1; p <= 5; ++p) {
#pragma unroll
        for (int i2 = 0; i2 < 32; ++i2)
            if (!(i2 & (1 << (p - 1)))) cex(k[i2], k[i2 ^ ((1 << p) - 1)]);
#pragma unroll
        for (int q = p - 2; q >= 0; --q) {
#pragma unroll
            for (int i2 = 0; i2 < 32; ++i2)
                if (!(i2 & (1 << q))) cex(k[i2], k[i2 | (1 << q)]);
        }
    }
    store_run(s, tid, k);
    Phases<LOG2N, 6, WIDE>::run(s, tid, k);
    // code of every original key: 0x0400 + lower_bound(sorted, key), 32 independent binary searches per thread. The
    // keys are fetched again (cache-resident: this workgroup read them a few microseconds ago) rather than held in 32
    // registers through the whole sort.
    unsigned orig[32], pos[32];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = 4 * tid + 4 * THREADS * j;
        const uint4 v = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, t * 4, 0, 0));
        orig[4 * j] = v.x; orig[4 * j + 1] = v.y; orig[4 * j + 2] = v.z; orig[4 * j + 3] = v.w;
    }
    // The search runs in two parts. Which LEAF of 16 keys: the levels with large steps probe keys that, in the padded
    // array, all fall into one or two banks (sixteen distinct addresses of one bank at the level of step 256: a
    // sixteen-fold replay); a copy of those N/16 - 1 keys in breadth-first order (node n: children 2n, 2n+1) puts the
    // 2^l probes of level l on consecutive words instead. Then four levels inside the leaf, on PHYSICAL indices biased
    // by the offset of the level's probe (key pos + step - 1): a probe is a ds_read of pos[e] itself, a compare, a
    // select and one three-input add (the move to the next level's bias). (cfg 2, the three kernels of the column sort:
    // no tree 0.191 ms; leaves of 32 keys 0.151, of 16 0.148, of 8 0.147 with 4 KB of tree.)
    constexpr int R = rank_leaf_bits(LOG2N), kTreeLevels = LOG2N - R, kNodes = N >> R;
    unsigned* tree = s + (N + N / 8);
    __syncthreads();                                           // the sorted column is complete
#pragma unroll
    for (int n = tid; n < kNodes; n += THREADS)
        if (n > 0) {
            const int l = 31 - __clz(n), j = n - (1 << l);     // node n: level l, j-th from the left
            tree[n] = s[phys((((2 * j + 1) << (LOG2N - 1 - l))) - 1)];
        }
    __syncthreads();
    {
#pragma unroll
        for (int e = 0; e < 32; ++e) pos[e] = 1;
#pragma unroll 1
        for (int l = 0; l < kTreeLevels; ++l) {
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const unsigned probe = tree[pos[e]];
                pos[e] = 2 * pos[e] + (probe < orig[e] ? 1 : 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 32; ++e) {                          // leaf -> physical index of its probe of step 2^(R-1)
            const int g = pos[e] - kNodes;
            pos[e] = (g << R) + ((g >> (5 - R)) << 2) + (1 << (R - 1)) - 1;
        }
#pragma unroll
        for (int m = R - 1; m >= 0; --m) {
            const int step = 1 << m, d = m > 0 ? -((step >> 1) + 0) : 0;      // (step/2 - 1) - (step - 1)
#pragma unroll
            for (int e = 0; e < 32; ++e) {
                const unsigned probe = s[pos[e]];
                pos[e] += (probe < orig[e] ? step : 0) + d;
            }
        }
    }
    // back to logical positions: physical p = i + 4 (i / 32)  =>  i = p - 4 (p / 36)
#pragma unroll
    for (int e = 0; e < 32; ++e) pos[e] -= 4 * (pos[e] / 36);
    // codes of frames 4 tid + 4 THREADS j + {0..3}: one 8-byte store each (frames >= T are dropped by the resource)
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.codes + col * a.vs_pitch, 0, (int)(round_up(a.T, 4) * 2), 0x00020000);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = 4 * tid + 4 * THREADS * j;
        const unsigned lo = (kRankCodeBase + pos[4 * j]) | ((kRankCodeBase + pos[4 * j + 1]) << 16);
        const unsigned hi = (kRankCodeBase + pos[4 * j + 2]) | ((kRankCodeBase + pos[4 * j + 3]) << 16);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned, make_uint2(lo, hi)),
                                              c_rsrc, t * 2, 0, 0);
    }
    // the sorted column over the original one: rank -> value table of the mask kernel (every thread has read its keys)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = 4 * tid + 4 * THREADS * j;
        if (t < a.vs_pitch) *reinterpret_cast<uint4*>(column + t) = *reinterpret_cast<const uint4*>(s + phys(t));
    }
}

__global__ void fill_rank_pad_rows_kernel(unsigned short* R, int64_t r_chan_stride, int64_t pad_row, int FS) {
    const int c = blockIdx.y;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * FS) return;
    R[c * r_chan_stride + pad_row * FS + k] = (k < FS) ? (unsigned short)0 : (unsigned short)0x7C00;
}

hipError_t launch_fill_rank_pad_rows(unsigned short* R, int64_t r_chan_stride, int32_t n_channels, int64_t pad_row,
                                     int32_t FS, hipStream_t s) {
    hipLaunchKernelGGL(fill_rank_pad_rows_kernel, dim3((unsigned)ceil_div(2 * FS, 256), (unsigned)n_channels), dim3(256), 0, s,
                       R, r_chan_stride, pad_row, FS);
    return hipGetLastError();
}

bool rank_columns_supported(int64_t T) { return T > kRankMinFrames && T <= kRankMaxFrames; }

template <int LOG2N>
static hipError_t launch_rank_n(const RankArgs& a, hipStream_t s, RankStepHook hook, void* user) {
    constexpr int N = 1 << LOG2N;
    constexpr int lds = (N + N / 8) * 4 + (N >> rank_leaf_bits(LOG2N)) * 4;        // the padded keys + the breadth-first copy of the leaf ends
    // REPET_RANK_TRIPS=5: the trips of up to five stages (WideTrips). Measured (profiles/r06_rank_trips_ab.txt): 19 trips instead of
    // 23 and NOT faster -- rank_columns_kernel<13> 132.6 us beside the peak picking against 125-130, the stage 0.2411-0.2415 ms
    // against 0.2392-0.2424: a trip's cost is its LDS INSTRUCTIONS (64 or 32 four- / eight-byte accesses per thread against 16
    // sixteen-byte ones), not its bytes. Default: the three-stage trips.
    // (2^15 keys: 1 024 threads at 128 registers each -- the wide trips spill there, that size keeps the three-stage form)
    static const bool wide_env = [] { const char* t = getenv("REPET_RANK_TRIPS"); return t && t[0] == '5'; }();
    constexpr bool kWideOk = LOG2N <= 14;
    const bool wide = kWideOk && wide_env;
    const void* fn = reinterpret_cast<const void*>(&rank_columns_kernel<LOG2N, false>);
    if constexpr (kWideOk) { if (wide) fn = reinterpret_cast<const void*>(&rank_columns_kernel<LOG2N, true>); }
    hipError_t e = ensure_dynamic_lds(fn, lds);
    if (e != hipSuccess) return e;
    const int64_t cols = (int64_t)a.n_channels * a.n_cols;
    // REPET_RANK_TILE=64: the 64-frame tiles of round 4 (A/B)
    static const bool tall = [] { const char* e = getenv("REPET_RANK_TILE"); return e && e[0] == '6'; }();
    const bool do_transpose = a.phases == 0 || (a.phases & 1), do_sort = a.phases == 0 || (a.phases & 2);
    if (do_transpose) {
        if (tall)
            hipLaunchKernelGGL(columns_from_rows_kernel<64>, dim3((unsigned)ceil_div(a.vs_pitch, 64), (unsigned)(a.n_cols / 64), (unsigned)a.n_channels),
                               dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(columns_from_rows_kernel<32>, dim3((unsigned)ceil_div(a.vs_pitch, 32), (unsigned)(a.n_cols / 64), (unsigned)a.n_channels),
                               dim3(256), 0, s, a);
        if (hook) hook(user, 0);
    }
    if (!do_sort) return hipGetLastError();
    bool launched = false;
    if constexpr (kWideOk) {
        if (wide) { hipLaunchKernelGGL((rank_columns_kernel<LOG2N, true>), dim3((unsigned)cols), dim3(N / 32), lds, s, a); launched = true; }
    }
    if (!launched) hipLaunchKernelGGL((rank_columns_kernel<LOG2N, false>), dim3((unsigned)cols), dim3(N / 32), lds, s, a);
    if (hook) hook(user, 1);
    if (a.P) {                           // the bit-sliced selection reads the planes only: no frame-major codes
        // REPET_RANK_TILE=64: the 64-frame workgroups of round 4 (A/B)
        if (tall) {
            constexpr int plane_lds_bytes = 1024 * 66 * 2;
            e = ensure_dynamic_lds(reinterpret_cast<const void*>(&code_planes_from_columns_kernel<64>), plane_lds_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(code_planes_from_columns_kernel<64>, dim3((unsigned)ceil_div(a.T, 64), 2), dim3(1024), plane_lds_bytes, s, a);
        } else {
            constexpr int plane_lds_bytes = 1024 * 34 * 2;
            e = ensure_dynamic_lds(reinterpret_cast<const void*>(&code_planes_from_columns_kernel<32>), plane_lds_bytes);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(code_planes_from_columns_kernel<32>, dim3((unsigned)ceil_div(a.T, 32), 2), dim3(1024), plane_lds_bytes, s, a);
        }
        if (hook) hook(user, 2);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(rows_from_code_columns_kernel, dim3((unsigned)ceil_div(a.vs_pitch, 128), (unsigned)(a.n_cols / 128), (unsigned)a.n_channels),
                       dim3(256), 0, s, a);
    if (hook) hook(user, 3);
    return hipGetLastError();
}

hipError_t launch_rank_columns(const RankArgs& a0, hipStream_t s, RankStepHook hook, void* user) {
    if (!rank_columns_supported(a0.T) || a0.n_cols <= 0 || (a0.n_cols & 127) || (a0.vs_pitch & 31) || (a0.FS & 1)) return hipErrorInvalidValue;
    if (a0.P && (a0.R || a0.n_planes != code_planes_for(a0.T) || a0.n_planes > 15 || a0.n_channels * (a0.n_cols >> 6) > 32)) return hipErrorInvalidValue;
    if (!a0.P && !a0.R) return hipErrorInvalidValue;
    RankArgs a = a0;
    if (a.T <= 2048) return launch_rank_n<11>(a, s, hook, user);
    if (a.T <= 4096) return launch_rank_n<12>(a, s, hook, user);
    if (a.T <= 8192) return launch_rank_n<13>(a, s, hook, user);
    if (a.T <= 16384) return launch_rank_n<14>(a, s, hook, user);
    return launch_rank_n<15>(a, s, hook, user);
}

}  // namespace repet
