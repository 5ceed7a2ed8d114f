// K8c: the median of `sim` (_simmask, repet.py:1511-1545) as a BIT-SLICED selection on the rank codes of rank.hip.
//
// The packed network of mask_sim_rank_kernel works on two cells (bin x channel) per lane and instruction and is bound by
// the issue rate of v_pk_min_u16 / v_pk_max_u16 (4.2 cycles per wave instruction): 1 426 of them per 128 cells. Here one
// 32-bit register holds ONE BIT of the codes of 32 cells -- P[t][plane][l] bit b = bit `plane` of the code of cell
// 64 b + l of frame t (code_planes_from_columns_kernel, rank.hip) -- so a boolean instruction works on 32 cells per lane, 2 048 per wave,
// and gfx950's v_bitop3_b32 (any function of three words, in the fast VALU class: 2.6 cycles) is a whole full adder's sum
// or carry. An order statistic is found by a radix descent over the planes, most significant first:
//   z_k    = alive_k & ~plane_k                  entries still in the running whose code has a 0 here
//   count  = z_0 + ... + z_{n-1}                 per cell, a carry-save counter of full adders (bitslice_count.inc)
//   s      = rank < count                        the wanted entry is among the zeros: its bit is 0 (else 1, rank -= count)
//   alive_k &= plane_k ^ s                       keep the entries on its side
// 3.9 instructions per entry and plane. np.median of an even list needs the entry of rank + 1 too: it walks the same path
// until the plane where exactly rank + 1 zeros are left (the lower median is the largest of them, the upper one the
// smallest of the ones); from there on it is the MINIMUM of its set, which needs "is any zero left" (an OR) instead of a
// count: 2.5 instructions per entry and plane. About 9 400 wave instructions per frame (13 planes, 100 entries, both
// channels) against 26 800 of the slower class, and 13 gathered bits per value instead of 16.
//
// One workgroup of four waves per frame: each wave owns a quarter of the list (its alive words stay in registers: 2 x 25
// and the planes of this step, 117 VGPRs -- four waves per SIMD, which the fast VALU class needs to reach its rate:
// tools/microbench/issue_rate.hip), counts ITS zeros, and the partial counts meet in LDS once per plane, where one wave (the
// `lead`, a different one from workgroup to workgroup) adds them up and decides for all four. Afterwards wave w turns the two code images back into numbers for the cells of bits
// [8 w, 8 w + 8) (a 16 x 16 bit transpose on both halves of the registers at once) and stores them, and a second kernel
// (mask_from_codes_kernel, scheduled by blocks of columns) looks the magnitudes up in the sorted columns Vs and applies the
// mask exactly as mask_sim_rank_kernel does: same codes, same table, same soft_mask -- the same bits
// (tests/test_gpu_variants.py::test_median_paths_agree_bit_for_bit, tests/test_gpu_stages.py::
// test_median_selection_on_rank_codes_against_numpy). What bounds it (tools/microbench/bitslice_select.hip, gather_rate.hip):
// the gathers -- 1 300 dword wave-loads per frame, of which a CU's vector memory path takes one every ~11 cycles whatever the
// cache says -- rather than the wave instructions.
#include "common.h"

namespace repet {

namespace {

// (the builtin, not inline asm: around opaque asm the compiler pads possible hazards with s_nop -- 31 per plane in the counter)
template <int IMM>
__device__ __forceinline__ unsigned bitop3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, IMM); }
// truth tables: operand 0 = 0xF0, operand 1 = 0xCC, operand 2 = 0xAA
__device__ __forceinline__ unsigned bs_xor3(unsigned a, unsigned b, unsigned c) { return bitop3<0x96>(a, b, c); }
__device__ __forceinline__ unsigned bs_maj(unsigned a, unsigned b, unsigned c) { return bitop3<0xE8>(a, b, c); }
__device__ __forceinline__ unsigned bs_or3(unsigned a, unsigned b, unsigned c) { return bitop3<0xFE>(a, b, c); }
__device__ __forceinline__ unsigned bs_and3(unsigned a, unsigned b, unsigned c) { return bitop3<0x80>(a, b, c); }
__device__ __forceinline__ unsigned bs_andn(unsigned a, unsigned b) { return bitop3<0x30>(a, b, b); }                  // a & ~b
__device__ __forceinline__ unsigned bs_keep(unsigned a, unsigned b, unsigned s) { return bitop3<0x60>(a, b, s); }      // a & (b ^ s)
__device__ __forceinline__ unsigned bs_borrow(unsigned r, unsigned c, unsigned bw) { return bitop3<0x8E>(r, c, bw); }  // maj(~r, c, bw)
__device__ __forceinline__ unsigned bs_sel(unsigned s, unsigned x, unsigned y) { return bitop3<0xCA>(s, x, y); }       // s ? x : y
__device__ __forceinline__ unsigned bs_or_andn(unsigned d, unsigned a, unsigned b) { return bitop3<0xF4>(d, a, b); }   // d | (a & ~b)
#include "bitslice_count.inc"

constexpr int bit_length(int v) { return v <= 0 ? 0 : 1 + bit_length(v >> 1); }

// 16 x 16 bit transpose of the low halves of x[0..15] and, in the same instructions, of the high halves:
// afterwards bit i of (either half of) x[j] is what bit j of x[i] was.
template <int J>
__device__ __forceinline__ void transpose16_stage(unsigned (&x)[16]) {
    constexpr unsigned m = J == 8 ? 0x00FF00FFu : J == 4 ? 0x0F0F0F0Fu : J == 2 ? 0x33333333u : 0x55555555u;
#pragma unroll
    for (int k = 0; k < 16; ++k)
        if (!(k & J)) {
            const unsigned t = ((x[k] >> J) ^ x[k + J]) & m;
            x[k + J] ^= t;
            x[k] ^= t << J;
        }
}
__device__ __forceinline__ void transpose16_halves(unsigned (&x)[16]) {
    transpose16_stage<8>(x); transpose16_stage<4>(x); transpose16_stage<2>(x); transpose16_stage<1>(x);
}

// H = list entries per wave (four waves: the list holds at most 4 H), NP = planes of the codes (bits of T - 1)
template <int H, int NP>
__global__ __launch_bounds__(256) void mask_sim_bits_kernel(MaskArgs a, const int* __restrict__ idx, int idx_pitch,
                                                            const int* __restrict__ count, int bpc_shift) {
    constexpr int kD = BitsliceCount<H>::kDigits;          // digits of a wave's own count
    constexpr int kR = bit_length(4 * H);                  // digits of the whole count and of the rank
    constexpr int kRowWords = 64, kRowBytes = NP * kRowWords * 4;
    __shared__ uint4 xch[4][2][64];                        // per wave and lane: the digits of its count, "one of my entries has a zero"
    __shared__ unsigned dec[2][64];                        // the leader's verdicts of this plane
    __shared__ unsigned lu[2][NP][64];                     // the code images of the lower and the upper median
    // Workgroup b lands on XCD b % 8: every XCD takes ONE contiguous run of frames instead of every eighth frame (0.215 ->
    // 0.204 ms in tools/microbench/bitslice_select.hip with the bench clip's lists; runs of frames sorted by their lists'
    // earliest entry, so that the frames of one repetition share an L2, 0.195 -- less than the sort would cost)
    const int64_t t_end = a.frame_end > 0 ? a.frame_end : a.T;
    const int64_t per_xcd = (t_end - a.frame0 + 7) >> 3;
    const int64_t t = a.frame0 + (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((int64_t)(blockIdx.x >> 3) >= per_xcd || t >= t_end) return;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = count[t];
    const int* list = idx + t * (int64_t)idx_pitch + w * H;
    // the row offsets of this wave's entries once in vector form (one coalesced load of the list; its rows hold at least
    // 128 entries), handed to the gathers by v_readlane; entries past the list's end read row 0 and are never in the running
    const int e_lane = list[lane < H ? lane : 0];
    const int off_v = (w * H + lane < n) ? e_lane * kRowBytes : 0;
    int off[H];
    unsigned A1[H], A2[H], B[H];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.P), 0, (int)(a.T * kRowBytes), 0x00020000);
#pragma unroll
    for (int k = 0; k < H; ++k) {
        off[k] = __builtin_amdgcn_readlane(off_v, k);
        A1[k] = (w * H + k < n) ? ~0u : 0u;
        A2[k] = A1[k];
        B[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4 + (NP - 1) * (kRowWords * 4) + off[k], 0, 0);
    }
    // the leader's state: the rank still wanted among the entries in the running, D = cells whose upper median has
    // left the lower one's path. An odd list has ONE middle entry: the two never part.
    const int lead = (blockIdx.x >> 8) & 3;                 // (not always wave 0: the waves of one number share a SIMD)
    unsigned r1[kR], D = 0u;
    const unsigned even = (n & 1) ? 0u : ~0u;
#pragma unroll
    for (int d = 0; d < kR; ++d) r1[d] = ((((n - 1) >> 1) >> d) & 1) ? ~0u : 0u;
#pragma unroll 1
    for (int p = NP - 1; p >= 0; --p) {
        unsigned c1[kD];
        BitsliceCount<H>::run([&](int k) { return bs_andn(A1[k], B[k]); }, c1);
        unsigned any2 = 0u;
        {
            unsigned z[H];
#pragma unroll
            for (int k = 0; k < H; ++k) z[k] = bs_andn(A2[k], B[k]);
#pragma unroll
            for (int k = 0; k + 1 < H; k += 2) any2 = bs_or3(any2, z[k], z[k + 1]);
            if (H & 1) any2 |= z[H - 1];
        }
        static_assert(kD + 1 <= 8, "two 16-byte words per lane");
        if (w != lead) {
            unsigned v[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) v[d] = d < kD ? c1[d] : d == kD ? any2 : 0u;
            xch[w][0][lane] = make_uint4(v[0], v[1], v[2], v[3]);
            xch[w][1][lane] = make_uint4(v[4], v[5], v[6], v[7]);
        }
        __syncthreads();
        if (w == lead) {                                    // the leader adds the partial counts up and decides for everybody
            __builtin_amdgcn_s_setprio(3);                  // (the other three wait for this: first in line on its SIMD)
            unsigned q[3][8];
#pragma unroll
            for (int oi = 0; oi < 3; ++oi) {
                const int ow = (lead + 1 + oi) & 3;
                const uint4 lo = xch[ow][0][lane], hi = xch[ow][1][lane];
                q[oi][0] = lo.x; q[oi][1] = lo.y; q[oi][2] = lo.z; q[oi][3] = lo.w;
                q[oi][4] = hi.x; q[oi][5] = hi.y; q[oi][6] = hi.z; q[oi][7] = hi.w;
            }
            any2 = bs_or3(any2, q[0][kD], q[1][kD]) | q[2][kD];
            // The four counts in carry-save form (every digit on its own, no carry chain): count = A + B. Then
            // rank - count = rank + ~A + ~B + 2 in kR + 1 digits of two's complement -- one more carry-save layer and ONE ripple;
            // its sign digit says rank < count.
            unsigned s1_[kD], c1_[kD], A[kR + 1], Bv[kR + 1];
#pragma unroll
            for (int d = 0; d < kD; ++d) { s1_[d] = bs_xor3(c1[d], q[0][d], q[1][d]); c1_[d] = bs_maj(c1[d], q[0][d], q[1][d]); }
#pragma unroll
            for (int d = 0; d <= kR; ++d) { A[d] = 0u; Bv[d] = 0u; }
#pragma unroll
            for (int d = 0; d < kD; ++d) {
                const unsigned cin = d > 0 ? c1_[d - 1] : 0u;
                A[d] = bs_xor3(s1_[d], q[2][d], cin);
                Bv[d + 1] = bs_maj(s1_[d], q[2][d], cin);
            }
            A[kD] = c1_[kD - 1];                            // the top carry of the first layer
            unsigned diff[kR + 1], cy = ~0u, prev = ~0u;     // (the + 2: a carry into the ripple and a 1 in the vacant digit 0 of the carries)
#pragma unroll
            for (int d = 0; d <= kR; ++d) {
                const unsigned rd = d < kR ? r1[d] : 0u;
                const unsigned s3 = bs_xor3(rd, A[d], Bv[d]);
                const unsigned c3 = bitop3<0x71>(rd, A[d], Bv[d]);      // maj(r, ~a, ~b)
                diff[d] = bs_xor3(s3, prev, cy);
                cy = bs_maj(s3, prev, cy);
                prev = c3;
            }
            const unsigned bw = diff[kR];                   // negative: rank < count
            unsigned all = diff[0];
#pragma unroll
            for (int d = 1; d <= kR; ++d) all &= diff[d];
            const unsigned part = all & bw & even;
            const unsigned s1 = bw, s2 = bs_sel(D, any2, bw & ~part);
            D = bs_or_andn(D, s1, s2);
#pragma unroll
            for (int d = 0; d < kR; ++d) r1[d] = bs_sel(bw, r1[d], diff[d]);
            dec[0][lane] = s1; dec[1][lane] = s2;
            lu[0][p][lane] = ~s1; lu[1][p][lane] = ~s2;
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
        const unsigned s1 = dec[0][lane], s2 = dec[1][lane];
        // The entry's row offset is ADDED to the lanes' offset (one v_add per load) rather than handed over as the load's scalar
        // offset: with a scalar offset operand the CU takes a dword wave-load every 11.4 cycles, without one every 8.4
        // (tools/microbench/gather_rate.hip) -- 0.206 -> 0.190 ms.
        const int next = lane * 4 + (p > 0 ? p - 1 : 0) * (kRowWords * 4);     // (the last round reads plane 0 again rather than branching)
#pragma unroll
        for (int k = 0; k < H; ++k) {
            A1[k] = bs_keep(A1[k], B[k], s1);
            A2[k] = bs_keep(A2[k], B[k], s2);
            B[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, next + off[k], 0, 0);
        }
    }
    // ---- the two code images back into numbers: wave w for the cells of bits [8 w, 8 w + 8) ----
    // One word per cell goes to median_codes (V's geometry): upper code << 16 | lower code, bit 15 = "the lower median is
    // below the frame's own value" (codes have 15 bits at most). mask_from_codes_kernel turns them into magnitudes: its
    // lookups in the sorted columns want the workgroups of an XCD on ONE block of columns at a time (the table is 64 MB;
    // from this kernel, where a wave covers a whole frame, they cost 148 us of 403 at cfg 2).
    const int n_bits = a.n_channels << bpc_shift;           // blocks of 64 bins over all channels (<= 32)
    unsigned own[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) own[p] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (unsigned)(t * kRowBytes) + lane * 4, p * (kRowWords * 4), 0);
    // need = lower median < own code: the borrow of (lower - own), least significant plane first
    unsigned need = 0u, x[16];
    const unsigned pick = 0x0C040C00u + 0x00010001u * w;   // v_perm_b32: byte w of the upper image | byte w of the lower one
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        if (p < NP) {
            const unsigned lo = lu[0][p][lane], hi = lu[1][p][lane];
            need = bs_borrow(lo, own[p], need);
            x[p] = __builtin_amdgcn_perm(hi, lo, pick);
        } else x[p] = 0u;
    }
    transpose16_halves(x);                                  // x[j]: upper code << 16 | lower code of the cell of bit 8 w + j
    const __amdgpu_buffer_rsrc_t c_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.median_codes, 0, (int)(a.n_channels * a.chan_stride * 4), 0x00020000);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int b = 8 * w + j;
        if (b < n_bits) {
            const int c = b >> bpc_shift, fb = b - (c << bpc_shift);
            const int row = (int)(c * a.chan_stride + t * a.FS) + fb * 64;            // elements; the lane adds itself
            const unsigned flag = ((need >> b) & 1u) << 15;
            __builtin_amdgcn_raw_buffer_store_b32(x[j] | flag, c_rsrc, lane * 4, row * 4, 2);      // nt: read once, by the next kernel
        }
    }
}

// Codes -> magnitudes -> mask: the tail of mask_sim_rank_kernel on the words the selection left. Same scheduling: the unit of
// work is (channel, block of 128 bins, 4 consecutive frames), and XCD x is given the combinations x, x + 8, ... one after
// the other over all frames, so its L2 holds the 4 MB of sorted columns of one combination while the lookups go there.
__global__ __launch_bounds__(256) void mask_from_codes_kernel(MaskArgs a, const int* __restrict__ count, int n_quads) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nfb = a.n_rank_cols >> 7;
    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
    const int combo = (i / n_quads) * 8 + xcd;
    if (combo >= a.n_channels * nfb) return;
    const int c = combo / nfb, fb = combo % nfb;
    const int64_t t_end = a.frame_end > 0 ? a.frame_end : a.T;
    const int64_t t = a.frame0 + 4 * (int64_t)(i % n_quads) + wave;
    if (t >= t_end) return;
    const int f0 = fb * 128 + 2 * lane;
    const float* vs = a.Vs + ((int64_t)c * a.n_rank_cols + f0) * a.vs_pitch;
    const int64_t o = c * a.chan_stride + t * a.FS + f0;
    const int n = count[t];
    // the streams pass through once: non-temporal, so that they do not push the table of sorted columns out of the L2
    // (0.117 -> 0.099 ms at cfg 2)
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const f2 v_nt = __builtin_nontemporal_load(reinterpret_cast<const f2*>(a.V + o));
    const u2 c_nt = __builtin_nontemporal_load(reinterpret_cast<const u2*>(a.median_codes + o));
    const float2 v_own = make_float2(v_nt.x, v_nt.y);
    const uint2 cw = make_uint2(c_nt.x, c_nt.y);
    float4 x_own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.X) { const f4 x_nt = __builtin_nontemporal_load(reinterpret_cast<const f4*>(a.X + o)); x_own = make_float4(x_nt.x, x_nt.y, x_nt.z, x_nt.w); }
    // lower median >= own value  =>  min(V, median) = V and the mask is exactly 1: V itself stands in for the model.
    // (One branch per bin, two table reads in each: 0.12 ms at cfg 2; all four behind one branch 0.133; two or four frames per
    // wave with all their loads up front 0.14.)
    float a0 = v_own.x, b0 = v_own.x, a1 = v_own.y, b1 = v_own.y;
    const bool need0 = cw.x & 0x8000u, need1 = cw.y & 0x8000u;
    if (need0) { a0 = vs[cw.x & 0x7fffu]; b0 = vs[cw.x >> 16]; }
    if (need1) { a1 = vs[a.vs_pitch + (cw.y & 0x7fffu)]; b1 = vs[a.vs_pitch + (cw.y >> 16)]; }
    float med0 = need0 ? ((n & 1) ? a0 : 0.5f * (a0 + b0)) : v_own.x;
    float med1 = need1 ? ((n & 1) ? a1 : 0.5f * (a1 + b1)) : v_own.y;
    if (n <= 0) med0 = med1 = __uint_as_float(0x7fc00000u);              // np.median of an empty slice
    const float m0 = soft_mask(v_own.x, med0, f0, a.cutoff), m1 = soft_mask(v_own.y, med1, f0 + 1, a.cutoff);
    if (a.mask) *reinterpret_cast<float2*>(a.mask + o) = make_float2(m0, m1);
    // (X itself is stored the ordinary way: the inverse STFT reads it next -- a non-temporal store gave the lookups 4 us and took 3 from it)
    if (a.X) *reinterpret_cast<float4*>(a.X + o) = make_float4(x_own.x * m0, x_own.y * m0, x_own.z * m1, x_own.w * m1);
}

template <int H>
hipError_t launch_bits_h(const MaskArgs& m, const int32_t* idx, int32_t idx_pitch, const int32_t* count, unsigned n_launch, int bpc_shift,
                         hipStream_t s) {
#define REPET_BITS_CASE(NP) case NP: hipLaunchKernelGGL((mask_sim_bits_kernel<H, NP>), dim3(8 * ((n_launch + 7) / 8)), dim3(256), 0, s, m, idx, idx_pitch, count, bpc_shift); break;
    switch (m.n_planes) {
        REPET_BITS_CASE(11) REPET_BITS_CASE(12) REPET_BITS_CASE(13) REPET_BITS_CASE(14) REPET_BITS_CASE(15)
        default: return hipErrorInvalidValue;
    }
#undef REPET_BITS_CASE
    return hipGetLastError();
}

}  // namespace

int code_planes_for(int64_t T) { int np = 11; while (((int64_t)1 << np) < T) ++np; return np; }

bool mask_sim_bits_supported(int64_t T, int32_t n_channels, int32_t n_cols, int32_t max_count) {
    const int bpc = n_cols >> 6;
    return rank_columns_supported(T) && n_cols > 0 && (n_cols & 127) == 0 && (bpc & (bpc - 1)) == 0 && n_channels * bpc <= 32 &&
           max_count <= 128 && code_planes_for(T) <= 15;
}

// wave instructions of the descent per frame (the bench's issue-rate figure): per plane every wave finds the zeros of both
// medians among its entries, counts the lower one's, ORs the upper one's and updates both alive sets; the leader adds the four
// counts up and does the rank arithmetic
int mask_sim_bits_instructions(int32_t max_count, int32_t n_planes) {
    const int h = max_count <= 100 ? 25 : 32;
    const int count = h == 25 ? BitsliceCount<25>::kInstructions : BitsliceCount<32>::kInstructions;
    const int kr = bit_length(4 * h);
    const int per_wave = 2 * h + count + h / 2 + 2 * h, leader = 3 * 2 * kr + 4 * kr + 6;
    return (4 * per_wave + leader) * n_planes;
}

hipError_t launch_mask_sim_bits(const MaskArgs& m, const int32_t* idx, int32_t idx_pitch, const int32_t* count, int32_t max_count,
                                unsigned n_launch, hipStream_t s) {
    if (!m.P || !mask_sim_bits_supported(m.T, m.n_channels, m.n_rank_cols, max_count) || m.n_planes != code_planes_for(m.T) ||
        idx_pitch < 128 || !m.median_codes || (int64_t)m.n_channels * m.chan_stride * 4 >= ((int64_t)1 << 31))
        return hipErrorInvalidValue;
    int bpc_shift = 0;
    while ((64 << bpc_shift) < m.n_rank_cols) ++bpc_shift;
    return max_count <= 100 ? launch_bits_h<25>(m, idx, idx_pitch, count, n_launch, bpc_shift, s)
                            : launch_bits_h<32>(m, idx, idx_pitch, count, n_launch, bpc_shift, s);
}

hipError_t launch_mask_from_codes(const MaskArgs& m, const int32_t* count, hipStream_t s) {
    const int64_t t_end = m.frame_end > 0 ? m.frame_end : m.T;
    if (!m.median_codes || !m.Vs || t_end <= m.frame0 || (m.n_rank_cols & 127)) return hipErrorInvalidValue;
    const int n_quads = (int)ceil_div(t_end - m.frame0, 4);
    const int combos = m.n_channels * (m.n_rank_cols >> 7);
    hipLaunchKernelGGL(mask_from_codes_kernel, dim3((unsigned)(8 * ceil_div(combos, 8) * n_quads)), dim3(256), 0, s, m, count, n_quads);
    return hipGetLastError();
}

}  // namespace repet
