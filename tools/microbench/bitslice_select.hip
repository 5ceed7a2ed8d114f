// The bit-sliced median selection of mask_bits.hip on its own: the lower and upper median of up to 100 gathered 13-bit rank
// codes for the 2 048 cells of a frame, one workgroup of four waves per frame -- each wave owns a quarter of the list entries,
// counts per plane and cell how many of ITS entries still in the running have a 0 there, the partial counts meet in LDS.
// Checks a few frames against a sort on the host and times the launch, for random lists (uniform over a 7 753-frame table,
// or inside a window of +- N frames) or the lists of a real run (tools/dump_sim_lists.py); -DSTAMPS: cycles per phase.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=131072 -I../../repet-python_amd/csrc bitslice_select.hip -o bitslice_select
//   ./bitslice_select [window, 0 = uniform] [list file]
// Measured on the way here (cfg-2 shapes, ms per launch): two waves of 50 entries 0.23-0.25; eight of 13 0.23-0.31; every
// wave adding all partial counts itself (one barrier per plane) 0.236 against 0.217 with a leader; the next plane's loads
// prefetched into a second register set 0.246 against 0.202 (a wave per SIMD less); planes in groups of four behind
// buffer_load_dwordx4 0.28 (two waves per SIMD). What bounds all of them: gather_rate.hip -- the CU takes one dword wave-load
// per 11.4 cycles whatever the cache says when the row comes as the load's scalar offset, per 8.4 when it is added into the
// lanes' offset (0.218 -> 0.20 here), and a frame needs 1 300 of them.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

template <int IMM>
__device__ __forceinline__ unsigned bitop3(unsigned a, unsigned b, unsigned c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, IMM);
}
// truth tables: src0 = 0xF0, src1 = 0xCC, src2 = 0xAA
__device__ __forceinline__ unsigned bs_xor3(unsigned a, unsigned b, unsigned c) { return bitop3<0x96>(a, b, c); }
__device__ __forceinline__ unsigned bs_maj(unsigned a, unsigned b, unsigned c) { return bitop3<0xE8>(a, b, c); }
__device__ __forceinline__ unsigned bs_andn(unsigned a, unsigned b) { return bitop3<0x30>(a, b, b); }              // a & ~b
__device__ __forceinline__ unsigned bs_keep(unsigned a, unsigned b, unsigned s) { return bitop3<0x60>(a, b, s); }  // a & (b ^ s)
__device__ __forceinline__ unsigned bs_borrow(unsigned r, unsigned c, unsigned bw) { return bitop3<0x8E>(r, c, bw); }   // maj(~r, c, bw)
__device__ __forceinline__ unsigned bs_sel(unsigned s, unsigned x, unsigned y) { return bitop3<0xCA>(s, x, y); }   // s ? x : y
__device__ __forceinline__ unsigned bs_or3(unsigned a, unsigned b, unsigned c) { return bitop3<0xFE>(a, b, c); }
__device__ __forceinline__ unsigned bs_and3(unsigned a, unsigned b, unsigned c) { return bitop3<0x80>(a, b, c); }
__device__ __forceinline__ unsigned bs_andn_or(unsigned d, unsigned a, unsigned b) { return bitop3<0xF4>(d, a, b); }   // d | (a & ~b)
#include "bitslice_count.inc"

constexpr int kH = 25;          // list entries per wave, four waves per frame
constexpr int kD = 5, kR = 7;   // digits of a wave's count, of the whole count and the rank
constexpr int kRowWords = 64;   // words of one plane of one frame

#ifdef STAMPS
__device__ unsigned long long g_phase[8192][4][8];
#define ST(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc[i] += now_ - last; last = now_; }
#else
#define ST(i)
#endif

// the selection of mask_sim_bits_kernel (mask_bits.hip) without what follows it: the two code images go to `out` (the compiler
// gives this copy 168 VGPRs -- three waves per SIMD -- where the kernel in the library has 118 and four)
template <int NP>
__global__ __launch_bounds__(256) void select_kernel(const unsigned* __restrict__ planes, const int* __restrict__ idx, int idx_pitch,
                                                     const int* __restrict__ count, unsigned* __restrict__ out, int T) {
    __shared__ uint4 xch[4][2][64];
    __shared__ unsigned dec[2][64];
    const int t = blockIdx.x;
    const int lead = (blockIdx.x >> 8) & 3;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = count[t];
    const int* list = idx + (long)t * idx_pitch + w * kH;
    constexpr int row_bytes = NP * kRowWords * 4;
    const int e_lane = list[lane < kH ? lane : 0];
    const int off_v = (w * kH + lane < n) ? e_lane * row_bytes : 0;
    int off[kH];
    unsigned A1[kH], A2[kH], B[kH];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(planes), 0, T * row_bytes, 0x00020000);
#ifdef STAMPS
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = __builtin_amdgcn_s_memtime();
    const unsigned long long start = last;
#endif
#pragma unroll
    for (int k = 0; k < kH; ++k) {
        off[k] = __builtin_amdgcn_readlane(off_v, k);
        A1[k] = (w * kH + k < n) ? ~0u : 0u;
        A2[k] = A1[k];
        B[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4 + (NP - 1) * (kRowWords * 4) + off[k], 0, 0);
    }
    unsigned r1[kR], D = 0u;
    const unsigned even = (n & 1) ? 0u : ~0u;
#pragma unroll
    for (int d = 0; d < kR; ++d) r1[d] = (((n - 1) >> 1) >> d & 1) ? ~0u : 0u;
    unsigned* o = out + (long)t * 2 * NP * 64;
#pragma unroll 1
    for (int p = NP - 1; p >= 0; --p) {
#ifdef STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ST(0)                                   // waiting for the plane
#endif
        unsigned c1[kD];
        BitsliceCount<kH>::run([&](int k) { return bs_andn(A1[k], B[k]); }, c1);
        unsigned any2 = 0u;
        {
            unsigned z[kH];
#pragma unroll
            for (int k = 0; k < kH; ++k) z[k] = bs_andn(A2[k], B[k]);
#pragma unroll
            for (int k = 0; k + 1 < kH; k += 2) any2 = bs_or3(any2, z[k], z[k + 1]);
            any2 |= z[kH - 1];
        }
        if (w != lead) {
            xch[w][0][lane] = make_uint4(c1[0], c1[1], c1[2], c1[3]);
            xch[w][1][lane] = make_uint4(c1[4], any2, 0u, 0u);
        }
#ifdef STAMPS
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        ST(1)                                   // counting
        __syncthreads();
        ST(2)                                   // first barrier
        if (w == lead) {                        // the leader adds the partial counts up and decides for everybody
            unsigned q[3][6];
#pragma unroll
            for (int oi = 0; oi < 3; ++oi) {
                const int ow = (lead + 1 + oi) & 3;
                const uint4 lo = xch[ow][0][lane], hi = xch[ow][1][lane];
                q[oi][0] = lo.x; q[oi][1] = lo.y; q[oi][2] = lo.z; q[oi][3] = lo.w; q[oi][4] = hi.x; q[oi][5] = hi.y;
            }
            any2 = bs_or3(any2, q[0][kD], q[1][kD]) | q[2][kD];
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
            A[kD] = c1_[kD - 1];
            unsigned diff[kR + 1], cy = ~0u, prev = ~0u;
#pragma unroll
            for (int d = 0; d <= kR; ++d) {
                const unsigned rd = d < kR ? r1[d] : 0u;
                const unsigned s3 = bs_xor3(rd, A[d], Bv[d]);
                const unsigned c3 = bitop3<0x71>(rd, A[d], Bv[d]);     // maj(r, ~a, ~b)
                diff[d] = bs_xor3(s3, prev, cy);
                cy = bs_maj(s3, prev, cy);
                prev = c3;
            }
            const unsigned bw = diff[kR];
            unsigned all = diff[0];
#pragma unroll
            for (int d = 1; d <= kR; ++d) all &= diff[d];
            const unsigned part = all & bw & even;
            const unsigned s1 = bw, s2 = bs_sel(D, any2, bw & ~part);
            D = bs_andn_or(D, s1, s2);
#pragma unroll
            for (int d = 0; d < kR; ++d) r1[d] = bs_sel(bw, r1[d], diff[d]);
            dec[0][lane] = s1; dec[1][lane] = s2;
            o[p * 64 + lane] = ~s1; o[(NP + p) * 64 + lane] = ~s2;
        }
        ST(3)                                   // the leader's arithmetic
        __syncthreads();
        ST(4)                                   // second barrier
        const unsigned s1 = dec[0][lane], s2 = dec[1][lane];
        const int next = lane * 4 + (p > 0 ? p - 1 : 0) * (kRowWords * 4);
#pragma unroll
        for (int k = 0; k < kH; ++k) {          // (the row in the lane offset: gather_rate.hip)
            A1[k] = bs_keep(A1[k], B[k], s1);
            A2[k] = bs_keep(A2[k], B[k], s2);
            B[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, next + off[k], 0, 0);
        }
        ST(5)                                   // update, issue of the next plane's loads
    }
#ifdef STAMPS
    if (blockIdx.x < 8192 && lane == 0) {
        acc[6] = __builtin_amdgcn_s_memtime() - start;
        for (int i = 0; i < 8; ++i) g_phase[blockIdx.x][w][i] = acc[i];
    }
#endif
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    constexpr int NP = 13;
    const int T = 7753, cells = 2048, pitch = 128;
    const int local = argc > 1 ? atoi(argv[1]) : 0;            // 0: lists uniform over the clip; > 0: within +-local frames
    std::mt19937 rng(5);
    std::vector<unsigned short> codes((size_t)T * cells);
    for (auto& c : codes) c = (unsigned short)(rng() % T);
    std::vector<unsigned> planes((size_t)T * NP * 64, 0u);
    for (int t = 0; t < T; ++t)
        for (int ci = 0; ci < cells; ++ci) {
            const int l = ci & 63, b = ci >> 6;
            for (int p = 0; p < NP; ++p)
                if (codes[(size_t)t * cells + ci] >> p & 1) planes[((size_t)t * NP + p) * 64 + l] |= 1u << b;
        }
#define LAUNCH(...) hipLaunchKernelGGL(select_kernel<NP>, dim3(T), dim3(256), 0, 0, __VA_ARGS__)
    std::vector<int> idx((size_t)T * pitch, 0), count(T);
    FILE* lf = (argc > 2) ? fopen(argv[2], "rb") : nullptr;      // real lists (tools/dump_sim_lists.py) instead of random ones
    if (lf) {
        int hdr[3];
        if (fread(hdr, 4, 3, lf) != 3 || hdr[0] != T || hdr[2] != pitch) { printf("list file does not fit\n"); return 1; }
        if (fread(idx.data(), 4, idx.size(), lf) != idx.size() || fread(count.data(), 4, count.size(), lf) != count.size()) return 1;
        fclose(lf);
    }
    for (int t = 0; t < T && !lf; ++t) {
        count[t] = (t % 97 == 5) ? 1 + (int)(rng() % 100) : 100;
        for (int k = 0; k < 100; ++k) {
            int j = local > 0 ? t - local + (int)(rng() % (2 * local + 1)) : (int)(rng() % T);
            idx[(size_t)t * pitch + k] = std::min(std::max(j, 0), T - 1);
        }
    }
    unsigned *d_planes, *d_out;
    int *d_idx, *d_count;
    CK(hipMalloc(&d_planes, planes.size() * 4));
    CK(hipMalloc(&d_out, (size_t)T * 2 * NP * 64 * 4));
    CK(hipMalloc(&d_idx, idx.size() * 4));
    CK(hipMalloc(&d_count, count.size() * 4));
    CK(hipMemcpy(d_planes, planes.data(), planes.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_count, count.data(), count.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) LAUNCH(d_planes, d_idx, pitch, d_count, d_out, T);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) LAUNCH(d_planes, d_idx, pitch, d_count, d_out, T);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> out((size_t)T * 2 * NP * 64);
    CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0, checked = 0;
    for (int t = 0; t < T; t += 61) {
        const int n = count[t];
        for (int ci = 0; ci < cells; ++ci) {
            std::vector<int> v(n);
            for (int k = 0; k < n; ++k) v[k] = codes[(size_t)idx[(size_t)t * pitch + k] * cells + ci];
            std::sort(v.begin(), v.end());
            const int lo = v[(n - 1) >> 1], hi = v[n >> 1];
            int glo = 0, ghi = 0;
            const int l = ci & 63, b = ci >> 6;
            for (int p = 0; p < NP; ++p) {
                glo |= (int)(out[((size_t)t * 2 * NP + p) * 64 + l] >> b & 1) << p;
                ghi |= (int)(out[((size_t)t * 2 * NP + NP + p) * 64 + l] >> b & 1) << p;
            }
            ++checked;
            if (glo != lo || ghi != hi) { if (bad < 5) printf("frame %d cell %d: got %d %d want %d %d (n %d)\n", t, ci, glo, ghi, lo, hi, n); ++bad; }
        }
    }
#ifdef STAMPS
    {
        static unsigned long long ph[8192][4][8];
        CK(hipMemcpyFromSymbol(ph, HIP_SYMBOL(g_phase), sizeof(ph)));
        const char* names[8] = {"wait for the plane", "count", "barrier 1", "leader", "barrier 2", "update + issue of the loads", "whole wave", "-"};
        for (int wv = 0; wv < 4; wv += 3) {
            printf("wave %d, cycles per frame (mean over %d frames):", wv, T < 8192 ? T : 8192);
            for (int i = 0; i < 8; ++i) { double sum = 0; int nn = T < 8192 ? T : 8192; for (int b = 0; b < nn; ++b) sum += (double)ph[b][wv][i]; printf("  %s %.0f", names[i], sum / nn); }
            printf("\n");
        }
    }
#endif
    printf("lists %s: %.4f ms per launch; %ld of %ld cells wrong\n", lf ? "from file" : local > 0 ? "local" : "uniform", ms / reps, bad, checked);
    return bad != 0;
}
