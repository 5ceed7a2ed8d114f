// Prototype of the bit-sliced median selection (tools/gen_bitslice_count.py): the lower and upper median of up to 100 gathered
// 13-bit rank codes for the 2 048 cells of a frame, one workgroup of two waves per frame -- each wave owns half of the list
// entries, counts per plane and cell how many of ITS entries still in the running have a 0 there, the two partial counts meet
// through LDS. Random lists over a 7 753-frame table: checks a few frames against a sort on the host and times the launch.
//   hipcc -O3 --offload-arch=gfx950 -I../../repet-python_amd/csrc bitslice_select.hip -o bitslice_select
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

template <int IMM>
__device__ __forceinline__ unsigned bitop3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:%4" : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(IMM));
    return r;
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

#ifndef WAVES
#define WAVES 4
#endif
constexpr int kW = WAVES;                   // waves per frame
constexpr int kH = (100 + kW - 1) / kW;     // list entries per wave
constexpr int kD = BitsliceCount<kH>::kDigits;
constexpr int kR = 7;
constexpr int kRowWords = 64;   // words of one plane of one frame

template <int NP>
__global__ __launch_bounds__(64 * kW) void select_kernel(const unsigned* __restrict__ planes, const int* __restrict__ idx, int idx_pitch,
                                                     const int* __restrict__ count, unsigned* __restrict__ out, int T) {
#ifdef SYM
    __shared__ unsigned xch2[2][kW][kD + 1][64];
#define xch xch2[p & 1]
#else
    __shared__ unsigned xch[kW][kD + 1][64];
#endif
    __shared__ unsigned dec[2][64];
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = count[t];
    const int* list = idx + (long)t * idx_pitch + w * kH;
    constexpr int row_bytes = NP * kRowWords * 4;
    const int e_lane = list[lane < kH ? lane : 0];
    const int off_v = (w * kH + lane < n) ? e_lane * row_bytes : 0;
    int off[kH];
    unsigned A1[kH], A2[kH];
#pragma unroll
    for (int k = 0; k < kH; ++k) {
        off[k] = __builtin_amdgcn_readlane(off_v, k);
        A1[k] = (w * kH + k < n) ? ~0u : 0u;
        A2[k] = A1[k];
    }
    unsigned r1[kR], D = 0u;
    const unsigned even = (n & 1) ? 0u : ~0u;
#pragma unroll
    for (int d = 0; d < kR; ++d) r1[d] = (((n - 1) >> 1) >> d & 1) ? ~0u : 0u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(planes), 0, T * row_bytes, 0x00020000);
    unsigned* o = out + (long)t * 2 * NP * 64;
    unsigned B[kH], Bn[kH];
#pragma unroll
    for (int k = 0; k < kH; ++k) B[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, off[k] + (NP - 1) * (kRowWords * 4), 0);
#pragma unroll 1
    for (int p = NP - 1; p >= 0; --p) {
#ifndef NO_DB
        const int pn = p > 0 ? p - 1 : 0;
#pragma unroll
        for (int k = 0; k < kH; ++k) Bn[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, off[k] + pn * (kRowWords * 4), 0);
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
            if (kH & 1) any2 |= z[kH - 1];
        }
#ifdef SYM
        {
#pragma unroll
            for (int d = 0; d < kD; ++d) xch[w][d][lane] = c1[d];
            xch[w][kD][lane] = any2;
        }
        __syncthreads();
        unsigned s1, s2;
        {
#else
        if (w != 0) {
#pragma unroll
            for (int d = 0; d < kD; ++d) xch[w][d][lane] = c1[d];
            xch[w][kD][lane] = any2;
        }
        __syncthreads();
        if (w == 0) {                           // the leader adds the partial counts up and decides for everybody
#endif
            unsigned tot[kR];
#pragma unroll
            for (int d = 0; d < kR; ++d) tot[d] = d < kD ? c1[d] : 0u;
#pragma unroll
            for (int oi = 1; oi < kW; ++oi) {
#ifdef SYM
                const int ow = (w + oi) % kW;
#else
                const int ow = oi;
#endif
                unsigned cy = 0u;
#pragma unroll
                for (int d = 0; d < kR; ++d) {
                    const unsigned x = tot[d], y = d < kD ? xch[ow][d][lane] : 0u;
                    tot[d] = bs_xor3(x, y, cy);
                    cy = bs_maj(x, y, cy);
                }
                any2 |= xch[ow][kD][lane];
            }
            unsigned diff[kR], bw = 0u, all = ~0u;
#pragma unroll
            for (int d = 0; d < kR; ++d) {
                diff[d] = bs_xor3(r1[d], tot[d], bw);
                bw = bs_borrow(r1[d], tot[d], bw);
                all &= diff[d];
            }
            const unsigned part = all & bw & even;
#ifdef SYM
            s1 = bw; s2 = bs_sel(D, any2, bw & ~part);
            D = bs_andn_or(D, s1, s2);
#pragma unroll
            for (int d = 0; d < kR; ++d) r1[d] = bs_sel(bw, r1[d], diff[d]);
            if (w < 2) o[(w * NP + p) * 64 + lane] = ~(w ? s2 : s1);
        }
#else
            const unsigned s1 = bw, s2 = bs_sel(D, any2, bw & ~part);
            D = bs_andn_or(D, s1, s2);
#pragma unroll
            for (int d = 0; d < kR; ++d) r1[d] = bs_sel(bw, r1[d], diff[d]);
            dec[0][lane] = s1; dec[1][lane] = s2;
            o[p * 64 + lane] = ~s1; o[(NP + p) * 64 + lane] = ~s2;
        }
        __syncthreads();
        const unsigned s1 = dec[0][lane], s2 = dec[1][lane];
#endif
#pragma unroll
        for (int k = 0; k < kH; ++k) { A1[k] = bs_keep(A1[k], B[k], s1); A2[k] = bs_keep(A2[k], B[k], s2); }
#ifndef NO_DB
#pragma unroll
        for (int k = 0; k < kH; ++k) B[k] = Bn[k];
#else
        if (p > 0) {
#pragma unroll
            for (int k = 0; k < kH; ++k) B[k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, off[k] + (p - 1) * (kRowWords * 4), 0);
        }
#endif
    }
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
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(select_kernel<NP>, dim3(T), dim3(64 * kW), 0, 0, d_planes, d_idx, pitch, d_count, d_out, T);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(select_kernel<NP>, dim3(T), dim3(64 * kW), 0, 0, d_planes, d_idx, pitch, d_count, d_out, T);
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
    printf("lists %s: %.4f ms per launch; %ld of %ld cells wrong\n", lf ? "from file" : local > 0 ? "local" : "uniform", ms / reps, bad, checked);
    return bad != 0;
}
