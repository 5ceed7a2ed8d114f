// Do the sources of v_pk_min_u16 / v_pk_maximum3_f16 care which VGPR "bank" (register number mod 4) they come from?
// One to four waves per SIMD run a long unrolled sequence of the instruction on fixed registers; cycles per instruction
// by s_memtime. Patterns: both sources in the same bank as each other, or in different banks (also vs the destination).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int PATTERN>
__global__ void probe(unsigned long long* out, int iters) {
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("v_mov_b32 v4, 1\n v_mov_b32 v5, 2\n v_mov_b32 v6, 3\n v_mov_b32 v7, 4\n v_mov_b32 v8, 5\n v_mov_b32 v9, 6\n v_mov_b32 v10, 7\n v_mov_b32 v11, 8\n"
                 "v_mov_b32 v12, 9\n v_mov_b32 v13, 10\n v_mov_b32 v14, 11\n v_mov_b32 v15, 12\n v_mov_b32 v16, 13\n v_mov_b32 v17, 14\n v_mov_b32 v18, 15\n v_mov_b32 v19, 16\n"
                 ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19");
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if constexpr (PATTERN == 0)        // sources same bank (4, 8), destinations rotate
            asm volatile(REP16("v_pk_min_u16 v20, v4, v8\n v_pk_max_u16 v21, v12, v16\n v_pk_min_u16 v22, v8, v12\n v_pk_max_u16 v23, v16, v4\n")
                         ::: "v20","v21","v22","v23");
        else if constexpr (PATTERN == 1)   // sources different banks (4, 9)
            asm volatile(REP16("v_pk_min_u16 v20, v4, v9\n v_pk_max_u16 v21, v13, v18\n v_pk_min_u16 v22, v8, v15\n v_pk_max_u16 v23, v17, v6\n")
                         ::: "v20","v21","v22","v23");
        else if constexpr (PATTERN == 2)   // three sources, same bank
            asm volatile(REP16("v_pk_minimum3_f16 v20, v4, v8, v12\n v_pk_maximum3_f16 v21, v12, v16, v4\n v_pk_minimum3_f16 v22, v8, v12, v16\n v_pk_maximum3_f16 v23, v16, v4, v8\n")
                         ::: "v20","v21","v22","v23");
        else                               // three sources, three banks
            asm volatile(REP16("v_pk_minimum3_f16 v20, v4, v9, v14\n v_pk_maximum3_f16 v21, v13, v18, v7\n v_pk_minimum3_f16 v22, v8, v15, v5\n v_pk_maximum3_f16 v23, v17, v6, v11\n")
                         ::: "v20","v21","v22","v23");
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
// pattern 4: sixteen different destinations (no write-after-write chain), sources in two banks
template <>
__global__ void probe<4>(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_pk_min_u16 v20, v4, v9\n v_pk_max_u16 v21, v13, v18\n v_pk_min_u16 v22, v8, v15\n v_pk_max_u16 v23, v17, v6\n")
                     REP16("v_pk_min_u16 v24, v5, v10\n v_pk_max_u16 v25, v14, v19\n v_pk_min_u16 v26, v9, v16\n v_pk_max_u16 v27, v18, v7\n")
                     REP16("v_pk_min_u16 v28, v6, v11\n v_pk_max_u16 v29, v15, v4\n v_pk_min_u16 v30, v10, v17\n v_pk_max_u16 v31, v19, v8\n")
                     REP16("v_pk_min_u16 v32, v7, v12\n v_pk_max_u16 v33, v16, v5\n v_pk_min_u16 v34, v11, v18\n v_pk_max_u16 v35, v4, v9\n")
                     ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (t1 - t0) / 4;       // four times the instructions of the other patterns
}
// pattern 5: a DEPENDENT chain (each instruction consumes the previous result): the latency
template <>
__global__ void probe<5>(unsigned long long* out, int iters) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i)
        asm volatile(REP16("v_pk_min_u16 v20, v20, v9\n v_pk_max_u16 v20, v20, v18\n v_pk_min_u16 v20, v20, v15\n v_pk_max_u16 v20, v20, v6\n") ::: "v20");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int P>
void run(const char* name, int waves_per_simd) {
    unsigned long long* out; hipMalloc(&out, 8);
    const int iters = 2000;
    // up to 4 waves per SIMD in one 1 024-thread workgroup per CU; 8: two such workgroups per CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<P>, dim3(waves_per_simd > 4 ? 512 : 256), dim3(256 * (waves_per_simd > 4 ? 4 : waves_per_simd)), 0, 0, out, iters);   // warm
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<P>, dim3(waves_per_simd > 4 ? 512 : 256), dim3(256 * (waves_per_simd > 4 ? 4 : waves_per_simd)), 0, 0, out, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)iters * 64 * (P == 4 ? 4 : 1) * waves_per_simd;
    printf("   whole launch %.3f ms -> %.2f cycles per instruction per SIMD at 2.4 GHz\n", ms, ms * 1e-3 * 2.4e9 / instr_per_simd);
    unsigned long long h = 0; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-34s %d wave(s)/SIMD: %.2f cycles per instruction per wave, %.2f per SIMD-issue\n", name, waves_per_simd,
           (double)h / (iters * 64.0), (double)h / (iters * 64.0) / waves_per_simd);
    hipFree(out);
}
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("pk_min/max, sources same bank", w);
        run<1>("pk_min/max, sources two banks", w);
        run<2>("pk_min3/max3, sources same bank", w);
        run<3>("pk_min3/max3, sources three banks", w);
        run<4>("pk_min/max, 16 destinations", w);
        run<5>("pk_min/max, dependent chain", w);
    }
    return 0;
}
