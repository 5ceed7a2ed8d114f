// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for the ops a
// compare-exchange network can be built from. 8 independent register chains per lane, 8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X X X X X X X X
#define OPS(NAME, ASM)                                                                              \
__global__ __launch_bounds__(512) void k_##NAME(float* out, int iters) {                            \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b = out[threadIdx.x & 3], c = out[(threadIdx.x & 3) + 4];                                 \
    for (int i = 0; i < iters; ++i) {                                                               \
        REP8(asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) \
    }                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;            \
}
#define E8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define E8_3(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
OPS(min_i32, E8("v_min_i32"))
OPS(max_u32, E8("v_max_u32"))
OPS(min_f32, E8("v_min_f32"))
OPS(add_f32, E8("v_add_f32"))
OPS(fma_f32, E8_3("v_fma_f32"))
OPS(med3_f32, E8_3("v_med3_f32"))
OPS(min3_f32, E8_3("v_min3_f32"))
OPS(max3_i32, E8_3("v_max3_i32"))
OPS(med3_i32, E8_3("v_med3_i32"))
OPS(pk_min_f16, E8("v_pk_min_f16"))

OPS(mul_f32, E8("v_mul_f32"))
OPS(add_u32, E8("v_add_u32"))
OPS(and_b32, E8("v_and_b32"))
OPS(xor_b32, E8("v_xor_b32"))
OPS(lshl_b32, "v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3\n v_lshlrev_b32 %4, 1, %4\n v_lshlrev_b32 %5, 1, %5\n v_lshlrev_b32 %6, 1, %6\n v_lshlrev_b32 %7, 1, %7\n")
OPS(mov_b32, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
OPS(cndmask, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
OPS(cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n")
OPS(swap_b32, "v_swap_b32 %0, %1\n v_swap_b32 %2, %3\n v_swap_b32 %4, %5\n v_swap_b32 %6, %7\n v_swap_b32 %0, %1\n v_swap_b32 %2, %3\n v_swap_b32 %4, %5\n v_swap_b32 %6, %7\n")
OPS(max_f32_dpp, E8("v_max_f32"))
OPS(sub_f32, E8("v_sub_f32"))
OPS(max_i16, E8("v_pk_max_i16"))

template <class K> void run(const char* name, K kern, float* d, int per_iter) {
    const int iters = 2000, blocks = 256 * 4;   // 512 threads = 8 waves; 4 blocks/CU -> 8 waves/SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 8 * iters * per_iter;       // wave-instructions
    const double per_simd = wave_instr / 1024.0;
    printf("%-12s %8.3f ms  %6.2f ns per wave-instr per SIMD  (= %5.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
}
int main() {
    float* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    run("v_min_i32", k_min_i32, d, 64); run("v_max_u32", k_max_u32, d, 64); run("v_min_f32", k_min_f32, d, 64);
    run("v_add_f32", k_add_f32, d, 64); run("v_fma_f32", k_fma_f32, d, 64); run("v_med3_f32", k_med3_f32, d, 64);
    run("v_min3_f32", k_min3_f32, d, 64); run("v_max3_i32", k_max3_i32, d, 64); run("v_med3_i32", k_med3_i32, d, 64);
    run("v_pk_min_f16", k_pk_min_f16, d, 64);
    run("v_mul_f32", k_mul_f32, d, 64); run("v_sub_f32", k_sub_f32, d, 64); run("v_add_u32", k_add_u32, d, 64); run("v_and_b32", k_and_b32, d, 64);
    run("v_xor_b32", k_xor_b32, d, 64); run("v_lshlrev_b32", k_lshl_b32, d, 64); run("v_mov_b32", k_mov_b32, d, 64);
    run("v_cndmask_b32", k_cndmask, d, 64); run("v_cmp_lt_f32", k_cmp_lt_f32, d, 64); run("v_swap_b32", k_swap_b32, d, 64);
    run("v_pk_max_i16", k_max_i16, d, 64);
    return 0;
}
