// Compare-exchange building blocks on gfx950: cycles per wave64 instruction per SIMD (8 waves/SIMD, 8 independent
// chains per lane). Complements valu_rate.hip with the candidates for a cheaper median comparator:
//   * v_sub_co_u32 + 2 x v_cndmask_b32 (borrow-driven select) against v_min_i32 + v_max_i32;
//   * packed 16-bit min/max (two rank-coded bins per lane), and gfx950's v_pk_minimum3_f16 / v_pk_maximum3_f16;
//   * cross-lane moves for an in-register column sort (ds_swizzle, v_permlane32_swap, DPP row_shr).
// Build: hipcc --offload-arch=gfx950 -O3 cex_rate.hip -o cex_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define KERNEL(NAME, BODY)                                                                          \
__global__ __launch_bounds__(512) void k_##NAME(unsigned* out, int iters) {                         \
    unsigned a0 = threadIdx.x * 2654435761u, a1 = a0 + 11, a2 = a0 * 3 + 2, a3 = a0 ^ 0x1234, a4 = a0 + 4, a5 = a0 * 7, a6 = a0 + 6, a7 = a0 ^ 77; \
    unsigned b = out[threadIdx.x & 3], c = out[(threadIdx.x & 3) + 4];                              \
    for (int i = 0; i < iters; ++i) { REP8(BODY) }                                                  \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b + c;    \
}
#define IO : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b), "+v"(c)
// 4 comparators per asm block on the pairs (a0,a1) (a2,a3) (a4,a5) (a6,a7)
KERNEL(cex_minmax, asm volatile(
    "v_min_u32 %8, %0, %1\n v_max_u32 %1, %0, %1\n v_mov_b32 %0, %8\n"
    "v_min_u32 %8, %2, %3\n v_max_u32 %3, %2, %3\n v_mov_b32 %2, %8\n"
    "v_min_u32 %8, %4, %5\n v_max_u32 %5, %4, %5\n v_mov_b32 %4, %8\n"
    "v_min_u32 %8, %6, %7\n v_max_u32 %7, %6, %7\n v_mov_b32 %6, %8\n" IO);)
KERNEL(cex_minmax_nomov, asm volatile(
    "v_min_u32 %8, %0, %1\n v_max_u32 %9, %0, %1\n"
    "v_min_u32 %0, %2, %3\n v_max_u32 %1, %2, %3\n"
    "v_min_u32 %2, %4, %5\n v_max_u32 %3, %4, %5\n"
    "v_min_u32 %4, %6, %7\n v_max_u32 %5, %6, %7\n" IO);)
KERNEL(cex_subco_cnd, asm volatile(
    "v_sub_co_u32 %8, vcc, %0, %1\n v_cndmask_b32 %8, %1, %0, vcc\n v_cndmask_b32 %9, %0, %1, vcc\n"
    "v_sub_co_u32 %0, vcc, %2, %3\n v_cndmask_b32 %0, %3, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n"
    "v_sub_co_u32 %2, vcc, %4, %5\n v_cndmask_b32 %2, %5, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"
    "v_sub_co_u32 %4, vcc, %6, %7\n v_cndmask_b32 %4, %7, %6, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n" IO : : "vcc");)
KERNEL(cex_cmp_cnd, asm volatile(
    "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %8, %1, %0, vcc\n v_cndmask_b32 %9, %0, %1, vcc\n"
    "v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %0, %3, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n"
    "v_cmp_lt_u32 vcc, %4, %5\n v_cndmask_b32 %2, %5, %4, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n"
    "v_cmp_lt_u32 vcc, %6, %7\n v_cndmask_b32 %4, %7, %6, vcc\n v_cndmask_b32 %5, %6, %7, vcc\n" IO : : "vcc");)
#define E8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define E8_3(OP) OP " %0, %0, %8, %9\n" OP " %1, %1, %8, %9\n" OP " %2, %2, %8, %9\n" OP " %3, %3, %8, %9\n" OP " %4, %4, %8, %9\n" OP " %5, %5, %8, %9\n" OP " %6, %6, %8, %9\n" OP " %7, %7, %8, %9\n"
KERNEL(sub_co, asm volatile(
    "v_sub_co_u32 %0, vcc, %0, %8\n v_sub_co_u32 %1, vcc, %1, %8\n v_sub_co_u32 %2, vcc, %2, %8\n v_sub_co_u32 %3, vcc, %3, %8\n"
    "v_sub_co_u32 %4, vcc, %4, %8\n v_sub_co_u32 %5, vcc, %5, %8\n v_sub_co_u32 %6, vcc, %6, %8\n v_sub_co_u32 %7, vcc, %7, %8\n" IO : : "vcc");)
KERNEL(pk_min_u16, asm volatile(E8("v_pk_min_u16") IO);)
KERNEL(pk_max_u16, asm volatile(E8("v_pk_max_u16") IO);)
KERNEL(pk_min_i16, asm volatile(E8("v_pk_min_i16") IO);)
KERNEL(pk_max_f16, asm volatile(E8("v_pk_max_f16") IO);)
KERNEL(pk_minimum3_f16, asm volatile(E8_3("v_pk_minimum3_f16") IO);)
KERNEL(pk_maximum3_f16, asm volatile(E8_3("v_pk_maximum3_f16") IO);)
KERNEL(minimum3_f32, asm volatile(E8_3("v_minimum3_f32") IO);)
KERNEL(bfi_b32, asm volatile(E8_3("v_bfi_b32") IO);)
KERNEL(perm_b32, asm volatile(E8_3("v_perm_b32") IO);)
// three-input boolean forms (bit-sliced counters): gfx950's v_bitop3_b32 (any truth table) beside the two-input ones
#define E8_3T(OP, TAIL) OP " %0, %0, %8, %9" TAIL "\n" OP " %1, %1, %8, %9" TAIL "\n" OP " %2, %2, %8, %9" TAIL "\n" OP " %3, %3, %8, %9" TAIL "\n" OP " %4, %4, %8, %9" TAIL "\n" OP " %5, %5, %8, %9" TAIL "\n" OP " %6, %6, %8, %9" TAIL "\n" OP " %7, %7, %8, %9" TAIL "\n"
KERNEL(bitop3_xor3, asm volatile(E8_3T("v_bitop3_b32", " bitop3:0x96") IO);)
KERNEL(bitop3_maj, asm volatile(E8_3T("v_bitop3_b32", " bitop3:0xe8") IO);)
KERNEL(or_b32, asm volatile(E8("v_or_b32") IO);)
KERNEL(xor_b32, asm volatile(E8("v_xor_b32") IO);)
KERNEL(xnor_b32, asm volatile(E8("v_xnor_b32") IO);)
KERNEL(bitop3_b16, asm volatile(E8_3T("v_bitop3_b16", " bitop3:0x96") IO);)
KERNEL(min_u32_dpp, asm volatile(
    "v_min_u32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_min_u32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_min_u32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
    "v_min_u32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf\n" IO);)
KERNEL(mov_dpp, asm volatile(
    "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
    "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" IO);)
KERNEL(permlane32_swap, asm volatile(
    "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
    "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n" IO);)
KERNEL(permlane16_swap, asm volatile(
    "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
    "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n" IO);)
KERNEL(ds_swizzle, asm volatile(
    "ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM, \"0000p\")\n ds_swizzle_b32 %1, %1 offset:swizzle(BITMASK_PERM, \"0000p\")\n"
    "ds_swizzle_b32 %2, %2 offset:swizzle(BITMASK_PERM, \"0000p\")\n ds_swizzle_b32 %3, %3 offset:swizzle(BITMASK_PERM, \"0000p\")\n"
    "ds_swizzle_b32 %4, %4 offset:swizzle(BITMASK_PERM, \"0000p\")\n ds_swizzle_b32 %5, %5 offset:swizzle(BITMASK_PERM, \"0000p\")\n"
    "ds_swizzle_b32 %6, %6 offset:swizzle(BITMASK_PERM, \"0000p\")\n ds_swizzle_b32 %7, %7 offset:swizzle(BITMASK_PERM, \"0000p\")\n"
    "s_waitcnt lgkmcnt(0)\n" IO);)

// round 3: is any packed 16-bit integer op in the fast (add / and / mov) class? A comparator as saturating subtract +
// subtract + add (d = sat(a - b); min = a - d; max = b + d) would take 3 x 2.3 cycles against 2 x 4.3 for min + max.
#define E8C(OP) OP " %0, %0, %8 clamp\n" OP " %1, %1, %8 clamp\n" OP " %2, %2, %8 clamp\n" OP " %3, %3, %8 clamp\n" OP " %4, %4, %8 clamp\n" OP " %5, %5, %8 clamp\n" OP " %6, %6, %8 clamp\n" OP " %7, %7, %8 clamp\n"
KERNEL(pk_add_u16, asm volatile(E8("v_pk_add_u16") IO);)
KERNEL(pk_sub_u16, asm volatile(E8("v_pk_sub_u16") IO);)
KERNEL(pk_sub_u16_clamp, asm volatile(E8C("v_pk_sub_u16") IO);)
KERNEL(pk_sub_i16, asm volatile(E8("v_pk_sub_i16") IO);)
KERNEL(pk_add_f16, asm volatile(E8("v_pk_add_f16") IO);)
KERNEL(pk_mul_f16, asm volatile(E8("v_pk_mul_f16") IO);)
KERNEL(pk_fma_f16, asm volatile(E8_3("v_pk_fma_f16") IO);)
KERNEL(pk_mul_lo_u16, asm volatile(E8("v_pk_mul_lo_u16") IO);)
KERNEL(pk_mad_u16, asm volatile(E8_3("v_pk_mad_u16") IO);)
KERNEL(pk_lshlrev_b16, asm volatile(E8("v_pk_lshlrev_b16") IO);)
KERNEL(add_u16, asm volatile(E8("v_add_u16") IO);)
KERNEL(sub_u32_clamp, asm volatile(E8C("v_sub_u32") IO);)
KERNEL(add3_u32, asm volatile(E8_3("v_add3_u32") IO);)
KERNEL(and_or_b32, asm volatile(E8_3("v_and_or_b32") IO);)
KERNEL(or3_b32, asm volatile(E8_3("v_or3_b32") IO);)
KERNEL(xad_u32, asm volatile(E8_3("v_xad_u32") IO);)
KERNEL(lshl_add_u32, asm volatile(E8_3("v_lshl_add_u32") IO);)
KERNEL(mad_u32_u24, asm volatile(E8_3("v_mad_u32_u24") IO);)
KERNEL(mul_u32_u24, asm volatile(E8("v_mul_u32_u24") IO);)
KERNEL(sad_u32, asm volatile(E8_3("v_sad_u32") IO);)
KERNEL(sad_u16, asm volatile(E8_3("v_sad_u16") IO);)
KERNEL(cex_satsub_pk, asm volatile(
    "v_pk_sub_u16 %8, %0, %1 clamp\n v_pk_sub_u16 %0, %0, %8\n v_pk_add_u16 %1, %1, %8\n"
    "v_pk_sub_u16 %9, %2, %3 clamp\n v_pk_sub_u16 %2, %2, %9\n v_pk_add_u16 %3, %3, %9\n"
    "v_pk_sub_u16 %8, %4, %5 clamp\n v_pk_sub_u16 %4, %4, %8\n v_pk_add_u16 %5, %5, %8\n"
    "v_pk_sub_u16 %9, %6, %7 clamp\n v_pk_sub_u16 %6, %6, %9\n v_pk_add_u16 %7, %7, %9\n" IO);)
KERNEL(cex_satsub_u32, asm volatile(
    "v_sub_u32 %8, %0, %1 clamp\n v_sub_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n"
    "v_sub_u32 %9, %2, %3 clamp\n v_sub_u32 %2, %2, %9\n v_add_u32 %3, %3, %9\n"
    "v_sub_u32 %8, %4, %5 clamp\n v_sub_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n"
    "v_sub_u32 %9, %6, %7 clamp\n v_sub_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n" IO);)

template <class K> void run(const char* name, K kern, unsigned* d, int per_iter, int units, const char* unit_name) {
    const int iters = 2000, blocks = 256 * 4;   // 512 threads = 8 waves; 4 blocks/CU -> 8 waves/SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 8 * iters * per_iter / 1024.0;     // wave-instructions per SIMD
    const double ns = ms * 1e6 / per_simd;
    printf("%-20s %8.3f ms  %5.2f cycles per wave-instr per SIMD at 2.4 GHz", name, ms, ns * 2.4);
    if (units > 0) printf("  -> %5.2f cycles per %s", ns * 2.4 * per_iter / units, unit_name);
    printf("\n");
}
int main() {
    unsigned* d; hipMalloc(&d, 1 << 22); hipMemset(d, 0, 1 << 22);
    run("cex min/max (+mov)", k_cex_minmax, d, 8 * 12, 8 * 4, "comparator");
    run("cex min/max", k_cex_minmax_nomov, d, 8 * 8, 8 * 4, "comparator");
    run("cex sub_co+2cndmask", k_cex_subco_cnd, d, 8 * 12, 8 * 4, "comparator");
    run("cex cmp+2cndmask", k_cex_cmp_cnd, d, 8 * 12, 8 * 4, "comparator");
    run("v_sub_co_u32", k_sub_co, d, 64, 0, "");
    run("v_pk_min_u16", k_pk_min_u16, d, 64, 0, ""); run("v_pk_max_u16", k_pk_max_u16, d, 64, 0, "");
    run("v_pk_min_i16", k_pk_min_i16, d, 64, 0, ""); run("v_pk_max_f16", k_pk_max_f16, d, 64, 0, "");
    run("v_pk_minimum3_f16", k_pk_minimum3_f16, d, 64, 0, ""); run("v_pk_maximum3_f16", k_pk_maximum3_f16, d, 64, 0, "");
    run("v_minimum3_f32", k_minimum3_f32, d, 64, 0, "");
    run("v_bfi_b32", k_bfi_b32, d, 64, 0, ""); run("v_perm_b32", k_perm_b32, d, 64, 0, "");
    run("v_bitop3_b32 xor3", k_bitop3_xor3, d, 64, 0, ""); run("v_bitop3_b32 maj", k_bitop3_maj, d, 64, 0, "");
    run("v_or_b32", k_or_b32, d, 64, 0, ""); run("v_xor_b32", k_xor_b32, d, 64, 0, ""); run("v_xnor_b32", k_xnor_b32, d, 64, 0, "");
    run("v_bitop3_b16", k_bitop3_b16, d, 64, 0, "");
    run("v_min_u32 dpp", k_min_u32_dpp, d, 64, 0, ""); run("v_mov_b32 dpp", k_mov_dpp, d, 64, 0, "");
    run("v_permlane32_swap", k_permlane32_swap, d, 64, 0, ""); run("v_permlane16_swap", k_permlane16_swap, d, 64, 0, "");
    run("ds_swizzle_b32", k_ds_swizzle, d, 64 + 8, 0, "");
    printf("# round 3: packed 16-bit integer classes and the saturating-subtract comparator\n");
    run("v_pk_add_u16", k_pk_add_u16, d, 64, 0, ""); run("v_pk_sub_u16", k_pk_sub_u16, d, 64, 0, "");
    run("v_pk_sub_u16 clamp", k_pk_sub_u16_clamp, d, 64, 0, ""); run("v_pk_sub_i16", k_pk_sub_i16, d, 64, 0, "");
    run("v_pk_add_f16", k_pk_add_f16, d, 64, 0, ""); run("v_pk_mul_f16", k_pk_mul_f16, d, 64, 0, "");
    run("v_pk_fma_f16", k_pk_fma_f16, d, 64, 0, ""); run("v_pk_mul_lo_u16", k_pk_mul_lo_u16, d, 64, 0, "");
    run("v_pk_mad_u16", k_pk_mad_u16, d, 64, 0, ""); run("v_pk_lshlrev_b16", k_pk_lshlrev_b16, d, 64, 0, "");
    run("v_add_u16", k_add_u16, d, 64, 0, ""); run("v_sub_u32 clamp", k_sub_u32_clamp, d, 64, 0, "");
    run("v_add3_u32", k_add3_u32, d, 64, 0, ""); run("v_and_or_b32", k_and_or_b32, d, 64, 0, "");
    run("v_or3_b32", k_or3_b32, d, 64, 0, ""); run("v_xad_u32", k_xad_u32, d, 64, 0, "");
    run("v_lshl_add_u32", k_lshl_add_u32, d, 64, 0, ""); run("v_mad_u32_u24", k_mad_u32_u24, d, 64, 0, "");
    run("v_mul_u32_u24", k_mul_u32_u24, d, 64, 0, ""); run("v_sad_u32", k_sad_u32, d, 64, 0, ""); run("v_sad_u16", k_sad_u16, d, 64, 0, "");
    run("cex satsub pk_u16", k_cex_satsub_pk, d, 8 * 12, 8 * 4, "comparator (two 16-bit values)");
    run("cex satsub u32", k_cex_satsub_u32, d, 8 * 12, 8 * 4, "comparator");
    return 0;
}
