// What the CU's vector memory path delivers for the gathers of the bit-sliced median (mask_bits.hip): every wave loads whole
// 256-byte (dword per lane), 512-byte (dwordx2) or 1-KB (dwordx4) rows of a table through a buffer resource, 25 rows per
// round with scalar row offsets, nothing else in the loop but an XOR per loaded register. Rows from a 1-MB window (L2 hits)
// or from the whole 26-MB table. Prints bytes per clock and CU at 2.4 GHz and TB/s chip-wide.
//   hipcc -O3 --offload-arch=gfx950 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>

// the same dword rows with the row offset ADDED INTO the lanes' offset (one v_add per load) instead of the scalar offset operand
__global__ __launch_bounds__(256) void kv(const unsigned* table, const int* rows, int n_rows_list, int iters, unsigned* out, int table_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, table_bytes, 0x00020000);
    unsigned acc = 0;
    const int* mine = rows + (wave * 25) % (n_rows_list - 25 * 16);
    for (int it = 0; it < iters; ++it) {
        int off[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) off[i] = __builtin_amdgcn_readfirstlane(mine[(it & 15) * 25 + i]);
#pragma unroll
        for (int i = 0; i < 25; ++i) acc ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4 + off[i], 0, 0);
    }
    out[wave * 64 + lane] = acc;
}

template <int WORDS>
__global__ __launch_bounds__(256) void k(const unsigned* table, const int* rows, int n_rows_list, int iters, unsigned* out, int table_bytes) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(table), 0, table_bytes, 0x00020000);
    unsigned acc = 0;
    const int* mine = rows + (wave * 25) % (n_rows_list - 25 * 16);
    for (int it = 0; it < iters; ++it) {
        int off[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) off[i] = __builtin_amdgcn_readfirstlane(mine[(it & 15) * 25 + i]);
#pragma unroll
        for (int i = 0; i < 25; ++i) {
            if constexpr (WORDS == 1) acc ^= (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rsrc, lane * 4, off[i], 0);
            else if constexpr (WORDS == 2) { auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane * 8, off[i], 0); acc ^= v[0] ^ v[1]; }
            else { auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, off[i], 0); acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
        }
    }
    out[wave * 64 + lane] = acc;
}

// the same rows through global_load_dword (64-bit lane addresses) instead of a buffer resource: 29 against 22 B/clk/CU here --
// in mask_sim_bits_kernel itself the address arithmetic and the register pairs cost more than that (222 against 208 us)
__global__ __launch_bounds__(256) void kg(const unsigned* table, const int* rows, int n_rows_list, int iters, unsigned* out) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned acc = 0;
    const int* mine = rows + (wave * 25) % (n_rows_list - 25 * 16);
    for (int it = 0; it < iters; ++it) {
        const unsigned* base[25];
#pragma unroll
        for (int i = 0; i < 25; ++i) base[i] = table + (__builtin_amdgcn_readfirstlane(mine[(it & 15) * 25 + i]) >> 2);
#pragma unroll
        for (int i = 0; i < 25; ++i) acc ^= base[i][lane];
    }
    out[wave * 64 + lane] = acc;
}
void run_global(const unsigned* table, const int* rows, int n_list, unsigned* out, int waves_per_simd, const char* what) {
    const int iters = 400, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kg, dim3(blocks), dim3(256), 0, 0, table, rows, n_list, 4, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kg, dim3(blocks), dim3(256), 0, 0, table, rows, n_list, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * 25 * 256;
    printf("%-8s global_load_dword  %d waves/SIMD: %7.3f ms  %6.1f B/clk/CU  %5.2f TB/s\n", what, waves_per_simd, ms, bytes / 256 / (ms * 1e-3 * 2.4e9), bytes / ms / 1e9);
}

void run_voff(const unsigned* table, const int* rows, int n_list, unsigned* out, int table_bytes, int waves_per_simd, const char* what) {
    const int iters = 400, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kv, dim3(blocks), dim3(256), 0, 0, table, rows, n_list, 4, out, table_bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kv, dim3(blocks), dim3(256), 0, 0, table, rows, n_list, iters, out, table_bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * 25 * 256;
    printf("%-8s buffer_load_dword, row in the lane offset  %d waves/SIMD: %7.3f ms  %6.1f B/clk/CU  %5.2f TB/s\n", what, waves_per_simd, ms, bytes / 256 / (ms * 1e-3 * 2.4e9), bytes / ms / 1e9);
}

template <int WORDS> void run(const unsigned* table, const int* rows, int n_list, unsigned* out, int table_bytes, int waves_per_simd, const char* what) {
    const int iters = 400, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<WORDS>, dim3(blocks), dim3(256), 0, 0, table, rows, n_list, 4, out, table_bytes);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<WORDS>, dim3(blocks), dim3(256), 0, 0, table, rows, n_list, iters, out, table_bytes);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 4 * iters * 25 * 256 * WORDS;
    printf("%-8s dwordx%d  %d waves/SIMD: %7.3f ms  %6.1f B/clk/CU  %5.2f TB/s\n", what, WORDS, waves_per_simd, ms, bytes / 256 / (ms * 1e-3 * 2.4e9), bytes / ms / 1e9);
}

int main() {
    const int T = 7753, row_bytes = 13 * 256;
    const int table_bytes = T * row_bytes;
    unsigned* table; hipMalloc(&table, table_bytes); hipMemset(table, 1, table_bytes);
    unsigned* out; hipMalloc(&out, 256 * 8 * 4 * 64 * 4);
    std::mt19937 rng(3);
    for (int local = 1; local >= 0; --local) {
        const int n_list = 1 << 20;
        std::vector<int> rows(n_list);
        for (auto& r : rows) r = (int)((local ? rng() % 300 : rng() % T) * row_bytes + (rng() % 10) * 256);     // a row of some plane (a KB-row may run into the next planes)
        int* d_rows; hipMalloc(&d_rows, n_list * 4); hipMemcpy(d_rows, rows.data(), n_list * 4, hipMemcpyHostToDevice);
        for (int w : {2, 4, 8}) {
            run<1>(table, d_rows, n_list, out, table_bytes, w, local ? "window" : "table");
            run<2>(table, d_rows, n_list, out, table_bytes, w, local ? "window" : "table");
            run<4>(table, d_rows, n_list, out, table_bytes, w, local ? "window" : "table");
            run_global(table, d_rows, n_list, out, w, local ? "window" : "table");
            run_voff(table, d_rows, n_list, out, table_bytes, w, local ? "window" : "table");
        }
        hipFree(d_rows);
    }
    return 0;
}
