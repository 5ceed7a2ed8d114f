// Host <-> device paths for a 127 MB float64 clip (cfg 2: 7 938 000 x 2 samples): what the drop-in call can use.
//   a) pageable hipMemcpy (what repet_ctx_upload did in round 1)
//   b) hipHostRegister in place + one DMA + hipHostUnregister
//   c) staged: N host threads convert f64 -> f32 into a pinned ring, chunk by chunk, each chunk DMA'd as soon as it is full
//      (and the mirror image for the result: DMA f32 chunks into the ring, threads widen them into the caller's array)
// Build: hipcc --offload-arch=gfx950 -O3 -pthread h2d_paths.hip -o h2d_paths
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t n = 7938000ull * 2;
    double* host = (double*)aligned_alloc(4096, n * 8);
    double* host_out = (double*)aligned_alloc(4096, n * 8);
    for (size_t i = 0; i < n; ++i) host[i] = (double)(i % 1000) * 1e-3;
    memset(host_out, 0, n * 8);
    void* dev; CK(hipMalloc(&dev, n * 8));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    printf("host threads available: %u\n", std::thread::hardware_concurrency());
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipMemcpy(dev, host, n * 8, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(host_out, dev, n * 8, hipMemcpyDeviceToHost)); double t2 = now();
        printf("a) pageable f64: H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, n * 8 / (t1 - t0) / 1e9, (t2 - t1) * 1e3, n * 8 / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); CK(hipHostRegister(host, n * 8, hipHostRegisterDefault)); double t1 = now();
        CK(hipMemcpyAsync(dev, host, n * 8, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); double t2 = now();
        CK(hipHostUnregister(host)); double t3 = now();
        printf("b) register %.2f ms, DMA H2D %.2f ms (%.1f GB/s), unregister %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3, n * 8 / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
    }
    // c) staged f64 -> f32
    const size_t chunk = 1 << 20;            // samples per chunk (4 MB f32)
    const int ring = 8;
    float* pinned; CK(hipHostMalloc((void**)&pinned, ring * chunk * 4, hipHostMallocDefault));
    hipEvent_t ev[ring]; for (int k = 0; k < ring; ++k) CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    const size_t n_chunks = (n + chunk - 1) / chunk;
    for (int nt : {1, 2, 4, 8, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            for (size_t c = 0; c < n_chunks; ++c) {
                const int slot = c % ring;
                if (c >= (size_t)ring) CK(hipEventSynchronize(ev[slot]));
                const size_t lo = c * chunk, cnt = std::min(chunk, n - lo);
                float* dst = pinned + slot * chunk;
                std::vector<std::thread> th;
                for (int t = 1; t < nt; ++t)
                    th.emplace_back([=] { for (size_t i = cnt * t / nt; i < cnt * (t + 1) / nt; ++i) dst[i] = (float)host[lo + i]; });
                for (size_t i = 0; i < cnt / nt; ++i) dst[i] = (float)host[lo + i];
                for (auto& x : th) x.join();
                CK(hipMemcpyAsync((float*)dev + lo, dst, cnt * 4, hipMemcpyHostToDevice, s));
                CK(hipEventRecord(ev[slot], s));
            }
            CK(hipStreamSynchronize(s));
            double t1 = now();
            // and back: f32 chunks into the ring, widened into the caller's float64 array
            for (size_t c = 0; c < n_chunks + 1; ++c) {
                if (c < n_chunks) {
                    const int slot = c % ring;
                    const size_t lo = c * chunk, cnt = std::min(chunk, n - lo);
                    CK(hipMemcpyAsync(pinned + slot * chunk, (float*)dev + lo, cnt * 4, hipMemcpyDeviceToHost, s));
                    CK(hipEventRecord(ev[slot], s));
                }
                if (c >= 1) {
                    const size_t cc = c - 1;
                    const int slot = cc % ring;
                    CK(hipEventSynchronize(ev[slot]));
                    const size_t lo = cc * chunk, cnt = std::min(chunk, n - lo);
                    const float* src = pinned + slot * chunk;
                    std::vector<std::thread> th;
                    for (int t = 1; t < nt; ++t)
                        th.emplace_back([=] { for (size_t i = cnt * t / nt; i < cnt * (t + 1) / nt; ++i) host_out[lo + i] = (double)src[i]; });
                    for (size_t i = 0; i < cnt / nt; ++i) host_out[lo + i] = (double)src[i];
                    for (auto& x : th) x.join();
                }
            }
            double t2 = now();
            printf("c) staged, %2d threads: in %.2f ms (%.1f GB/s of f64), out %.2f ms\n", nt, (t1 - t0) * 1e3, n * 8 / (t1 - t0) / 1e9, (t2 - t1) * 1e3);
        }
    }
    return 0;
}
