// Pinned-memory copies by NUMA node: where does a pinned buffer have to live (and who has to have allocated it) for the
// H2D / D2H DMA of this GPU to run at link speed?   hipcc -O2 --offload-arch=gfx950 h2d_numa.hip -o h2d_numa -lnuma? (no: sched only)
#include <hip/hip_runtime.h>
#include <sched.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

static std::vector<int> cpus_of_node(int node) {
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string s; std::getline(f, s);
    std::vector<int> out;
    size_t i = 0;
    while (i < s.size()) {
        size_t j = s.find(',', i); if (j == std::string::npos) j = s.size();
        std::string part = s.substr(i, j - i);
        size_t d = part.find('-');
        int a = std::stoi(part.substr(0, d)), b = d == std::string::npos ? a : std::stoi(part.substr(d + 1));
        for (int c = a; c <= b; ++c) out.push_back(c);
        i = j + 1;
    }
    return out;
}
static void pin(const std::vector<int>& cpus) {
    cpu_set_t set; CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    sched_setaffinity(0, sizeof(set), &set);
}
int main() {
    const size_t bytes = 64u << 20;
    void* dev; hipMalloc(&dev, bytes);
    hipStream_t s; hipStreamCreate(&s);
    for (int node = 0; node < 2; ++node) {
        auto cpus = cpus_of_node(node);
        if (cpus.empty()) continue;
        pin(cpus);
        for (unsigned flags : {0u, (unsigned)hipHostMallocNumaUser}) {
            void* host = nullptr;
            if (hipHostMalloc(&host, bytes, flags) != hipSuccess) { printf("alloc failed\n"); continue; }
            memset(host, 1, bytes);
            for (int dir = 0; dir < 2; ++dir) {
                double best = 1e9;
                for (int rep = 0; rep < 6; ++rep) {
                    auto t0 = std::chrono::steady_clock::now();
                    if (dir == 0) hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, s);
                    else hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s);
                    hipStreamSynchronize(s);
                    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (rep > 0 && dt < best) best = dt;
                }
                printf("thread on node %d, hipHostMalloc flags %s, %s: %.2f ms for 64 MiB = %.1f GB/s\n", node, flags ? "NumaUser" : "default ", dir ? "D2H" : "H2D",
                       best * 1e3, bytes / best / 1e9);
            }
            // a host thread narrowing float64 -> float32 into this buffer (one thread, its own node): GB/s of source read
            {
                std::vector<double> src(bytes / 4, 0.5);
                float* dstf = static_cast<float*>(host);
                double best = 1e9;
                for (int rep = 0; rep < 4; ++rep) {
                    auto t0 = std::chrono::steady_clock::now();
                    for (size_t i = 0; i < src.size(); ++i) dstf[i] = (float)src[i];
                    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (rep > 0 && dt < best) best = dt;
                }
                printf("    one thread narrowing %zu MB of float64 into it: %.2f ms (%.1f GB/s read)\n", src.size() * 8 >> 20, best * 1e3, src.size() * 8 / best / 1e9);
            }
            hipHostFree(host);
        }
    }
    return 0;
}
