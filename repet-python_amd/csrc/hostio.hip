// Host <-> device movement of waveforms for the drop-in call (repet.sim(audio_signal, fs): float64 NumPy in, float64
// NumPy out -- repet.py:571-709 keeps everything in host RAM, so for the engine the two copies ARE the call).
//
// Measured on the MI355X host (tools/microbench/h2d_paths.hip, 127 MB of float64 = the cfg-2 clip):
//   * a pageable hipMemcpy of memory the runtime has not pinned before: 19 ms (6.6 GB/s); explicit hipHostRegister 4.2 ms
//     + DMA 2.25 ms; a result array fresh from np.empty additionally page-faults on first touch;
//   * a few host threads narrowing float64 -> float32 into a ring of pinned 4 MB chunks, each chunk DMA'd as soon as it
//     is full: 2.6 ms, independent of what the runtime has cached -- and half the bytes cross PCIe.
// So uploads and downloads are STAGED: a small process-wide pool of worker threads converts between the caller's
// array (any dtype the ABI takes) and fp32 in a per-context pinned ring while the previous chunks are in flight.
// The caller's memory is only touched by those threads, during the call: no pointer is retained.
// Results can be written into pinned host buffers handed out by repet_host_alloc (a recycling pool): the Python
// module wraps them as NumPy arrays, so a result array is already faulted-in and pinned when the next call fills it.
#include "common.h"

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <sched.h>
#include <cstring>
#include <cmath>
#include <functional>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

namespace repet {

// ---- worker pool -----------------------------------------------------------------------------------------------
// The CPUs of the NUMA node the current GPU hangs off, as far as this process may use them (empty: unknown, one node only,
// or REPET_HOST_NUMA=0). Measured on the 2-socket MI355X host (tools/dropin_probe.py under taskset): with the conversion
// threads on the GPU's node a 180-s clip goes up in 1.50 ms and the drop-in call takes 3.84 ms; on the other node 1.86 and
// 4.95; left to the scheduler 1.78 and 4.33. The pool's own threads are therefore kept on that node -- the caller's thread
// (which converts one part in eight and issues the copies) is the caller's business and is left where it is.
std::vector<int> host_cpus_near_device(int dev) {
    std::vector<int> out;
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus) - 1, dev) != hipSuccess) return out;
    for (char* q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
    auto read_line = [](const std::string& path) -> std::string {
        std::string line;
        if (FILE* f = fopen(path.c_str(), "r")) { char buf[4096]; if (fgets(buf, sizeof(buf), f)) line = buf; fclose(f); }
        while (!line.empty() && (line.back() == '\n' || line.back() == ' ')) line.pop_back();
        return line;
    };
    const std::string node = read_line(std::string("/sys/bus/pci/devices/") + bus + "/numa_node");
    if (node.empty() || node[0] == '-') return out;
    const std::string list = read_line("/sys/devices/system/node/node" + node + "/cpulist");
    cpu_set_t allowed;
    if (list.empty() || sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
    size_t i = 0;
    while (i < list.size()) {                                   // "64-127,192-255"
        size_t j = list.find(',', i);
        if (j == std::string::npos) j = list.size();
        const std::string part = list.substr(i, j - i);
        const size_t dash = part.find('-');
        const int a = atoi(part.substr(0, dash).c_str());
        const int b = dash == std::string::npos ? a : atoi(part.substr(dash + 1).c_str());
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &allowed)) out.push_back(c);
        i = j + 1;
    }
    if ((int)out.size() == CPU_COUNT(&allowed)) out.clear();    // (the whole mask: nothing to choose)
#endif
    return out;
}

namespace {

class HostWorkers {
public:
    static HostWorkers& get() {
        static HostWorkers* pool = new HostWorkers();      // never destroyed: threads may outlive static destructors
        return *pool;
    }
    int size() const { return n_threads_; }
    // fn(part, n_parts) on n_parts = size() parts, the caller running part 0; returns when all are done.
    // One job at a time (callers from different host threads take turns).
    // A staged copy is thirty-odd such jobs back to back, one per 4-MB chunk: a worker that has just finished a part SPINS
    // for the next job for a while before it goes back to sleep on the condition variable, and the caller spins for the parts
    // -- woken through the condition variable every time, the hand-over cost 30 .. 50 us per chunk, a millisecond per clip.
    void run(const std::function<void(int, int)>& fn, size_t work_items = ~(size_t)0) {
        if (!start(fn, work_items)) { fn(0, 1); return; }
        finish(fn);
    }
    // The same in two steps: start() hands the job to the workers and returns (false: too small, the caller runs fn(0, 1)
    // itself at finish time); the caller does something else -- enqueue the previous chunk's DMA -- and then finish() runs its
    // own part and waits for the others. A job that was started must be finished by the same thread.
    bool start(const std::function<void(int, int)>& fn, size_t work_items = ~(size_t)0) {
        if (n_threads_ <= 1 || work_items < 65536) return false;     // not worth waking anybody up
        job_mutex_.lock();
        note_device();
        fn_ = &fn;
        pending_.store(n_threads_ - 1, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> lk(m_);
            generation_.fetch_add(1, std::memory_order_release);
        }
        if (sleepers_.load(std::memory_order_acquire) > 0) cv_.notify_all();
        return true;
    }
    void finish(const std::function<void(int, int)>& fn) {
        fn(0, n_threads_);
        for (int spins = 0; pending_.load(std::memory_order_acquire) != 0; ++spins) {
            if (spin_ok_ && spins < 20000) cpu_relax(); else std::this_thread::yield();
        }
        fn_ = nullptr;
        job_mutex_.unlock();
    }

private:
    static void cpu_relax() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        __builtin_ia32_pause();
#else
        std::this_thread::yield();
#endif
    }
    // The workers are pinned to the NUMA node of the device that was current when the pool was made. A process that drives
    // devices on BOTH sockets (repet_run_batch, one host thread per device, all sharing this pool) would convert half its
    // clips on the far node -- measured slower than leaving the placement to the scheduler (4.95 against 4.33 ms per call):
    // the first job that arrives for a device on another node lifts the pinning for good.
    void note_device() {
        if (near_cpus_.empty() || unpin_.load(std::memory_order_relaxed)) return;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev == pinned_device_) return;
        if (dev == last_other_device_) return;                      // (already compared: same node)
        if (host_cpus_near_device(dev) != near_cpus_) unpin_.store(true, std::memory_order_release);
        else last_other_device_ = dev;
    }
    static std::vector<int> gpu_node_cpus() {
        const char* off = getenv("REPET_HOST_NUMA");
        if (off && off[0] == '0') return {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return {};
        return host_cpus_near_device(dev);
    }
    // CPUs this process may run on (its affinity mask / container quota as the scheduler reports it)
    static int usable_cpus() {
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) return std::max(1, CPU_COUNT(&set));
#endif
        return std::max(1u, std::thread::hardware_concurrency());
    }
    HostWorkers() {
        const char* e = getenv("REPET_HOST_THREADS");
        int n = e ? atoi(e) : 0;
        if (n <= 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            // measured on the MI355X host (256 hardware threads), repet.sim of a 180-s stereo clip end to end:
            // 2 threads 9.3 ms, 4: 6.2, 6: 5.9, 8: 4.7, 12: 6.2 -- a handful saturates one PCIe link
            n = hw >= 32 ? 8 : (hw >= 8 ? 4 : (hw >= 4 ? 2 : 1));
        }
        n_threads_ = std::min(n, 32);
        (void)hipGetDevice(&pinned_device_);
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
        have_all_cpus_ = sched_getaffinity(0, sizeof(all_cpus_), &all_cpus_) == 0;
#endif
        near_cpus_ = gpu_node_cpus();
        if ((int)near_cpus_.size() < n_threads_) near_cpus_.clear();
        // more threads than CPUs (REPET_HOST_THREADS above a container's share): spinners would take the CPU from the thread
        // that converts part 0 -- hand over through the condition variable then
        spin_ok_ = n_threads_ <= usable_cpus();
        for (int k = 1; k < n_threads_; ++k) std::thread([this, k] { loop(k); }).detach();
    }
    void loop(int part) {
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
        if (!near_cpus_.empty()) {
            cpu_set_t set;
            CPU_ZERO(&set);
            for (int c : near_cpus_) CPU_SET(c, &set);
            (void)sched_setaffinity(0, sizeof(set), &set);          // (this worker only: tid 0 = the calling thread)
        }
#endif
        unsigned seen = 0;
        [[maybe_unused]] bool pinned = !near_cpus_.empty();
        for (;;) {
            // the next job: spin for about 200 us of WALL time (the gaps between the chunks of one copy), then sleep
            bool have = generation_.load(std::memory_order_acquire) != seen;
            if (!have && spin_ok_) {
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(200);
                for (int spins = 0; !have; ++spins) {
                    if (generation_.load(std::memory_order_acquire) != seen) { have = true; break; }
                    if ((spins & 255) == 255 && std::chrono::steady_clock::now() >= until) break;
                    cpu_relax();
                }
            }
            if (!have) {
                std::unique_lock<std::mutex> lk(m_);
                sleepers_.fetch_add(1, std::memory_order_release);
                cv_.wait(lk, [&] { return generation_.load(std::memory_order_acquire) != seen; });
                sleepers_.fetch_sub(1, std::memory_order_release);
            }
            seen = generation_.load(std::memory_order_acquire);
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
            if (pinned && unpin_.load(std::memory_order_acquire)) {       // devices on several nodes: back to the process's own mask
                if (have_all_cpus_) (void)sched_setaffinity(0, sizeof(all_cpus_), &all_cpus_);
                pinned = false;
            }
#endif
            const std::function<void(int, int)>* fn = fn_;
            (*fn)(part, n_threads_);
            pending_.fetch_sub(1, std::memory_order_release);
        }
    }
    int n_threads_ = 1;
    bool spin_ok_ = true;
    std::vector<int> near_cpus_;
    int pinned_device_ = 0, last_other_device_ = -1;
    std::atomic<bool> unpin_{false};
    [[maybe_unused]] bool have_all_cpus_ = false;
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
    cpu_set_t all_cpus_;
#endif
    std::mutex job_mutex_, m_;
    std::condition_variable cv_;
    const std::function<void(int, int)>* fn_ = nullptr;
    std::atomic<int> pending_{0}, sleepers_{0};
    std::atomic<unsigned> generation_{0};
};

// The conversion loops are compiled three times (baseline x86-64, AVX2, AVX-512) and picked at load time: at two doubles per
// instruction the narrowing of a 127-MB clip was instruction-bound on the eight threads (1.7 ms for a copy that takes 1.15).
#if defined(__HIP_DEVICE_COMPILE__) || !defined(__x86_64__)
#define REPET_HOST_CLONES
#else
#define REPET_HOST_CLONES __attribute__((target_clones("default", "avx2", "avx512f")))
#endif
// (every converter also says whether it met a sample that is not finite: the drop-in rejects such input -- INTEGRATION.md)
REPET_HOST_CLONES bool narrow_f64(const double* src, float* dst, size_t lo, size_t hi) {
    bool bad = false;
    for (size_t i = lo; i < hi; ++i) {
        const double x = src[i];
        dst[i] = (float)x;
        bad |= !(std::fabs(x) <= std::numeric_limits<double>::max());
    }
    return bad;
}
REPET_HOST_CLONES bool narrow_f32(const float* src, float* dst, size_t lo, size_t hi) {
    bool bad = false;
    for (size_t i = lo; i < hi; ++i) {
        const float x = src[i];
        dst[i] = x;
        bad |= !(std::fabs(x) <= std::numeric_limits<float>::max());
    }
    return bad;
}
REPET_HOST_CLONES void narrow_i16(const int16_t* src, float* dst, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) dst[i] = (float)src[i];
}
REPET_HOST_CLONES void widen_f32_plain(const float* src, double* dst, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) dst[i] = (double)src[i];
}
// Round 6: the big conversions write with NON-TEMPORAL stores. A plain store of a line the core does not own reads the line
// first: widening a 3-minute stereo result (127 MB of float64) read those 127 MB for nothing and pushed the caller's other
// data out of the caches; the staging ring is read by the DMA engine only. Streams of whole 64-byte lines (the head of a part
// goes the plain way up to the first line boundary); a store fence closes every part -- the hand-over to the thread that
// issues the DMA (or returns the result) is an ordinary atomic, which non-temporal stores are not ordered with.
// REPET_HOST_NT=0: plain stores everywhere (A/B).
static bool host_nt() {
    static const bool on = [] { const char* e = getenv("REPET_HOST_NT"); return !(e && e[0] == '0'); }();
    return on;
}
static inline void store_fence() {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_sfence();
#endif
}
// (explicit intrinsics: hipcc's host pass drops __builtin_nontemporal_store's hint AND leaves such a loop scalar)
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define REPET_HOST_X86 1
__attribute__((target("avx512f"))) static void widen_lines_512(const float* s, double* d, size_t n) {      // d on a 64-byte boundary
    size_t k = 0;
    for (; k + 16 <= n; k += 16) {
        _mm512_stream_pd(d + k, _mm512_cvtps_pd(_mm256_loadu_ps(s + k)));
        _mm512_stream_pd(d + k + 8, _mm512_cvtps_pd(_mm256_loadu_ps(s + k + 8)));
    }
    for (; k < n; ++k) d[k] = (double)s[k];
}
__attribute__((target("avx2"))) static void widen_lines_256(const float* s, double* d, size_t n) {
    size_t k = 0;
    for (; k + 8 <= n; k += 8) {
        _mm256_stream_pd(d + k, _mm256_cvtps_pd(_mm_loadu_ps(s + k)));
        _mm256_stream_pd(d + k + 4, _mm256_cvtps_pd(_mm_loadu_ps(s + k + 4)));
    }
    for (; k < n; ++k) d[k] = (double)s[k];
}
// float64 -> fp32 sample + fp32 remainder, both planes as whole lines (ph, pl on 64-byte boundaries); true: a sample was not finite
__attribute__((target("avx512f"))) static bool split_lines_512(const double* s, float* ph, float* pl, size_t n) {
    const __m512d big = _mm512_set1_pd(std::numeric_limits<double>::max());
    __mmask8 fine = 0xff;
    size_t k = 0;
    for (; k + 8 <= n; k += 8) {
        const __m512d x = _mm512_loadu_pd(s + k);
        const __m256 h = _mm512_cvtpd_ps(x);
        _mm256_stream_ps(ph + k, h);
        _mm256_stream_ps(pl + k, _mm512_cvtpd_ps(_mm512_sub_pd(x, _mm512_cvtps_pd(h))));
        fine &= _mm512_cmp_pd_mask(_mm512_abs_pd(x), big, _CMP_LE_OQ);
    }
    bool bad = fine != 0xff;
    for (; k < n; ++k) {
        const double x = s[k];
        const float h = (float)x;
        ph[k] = h;
        pl[k] = (float)(x - (double)h);
        bad |= !(std::fabs(x) <= std::numeric_limits<double>::max());
    }
    return bad;
}
__attribute__((target("avx2"))) static bool split_lines_256(const double* s, float* ph, float* pl, size_t n) {
    const __m256d big = _mm256_set1_pd(std::numeric_limits<double>::max());
    const __m256d sign = _mm256_set1_pd(-0.0);
    int fine = 0xf;
    size_t k = 0;
    for (; k + 4 <= n; k += 4) {
        const __m256d x = _mm256_loadu_pd(s + k);
        const __m128 h = _mm256_cvtpd_ps(x);
        _mm_stream_ps(ph + k, h);
        _mm_stream_ps(pl + k, _mm256_cvtpd_ps(_mm256_sub_pd(x, _mm256_cvtps_pd(h))));
        fine &= _mm256_movemask_pd(_mm256_cmp_pd(_mm256_andnot_pd(sign, x), big, _CMP_LE_OQ));
    }
    bool bad = fine != 0xf;
    for (; k < n; ++k) {
        const double x = s[k];
        const float h = (float)x;
        ph[k] = h;
        pl[k] = (float)(x - (double)h);
        bad |= !(std::fabs(x) <= std::numeric_limits<double>::max());
    }
    return bad;
}
// float64 -> fp32 samples only (ph on a 64-byte boundary), the remainders TESTED: bit 0 of the result = a remainder is not
// zero, bit 1 = a sample is not finite
__attribute__((target("avx512f"))) static int narrow_test_lines_512(const double* s, float* ph, size_t n) {
    const __m512d big = _mm512_set1_pd(std::numeric_limits<double>::max());
    __mmask8 fine = 0xff, exact = 0xff;
    size_t k = 0;
    for (; k + 8 <= n; k += 8) {
        const __m512d x = _mm512_loadu_pd(s + k);
        const __m256 h = _mm512_cvtpd_ps(x);
        _mm256_stream_ps(ph + k, h);
        exact &= _mm512_cmp_pd_mask(x, _mm512_cvtps_pd(h), _CMP_EQ_OQ);        // (NaN: not equal -- its remainder must travel)
        fine &= _mm512_cmp_pd_mask(_mm512_abs_pd(x), big, _CMP_LE_OQ);
    }
    int out = (exact != 0xff ? 1 : 0) | (fine != 0xff ? 2 : 0);
    for (; k < n; ++k) {
        const double x = s[k];
        const float h = (float)x;
        ph[k] = h;
        out |= (x != (double)h) ? 1 : 0;
        out |= !(std::fabs(x) <= std::numeric_limits<double>::max()) ? 2 : 0;
    }
    return out;
}
__attribute__((target("avx2"))) static int narrow_test_lines_256(const double* s, float* ph, size_t n) {
    const __m256d big = _mm256_set1_pd(std::numeric_limits<double>::max());
    const __m256d sign = _mm256_set1_pd(-0.0);
    int fine = 0xf, exact = 0xf;
    size_t k = 0;
    for (; k + 4 <= n; k += 4) {
        const __m256d x = _mm256_loadu_pd(s + k);
        const __m128 h = _mm256_cvtpd_ps(x);
        _mm_stream_ps(ph + k, h);
        exact &= _mm256_movemask_pd(_mm256_cmp_pd(x, _mm256_cvtps_pd(h), _CMP_EQ_OQ));
        fine &= _mm256_movemask_pd(_mm256_cmp_pd(_mm256_andnot_pd(sign, x), big, _CMP_LE_OQ));
    }
    int out = (exact != 0xf ? 1 : 0) | (fine != 0xf ? 2 : 0);
    for (; k < n; ++k) {
        const double x = s[k];
        const float h = (float)x;
        ph[k] = h;
        out |= (x != (double)h) ? 1 : 0;
        out |= !(std::fabs(x) <= std::numeric_limits<double>::max()) ? 2 : 0;
    }
    return out;
}
static int host_simd() {            // 2: AVX-512F, 1: AVX2, 0: neither
    static const int level = __builtin_cpu_supports("avx512f") ? 2 : (__builtin_cpu_supports("avx2") ? 1 : 0);
    return level;
}
#endif
void widen_f32(const float* src, double* dst, size_t lo, size_t hi) {
#ifdef REPET_HOST_X86
    if (host_nt() && hi - lo >= 4096 && host_simd() > 0) {
        size_t i = lo;
        for (; i < hi && (reinterpret_cast<uintptr_t>(dst + i) & 63); ++i) dst[i] = (double)src[i];
        if (host_simd() == 2) widen_lines_512(src + i, dst + i, hi - i); else widen_lines_256(src + i, dst + i, hi - i);
        store_fence();
        return;
    }
#endif
    widen_f32_plain(src, dst, lo, hi);
}

// float64 -> the fp32 sample and the fp32 remainder (hi + lo carries 48 bits of the double); true when a remainder is not zero
// Samples that came from PCM or fp32 data (what wavread returns) have no remainders at all: the part is narrowed in blocks
// whose remainders are only TESTED (one OR per block, nothing stored), and the remainder plane of the part is written -- zeros
// up to there, values from there on -- once a block has met one that is not zero.
REPET_HOST_CLONES bool split_part(const double* src, float* dst_hi, float* dst_lo, size_t lo, size_t hi, bool* not_finite) {
    constexpr size_t kBlock = 1024;
    size_t i = lo;
    bool bad = false;
#ifdef REPET_HOST_X86
    const bool lines = host_nt() && host_simd() > 0 && hi - lo >= 4096;
    if (lines) {                                  // the head of the part up to the first line boundary, the plain way
        bool any = false;
        size_t k = lo;
        for (; k < hi && (reinterpret_cast<uintptr_t>(dst_hi + k) & 63); ++k) {
            const double x = src[k];
            const float h = (float)x;
            dst_hi[k] = h;
            any |= (x != (double)h);
            bad |= !(std::fabs(x) <= std::numeric_limits<double>::max());
        }
        if (any) i = lo;                          // (a remainder already: the second loop redoes the part from its start)
        else {
            for (i = k; i < hi; i += kBlock) {
                const size_t end = std::min(hi, i + kBlock);
                const int r = host_simd() == 2 ? narrow_test_lines_512(src + i, dst_hi + i, end - i) : narrow_test_lines_256(src + i, dst_hi + i, end - i);
                bad |= (r & 2) != 0;
                if (r & 1) break;
            }
            store_fence();
        }
    }
    if (!lines)
#endif
    for (; i < hi; i += kBlock) {
        const size_t end = std::min(hi, i + kBlock);
        bool any = false;
        for (size_t k = i; k < end; ++k) {
            const double x = src[k];
            const float h = (float)x;
            dst_hi[k] = h;
            any |= (x != (double)h);    // (a NaN sample counts: its remainder is NaN, which is what hi + lo must say)
            bad |= !(std::fabs(x) <= std::numeric_limits<double>::max());
        }
        if (any) break;
    }
    if (i < hi) {
        std::memset(dst_lo + lo, 0, (i - lo) * sizeof(float));
        size_t k = i;
#ifdef REPET_HOST_X86
        if (host_nt() && host_simd() > 0 && ((reinterpret_cast<uintptr_t>(dst_hi) ^ reinterpret_cast<uintptr_t>(dst_lo)) & 63) == 0) {
            // both planes as streams of whole lines (they share their alignment: same offset into two 64-byte-aligned buffers)
            for (; k < hi && (reinterpret_cast<uintptr_t>(dst_hi + k) & 63); ++k) {
                const double x = src[k];
                const float h = (float)x;
                dst_hi[k] = h;
                dst_lo[k] = (float)(x - (double)h);
                bad |= !(std::fabs(x) <= std::numeric_limits<double>::max());
            }
            bad |= host_simd() == 2 ? split_lines_512(src + k, dst_hi + k, dst_lo + k, hi - k) : split_lines_256(src + k, dst_hi + k, dst_lo + k, hi - k);
            store_fence();
            k = hi;
        }
#endif
        for (; k < hi; ++k) {
            const double x = src[k];
            const float h = (float)x;
            dst_hi[k] = h;
            dst_lo[k] = (float)(x - (double)h);
            bad |= !(std::fabs(x) <= std::numeric_limits<double>::max());
        }
    }
    if (bad) *not_finite = true;
    return i < hi;
}

}  // namespace

// ---- staging ring ------------------------------------------------------------------------------------------------
StagingRing::~StagingRing() { release(); }

void StagingRing::release() {
    for (int k = 0; k < kSlots; ++k) {
        if (event[k]) { (void)hipEventSynchronize(event[k]); (void)hipEventDestroy(event[k]); event[k] = nullptr; }
        busy[k] = false;
    }
    if (base) { (void)hipHostFree(base); base = nullptr; }
    if (base_lo) { (void)hipHostFree(base_lo); base_lo = nullptr; }
    if (lo_done) { (void)hipEventSynchronize(lo_done); (void)hipEventDestroy(lo_done); lo_done = nullptr; }
    if (hi_done) { (void)hipEventDestroy(hi_done); hi_done = nullptr; }
    if (lo_full) { (void)hipHostFree(lo_full); lo_full = nullptr; lo_full_elems = 0; }
    lo_in_flight = false;
}

hipError_t StagingRing::ensure_lo_full(size_t elems) {
    if (!lo_done) {
        hipError_t e = hipEventCreateWithFlags(&lo_done, hipEventDisableTiming);
        if (e != hipSuccess) { lo_done = nullptr; return e; }
        e = hipEventCreateWithFlags(&hi_done, hipEventDisableTiming);
        if (e != hipSuccess) { hi_done = nullptr; return e; }
    }
    if (lo_in_flight) { const hipError_t w = hipEventSynchronize(lo_done); if (w != hipSuccess) return w; lo_in_flight = false; }   // the last plane has left
    if (lo_full_elems >= elems) return hipSuccess;
    if (lo_full) { (void)hipHostFree(lo_full); lo_full = nullptr; lo_full_elems = 0; }
    const size_t want = elems + elems / 8;
    const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&lo_full), want * sizeof(float), hipHostMallocDefault);
    if (e != hipSuccess) { lo_full = nullptr; return e; }
    lo_full_elems = want;
    return hipSuccess;
}

hipError_t StagingRing::ensure_lo() {
    if (base_lo) return hipSuccess;
    const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&base_lo), (size_t)kSlots * kSlotElems * sizeof(float), hipHostMallocDefault);
    if (e != hipSuccess) base_lo = nullptr;
    return e;
}

hipError_t StagingRing::ensure() {
    if (base) return hipSuccess;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&base), (size_t)kSlots * kSlotElems * sizeof(float), hipHostMallocDefault);
    if (e != hipSuccess) { base = nullptr; return e; }
    for (int k = 0; k < kSlots; ++k) {
        e = hipEventCreateWithFlags(&event[k], hipEventDisableTiming);
        if (e != hipSuccess) { release(); return e; }      // (a half-built ring must not pass the next ensure())
    }
    return hipSuccess;
}

// src (host, `dtype` elements) -> dst (device fp32), `count` elements; returns once the caller's memory has been read
// (the last DMAs out of the ring may still be in flight on `s`).
// dst_lo (nullable, float64 sources only): the fp32 remainders src - (double)(float)src go there, chunk by chunk -- a chunk
// whose remainders are all zero (samples that came from PCM or fp32 data) is cleared on the device instead of travelling;
// *any_lo says whether any chunk travelled.
hipError_t staged_upload(StagingRing& ring, const void* src, int dtype, float* dst, size_t count, hipStream_t s,
                         float* dst_lo, bool* any_lo, hipStream_t s_lo, bool* not_finite) {
    hipError_t e = ring.ensure();
    if (e != hipSuccess) return e;
    const bool split = dst_lo && dtype == 1;
    const bool deferred = split && s_lo != nullptr;
    if (any_lo) *any_lo = false;
    if (split && !deferred) { e = ring.ensure_lo(); if (e != hipSuccess) return e; }
    if (deferred) { e = ring.ensure_lo_full(count); if (e != hipSuccess) return e; }
    HostWorkers& pool = HostWorkers::get();
    const size_t n_chunks = (count + StagingRing::kSlotElems - 1) / StagingRing::kSlotElems;
    std::vector<char> chunk_has_lo(deferred ? n_chunks : 0, 0);
    std::atomic<bool> met_not_finite{false};
    // Chunk c is converted by the workers WHILE this thread enqueues chunk c - 1 (a copy and an event: 10 .. 20 us of driver
    // calls per chunk, a third of what converting a chunk takes): convert(c) is started, issue(c - 1) runs, then this thread
    // takes its own part of chunk c.
    struct Chunk { size_t lo = 0, cnt = 0; int slot = 0; float* stage = nullptr; float* stage_lo = nullptr; unsigned with_lo = 0; int parts = 1; };
    auto issue = [&](const Chunk& k, size_t c) -> hipError_t {
        // a chunk's remainders travel whole or not at all: a part that met none wrote nothing, so when another part did, its
        // share of the staged plane is cleared here (mixed chunks are rare: clips are PCM-exact or they are not)
        const bool chunk_lo = k.with_lo != 0;
        if (chunk_lo)
            for (int part = 0; part < k.parts; ++part)
                if (!(k.with_lo >> part & 1u)) {
                    const size_t a = k.cnt * part / k.parts, b = k.cnt * (part + 1) / k.parts;
                    std::memset(k.stage_lo + a, 0, (b - a) * sizeof(float));
                }
        hipError_t r = hipMemcpyAsync(dst + k.lo, k.stage, k.cnt * sizeof(float), hipMemcpyHostToDevice, s);
        if (r != hipSuccess) return r;
        if (split && !deferred) {
            if (chunk_lo) { r = hipMemcpyAsync(dst_lo + k.lo, k.stage_lo, k.cnt * sizeof(float), hipMemcpyHostToDevice, s); if (any_lo) *any_lo = true; }
            else r = hipMemsetAsync(dst_lo + k.lo, 0, k.cnt * sizeof(float), s);
            if (r != hipSuccess) return r;
        }
        if (deferred && chunk_lo) {
            chunk_has_lo[c] = 1;
            if (any_lo) *any_lo = true;
        }
        r = hipEventRecord(ring.event[k.slot], s);
        if (r != hipSuccess) return r;
        ring.busy[k.slot] = true;
        return hipSuccess;
    };
    Chunk prev;
    bool have_prev = false;
    for (size_t c = 0; c < n_chunks; ++c) {
        Chunk k;
        k.slot = (int)(c % StagingRing::kSlots);
        if (ring.busy[k.slot]) { e = hipEventSynchronize(ring.event[k.slot]); if (e != hipSuccess) return e; }
        k.lo = c * StagingRing::kSlotElems;
        k.cnt = std::min(StagingRing::kSlotElems, count - k.lo);
        k.stage = ring.base + (size_t)k.slot * StagingRing::kSlotElems;
        k.stage_lo = !split ? nullptr : (deferred ? ring.lo_full + k.lo : ring.base_lo + (size_t)k.slot * StagingRing::kSlotElems);
        std::atomic<unsigned> parts_with_lo{0};                 // (at most 32 parts)
        int parts_run = 1;
        const std::function<void(int, int)> convert = [&](int part, int parts) {
            const size_t a = k.cnt * part / parts, b = k.cnt * (part + 1) / parts;
            if (part == 0) parts_run = parts;
            bool bad = false;
            if (split) {
                if (split_part(static_cast<const double*>(src) + k.lo, k.stage, k.stage_lo, a, b, &bad)) parts_with_lo.fetch_or(1u << part, std::memory_order_relaxed);
            }
            else if (dtype == 1) bad = narrow_f64(static_cast<const double*>(src) + k.lo, k.stage, a, b);
            else if (dtype == 2) narrow_i16(static_cast<const int16_t*>(src) + k.lo, k.stage, a, b);
            else bad = narrow_f32(static_cast<const float*>(src) + k.lo, k.stage, a, b);
            if (bad) met_not_finite.store(true, std::memory_order_relaxed);
        };
        const bool started = pool.start(convert, k.cnt);
        if (have_prev) { e = issue(prev, c - 1); if (e != hipSuccess) { if (started) pool.finish(convert); return e; } }
        if (started) pool.finish(convert); else convert(0, 1);
        k.with_lo = parts_with_lo.load();
        k.parts = parts_run;
        prev = k;
        have_prev = true;
    }
    if (have_prev) { e = issue(prev, n_chunks - 1); if (e != hipSuccess) return e; }
    if (deferred) {
        // the remainder plane: behind the last chunk of samples, on the other stream
        e = hipEventRecord(ring.hi_done, s);
        if (e != hipSuccess) return e;
        e = hipStreamWaitEvent(s_lo, ring.hi_done, 0);
        if (e != hipSuccess) return e;
        for (size_t c = 0; c < n_chunks; ++c) {
            const size_t lo = c * StagingRing::kSlotElems, cnt = std::min(StagingRing::kSlotElems, count - lo);
            if (chunk_has_lo[c]) e = hipMemcpyAsync(dst_lo + lo, ring.lo_full + lo, cnt * sizeof(float), hipMemcpyHostToDevice, s_lo);
            else e = hipMemsetAsync(dst_lo + lo, 0, cnt * sizeof(float), s_lo);
            if (e != hipSuccess) return e;
        }
        e = hipEventRecord(ring.lo_done, s_lo);
        if (e != hipSuccess) return e;
        ring.lo_in_flight = true;
    }
    if (not_finite) *not_finite = met_not_finite.load();
    return hipSuccess;
}

// raw bytes (a file's PCM payload) -> device, through the same ring
hipError_t staged_upload_bytes(StagingRing& ring, const void* src, void* dst, size_t n_bytes, hipStream_t s) {
    hipError_t e = ring.ensure();
    if (e != hipSuccess) return e;
    HostWorkers& pool = HostWorkers::get();
    const size_t slot_bytes = StagingRing::kSlotElems * sizeof(float);
    const size_t n_chunks = (n_bytes + slot_bytes - 1) / slot_bytes;
    for (size_t c = 0; c < n_chunks; ++c) {
        const int slot = (int)(c % StagingRing::kSlots);
        if (ring.busy[slot]) { e = hipEventSynchronize(ring.event[slot]); if (e != hipSuccess) return e; }
        const size_t lo = c * slot_bytes, cnt = std::min(slot_bytes, n_bytes - lo);
        unsigned char* stage = reinterpret_cast<unsigned char*>(ring.base + (size_t)slot * StagingRing::kSlotElems);
        pool.run([&](int part, int parts) {
            const size_t a = cnt * part / parts, b = cnt * (part + 1) / parts;
            std::memcpy(stage + a, static_cast<const unsigned char*>(src) + lo + a, b - a);
        }, cnt / 4);
        e = hipMemcpyAsync(static_cast<unsigned char*>(dst) + lo, stage, cnt, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        e = hipEventRecord(ring.event[slot], s);
        if (e != hipSuccess) return e;
        ring.busy[slot] = true;
    }
    return hipSuccess;
}

// src (device fp32, produced by work already enqueued on `s`) -> dst (host float64); returns when dst is complete.
hipError_t staged_download(StagingRing& ring, const float* src, double* dst, size_t count, hipStream_t s) {
    hipError_t e = ring.ensure();
    if (e != hipSuccess) return e;
    HostWorkers& pool = HostWorkers::get();
    const size_t n_chunks = (count + StagingRing::kSlotElems - 1) / StagingRing::kSlotElems;
    auto issue = [&](size_t c) -> hipError_t {
        const int slot = (int)(c % StagingRing::kSlots);
        if (ring.busy[slot]) { hipError_t w = hipEventSynchronize(ring.event[slot]); if (w != hipSuccess) return w; }
        const size_t lo = c * StagingRing::kSlotElems, cnt = std::min(StagingRing::kSlotElems, count - lo);
        hipError_t r = hipMemcpyAsync(ring.base + (size_t)slot * StagingRing::kSlotElems, src + lo, cnt * sizeof(float),
                                      hipMemcpyDeviceToHost, s);
        if (r != hipSuccess) return r;
        ring.busy[slot] = true;
        return hipEventRecord(ring.event[slot], s);
    };
    for (size_t c = 0; c < n_chunks && c < (size_t)StagingRing::kSlots; ++c) { e = issue(c); if (e != hipSuccess) return e; }
    for (size_t c = 0; c < n_chunks; ++c) {
        const int slot = (int)(c % StagingRing::kSlots);
        e = hipEventSynchronize(ring.event[slot]);
        if (e != hipSuccess) return e;
        ring.busy[slot] = false;
        const size_t lo = c * StagingRing::kSlotElems, cnt = std::min(StagingRing::kSlotElems, count - lo);
        const float* stage = ring.base + (size_t)slot * StagingRing::kSlotElems;
        pool.run([&](int part, int parts) {
            const size_t a = cnt * part / parts, b = cnt * (part + 1) / parts;
            widen_f32(stage, dst + lo, a, b);
        }, cnt);
        if (c + StagingRing::kSlots < n_chunks) { e = issue(c + StagingRing::kSlots); if (e != hipSuccess) return e; }
    }
    return hipSuccess;
}

// ---- pinned result buffers ---------------------------------------------------------------------------------------
namespace {
struct HostPool {
    std::mutex m;
    std::map<void*, size_t> outstanding;                  // handed out: ptr -> capacity
    std::vector<std::pair<void*, size_t>> idle;           // returned, ready for reuse (oldest first)
    size_t idle_bytes = 0, live_bytes = 0;
};
HostPool& host_pool() { static HostPool* p = new HostPool(); return *p; }
constexpr size_t kIdleCap = (size_t)2 << 30;              // at most 2 GiB of returned buffers are kept pinned
constexpr size_t kLiveCap = (size_t)16 << 30;             // beyond 16 GiB outstanding the caller gets ordinary memory
}  // namespace

void* host_alloc(size_t bytes) {
    if (bytes == 0) bytes = 1;
    HostPool& p = host_pool();
    {
        std::lock_guard<std::mutex> lk(p.m);
        size_t best = p.idle.size();
        for (size_t k = 0; k < p.idle.size(); ++k)
            if (p.idle[k].second >= bytes && p.idle[k].second <= 2 * bytes + (1 << 20) &&
                (best == p.idle.size() || p.idle[k].second < p.idle[best].second))
                best = k;
        if (best != p.idle.size()) {
            const auto blk = p.idle[best];
            p.idle.erase(p.idle.begin() + best);
            p.idle_bytes -= blk.second;
            p.outstanding[blk.first] = blk.second;
            p.live_bytes += blk.second;
            return blk.first;
        }
        if (p.live_bytes + bytes > kLiveCap) return nullptr;
    }
    const size_t cap = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
    void* ptr = nullptr;
    if (hipHostMalloc(&ptr, cap, hipHostMallocDefault) != hipSuccess || !ptr) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(p.m);
    p.outstanding[ptr] = cap;
    p.live_bytes += cap;
    return ptr;
}

void host_free(void* ptr) {
    if (!ptr) return;
    HostPool& p = host_pool();
    std::vector<void*> drop;
    {
        std::lock_guard<std::mutex> lk(p.m);
        auto it = p.outstanding.find(ptr);
        if (it == p.outstanding.end()) return;             // not ours
        const size_t cap = it->second;
        p.outstanding.erase(it);
        p.live_bytes -= cap;
        p.idle.emplace_back(ptr, cap);
        p.idle_bytes += cap;
        while (p.idle_bytes > kIdleCap && !p.idle.empty()) {
            drop.push_back(p.idle.front().first);
            p.idle_bytes -= p.idle.front().second;
            p.idle.erase(p.idle.begin());
        }
    }
    for (void* d : drop) (void)hipHostFree(d);
}

}  // namespace repet

// (ABI 4) Self-test of the host conversions, no GPU needed: the routines a staged upload / download runs (float64 -> fp32
// samples + fp32 remainders, fp32 -> float64; non-temporal AVX-512 / AVX2 lines where the CPU has them) against scalar
// loops, on n values with NaN, infinities, denormals, PCM-exact runs and every misalignment of the part's first element.
// Returns the number of values that differ (0 = pass), -1 for n < 1.
extern "C" int64_t repet_host_conversion_selftest(int64_t n, uint32_t seed) {
    using namespace repet;
    if (n < 1) return -1;
    const size_t count = (size_t)n;
    std::vector<double> src(count + 64);
    uint64_t st = 0x9E3779B97F4A7C15ull ^ seed;
    auto rnd = [&st]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (size_t i = 0; i < src.size(); ++i) {
        const uint64_t r = rnd();
        double x = (double)(int64_t)(r >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0;          // 53 random bits in [-1, 1)
        const unsigned kind = (unsigned)(r & 1023u);
        if (kind == 0) x = std::numeric_limits<double>::quiet_NaN();
        else if (kind == 1) x = std::numeric_limits<double>::infinity();
        else if (kind == 2) x = -std::numeric_limits<double>::infinity();
        else if (kind == 3) x = 4.9e-324;
        else if (kind == 4) x = 1.0e-42;                                                       // an fp32 denormal
        else if (kind < 40) x = std::round(x * 32768.0) / 32768.0;                              // PCM-exact: no remainder
        src[i] = x;
    }
    // a PCM-exact stretch at the start of some parts (the "remainders only tested" blocks of split_part)
    for (size_t i = 0; i < std::min<size_t>(count, 5000); ++i) src[i] = std::round(src[i] == src[i] && std::fabs(src[i]) <= 1.0 ? src[i] * 32768.0 : 0.0) / 32768.0;
    int64_t bad = 0;
    auto same = [](float a, float b) { return std::memcmp(&a, &b, 4) == 0 || (a != a && b != b); };
    auto same_d = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0 || (a != a && b != b); };
    const size_t pad = 16;                                      // floats in front of the outputs: every misalignment of a line
    std::vector<float> hi(count + 2 * pad + 64), lo(count + 2 * pad + 64), back(count + 64);
    std::vector<double> wide(count + 2 * pad + 64);
    for (size_t shift = 0; shift < 16; shift += 3) {
        const size_t a = shift, b = count;                      // the part [a, b)
        if (a >= b) break;
        // float64 -> hi + lo
        std::fill(hi.begin(), hi.end(), -7.f);
        std::fill(lo.begin(), lo.end(), -7.f);
        bool nf = false;
        const bool any = split_part(src.data(), hi.data() + pad, lo.data() + pad, a, b, &nf);
        bool want_any = false, want_nf = false;
        for (size_t k = a; k < b; ++k) {
            const double x = src[k];
            const float h = (float)x;
            want_any |= (x != (double)h);
            want_nf |= !(std::fabs(x) <= std::numeric_limits<double>::max());
        }
        if (any != want_any || nf != want_nf) ++bad;
        size_t first_lo = b;                                    // remainders are written from the first block that holds one
        for (size_t k = a; k < b; ++k) if (src[k] != (double)(float)src[k]) { first_lo = k; break; }
        for (size_t k = a; k < b; ++k) {
            const double x = src[k];
            const float h = (float)x;
            if (!same(hi[pad + k], h)) ++bad;
            if (any) {
                const float l = (float)(x - (double)h);
                const float got = lo[pad + k];
                if (k >= first_lo ? !same(got, l) : !(same(got, 0.f) || same(got, l))) ++bad;
            }
        }
        // fp32 -> float64
        for (size_t k = 0; k < count; ++k) back[k] = hi[pad + std::max(k, a)];
        std::fill(wide.begin(), wide.end(), -7.0);
        widen_f32(back.data(), wide.data() + (shift & 7), a, b);
        for (size_t k = a; k < b; ++k) if (!same_d(wide[(shift & 7) + k], (double)back[k])) ++bad;
        if (a > 0 && wide[(shift & 7) + a - 1] != -7.0) ++bad;   // nothing written outside the part
        if (wide[(shift & 7) + b] != -7.0) ++bad;
    }
    return bad;
}
