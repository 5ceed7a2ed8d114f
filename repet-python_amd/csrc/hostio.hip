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

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace repet {

// ---- worker pool -----------------------------------------------------------------------------------------------
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
    void run(const std::function<void(int, int)>& fn, size_t work_items = ~(size_t)0) {
        if (n_threads_ <= 1 || work_items < 65536) { fn(0, 1); return; }     // not worth waking anybody up
        std::lock_guard<std::mutex> one_job(job_mutex_);
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = &fn; pending_ = n_threads_ - 1; ++generation_;
        }
        cv_.notify_all();
        fn(0, n_threads_);
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
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
        for (int k = 1; k < n_threads_; ++k) std::thread([this, k] { loop(k); }).detach();
    }
    void loop(int part) {
        unsigned seen = 0;
        for (;;) {
            const std::function<void(int, int)>* fn;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return generation_ != seen; });
                seen = generation_;
                fn = fn_;
            }
            (*fn)(part, n_threads_);
            {
                std::lock_guard<std::mutex> lk(m_);
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    int n_threads_ = 1;
    std::mutex job_mutex_, m_;
    std::condition_variable cv_, done_;
    const std::function<void(int, int)>* fn_ = nullptr;
    int pending_ = 0;
    unsigned generation_ = 0;
};

template <typename T>
void narrow_part(const T* src, float* dst, size_t lo, size_t hi) {
    for (size_t i = lo; i < hi; ++i) dst[i] = (float)src[i];
}

// float64 -> the fp32 sample and the fp32 remainder (hi + lo carries 48 bits of the double); true when a remainder is not zero
bool split_part(const double* src, float* dst_hi, float* dst_lo, size_t lo, size_t hi) {
    bool any = false;
    for (size_t i = lo; i < hi; ++i) {
        const double x = src[i];
        const float h = (float)x;
        const float l = (float)(x - (double)h);
        dst_hi[i] = h;
        dst_lo[i] = l;
        any = any || (l != 0.0f);       // (a NaN or infinite sample gives a NaN remainder: kept, it is what hi + lo must say)
    }
    return any;
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
                         float* dst_lo, bool* any_lo) {
    hipError_t e = ring.ensure();
    if (e != hipSuccess) return e;
    const bool split = dst_lo && dtype == 1;
    if (any_lo) *any_lo = false;
    if (split) { e = ring.ensure_lo(); if (e != hipSuccess) return e; }
    HostWorkers& pool = HostWorkers::get();
    const size_t n_chunks = (count + StagingRing::kSlotElems - 1) / StagingRing::kSlotElems;
    for (size_t c = 0; c < n_chunks; ++c) {
        const int slot = (int)(c % StagingRing::kSlots);
        if (ring.busy[slot]) { e = hipEventSynchronize(ring.event[slot]); if (e != hipSuccess) return e; }
        const size_t lo = c * StagingRing::kSlotElems, cnt = std::min(StagingRing::kSlotElems, count - lo);
        float* stage = ring.base + (size_t)slot * StagingRing::kSlotElems;
        float* stage_lo = split ? ring.base_lo + (size_t)slot * StagingRing::kSlotElems : nullptr;
        std::atomic<bool> chunk_lo{false};
        pool.run([&](int part, int parts) {
            const size_t a = cnt * part / parts, b = cnt * (part + 1) / parts;
            if (split) { if (split_part(static_cast<const double*>(src) + lo, stage, stage_lo, a, b)) chunk_lo.store(true, std::memory_order_relaxed); }
            else if (dtype == 1) narrow_part(static_cast<const double*>(src) + lo, stage, a, b);
            else if (dtype == 2) narrow_part(static_cast<const int16_t*>(src) + lo, stage, a, b);
            else std::memcpy(stage + a, static_cast<const float*>(src) + lo + a, (b - a) * sizeof(float));
        }, cnt);
        e = hipMemcpyAsync(dst + lo, stage, cnt * sizeof(float), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) return e;
        if (split) {
            if (chunk_lo.load()) { e = hipMemcpyAsync(dst_lo + lo, stage_lo, cnt * sizeof(float), hipMemcpyHostToDevice, s); if (any_lo) *any_lo = true; }
            else e = hipMemsetAsync(dst_lo + lo, 0, cnt * sizeof(float), s);
            if (e != hipSuccess) return e;
        }
        e = hipEventRecord(ring.event[slot], s);
        if (e != hipSuccess) return e;
        ring.busy[slot] = true;
    }
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
            double* out = dst + lo;
            for (size_t i = a; i < b; ++i) out[i] = (double)stage[i];
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
