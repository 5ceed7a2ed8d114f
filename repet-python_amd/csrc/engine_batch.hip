// repet_run_batch: clips dealt over the devices of this process, host transport and RCCL transport (see engine.h for the map of the engine's files)
#include "engine.h"

using namespace repet;
using namespace repet_eng;

#include <dlfcn.h>

#include <atomic>

namespace repet_eng {

// Logical devices (test switch): REPET_LOGICAL_DEVICES=n lets repet_run_batch deal its clips over n "devices" although
// fewer GPUs are visible -- logical device d runs on physical device d % visible, in its own thread, context and stream.
// Dealing, per-device threads and result placement of the multi-GPU path can then be exercised on a one-GPU box.
// what the last repet_run_batch / repet_run_batch_rccl call of this thread did (repet_last_batch_info)
thread_local BatchInfo g_batch_info;

int logical_device_count(int physical) {
    const char* e = getenv("REPET_LOGICAL_DEVICES");
    const int n = e ? atoi(e) : 0;
    return n > physical ? n : physical;
}

// ---- RCCL over xGMI, inside the library --------------------------------------------------------------
// librccl is opened on first use (dlopen by soname: a process that has PyTorch's RCCL loaded gets that one) -- the library
// carries no link-time dependency on it. One communicator per physical device from ncclCommInitAll, one process.
struct Rccl {
    using comm_t = void*;
    int (*CommInitAll)(comm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(comm_t) = nullptr;
    int (*CommAbort)(comm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
    static constexpr int kFloat = 7;          // ncclFloat32
    static Rccl& get() {
        static Rccl r = [] {
            Rccl x;
            void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
            if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
            if (!h) return x;
            auto sym = [&](const char* name) { return dlsym(h, name); };
            x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(sym("ncclCommInitAll"));
            x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(sym("ncclCommDestroy"));
            x.CommAbort = reinterpret_cast<decltype(x.CommAbort)>(sym("ncclCommAbort"));
            x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(sym("ncclGroupStart"));
            x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(sym("ncclGroupEnd"));
            x.Send = reinterpret_cast<decltype(x.Send)>(sym("ncclSend"));
            x.Recv = reinterpret_cast<decltype(x.Recv)>(sym("ncclRecv"));
            x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(sym("ncclGetErrorString"));
            x.ok = x.CommInitAll && x.CommDestroy && x.GroupStart && x.GroupEnd && x.Send && x.Recv;
            return x;
        }();
        return r;
    }
};

#define NCCL_TRY(expr)                                                                                           \
    do {                                                                                                         \
        const int r_ = (expr);                                                                                   \
        if (r_ != 0) return fail(REPET_ERR_HIP, std::string(#expr) + ": " + (rc.GetErrorString ? rc.GetErrorString(r_) : "RCCL error")); \
    } while (0)

// Contexts of repet_run_stream, kept between calls: per device a list of idle ones. A call takes `depth` of them (creating what
// is missing) and gives them back; concurrent calls get disjoint sets.
struct StreamPool {
    std::mutex mu;
    std::map<int, std::vector<repet_ctx*>> idle;
    static StreamPool& get() { static StreamPool* p = new StreamPool(); return *p; }
    int take(int device, int n, std::vector<repet_ctx*>* out) {
        std::lock_guard<std::mutex> lk(mu);
        std::vector<repet_ctx*>& v = idle[device];
        while ((int)out->size() < n && !v.empty()) { out->push_back(v.back()); v.pop_back(); }
        while ((int)out->size() < n) {
            repet_ctx* c = nullptr;
            RP_TRY(repet_ctx_create(device, &c));
            out->push_back(c);
        }
        return REPET_OK;
    }
    void give_back(int device, const std::vector<repet_ctx*>& cs) {
        std::lock_guard<std::mutex> lk(mu);
        for (repet_ctx* c : cs) idle[device].push_back(c);
    }
    void release() {
        std::lock_guard<std::mutex> lk(mu);
        for (auto& kv : idle) for (repet_ctx* c : kv.second) repet_ctx_destroy(c);
        idle.clear();
    }
};
void release_stream_contexts() { StreamPool::get().release(); }

// transport 0: every device's worker thread uploads its own clips from the caller's host arrays and downloads its own
// results (with the data in host RAM this uses every device's own PCIe link: SURVEY 8e's "honest comparison").
// transport 1: the clips enter through device 0, travel to their devices as ONE group of ncclSend / ncclRecv over xGMI
// (fp32, interleaved), are separated there from the received device buffers, and the results come back the same way.
// transport 2 (repet_run_stream): ONE device (`one_device`), n_devices = clips in flight -- the workers of transport 0, all on
// that device, each with its own context and stream.
int run_batch_impl(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                   const int32_t* n_channels, const repet_params* p, double* const* out, int32_t n_devices, int transport,
                   int one_device) {
    if (n_clips < 0 || (n_clips > 0 && (!audio || !n_samples || !n_channels || !out)))
        return fail(REPET_ERR_BAD_ARG, "null argument");
    const int physical = repet_device_count();
    if (physical < 1) return fail(REPET_ERR_HIP, "no HIP device");
    const int avail = transport == 1 ? physical : transport == 2 ? 8 : logical_device_count(physical);     // (RCCL needs distinct physical devices)
    if (n_devices < 1 || n_devices > avail) return fail(REPET_ERR_BAD_ARG, transport == 2 ? "depth must be 1 .. 8" : "n_devices out of range");
    if (transport == 2 && (one_device < 0 || one_device >= physical)) return fail(REPET_ERR_BAD_ARG, "no such device");
    // longest first, dealt round-robin: clip order[i] -> device i % n_devices
    std::vector<int> order(n_clips);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n_samples[a] > n_samples[b]; });
    std::vector<int> device_of(n_clips);
    for (int i = 0; i < n_clips; ++i) device_of[order[i]] = i % n_devices;
    std::vector<int> rcs(n_devices, REPET_OK);
    std::vector<std::string> msgs(n_devices);
    auto run_threads = [&](const std::function<void(int)>& worker) {
        if (n_devices == 1) { worker(0); return; }
        std::vector<std::thread> th;
        for (int d = 0; d < n_devices; ++d) th.emplace_back(worker, d);
        for (auto& t : th) t.join();
    };
    auto first_error = [&]() -> int {
        for (int d = 0; d < n_devices; ++d)
            if (rcs[d] != REPET_OK) return fail(rcs[d], msgs[d]);
        return REPET_OK;
    };

    if (transport == 2) {
        // `depth` clips in flight on one device: worker j takes clips j, j + depth, ... through a context of its own (stream,
        // pinned ring, workspaces), so that while one clip is being separated the next is narrowed and uploaded and the one
        // before is copied back and widened -- PCIe both ways and the kernels side by side, the host conversions taking turns
        // chunk by chunk (hostio.hip). The contexts are kept for the next call (a context's workspaces are half a gigabyte
        // for a 3-minute clip; repet_release_thread_ctx frees them).
        g_batch_info = BatchInfo{};
        std::vector<repet_ctx*> taken;
        const int rc_take = StreamPool::get().take(one_device, n_devices, &taken);
        if (rc_take != REPET_OK) { StreamPool::get().give_back(one_device, taken); return rc_take; }
        std::atomic<int> next{0};
        run_threads([&](int j) {
            repet_ctx* c = taken[j];
            c->strict = !(p && (p->flags & REPET_FLAG_REFUSE_NONFINITE));
            int rc = REPET_OK;
            // (clips in the caller's order, whichever worker is free takes the next: the results of equal clips come back in order)
            for (int i = next.fetch_add(1); rc == REPET_OK && i < n_clips; i = next.fetch_add(1)) {
                rc = repet_ctx_upload(c, audio[i], dtype, n_samples[i], n_channels[i]);
                if (rc == REPET_OK) rc = repet_ctx_execute_async(c, algo, p);
                if (rc == REPET_OK) rc = repet_ctx_download(c, out[i]);
            }
            if (rc != REPET_OK) msgs[j] = g_last_error;
            rcs[j] = rc;
        });
        StreamPool::get().give_back(one_device, taken);
        return first_error();
    }
    if (transport != 1) {
        g_batch_info = BatchInfo{};
        run_threads([&](int dev) {
            repet_ctx* c = nullptr;
            int rc = repet_ctx_create(dev % physical, &c);
            if (rc == REPET_OK) c->strict = !(p && (p->flags & REPET_FLAG_REFUSE_NONFINITE));
            for (int i = dev; rc == REPET_OK && i < n_clips; i += n_devices) {
                const int k = order[i];
                rc = repet_ctx_upload(c, audio[k], dtype, n_samples[k], n_channels[k]);
                if (rc == REPET_OK) rc = repet_ctx_execute(c, algo, p, nullptr);
                if (rc == REPET_OK) rc = repet_ctx_download(c, out[k]);
            }
            if (rc != REPET_OK) msgs[dev] = g_last_error;
            rcs[dev] = rc;
            repet_ctx_destroy(c);
        });
        return first_error();
    }

    // ---- transport 1 -------------------------------------------------------------------------------------------------
    // One call at a time (the communicators are shared, cached per device count and kept until the process ends: creating
    // them costs hundreds of milliseconds). The clips are worked through in ROUNDS of one clip per device: while the devices
    // separate round r, the root narrows / uploads round r + 1 and its scatter group is already enqueued; the results of
    // round r return in their own group and their buffers are freed before round r + 2 is staged -- the root never holds more
    // than two rounds. A float64 clip travels as TWO fp32 planes, samples and remainders (x - (double)(float)x, only where
    // one is not zero): the second level of the peak picking then decides on the same 48 bits as the single-GPU call.
    // REPET_RCCL_SELF=1 with n_devices == 1 (test switch): every clip takes the send / receive path, device 0 to itself
    // inside the group, so that the transport's lines run on a one-GPU box.
    Rccl& rc = Rccl::get();
    if (!rc.ok) return fail(REPET_ERR_HIP, "librccl could not be loaded (RCCL transport of repet_run_batch)");
    static std::mutex call_mu;
    static std::map<int, std::vector<Rccl::comm_t>> comm_cache;
    std::lock_guard<std::mutex> call_lock(call_mu);
    const bool self_test = n_devices == 1 && [] { const char* e = getenv("REPET_RCCL_SELF"); return e && e[0] == '1'; }();
    g_batch_info = BatchInfo{};
    g_batch_info.transport = 1;
    auto it = comm_cache.find(n_devices);
    if (it == comm_cache.end()) {
        std::vector<int> devs(n_devices);
        std::iota(devs.begin(), devs.end(), 0);
        std::vector<Rccl::comm_t> fresh(n_devices, nullptr);
        NCCL_TRY(rc.CommInitAll(fresh.data(), n_devices, devs.data()));
        it = comm_cache.emplace(n_devices, std::move(fresh)).first;
    }
    std::vector<Rccl::comm_t>& comms = it->second;
    bool comms_broken = false;

    struct ClipBufs { float *in_root = nullptr, *lo_root = nullptr, *out_root = nullptr, *in_dev = nullptr, *lo_dev = nullptr, *out_dev = nullptr; bool has_lo = false; };
    std::vector<ClipBufs> bufs(n_clips);
    std::vector<repet_ctx*> ctx(n_devices, nullptr);
    repet_ctx* io = nullptr;                          // the root's own context for staging and transport (ctx[0] computes)
    auto travels = [&](int k) { return device_of[k] != 0 || self_test; };
    auto free_clip = [&](int k) {
        ClipBufs& q = bufs[k];
        { DeviceGuard g(0); for (float** ptr : {&q.in_root, &q.lo_root, &q.out_root}) if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; } }
        { DeviceGuard g(device_of[k]); for (float** ptr : {&q.in_dev, &q.lo_dev, &q.out_dev}) if (*ptr) { (void)hipFree(*ptr); *ptr = nullptr; } }
    };
    struct Xfer { const float* src; int src_dev; float* dst; int dst_dev; size_t count; };
    // One group of sends and receives. The group is CLOSED whatever happens inside it (an error between ncclGroupStart and
    // ncclGroupEnd used to leave it open under the communicators' destruction); a failed group marks the communicators broken.
    auto exchange = [&](const std::vector<Xfer>& xs) -> int {
        if (xs.empty()) return REPET_OK;
        int err = rc.GroupStart();
        if (err != 0) { comms_broken = true; return fail(REPET_ERR_HIP, std::string("ncclGroupStart: ") + (rc.GetErrorString ? rc.GetErrorString(err) : "RCCL error")); }
        const char* what = nullptr;
        for (const Xfer& x : xs) {
            hipStream_t send_stream = x.src_dev == 0 ? io->stream : ctx[x.src_dev]->stream;
            hipStream_t recv_stream = x.dst_dev == 0 ? io->stream : ctx[x.dst_dev]->stream;
            err = rc.Send(x.src, x.count, Rccl::kFloat, x.dst_dev, comms[x.src_dev], send_stream);
            if (err != 0) { what = "ncclSend"; break; }
            err = rc.Recv(x.dst, x.count, Rccl::kFloat, x.src_dev, comms[x.dst_dev], recv_stream);
            if (err != 0) { what = "ncclRecv"; break; }
        }
        const int end = rc.GroupEnd();
        if (err == 0 && end != 0) { err = end; what = "ncclGroupEnd"; }
        if (err != 0) { comms_broken = true; return fail(REPET_ERR_HIP, std::string(what) + ": " + (rc.GetErrorString ? rc.GetErrorString(err) : "RCCL error")); }
        ++g_batch_info.groups;
        return REPET_OK;
    };
    const int n_rounds = (n_clips + n_devices - 1) / n_devices;
    auto round_clips = [&](int r) { std::vector<int> ks; for (int i = r * n_devices; i < std::min(n_clips, (r + 1) * n_devices); ++i) ks.push_back(order[i]); return ks; };
    // A(r): the round's clips enter through the root (fp32 samples + remainders), the travelling ones leave in one group
    auto stage_round = [&](int r) -> int {
        std::vector<Xfer> xs;
        for (int k : round_clips(r)) {
            ClipBufs& q = bufs[k];
            const size_t count = (size_t)n_samples[k] * n_channels[k];
            const size_t bytes = std::max<size_t>(count * sizeof(float), 256);
            const bool want_lo = dtype == REPET_F64 && count > 0;
            {
                DeviceGuard g(0);
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.in_root), bytes));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.out_root), bytes));
                if (want_lo) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.lo_root), bytes));
                bool not_finite = false;
                HIP_TRY(staged_upload(io->ring, audio[k], dtype, q.in_root, count, io->stream, q.lo_root, &q.has_lo, nullptr, &not_finite));
                if (not_finite && p && (p->flags & REPET_FLAG_REFUSE_NONFINITE))
                    return fail(REPET_ERR_BAD_ARG, "audio_signal contains NaN or infinite samples");
            }
            if (q.has_lo) ++g_batch_info.clips_with_remainders;
            if (!travels(k) || count == 0) continue;
            const int g = device_of[k];
            {
                DeviceGuard gd(g);
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.in_dev), bytes));
                HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.out_dev), bytes));
                if (q.has_lo) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&q.lo_dev), bytes));
            }
            xs.push_back({q.in_root, 0, q.in_dev, g, count});
            if (q.has_lo) xs.push_back({q.lo_root, 0, q.lo_dev, g, count});
            ++g_batch_info.clips_sent;
        }
        RP_TRY(exchange(xs));
        DeviceGuard g(0);
        HIP_TRY(hipStreamSynchronize(io->stream));          // the root's copies are complete (and the sends have been matched)
        return REPET_OK;
    };
    // B(r): every device separates its clip of the round from device memory; the result stays on the device
    auto compute_clip = [&](int k) -> int {
        ClipBufs& q = bufs[k];
        const int dev = device_of[k];
        const bool moved = travels(k) && (size_t)n_samples[k] * n_channels[k] > 0;
        RP_TRY(repet_ctx_upload_device_split(ctx[dev], moved ? q.in_dev : q.in_root, q.has_lo ? (moved ? q.lo_dev : q.lo_root) : nullptr,
                                             n_samples[k], n_channels[k], 1));
        RP_TRY(repet_ctx_execute(ctx[dev], algo, p, nullptr));
        return repet_ctx_download_device(ctx[dev], moved ? q.out_dev : q.out_root);
    };
    // C(r): the travelling results return in one group; the root widens them into the caller's arrays; the round is freed
    auto finish_round = [&](int r) -> int {
        std::vector<Xfer> xs;
        for (int k : round_clips(r)) {
            const size_t count = (size_t)n_samples[k] * n_channels[k];
            if (travels(k) && count > 0) xs.push_back({bufs[k].out_dev, device_of[k], bufs[k].out_root, 0, count});
        }
        RP_TRY(exchange(xs));
        {
            DeviceGuard g(0);
            for (int k : round_clips(r))
                HIP_TRY(staged_download(io->ring, bufs[k].out_root, out[k], (size_t)n_samples[k] * n_channels[k], io->stream));
            HIP_TRY(hipStreamSynchronize(io->stream));
        }
        for (int d = 1; d < n_devices; ++d) { DeviceGuard g(d); HIP_TRY(hipStreamSynchronize(ctx[d]->stream)); }     // (their sends)
        for (int k : round_clips(r)) free_clip(k);
        return REPET_OK;
    };
    auto body = [&]() -> int {
        RP_TRY(repet_ctx_create(0, &io));
        for (int d = 0; d < n_devices; ++d) {
            RP_TRY(repet_ctx_create(d, &ctx[d]));
            // (the planes arrive through repet_ctx_upload_device_split, unscanned: with `strict` the passes that reproduce
            // repet.py on NaN / infinite samples run on every clip, so both transports give the same result)
            ctx[d]->strict = !(p && (p->flags & REPET_FLAG_REFUSE_NONFINITE));
        }
        if (n_rounds > 0) RP_TRY(stage_round(0));
        for (int r = 0; r < n_rounds; ++r) {
            // the devices work on round r in their own threads while this one stages round r + 1
            const std::vector<int> ks = round_clips(r);
            std::vector<int> round_rc(ks.size(), REPET_OK);
            std::vector<std::string> round_msg(ks.size());
            std::vector<std::thread> th;
            for (size_t i = 0; i < ks.size(); ++i)
                th.emplace_back([&, i] { round_rc[i] = compute_clip(ks[i]); if (round_rc[i] != REPET_OK) round_msg[i] = g_last_error; });
            const int staged = r + 1 < n_rounds ? stage_round(r + 1) : REPET_OK;
            const std::string staged_msg = g_last_error;
            for (auto& t : th) t.join();
            for (size_t i = 0; i < ks.size(); ++i) if (round_rc[i] != REPET_OK) return fail(round_rc[i], round_msg[i]);
            if (staged != REPET_OK) return fail(staged, staged_msg);
            RP_TRY(finish_round(r));
        }
        return REPET_OK;
    };
    const int status = body();
    const std::string keep = g_last_error;
    for (int d = 0; d < n_devices; ++d) if (ctx[d]) { DeviceGuard g(d); (void)hipStreamSynchronize(ctx[d]->stream); }
    if (io) { DeviceGuard g(0); (void)hipStreamSynchronize(io->stream); }
    for (int k = 0; k < n_clips; ++k) free_clip(k);
    for (int d = 0; d < n_devices; ++d) if (ctx[d]) repet_ctx_destroy(ctx[d]);
    if (io) repet_ctx_destroy(io);
    if (comms_broken) {                               // do not hand a communicator with a failed group to the next call
        for (Rccl::comm_t cm : comms) if (cm) (void)(rc.CommAbort ? rc.CommAbort(cm) : rc.CommDestroy(cm));
        comm_cache.erase(n_devices);
    }
    if (status != REPET_OK) g_last_error = keep;
    return status;
}

}  // namespace repet_eng

extern "C" {

int repet_last_batch_info(int64_t out[4]) {
    if (!out) return fail(REPET_ERR_BAD_ARG, "null argument");
    out[0] = g_batch_info.transport; out[1] = g_batch_info.clips_sent; out[2] = g_batch_info.clips_with_remainders; out[3] = g_batch_info.groups;
    return REPET_OK;
}

int repet_run_batch(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                    const int32_t* n_channels, const repet_params* p, double* const* out, int32_t n_devices) {
    return run_batch_impl(algo, n_clips, audio, dtype, n_samples, n_channels, p, out, n_devices, 0);
}

int repet_run_stream(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                     const int32_t* n_channels, const repet_params* p, double* const* out, int device, int32_t depth) {
    return run_batch_impl(algo, n_clips, audio, dtype, n_samples, n_channels, p, out, depth, 2, device);
}

int repet_run_batch_rccl(int algo, int32_t n_clips, const void* const* audio, int dtype, const int64_t* n_samples,
                         const int32_t* n_channels, const repet_params* p, double* const* out, int32_t n_devices) {
    return run_batch_impl(algo, n_clips, audio, dtype, n_samples, n_channels, p, out, n_devices, 1);
}

}  // extern "C"
