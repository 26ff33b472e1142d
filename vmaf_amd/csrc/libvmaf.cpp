// libvmaf.cpp - host layer of the MI355X VMAF engine: compute_vmaf() (include/libvmaf.h).
//
// Mirrors the reference's wrapper: libvmaf.cpp:28-67 (flag handling) -> vmaf.cpp:631-753 (RunVmaf:
// timer, pooling, stdout lines, XML/JSON logs) -> vmaf.cpp:140-624 (VmafRunner::run: model, per-frame
// feature loop, derived features, normalise, SVR, de-normalise, transform, clip).  The per-frame C
// feature functions and svm_predict are replaced by the kernels behind include/vmaf_hip.h.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include <atomic>

#include <sched.h>

#include "../../include/libvmaf.h"
#include "../../include/vmaf_hip.h"
#include "model.h"

namespace {

// wrapper/src/vmaf.h:144-151
class VmafException : public std::exception {
public:
    explicit VmafException(const std::string &m) : msg(m) {}
    const char *what() const noexcept override { return msg.c_str(); }
private:
    std::string msg;
};

constexpr int kAtoms = VMAF_HIP_ATOMS;

// VMAF_HIP_TRACE=1: wall-clock milestones of a compute_vmaf* call on stderr (where the fixed cost of a short run goes)
void trace(const char *what)
{
    static const bool on = std::getenv("VMAF_HIP_TRACE") && std::atoi(std::getenv("VMAF_HIP_TRACE")) != 0;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[vmaf_hip %8.3f ms] %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}

struct EngineGuard {
    std::vector<vmaf_hip_engine *> engines;
    std::vector<void *> pinned;
    ~EngineGuard()
    {
        for (auto e : engines) vmaf_hip_engine_destroy(e);
        for (auto p : pinned) vmaf_hip_host_free(p);
    }
};

std::vector<int> parse_devices()
{
    std::vector<int> d;
    const char *env = std::getenv("VMAF_HIP_DEVICES");
    if (env && *env) {
        const char *p = env;
        while (*p) {
            char *end = nullptr;
            const long v = std::strtol(p, &end, 10);
            if (end == p) break;
            d.push_back((int)v);
            p = (*end == ',') ? end + 1 : end;
        }
    }
    if (d.empty()) d.push_back(0);
    return d;
}

// The per-frame feature loop of combo() (wrapper/src/combo.c:100-371), batched: frames are pulled
// serially through read_frame into page-locked buffers and dealt, a batch at a time, round-robin to
// the engines (one per GPU); each batch after the first gets its motion halo from the previous
// batch's last ref frame.  Fills atoms [frames][20].
// Where the frames come from: the reference's fp32 callback (bps = 0), or a raw-luma callback that hands over the
// samples as the file holds them (bps = 1 / 2 bytes) and leaves the widening to the GPU.
struct FrameSource {
    int bps = 0;
    int (*read_frame)(float *, float *, float *, int, void *) = nullptr;
    int (*read_raw)(void *, void *, int, void *) = nullptr;
    // seekable source (compute_vmaf_raw_seekable): frame `index` of nframes, callable from several threads at once
    int (*read_raw_at)(void *, void *, int, long, void *) = nullptr;
    long nframes = 0;
    void *user = nullptr;
};

struct MetricFlags { bool psnr = false, ssim = false, ms_ssim = false, ten_bit = false; bool any() const { return psnr || ssim || ms_ssim; } };

int batch_frames(size_t plane);

void extract_atoms(const FrameSource &src, int w, int h, bool scalar_order, const MetricFlags &mf, std::vector<double> &atoms,
                   std::vector<double> &metrics /* [frames][VMAF_HIP_METRICS] when mf.any() */)
{
    if (w <= 0 || h <= 0) throw VmafException("wrong width " + std::to_string(w) + " or height " + std::to_string(h) + ".");
    const size_t stride = src.bps ? (size_t)w * src.bps : ((size_t)w * 4 + 31) / 32 * 32;   // ALIGN_CEIL, combo.c:106
    const size_t plane = stride * h;

    const std::vector<int> devices = parse_devices();
    const int G = (int)devices.size();
    // frames per batch: ~32 MB of luma per plane and batch (16 frames of raw 8-bit 1080p).  Larger batches only add
    // page-locked memory to set up and latency before the first launch; measured with tools/e2e_cli_rate.py
    const int batch = batch_frames(plane);

    EngineGuard guard;
    struct Slot { float *ref[2], *dis[2]; int fill = 0; bool pending = false; size_t pend_start = 0; int pend_n = 0; std::vector<double> out; };
    std::vector<Slot> slots(G);
    for (int g = 0; g < G; ++g) {
        vmaf_hip_engine *e = nullptr;
        if (vmaf_hip_engine_create(&e, devices[g], w, h, batch)) throw VmafException(vmaf_hip_last_error());
        guard.engines.push_back(e);
        // the reference's two orders (disable_avx picks the scalar one); VMAF_HIP_FAST=1 opts into fused multiply-adds
        // instead: not bit-identical to either, inside the 1e-4 / 1e-3 tolerances, about a third faster
        static const bool fast = std::getenv("VMAF_HIP_FAST") && std::atoi(std::getenv("VMAF_HIP_FAST")) != 0;
        if (vmaf_hip_engine_set_conv_order(e, scalar_order ? 0 : fast ? 2 : 1) || vmaf_hip_engine_set_sums_exact(e, scalar_order))
            throw VmafException(vmaf_hip_last_error());
        if (mf.any() && (vmaf_hip_engine_enable_metrics(e, mf.psnr, mf.ssim, mf.ms_ssim, mf.ten_bit) || vmaf_hip_engine_set_metrics_exact(e, scalar_order)))
            throw VmafException(vmaf_hip_last_error());
        for (int b = 0; b < 2; ++b) {
            float **pp[2] = { &slots[g].ref[b], &slots[g].dis[b] };
            for (auto p : pp) {
                *p = (float *)vmaf_hip_host_alloc(plane * batch);
                if (!*p) throw VmafException(vmaf_hip_last_error());
                guard.pinned.push_back(*p);
            }
        }
        slots[g].out.resize((size_t)batch * kAtoms);
    }
    float *temp = (float *)vmaf_hip_host_alloc(plane * 2);        // combo.c:139-144
    if (!temp) throw VmafException(vmaf_hip_last_error());
    guard.pinned.push_back(temp);

    auto collect = [&](int g) {
        Slot &s = slots[g];
        if (!s.pending) return;
        if (vmaf_hip_engine_sync(guard.engines[g])) throw VmafException(vmaf_hip_last_error());
        if (atoms.size() < (s.pend_start + s.pend_n) * kAtoms) atoms.resize((s.pend_start + s.pend_n) * kAtoms);
        std::copy(s.out.begin(), s.out.begin() + (size_t)s.pend_n * kAtoms, atoms.begin() + s.pend_start * kAtoms);
        if (mf.any()) {
            if (metrics.size() < (s.pend_start + s.pend_n) * VMAF_HIP_METRICS) metrics.resize((s.pend_start + s.pend_n) * VMAF_HIP_METRICS);
            if (vmaf_hip_engine_get_metrics(guard.engines[g], metrics.data() + s.pend_start * VMAF_HIP_METRICS, s.pend_n))
                throw VmafException(vmaf_hip_last_error());
        }
        s.pending = false;
    };

    size_t total = 0;
    int g = 0;
    const float *prev_last_ref = nullptr;                          // last ref frame of the previous batch (host)
    bool eof = false;
    while (!eof) {
        Slot &s = slots[g];
        float *rb = s.ref[s.fill], *db = s.dis[s.fill];
        int n = 0;
        while (n < batch) {
            const int ret = src.bps ? src.read_raw((char *)rb + plane * n, (char *)db + plane * n, (int)stride, src.user)
                                    : src.read_frame((float *)((char *)rb + plane * n), (float *)((char *)db + plane * n), temp, (int)stride, src.user);
            if (ret == 1) throw VmafException("read_frame failed.");   // combo.c:151-153
            if (ret == 2) { eof = true; break; }
            ++n;
        }
        if (n == 0) break;
        collect(g);                                                // engine g is free again, its other buffer reusable
        const bool cont = total > 0;
        if (cont && G > 1) {
            const int rc = src.bps ? vmaf_hip_motion_seed_host_raw(guard.engines[g], prev_last_ref, src.bps, stride)
                                   : vmaf_hip_motion_seed_host(guard.engines[g], prev_last_ref, stride);
            if (rc) throw VmafException(vmaf_hip_last_error());
        }
        const int rc = src.bps ? vmaf_hip_extract_host_raw_async(guard.engines[g], rb, db, src.bps, stride, plane, n, cont ? 1 : 0, s.out.data())
                               : vmaf_hip_extract_host_async(guard.engines[g], rb, db, stride, plane, n, cont ? 1 : 0, s.out.data());
        if (rc) throw VmafException(vmaf_hip_last_error());
        s.pending = true; s.pend_start = total; s.pend_n = n;
        s.fill ^= 1;
        prev_last_ref = (const float *)((const char *)rb + plane * (n - 1));
        total += n;
        g = (g + 1) % G;
    }
    for (int i = 0; i < G; ++i) collect(i);
    atoms.resize(total * kAtoms);
    if (mf.any()) metrics.resize(total * VMAF_HIP_METRICS);
}

// Helper threads of one shard: the frames of a batch are fetched side by side (one positional read_frame call per frame,
// frame i on thread i % N), because a single thread copies only ~8 GB/s out of the page cache - less than half of what
// the PCIe link then moves.  One hand-off per batch, not per frame.
class FrameFetchers {
public:
    explicit FrameFetchers(int helpers)
    {
        for (int i = 0; i < helpers; ++i) th_.emplace_back([this, i] { work(i + 1); });
    }
    ~FrameFetchers()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    // calls fn(i) for i in [0, n) on the helpers and the calling thread; 0 = all fine, else see worse()
    int run(int n, const std::function<int(int)> &fn)
    {
        const int parts = (int)th_.size() + 1;
        if (th_.empty() || n <= 1) {
            for (int i = 0; i < n; ++i) if (int rc = fn(i)) return rc;
            return 0;
        }
        { std::lock_guard<std::mutex> l(m_); fn_ = &fn; n_ = n; rc_ = 0; left_ = (int)th_.size(); ++gen_; }
        cv_.notify_all();
        int rc = 0;
        for (int i = 0; i < n; i += parts) if (int r = fn(i)) { rc = r; break; }
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [this] { return left_ == 0; });
        return worse(rc, rc_);
    }
    // of two results the one to report: the smallest positive one (fn returns index + 1 of a failed item), else any non-zero
    static int worse(int a, int b)
    {
        if (a > 0 && b > 0) return std::min(a, b);
        return a > 0 ? a : b > 0 ? b : a ? a : b;
    }

private:
    void work(int me)
    {
        unsigned long seen = 0;
        for (;;) {
            const std::function<int(int)> *fn;
            int n;
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                fn = fn_; n = n_;
            }
            const int parts = (int)th_.size() + 1;
            int rc = 0;
            for (int i = me; i < n; i += parts) if ((rc = (*fn)(i)) != 0) break;
            std::lock_guard<std::mutex> l(m_);
            rc_ = worse(rc_, rc);
            if (--left_ == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<int(int)> *fn_ = nullptr;
    int n_ = 0, rc_ = 0, left_ = 0;
    unsigned long gen_ = 0;
    bool stop_ = false;
};

// Reader threads PER SHARD: VMAF_HIP_READ_THREADS, default 8 (never fewer than 4 because several shards share the host:
// one copying thread moves ~8 GB/s out of the page cache, a shard's PCIe link ~51 GB/s - a shard with one or two readers
// would run at a fraction of its GPU's rate, and eight such shards together at about the rate of ONE well-fed GPU).
// Bounded by the CPUs this process may run on (sched_getaffinity; hardware_concurrency counts the whole machine).
int fetch_threads(int shards)
{
    const char *env = std::getenv("VMAF_HIP_READ_THREADS");
    if (env && std::atoi(env) > 0) return std::min(std::atoi(env), 64);
    int cpus = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
    if (cpus <= 0) cpus = (int)std::thread::hardware_concurrency();
    int n = cpus > 0 ? cpus / std::max(1, shards) : 8;
    return std::max(4, std::min(8, n));
}

int batch_frames(size_t plane)
{
    // frames per batch: ~32 MB of luma per plane and batch (16 frames of raw 8-bit 1080p), see extract_atoms
    int batch = (int)std::min<size_t>(64, std::max<size_t>(4, ((size_t)32 << 20) / plane));
    if (const char *env = std::getenv("VMAF_HIP_BATCH")) { const int v = std::atoi(env); if (v > 0) batch = std::min(v, 4096); }
    return batch;
}

// What the shards of one video share: the first failure (the one with the lowest frame index wins) and the flag that makes
// every other shard and reader thread stop at its next frame instead of scoring its whole range first.
struct ShardAbort {
    std::atomic<bool> stop{false};
    std::mutex m;
    long frame = -1;                                               // lowest failing frame index (-1: a failure without one)
    std::string msg;
    void fail(long at, const std::string &what)
    {
        std::lock_guard<std::mutex> l(m);
        if (msg.empty() || (at >= 0 && (frame < 0 || at < frame))) { frame = at; msg = what; }
        stop.store(true);
    }
};

// a failure of one shard: the frame it belongs to (-1 = none) and the message
struct ShardFailure { long frame; std::string msg; };

// One contiguous frame range [start, stop) of a seekable source on one engine: the shard
// reads and blurs ref frame start-1 itself (motion halo: ref only, no VIF/ADM), then walks its range in batches through a
// ring of kRing page-locked buffer sets: a READER stage (one thread + the helper threads of FrameFetchers) fills free sets
// ahead of the GPU, this thread hands ready sets to the engine (at most two calls in flight) and frees a set when its
// call has been collected - reading, the H2D copies and the kernels of different batches all overlap, and this thread's
// enqueue time (~0.5 ms per batch) no longer adds to the reading time.  Writes atoms[start..stop).
// Start-up order: the engine is created on a helper thread (runtime initialisation, device context, streams, the kernels'
// code object: ~30 ms) WHILE this thread page-locks the buffers and the reader stage already fetches the first batches.
constexpr int kRing = 4;

void extract_shard(const FrameSource &src, int device, int w, int h, bool scalar_order, const MetricFlags &mf, long start, long stop,
                   int readers, double *atoms, double *metrics, ShardAbort &abort)
{
    const size_t stride = (size_t)w * src.bps, plane = stride * h;
    // ~16 MB of luma per plane and batch (8 frames of 8-bit 1080p): with kRing sets that is the page-locked memory two sets
    // of 32 MB took, and the link is what bounds the run either way (tools/trace_run.sh, VMAF_HIP_BATCH=8 / 16)
    const int batch = (int)std::min<long>(batch_frames(plane * 2), stop - start);
    const long nb = (stop - start + batch - 1) / batch;
    const int ring = (int)std::min<long>(kRing, nb);
    EngineGuard guard;
    char *ref[kRing] = {}, *dis[kRing] = {};
    vmaf_hip_engine *e = nullptr;
    std::string create_err;
    std::thread creator([&] {
        if (vmaf_hip_engine_create(&e, device, w, h, batch)) { create_err = vmaf_hip_last_error(); return; }
        static const bool fast = std::getenv("VMAF_HIP_FAST") && std::atoi(std::getenv("VMAF_HIP_FAST")) != 0;
        if (vmaf_hip_engine_set_conv_order(e, scalar_order ? 0 : fast ? 2 : 1) || vmaf_hip_engine_set_sums_exact(e, scalar_order) ||
            (mf.any() && (vmaf_hip_engine_enable_metrics(e, mf.psnr, mf.ssim, mf.ms_ssim, mf.ten_bit) || vmaf_hip_engine_set_metrics_exact(e, scalar_order))))
            create_err = vmaf_hip_last_error();
    });
    try {
        for (int b = 0; b < ring; ++b) {
            char **pp[2] = { &ref[b], &dis[b] };
            for (auto p : pp) {
                *p = (char *)vmaf_hip_host_alloc(plane * batch);
                if (!*p) throw VmafException(vmaf_hip_last_error());
                guard.pinned.push_back(*p);
            }
        }
    } catch (...) {
        creator.join();
        if (e) guard.engines.push_back(e);
        throw;
    }
    trace("shard: page-locked buffers ready");

    // ---- reader stage
    struct Ring {
        std::mutex m;
        std::condition_variable cv;
        bool ready[kRing] = {};                                    // set filled, not yet handed to the engine / collected
        bool quit = false, failed = false;
        ShardFailure failure{ -1, "" };
    } rs;
    char msg[160];
    std::snprintf(msg, sizeof msg, "shard: frames [%ld, %ld), %d reader threads, %ld batches of %d, ring of %d", start, stop, readers, nb, batch, ring);
    trace(msg);
    std::vector<char> halo;                                        // the halo frame has a buffer of its own
    std::thread reader([&] {
        FrameFetchers fetch(readers - 1);
        auto give_up = [&](long frame, const std::string &what) {
            std::lock_guard<std::mutex> l(rs.m);
            rs.failed = true; rs.failure = ShardFailure{ frame, what };
            rs.cv.notify_all();
        };
        if (start > 0) {
            halo.resize(plane * 2);
            if (src.read_raw_at(halo.data(), halo.data() + plane, (int)stride, start - 1, src.user))
                return give_up(start - 1, "read_frame failed at frame " + std::to_string(start - 1) + ".");
        }
        for (long k = 0; k < nb; ++k) {
            const int b = (int)(k % ring);
            {
                std::unique_lock<std::mutex> l(rs.m);
                rs.cv.wait(l, [&] { return !rs.ready[b] || rs.quit; });
                if (rs.quit) return;
            }
            const long at = start + k * batch;
            const int n = (int)std::min<long>(batch, stop - at);
            char *rb = ref[b], *db = dis[b];
            const int bad = fetch.run(n, [&](int i) {
                if (abort.stop.load(std::memory_order_relaxed)) return -1;
                return src.read_raw_at(rb + plane * i, db + plane * i, (int)stride, at + i, src.user) ? i + 1 : 0;
            });
            if (bad > 0) return give_up(at + bad - 1, "read_frame failed at frame " + std::to_string(at + bad - 1) + ".");
            if (bad < 0) return give_up(-1, "stopped: another shard failed");
            std::lock_guard<std::mutex> l(rs.m);
            rs.ready[b] = true;
            rs.cv.notify_all();
        }
    });
    // whatever happens below: both helper threads are joined, and the engine the creator made is destroyed with the guard
    struct Joiner {
        std::thread &creator, &reader; vmaf_hip_engine *&e; EngineGuard &g; Ring &rs; bool created = false;
        void join_creator() { if (!created) { creator.join(); if (e) g.engines.push_back(e); created = true; } }
        ~Joiner()
        {
            { std::lock_guard<std::mutex> l(rs.m); rs.quit = true; }
            rs.cv.notify_all();
            reader.join();
            join_creator();
        }
    } joiner{creator, reader, e, guard, rs};

    joiner.join_creator();
    if (!create_err.empty()) throw VmafException(create_err);
    trace("shard: engine created");

    std::vector<double> out[2];
    out[0].resize((size_t)batch * kAtoms); out[1].resize((size_t)batch * kAtoms);
    struct Pend { long at = 0; int n = 0, set = 0; } pend[2];
    int inflight = 0, slot = 0;                                    // slot: which of out[] / pend[] the next call takes
    auto collect = [&]() {                                         // oldest call in flight -> atoms / metrics rows; its buffer set is free
        const int o = (slot + 2 - inflight) % 2;
        if (vmaf_hip_engine_sync_oldest(e)) throw VmafException(vmaf_hip_last_error());
        std::copy(out[o].begin(), out[o].begin() + (size_t)pend[o].n * kAtoms, atoms + (size_t)pend[o].at * kAtoms);
        if (mf.any() && vmaf_hip_engine_get_metrics(e, metrics + (size_t)pend[o].at * VMAF_HIP_METRICS, pend[o].n))
            throw VmafException(vmaf_hip_last_error());
        --inflight;
        { std::lock_guard<std::mutex> l(rs.m); rs.ready[pend[o].set] = false; }
        rs.cv.notify_all();
    };
    double t_gpu = 0, t_read = 0, t_enq = 0;                       // VMAF_HIP_TRACE: what this thread waits for
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    for (long k = 0; k < nb; ++k) {
        if (abort.stop.load(std::memory_order_relaxed)) throw ShardFailure{ -1, "stopped: another shard failed" };
        const int b = (int)(k % ring);
        const long at = start + k * batch;
        const int n = (int)std::min<long>(batch, stop - at);
        const auto t0 = now();
        if (inflight == 2) collect();
        const auto t1 = now();
        {
            std::unique_lock<std::mutex> l(rs.m);
            rs.cv.wait(l, [&] { return rs.ready[b] || rs.failed; });
            if (rs.failed) throw rs.failure;
        }
        const auto t2 = now();
        if (k == 0 && start > 0 && vmaf_hip_motion_seed_host_raw(e, halo.data(), src.bps, stride)) throw VmafException(vmaf_hip_last_error());
        if (vmaf_hip_extract_host_raw_async(e, ref[b], dis[b], src.bps, stride, plane, n, at > 0 ? 1 : 0, out[slot].data()))
            throw VmafException(vmaf_hip_last_error());
        pend[slot] = { at, n, b };
        ++inflight;
        slot ^= 1;
        t_gpu += ms(t0, t1); t_read += ms(t1, t2); t_enq += ms(t2, now());
    }
    std::snprintf(msg, sizeof msg, "shard: all batches enqueued (this thread waited %.1f ms for the GPU, %.1f ms for the readers, enqueued for %.1f ms)",
                  t_gpu, t_read, t_enq);
    trace(msg);
    while (inflight) collect();
    trace("shard: done");
}

// Frame-range sharding of ONE video over the devices of VMAF_HIP_DEVICES: device g scores frames [g*F/G, (g+1)*F/G) on its
// own host thread with its own engine and reads; no exchange between shards (the halo is re-read), motion2 and the fusion
// run after all shards have delivered (vmaf.cpp:375-379 needs motion[i+1] across the shard edge).  A shard that fails
// stops the others at their next frame; the failure with the lowest frame index is the one reported.
void extract_atoms_sharded(const FrameSource &src, int w, int h, bool scalar_order, const MetricFlags &mf, std::vector<double> &atoms,
                           std::vector<double> &metrics)
{
    if (w <= 0 || h <= 0) throw VmafException("wrong width " + std::to_string(w) + " or height " + std::to_string(h) + ".");
    const long F = src.nframes;
    if (F <= 0) throw VmafException("no frames delivered by read_frame");
    std::vector<int> devices = parse_devices();
    if ((long)devices.size() > F) devices.resize((size_t)F);
    const int G = (int)devices.size();
    atoms.assign((size_t)F * kAtoms, 0.0);
    if (mf.any()) metrics.assign((size_t)F * VMAF_HIP_METRICS, 0.0);
    ShardAbort abort;
    auto shard = [&](int g) {
        const long start = F * g / G, stop = F * (g + 1) / G;
        try { extract_shard(src, devices[g], w, h, scalar_order, mf, start, stop, fetch_threads(G), atoms.data(), mf.any() ? metrics.data() : nullptr, abort); }
        catch (const ShardFailure &f) { if (f.frame >= 0 || !abort.stop.load()) abort.fail(f.frame, f.msg); }
        catch (const std::exception &ex) { abort.fail(-1, *ex.what() ? ex.what() : "unknown error"); }
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(shard, g);
    shard(0);                                                      // the first shard on the calling thread: one thread start less
    for (auto &t : th) t.join();
    if (abort.stop.load()) throw VmafException(abort.msg.empty() ? std::string("unknown error") : abort.msg);
}

const char *kKnownFeatures[] = { "adm2", "adm_scale0", "adm_scale1", "adm_scale2", "adm_scale3", "motion", "vif_scale0",
                                 "vif_scale1", "vif_scale2", "vif_scale3", "vif", "motion2" };

// vmaf.cpp:399-427: "VMAF_feature_<name>_score" -> short name, or throw "Unknown feature name"
std::string short_feature_name(const std::string &full)
{
    const std::string pre = "VMAF_feature_", post = "_score";
    if (full.size() > pre.size() + post.size() && full.compare(0, pre.size(), pre) == 0 &&
        full.compare(full.size() - post.size(), post.size(), post) == 0) {
        const std::string mid = full.substr(pre.size(), full.size() - pre.size() - post.size());
        for (auto k : kKnownFeatures) if (mid == k) return mid;
    }
    std::printf("Unknown feature name: '%s'.\n", full.c_str());
    throw VmafException("Unknown feature name");
}

// vmaf.cpp:351-559
void fuse(const vmaf_amd::VmafModel &model, const std::vector<double> &atoms, size_t F, bool disable_clip, bool enable_transform,
          int device, std::vector<double> &vmaf, std::map<std::string, std::vector<double>> &feats)
{
    std::map<std::string, std::vector<double>> all;
    for (auto k : kKnownFeatures) all[k].resize(F);
    for (size_t i = 0; i < F; ++i) {
        const double *a = &atoms[i * kAtoms];
        all["adm2"][i] = a[0] / a[1];                              // ADM2_CONSTANT = 0, vmaf.cpp:358
        for (int s = 0; s < 4; ++s) {
            all["adm_scale" + std::to_string(s)][i] = a[2 + 2 * s] / a[3 + 2 * s];
            all["vif_scale" + std::to_string(s)][i] = a[11 + 2 * s] / a[12 + 2 * s];
        }
        all["motion"][i] = a[10];
        all["vif"][i] = a[19];
    }
    for (size_t i = 0; i + 1 < F; ++i) all["motion2"][i] = std::min(atoms[i * kAtoms + 10], atoms[(i + 1) * kAtoms + 10]);
    if (F) all["motion2"][F - 1] = atoms[(F - 1) * kAtoms + 10];   // vmaf.cpp:375-379

    const size_t D = model.feature_names.size();
    std::vector<std::string> names;
    for (auto &fn : model.feature_names) names.push_back(short_feature_name(fn));
    std::vector<double> x(F * D);
    for (size_t j = 0; j < D; ++j) {
        const std::vector<double> &v = all[names[j]];
        for (size_t i = 0; i < F; ++i)
            x[i * D + j] = model.linear_rescale ? model.slopes[j + 1] * v[i] + model.intercepts[j + 1] : v[i];
    }
    vmaf.assign(F, 0.0);
    if (F && vmaf_hip_svm_predict(device, model.sv.data(), model.coef.data(), model.n_sv, model.dim, model.gamma, model.rho,
                                  x.data(), (int)F, vmaf.data()))
        throw VmafException(vmaf_hip_last_error());
    for (size_t i = 0; i < F; ++i) {
        double p = vmaf[i];
        if (model.linear_rescale) p = (p - model.intercepts[0]) / model.slopes[0];
        if (enable_transform && model.transform.present) {        // vmaf.cpp:481-516
            const vmaf_amd::ScoreTransform &t = model.transform;
            double v = 0.0;
            if (t.has_p0) v += t.p0;
            if (t.has_p1) v += t.p1 * p;
            if (t.has_p2) v += t.p2 * p * p;
            if (t.out_lte_in && v > p) v = p;
            if (t.out_gte_in && v < p) v = p;
            p = v;
        }
        if (!disable_clip && model.has_clip) {                    // vmaf.cpp:519-529
            if (p < model.clip_lo) p = model.clip_lo;
            else if (p > model.clip_hi) p = model.clip_hi;
        }
        vmaf[i] = p;
    }
    feats.clear();
    for (auto &n : names) feats[n] = all[n];
}

// vmaf.h:59-88
double pool(const std::vector<double> &v, const char *method)
{
    if (method && std::strcmp(method, "min") == 0) {
        double m = v[0];
        for (double e : v) if (e < m) m = e;
        return m;
    }
    if (method && std::strcmp(method, "harmonic_mean") == 0) {
        double s = 0.0;
        for (double e : v) s += 1.0 / (e + 1.0);
        return 1.0 / (s / v.size()) - 1.0;
    }
    double s = 0.0;
    for (double e : v) s += e;
    return s / v.size();
}

std::string file_name_of(const std::string &s)                     // vmaf.cpp:767-778
{
    const size_t i = s.rfind('/');
    return i == std::string::npos ? std::string("") : s.substr(i + 1);
}

std::string fmt_g(double v) { char b[64]; std::snprintf(b, sizeof b, "%g", v); return b; }   // pugixml double attribute

std::string fmt_json(double v)                                     // ptools JSONPrint of a real_8
{
    char b[64];
    std::snprintf(b, sizeof b, "%.16g", v);
    std::string s = b;
    if (s.find_first_of(".en") == std::string::npos) s += ".0";
    return s;
}

double round5(double v)                                            // vmaf.cpp:755-765
{
    const double m = 100000.0;
    const double x = v * m;
    return (x < 0 ? std::ceil(x - 0.5) : std::floor(x + 0.5)) / m;
}

const char kLogVersion[] = "0.3.2";                                // vmaf.cpp:629

void write_logs(const char *log_path, const char *log_fmt, const char *model_path, int width, int height, size_t F, double aggregate,
                double exec_fps, const std::map<std::string, std::vector<double>> &result)
{
    FILE *fp = std::fopen(log_path, "w");
    if (!fp) return;                                               // the reference ignores a failed log open too
    const std::string model = file_name_of(model_path);
    if (log_fmt && std::strcmp(log_fmt, "json") == 0) {           // vmaf.cpp:675-718
        std::fprintf(fp, "{\n  \"version\":\"%s\",\n  \"params\":{\n    \"model\":\"%s\",\n    \"scaledWidth\":%d,\n    \"scaledHeight\":%d\n  },\n",
                     kLogVersion, model.c_str(), width, height);
        std::fprintf(fp, "  \"metrics\":[\n");
        size_t k = 0;
        for (auto &kv : result) std::fprintf(fp, "    \"%s\"%s\n", kv.first.c_str(), ++k < result.size() ? "," : "");
        std::fprintf(fp, "  ],\n  \"frames\":[\n");
        for (size_t i = 0; i < F; ++i) {
            std::fprintf(fp, "    {\n      \"frameNum\":%zu,\n      \"metrics\":{\n", i);
            k = 0;
            for (auto &kv : result)
                std::fprintf(fp, "        \"%s\":%s%s\n", kv.first.c_str(), fmt_json(round5(kv.second[i])).c_str(), ++k < result.size() ? "," : "");
            std::fprintf(fp, "      }\n    }%s\n", i + 1 < F ? "," : "");
        }
        std::fprintf(fp, "  ]\n}\n");
    } else {                                                       // vmaf.cpp:719-750
        std::fprintf(fp, "<?xml version=\"1.0\"?>\n<VMAF version=\"%s\">\n", kLogVersion);
        std::fprintf(fp, "\t<params model=\"%s\" scaledWidth=\"%d\" scaledHeight=\"%d\" />\n", model.c_str(), width, height);
        std::fprintf(fp, "\t<fyi numOfFrames=\"%d\" aggregateVMAF=\"%s\" execFps=\"%s\" />\n", (int)F, fmt_g(aggregate).c_str(), fmt_g(exec_fps).c_str());
        std::fprintf(fp, "\t<frames>\n");
        for (size_t i = 0; i < F; ++i) {
            std::fprintf(fp, "\t\t<frame frameNum=\"%d\"", (int)i);
            for (auto &kv : result) std::fprintf(fp, " %s=\"%s\"", kv.first.c_str(), fmt_g(kv.second[i]).c_str());
            std::fprintf(fp, " />\n");
        }
        std::fprintf(fp, "\t</frames>\n</VMAF>\n");
    }
    std::fclose(fp);
}

double run_vmaf(const char *fmt, int width, int height, const FrameSource &src,
                const char *model_path, const char *log_path, const char *log_fmt, bool disable_clip, bool disable_avx,
                bool enable_transform, bool do_psnr, bool do_ssim, bool do_ms_ssim, const char *pool_method)
{
    trace("run_vmaf: start");
    std::printf("Start calculating VMAF score...\n");             // vmaf.cpp:638
    if (!model_path) throw VmafException("Error loading model (.pkl): null path");
    if (!(src.read_raw_at ? (void *)src.read_raw_at : src.bps ? (void *)src.read_raw : (void *)src.read_frame)) throw VmafException("read_frame callback is null");

    vmaf_amd::VmafModel model;
    try { model = vmaf_amd::load_vmaf_model(model_path); }
    catch (const std::runtime_error &e) { throw VmafException(e.what()); }

    trace("run_vmaf: model loaded");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<double> atoms;
    MetricFlags mf;
    mf.psnr = do_psnr; mf.ssim = do_ssim; mf.ms_ssim = do_ms_ssim;
    const size_t fl = fmt ? std::strlen(fmt) : 0;
    mf.ten_bit = fl > 4 && std::strcmp(fmt + fl - 4, "10le") == 0;   // combo.c:171-186: the PSNR peak follows the format
    std::vector<double> metrics;
    if (src.read_raw_at) extract_atoms_sharded(src, width, height, disable_avx, mf, atoms, metrics);
    else extract_atoms(src, width, height, disable_avx, mf, atoms, metrics);
    const size_t F = atoms.size() / kAtoms;
    if (F == 0) throw VmafException("no frames delivered by read_frame");
    std::vector<double> vmaf;
    std::map<std::string, std::vector<double>> result;            // std::map: keys in alphabetical order, vmaf.h:126-134
    trace("run_vmaf: atoms extracted");
    fuse(model, atoms, F, disable_clip, enable_transform, parse_devices()[0], vmaf, result);
    trace("run_vmaf: fused");
    result["vmaf"] = vmaf;
    const char *mnames[3] = { "psnr", "ssim", "ms_ssim" };            // vmaf.cpp:595-597
    const bool mon[3] = { do_psnr, do_ssim, do_ms_ssim };
    for (int k = 0; k < 3; ++k)
        if (mon[k]) {
            std::vector<double> &v = result[mnames[k]];
            v.resize(F);
            for (size_t i = 0; i < F; ++i) v[i] = metrics[i * VMAF_HIP_METRICS + k];
        }
    const double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    const double aggregate = pool(vmaf, pool_method);
    const double exec_fps = (double)F / elapsed;
    std::printf("Exec FPS: %f\n", exec_fps);                      // vmaf.cpp:664-673
    if (pool_method) std::printf("VMAF score (%s) = %f\n", pool_method, aggregate);
    else std::printf("VMAF score = %f\n", aggregate);
    std::fflush(stdout);
    if (log_path) write_logs(log_path, log_fmt, model_path, width, height, F, aggregate, exec_fps, result);
    return aggregate;
}

void copy_err(char *errbuf, int len, const char *msg)
{
    if (errbuf && len > 0) { std::strncpy(errbuf, msg, (size_t)len - 1); errbuf[len - 1] = 0; }
}

}  // namespace

extern "C" {

double compute_vmaf(char *fmt, int width, int height, int (*read_frame)(float *, float *, float *, int, void *), void *user_data,
                    char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform,
                    int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method)
{
    // disable_avx selects the reference's scalar summation order instead of its AVX order (libvmaf.cpp:56-61)
    FrameSource src;
    src.read_frame = read_frame; src.user = user_data;
    return run_vmaf(fmt, width, height, src, model_path, log_path, log_fmt, disable_clip != 0, disable_avx != 0,
                    (enable_transform || phone_model), do_psnr != 0, do_ssim != 0, do_ms_ssim != 0, pool_method);
}

double compute_vmaf_raw(char *fmt, int width, int height, int (*read_frame_raw)(void *, void *, int, void *), void *user_data,
                        char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform,
                        int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method)
{
    FrameSource src;
    const size_t n = fmt ? std::strlen(fmt) : 0;
    src.bps = (n > 4 && std::strcmp(fmt + n - 4, "10le") == 0) ? 2 : 1;      // the *10le formats hold 16-bit words (file_io.c:137-172)
    src.read_raw = read_frame_raw; src.user = user_data;
    return run_vmaf(fmt, width, height, src, model_path, log_path, log_fmt, disable_clip != 0, disable_avx != 0,
                    (enable_transform || phone_model), do_psnr != 0, do_ssim != 0, do_ms_ssim != 0, pool_method);
}

double compute_vmaf_raw_seekable(char *fmt, int width, int height, long nframes,
                                 int (*read_frame_raw_at)(void *, void *, int, long, void *), void *user_data, char *model_path, char *log_path,
                                 char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr,
                                 int do_ssim, int do_ms_ssim, char *pool_method)
{
    FrameSource src;
    const size_t n = fmt ? std::strlen(fmt) : 0;
    src.bps = (n > 4 && std::strcmp(fmt + n - 4, "10le") == 0) ? 2 : 1;
    src.read_raw_at = read_frame_raw_at; src.nframes = nframes; src.user = user_data;
    if (!read_frame_raw_at) throw VmafException("read_frame callback is null");
    return run_vmaf(fmt, width, height, src, model_path, log_path, log_fmt, disable_clip != 0, disable_avx != 0,
                    (enable_transform || phone_model), do_psnr != 0, do_ssim != 0, do_ms_ssim != 0, pool_method);
}

int compute_vmaf_raw_seekable_noexcept(double *score, char *fmt, int width, int height, long nframes,
                                       int (*read_frame_raw_at)(void *, void *, int, long, void *), void *user_data, char *model_path,
                                       char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform,
                                       int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method, char *errbuf, int errbuf_len)
{
    try {
        const double s = compute_vmaf_raw_seekable(fmt, width, height, nframes, read_frame_raw_at, user_data, model_path, log_path, log_fmt,
                                                   disable_clip, disable_avx, enable_transform, phone_model, do_psnr, do_ssim, do_ms_ssim, pool_method);
        if (score) *score = s;
        return 0;
    } catch (const std::exception &e) {
        copy_err(errbuf, errbuf_len, e.what());
        return 1;
    } catch (...) {
        copy_err(errbuf, errbuf_len, "unknown error");
        return 1;
    }
}

int compute_vmaf_raw_noexcept(double *score, char *fmt, int width, int height, int (*read_frame_raw)(void *, void *, int, void *),
                              void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx,
                              int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method,
                              char *errbuf, int errbuf_len)
{
    try {
        const double s = compute_vmaf_raw(fmt, width, height, read_frame_raw, user_data, model_path, log_path, log_fmt, disable_clip,
                                          disable_avx, enable_transform, phone_model, do_psnr, do_ssim, do_ms_ssim, pool_method);
        if (score) *score = s;
        return 0;
    } catch (const std::exception &e) {
        copy_err(errbuf, errbuf_len, e.what());
        return 1;
    } catch (...) {
        copy_err(errbuf, errbuf_len, "unknown error");
        return 1;
    }
}

int compute_vmaf_noexcept(double *score, char *fmt, int width, int height, int (*read_frame)(float *, float *, float *, int, void *),
                          void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx,
                          int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method,
                          char *errbuf, int errbuf_len)
{
    try {
        const double s = compute_vmaf(fmt, width, height, read_frame, user_data, model_path, log_path, log_fmt, disable_clip, disable_avx,
                                      enable_transform, phone_model, do_psnr, do_ssim, do_ms_ssim, pool_method);
        if (score) *score = s;
        return 0;
    } catch (const std::exception &e) {
        copy_err(errbuf, errbuf_len, e.what());
        return 1;
    } catch (...) {
        copy_err(errbuf, errbuf_len, "unknown error");
        return 1;
    }
}

// repeated calls with one model (a rank fusing batch after batch) parse the two model files once
static vmaf_amd::VmafModel cached_model(const char *model_path)
{
    static std::mutex mu;
    static std::map<std::string, vmaf_amd::VmafModel> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(model_path);
    if (it == cache.end()) it = cache.emplace(model_path, vmaf_amd::load_vmaf_model(model_path)).first;
    return it->second;
}

int vmaf_amd_model_feature_names(const char *model_path, int *n_features, char *names_out, int names_len, char *errbuf, int errbuf_len)
{
    try {
        if (!model_path || !n_features) throw VmafException("bad argument");
        const vmaf_amd::VmafModel model = cached_model(model_path);
        *n_features = (int)model.feature_names.size();
        if (names_out) {
            std::string all;
            for (size_t j = 0; j < model.feature_names.size(); ++j) all += (j ? "\n" : "") + model.feature_names[j];
            if (names_len <= 0 || all.size() + 1 > (size_t)names_len) throw VmafException("names_out too small: " + std::to_string(all.size() + 1) + " bytes needed");
            std::memcpy(names_out, all.c_str(), all.size() + 1);
        }
        return 0;
    } catch (const std::exception &e) {
        copy_err(errbuf, errbuf_len, e.what());
        return 1;
    }
}

int vmaf_amd_predict_from_atoms_v2(const char *model_path, const double *atoms, int nframes, int disable_clip, int enable_transform,
                                   int device, double *vmaf_out, double *features_out, int features_capacity, char *errbuf, int errbuf_len)
{
    try {
        if (!model_path || !atoms || !vmaf_out || nframes <= 0) throw VmafException("bad argument");
        const vmaf_amd::VmafModel model = cached_model(model_path);
        const size_t D = model.feature_names.size();
        if (features_out && (features_capacity < 0 || (size_t)features_capacity < D))
            throw VmafException("features_out holds " + std::to_string(features_capacity) + " columns, the model has " + std::to_string(D) + " features");
        std::vector<double> a(atoms, atoms + (size_t)nframes * kAtoms), vmaf;
        std::map<std::string, std::vector<double>> feats;
        fuse(model, a, (size_t)nframes, disable_clip != 0, enable_transform != 0, device, vmaf, feats);
        std::copy(vmaf.begin(), vmaf.end(), vmaf_out);
        if (features_out) {
            for (size_t j = 0; j < D; ++j) {
                const std::vector<double> &v = feats[short_feature_name(model.feature_names[j])];
                for (int i = 0; i < nframes; ++i) features_out[(size_t)i * features_capacity + j] = v[i];
            }
        }
        return 0;
    } catch (const std::exception &e) {
        copy_err(errbuf, errbuf_len, e.what());
        return 1;
    }
}

// the prototype this symbol was first shipped with: features_out rows are exactly as wide as the model has features
int vmaf_amd_predict_from_atoms(const char *model_path, const double *atoms, int nframes, int disable_clip, int enable_transform,
                                int device, double *vmaf_out, double *features_out, char *errbuf, int errbuf_len)
{
    int D = 0;
    if (features_out && vmaf_amd_model_feature_names(model_path, &D, nullptr, 0, errbuf, errbuf_len)) return 1;
    return vmaf_amd_predict_from_atoms_v2(model_path, atoms, nframes, disable_clip, enable_transform, device, vmaf_out, features_out, D,
                                          errbuf, errbuf_len);
}

}  // extern "C"
