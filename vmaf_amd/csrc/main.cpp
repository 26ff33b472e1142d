// main.cpp - vmafossexec for the MI355X engine.  Same argv contract, messages and exit codes as the
// reference CLI (wrapper/src/main.cpp:49-232), with its own file-backed read_frame callback
// (behaviour of feature/src/common/frame.c:26-129 + file_io.c:86-172).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "../../include/libvmaf.h"

namespace {

// One helper thread for the whole run of the sequential (FIFO) path: it reads the dis plane of every frame while the calling
// thread reads the ref plane (a thread started per frame costs more than it saves on small frames).
class SideReader {
public:
    ~SideReader()
    {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        if (t_.joinable()) t_.join();
    }
    void start(std::function<int()> job)
    {
        if (!t_.joinable()) t_ = std::thread([this] { loop(); });
        { std::lock_guard<std::mutex> l(m_); job_ = std::move(job); busy_ = true; }
        cv_.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [this] { return !busy_; });
        return rc_;
    }

private:
    void loop()
    {
        std::unique_lock<std::mutex> l(m_);
        for (;;) {
            cv_.wait(l, [this] { return stop_ || (busy_ && job_); });
            if (stop_) return;
            std::function<int()> job = std::move(job_);
            job_ = nullptr;
            l.unlock();
            const int rc = job();
            l.lock();
            rc_ = rc; busy_ = false;
            cv_.notify_all();
        }
    }
    std::thread t_;
    std::mutex m_;
    std::condition_variable cv_;
    std::function<int()> job_;
    bool busy_ = false, stop_ = false;
    int rc_ = 0;
};

struct YuvPair {
    FILE *ref = nullptr, *dis = nullptr;
    int width = 0, height = 0;
    bool ten_bit = false;
    size_t chroma_samples = 0;          // samples to skip after each luma plane (main.cpp:69-86)
    std::vector<unsigned char> row;
    // regular files (compute_vmaf_raw_seekable): frames are read by position, from several shard threads at once
    size_t luma_bytes = 0, frame_bytes = 0;
    SideReader side;                    // sequential raw path only
    std::vector<unsigned char> dis_scratch;
};

// One luma plane -> fp32 in 8-bit units (10-bit: value / 4.0, file_io.c:159-163). 0 ok, 1 short read.
int read_luma(FILE *f, YuvPair &s, float *dst, int stride_byte)
{
    const size_t bpp = s.ten_bit ? 2 : 1;
    s.row.resize((size_t)s.width * bpp);
    for (int i = 0; i < s.height; ++i) {
        if (std::fread(s.row.data(), bpp, (size_t)s.width, f) != (size_t)s.width) return 1;
        float *out = (float *)((char *)dst + (size_t)i * stride_byte);
        if (s.ten_bit) {
            for (int j = 0; j < s.width; ++j) {
                const unsigned v = s.row[2 * j] | ((unsigned)s.row[2 * j + 1] << 8);   // little-endian words
                out[j] = (float)(v / 4.0);
            }
        } else {
            for (int j = 0; j < s.width; ++j) out[j] = (float)s.row[j];
        }
    }
    return 0;
}

// Skip the chroma planes; seek when the stream allows it, otherwise read and discard (FIFOs).
void skip_chroma(FILE *f, YuvPair &s, float *scratch, size_t scratch_bytes)
{
    size_t bytes = s.chroma_samples * (s.ten_bit ? 2 : 1);
    if (std::fseek(f, (long)bytes, SEEK_CUR) == 0) return;
    while (bytes) {
        const size_t n = std::min(bytes, scratch_bytes);
        if (std::fread(scratch, 1, n, f) != n) { std::fprintf(stderr, "fread u and v failed.\n"); return; }
        bytes -= n;
    }
}

// libvmaf read_frame contract: 0 frame delivered, 2 end of stream, 1 failure (frame.c:48-78)
int read_frame_cb(float *ref_data, float *dis_data, float *temp_data, int stride_byte, void *user)
{
    YuvPair &s = *(YuvPair *)user;
    if (read_luma(s.ref, s, ref_data, stride_byte)) return std::feof(s.ref) ? 2 : 1;
    if (read_luma(s.dis, s, dis_data, stride_byte)) return std::feof(s.dis) ? 2 : 1;
    const size_t scratch = (size_t)stride_byte * s.height * 2;
    skip_chroma(s.ref, s, temp_data, scratch);
    skip_chroma(s.dis, s, temp_data, scratch);
    return 0;
}

// The same frame rule with raw hand-over (compute_vmaf_raw): the luma plane goes to the library as the file holds it,
// the GPU does file_io.c's conversion.  One fread per plane.
int read_frame_raw_cb(void *ref_luma, void *dis_luma, int stride_byte, void *user)
{
    YuvPair &s = *(YuvPair *)user;
    const size_t bytes = (size_t)stride_byte * s.height;
    // the two files are read side by side: one copy from the page cache into page-locked memory per thread is what
    // bounds the file-to-score rate of this CLI
    s.side.start([&s, dis_luma, bytes]() -> int {
        if (std::fread(dis_luma, 1, bytes, s.dis) != bytes) return std::feof(s.dis) ? 2 : 1;
        s.dis_scratch.resize(1 << 20);
        skip_chroma(s.dis, s, (float *)s.dis_scratch.data(), s.dis_scratch.size());
        return 0;
    });
    int ref_rc = 0;
    if (std::fread(ref_luma, 1, bytes, s.ref) != bytes) ref_rc = std::feof(s.ref) ? 2 : 1;
    else {
        s.row.resize(1 << 20);
        skip_chroma(s.ref, s, (float *)s.row.data(), s.row.size());
    }
    const int dis_rc = s.side.wait();
    return ref_rc ? ref_rc : dis_rc;                             // frame.c:48-78: the ref read is checked first
}

// Positional form for regular files (compute_vmaf_raw_seekable): the two luma planes of frame `index` by pread().  The
// library calls it from several threads at once (shards, and the reader threads of a shard: VMAF_HIP_READ_THREADS);
// it touches no shared state but the two descriptors.
int read_frame_raw_at_cb(void *ref_luma, void *dis_luma, int stride_byte, long index, void *user)
{
    const YuvPair &s = *(const YuvPair *)user;
    const size_t bytes = (size_t)stride_byte * s.height;
    if (bytes != s.luma_bytes) return 1;
    const off_t at = (off_t)index * (off_t)s.frame_bytes;
    for (int f = 0; f < 2; ++f) {
        char *dst = (char *)(f ? dis_luma : ref_luma);
        const int fd = fileno(f ? s.dis : s.ref);
        size_t got = 0;
        while (got < bytes) {                                    // pread may return less than asked
            const ssize_t r = ::pread(fd, dst + got, bytes - got, at + (off_t)got);
            if (r <= 0) return 1;                                // the frame count was taken from the file sizes: a short read is a failure
            got += (size_t)r;
        }
    }
    return 0;
}

// Frames a raw YUV file of `size` bytes holds under the reference's rule (frame.c:48-87): a frame counts when its luma
// plane is complete; its chroma may be cut short.
long frames_in(off_t size, size_t luma_bytes, size_t frame_bytes)
{
    long n = (long)(size / (off_t)frame_bytes);
    if ((size_t)(size - (off_t)n * (off_t)frame_bytes) >= luma_bytes) ++n;
    return n;
}

const char *opt_value(char **b, char **e, const std::string &name)
{
    char **it = std::find(b, e, name);
    return (it != e && ++it != e) ? *it : nullptr;
}
bool opt_present(char **b, char **e, const std::string &name) { return std::find(b, e, name) != e; }

void usage(const char *prog)
{
    std::fprintf(stderr, "Usage: %s fmt width height ref_path dis_path model_path [--log log_path] [--log-fmt log_fmt] [--disable-clip] "
                         "[--disable-avx] [--psnr] [--ssim] [--ms-ssim] [--phone-model]\n", prog);
    std::fprintf(stderr, "fmt:\n\tyuv420p\n\tyuv422p\n\tyuv444p\n\tyuv420p10le\n\tyuv422p10le\n\tyuv444p10le\n\n");
    std::fprintf(stderr, "log_fmt:\n\txml (default)\n\tjson\n\n");
}

// VMAF_HIP_TRACE=1: wall-clock (CLOCK_REALTIME) stamps of the process itself, so that a parent that noted the time before
// the exec can tell loader time, library time and teardown apart (tools/trace_run.sh)
void stamp(const char *what)
{
    static const bool on = std::getenv("VMAF_HIP_TRACE") && std::atoi(std::getenv("VMAF_HIP_TRACE")) != 0;
    if (!on) return;
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    std::fprintf(stderr, "[vmafossexec %.6f] %s\n", (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec, what);
}

}  // namespace

int main(int argc, char *argv[])
{
    stamp("main entered");
    if (argc < 7) { usage(argv[0]); return -1; }
    char *fmt = argv[1];
    int width = 0, height = 0;
    try { width = std::stoi(argv[2]); height = std::stoi(argv[3]); }
    catch (const std::exception &e) { std::fprintf(stderr, "Error: %s\n", e.what()); usage(argv[0]); return -1; }
    char *ref_path = argv[4], *dis_path = argv[5], *model_path = argv[6];
    if (width <= 0 || height <= 0) {
        std::fprintf(stderr, "Error: Invalid frame resolution %d x %d\n", width, height);
        usage(argv[0]);
        return -1;
    }
    char **ob = argv + 7, **oe = argv + argc;
    const char *log_path = opt_value(ob, oe, "--log");
    const char *log_fmt = opt_value(ob, oe, "--log-fmt");
    if (log_fmt && std::strcmp(log_fmt, "xml") && std::strcmp(log_fmt, "json")) {
        std::fprintf(stderr, "Error: log_fmt must be xml or json, but is %s\n", log_fmt);
        return -1;
    }
    const bool disable_clip = opt_present(ob, oe, "--disable-clip"), disable_avx = opt_present(ob, oe, "--disable-avx");
    const bool enable_transform = opt_present(ob, oe, "--enable-transform"), phone_model = opt_present(ob, oe, "--phone-model");
    const bool do_psnr = opt_present(ob, oe, "--psnr"), do_ssim = opt_present(ob, oe, "--ssim"), do_ms_ssim = opt_present(ob, oe, "--ms-ssim");
    const char *pool_method = opt_value(ob, oe, "--pool");
    if (pool_method && std::strcmp(pool_method, "min") && std::strcmp(pool_method, "harmonic_mean") && std::strcmp(pool_method, "mean")) {
        std::fprintf(stderr, "Error: pool_method must be min, harmonic_mean or mean, but is %s\n", pool_method);
        return -1;
    }

    YuvPair s;
    s.width = width; s.height = height;
    const std::string f = fmt;
    const size_t n = (size_t)width * height;
    if (f == "yuv420p" || f == "yuv420p10le") {
        if (n % 2 != 0) { std::fprintf(stderr, "(width * height) %% 2 != 0, width = %d, height = %d.\n", width, height); return 1; }
        s.chroma_samples = n / 2;
    } else if (f == "yuv422p" || f == "yuv422p10le") s.chroma_samples = n;
    else if (f == "yuv444p" || f == "yuv444p10le") s.chroma_samples = n * 2;
    else { std::fprintf(stderr, "unknown format %s.\n", fmt); return 1; }
    s.ten_bit = f.size() > 4 && f.compare(f.size() - 4, 4, "10le") == 0;

    int ret = 0;
    if (!(s.ref = std::fopen(ref_path, "rb"))) { std::fprintf(stderr, "fopen ref_path %s failed.\n", ref_path); ret = 1; }
    else if (!(s.dis = std::fopen(dis_path, "rb"))) { std::fprintf(stderr, "fopen ref_path %s failed.\n", dis_path); ret = 1; }
    bool seekable = false;
    long nframes = 0;
    if (!ret) {
        // regular files are read by position: the library fetches the frames of a batch on several threads
        // (VMAF_HIP_READ_THREADS, default 8; 0 or 1 = the shard's own thread only) and splits the video into frame-range
        // shards over the devices of VMAF_HIP_DEVICES.  Pipes and FIFOs keep the
        // sequential path.  VMAF_HIP_SEEKABLE=0 forces the sequential path for files too.
        struct stat sr, sd;
        const char *sk = std::getenv("VMAF_HIP_SEEKABLE");
        const bool allow = !(sk && *sk == '0');
        if (allow && fstat(fileno(s.ref), &sr) == 0 && fstat(fileno(s.dis), &sd) == 0 && S_ISREG(sr.st_mode) && S_ISREG(sd.st_mode)) {
            const size_t bps = s.ten_bit ? 2 : 1;
            s.luma_bytes = n * bps;
            s.frame_bytes = (n + s.chroma_samples) * bps;
            nframes = std::min(frames_in(sr.st_size, s.luma_bytes, s.frame_bytes), frames_in(sd.st_size, s.luma_bytes, s.frame_bytes));
            seekable = nframes > 0;                              // an empty pair takes the sequential path and its error message
        }
    }
    if (!ret) {
        try {
            const char *fp32 = std::getenv("VMAF_HIP_FP32_CALLBACK");      // 1: the reference's fp32 read_frame contract instead
            if (fp32 && *fp32 && *fp32 != '0')
                compute_vmaf(fmt, width, height, read_frame_cb, &s, model_path, (char *)log_path, (char *)log_fmt, disable_clip, disable_avx,
                             enable_transform, phone_model, do_psnr, do_ssim, do_ms_ssim, (char *)pool_method);
            else if (seekable)
                compute_vmaf_raw_seekable(fmt, width, height, nframes, read_frame_raw_at_cb, &s, model_path, (char *)log_path, (char *)log_fmt,
                                          disable_clip, disable_avx, enable_transform, phone_model, do_psnr, do_ssim, do_ms_ssim, (char *)pool_method);
            else
                compute_vmaf_raw(fmt, width, height, read_frame_raw_cb, &s, model_path, (char *)log_path, (char *)log_fmt, disable_clip,
                                 disable_avx, enable_transform, phone_model, do_psnr, do_ssim, do_ms_ssim, (char *)pool_method);
        } catch (const std::exception &e) {
            std::fprintf(stderr, "Error: %s\n", e.what());
            usage(argv[0]);
            ret = -1;
        }
    }
    if (s.ref) std::fclose(s.ref);
    if (s.dis) std::fclose(s.dis);
    stamp("main returns");
    // (Leaving with _exit() to skip the HIP / HSA exit handlers, ~23 ms, is no shortcut: without the orderly shutdown the
    // kernel driver tears the process's queues down itself and the parent waits ~220 ms for the exit.)
    return ret;
}
