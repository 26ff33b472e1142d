// vmaf_engine.cpp - the vmaf_hip_* C-ABI (include/vmaf_hip.h): device memory, streams, launch order
// and the host-side finishing of the raw device sums into the 20 reference atoms.
//
// HBM layout per engine (B = max_batch frames, all fp32, dense rows):
//   vif pyramid  scale 1..3: ref + dis planes, (w>>s) x (h>>s) each, B of them      (vif.c:157-186)
//   adm LL bands level 0..2: ref + dis planes, ceil-halved sizes, B of them         (adm.c:241-247)
//   carry[2]     blurred ref of the last frame of the previous call, w x h          (combo.c:326)
//   staging      on first use only: raw 8/16-bit planes of a host-pointer raw call, B fp32 ref + dis planes for the fp32
//                host-pointer entry point (and for the optional metrics when the input is raw luma)
//   partials     per-block fp64 partial sums of every kernel of the batch
//   sums         B x VK_NSUMS fp64, copied to pinned host memory
// The VIF chain, the ADM chain and motion run on three streams and join before the reduction.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <vector>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>

#include "../../include/vmaf_hip.h"
#include "../../include/vmaf_hip_debug.h"
#include "vmaf_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

// VMAF_HIP_TRACE=1: wall-clock milestones on stderr (where the fixed cost of a short run goes; libvmaf.cpp has the same switch)
void trace(const char *what)
{
    static const bool on = getenv("VMAF_HIP_TRACE") && atoi(getenv("VMAF_HIP_TRACE")) != 0;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    fprintf(stderr, "[vmaf_hip engine %8.3f ms] %s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), what);
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(_e));      \
    } while (0)

// adm_tools.h:105-115 (Watson DWT quantisation step, luma, 7/9 basis amplitudes) -> 1/Q as adm_csf uses it
float adm_rfactor(int lambda, int theta)
{
    static const float a = 0.495f, k = 0.466f, f0 = 0.401f;
    static const float g[4] = { 1.501f, 1.0f, 0.534f, 1.0f };
    static const float amp[6][4] = {
        { 0.62171f, 0.67234f, 0.72709f, 0.67234f }, { 0.34537f, 0.41317f, 0.49428f, 0.41317f },
        { 0.18004f, 0.22727f, 0.28688f, 0.22727f }, { 0.091401f, 0.11792f, 0.15214f, 0.11792f },
        { 0.045943f, 0.059758f, 0.077727f, 0.059758f }, { 0.023013f, 0.030018f, 0.039156f, 0.030018f } };
    const float r = (float)(3.0f * 1080 * 3.14159265358979323846264338327 / 180.0);
    const float t = (float)std::log10(std::pow(2.0, lambda + 1) * f0 * g[theta] / r);
    const float q = (float)(2.0 * a * std::pow(10.0, k * t * t) / amp[lambda][theta]);
    return 1.0f / q;
}

}  // namespace

constexpr int kDepth = 2;

// Everything that follows from the frame size and the batch size alone: plane sizes per scale, tiles per kernel, where each
// kernel's per-workgroup partial sums start in the partial buffer, ADM constants.  Shared by the engine and by the
// allocation-free entry point, which lays the same buffers out in the static scratch area.
struct Geometry {
    int w = 0, h = 0, max_batch = 0;
    int vw[4], vh[4];              // VIF plane size per scale
    int aw[5], ah[5];              // ADM input size per scale (index 4 = size of the last LL, unused)
    size_t part_off_vif[4], part_off_adm[4], part_off_mot, part_off_vifx, part_total;   // vifx: 8 per frame, the VIF sums in vif_sum_s's order
    int tiles_vif[4], tiles_adm[4], tiles_mot;
    float rfac[8];
    int win[4][4];                 // ADM centre window per scale: left, top, right, bottom
};

// Device buffers of one batch in flight (engine: hipMalloc'd at create; no-alloc entry: carved from the static scratch)
struct FeatureBuffers {
    float *vif_ref[4] = {}, *vif_dis[4] = {};   // [1..3]
    float *adm_ref[4] = {}, *adm_dis[4] = {};   // [1..3] = LL of level 0..2
    double *partials = nullptr, *sums_dev = nullptr;
    // set when the VIF plane sums are wanted in vif_sum_s's own order (vmaf_hip_engine_set_sums_exact): per-pixel num / den
    // planes of the scale being worked on ([frame][2][h][w], sized for scale 0) and the row sums of such a pair of planes
    float *vif_nd = nullptr, *vif_rowsum = nullptr;
};

struct vmaf_hip_engine : Geometry, FeatureBuffers {
    int device = 0;
    hipStream_t s_main = nullptr, s_adm = nullptr, s_mot = nullptr;
    // the join of the three chains, the reduction and the D2H copy of a batch run on their own stream: s_main goes straight
    // on to the next batch's scale-0 VIF launch while the short ADM tail of this batch is still running.  That is why the
    // per-workgroup partial sums exist twice (one set per pending slot): batch k+1 writes its set before batch k's is reduced
    hipStream_t s_red = nullptr;
    hipEvent_t ev_vif = nullptr;
    double *partials_set[kDepth] = {}, *sums_set[kDepth] = {};
    hipEvent_t ev_in = nullptr, ev_adm = nullptr, ev_mot = nullptr;
    hipEvent_t ev_t0[kDepth] = {}, ev_t1[kDepth] = {}, ev_k0[kDepth] = {}, ev_k1[kDepth] = {}, ev_done[kDepth] = {};
    bool timed = false;
    bool serial = false;
    int order = 1;                 // convolution tap order: 1 = reference AVX path (default), 0 = scalar path, 2 = fused (fast, not bit-exact)

    // host-pointer entry points: the H2D copies of call k+1 run on s_copy while the kernels of call k run, into the
    // staging set of their own pending slot (raw 8/16-bit luma, or fp32 planes for the reference's read_frame contract)
    hipStream_t s_copy = nullptr;
    hipEvent_t ev_copy[kDepth] = {};
    unsigned char *raw_ref[kDepth] = {}, *raw_dis[kDepth] = {};   // allocated on first use
    float *carry[2] = {};
    unsigned char *arena = nullptr;                              // ONE device allocation: pyramid planes, LL bands, carry planes, reduced sums
    int carry_cur = 0;
    bool carry_valid = false;
    float *stage_ref[kDepth] = {}, *stage_dis[kDepth] = {};      // fp32 planes, allocated on first use: the fp32 host path, and raw input with metrics on
    void *seed_buf = nullptr;                                    // one plane for the motion seed of a host frame, on first use
    double *sums_host[kDepth] = {};

    // optional metrics (combo.c:168-230): PSNR, SSIM, MS-SSIM on the un-offset planes
    bool m_psnr = false, m_ssim = false, m_ms = false, m_ansnr = false;
    int m_exact = 0;                               // SSIM / MS-SSIM window sums and per-pixel terms as the reference's own expressions
    double psnr_peak = 255.0, psnr_max = 60.0;
    hipStream_t s_iqa = nullptr;
    hipEvent_t ev_iqa = nullptr;
    int ssim_f = 1, ssim_w = 0, ssim_h = 0;        // ssim.c:86 decimation factor and the size _iqa_ssim sees
    float *ssim_ref = nullptr, *ssim_dis = nullptr;
    int ms_w[5] = {}, ms_h[5] = {};
    float *ms_ref[5] = {}, *ms_dis[5] = {};        // [1..4]
    size_t part_off_psnr = 0, part_off_ssim = 0, part_off_ms[5] = {}, part_off_ansnr = 0;
    int tiles_psnr = 0, tiles_ssim = 0, tiles_ms[5] = {}, tiles_ansnr = 0;
    std::vector<double> metrics;                   // [frames of the last synchronised call][VMAF_HIP_METRICS]

    // asynchronous calls in flight (at most kDepth: the host can enqueue batch k+1 while batch k runs, so the GPU
    // does not idle between batches); slot i owns sums_host[i] and the events [i]
    struct Pending { int frames = 0; bool first_has_motion = false; double *out = nullptr; } pend[kDepth];
    int pend_head = 0, pend_count = 0, last_slot = 0;
};

namespace {

// Finish one frame: VK_NSUMS raw fp64 sums -> 20 atoms, in the reference's own precision steps.
void finish_atoms(const Geometry *e, const double *s, bool has_motion, double *atoms)
{
    // vif.c:265-291: vif_sum returns float; per-scale values widen to double, totals add in double
    double vnum = 0, vden = 0;
    for (int sc = 0; sc < 4; ++sc) {
        const double n = (double)(float)s[2 * sc], d = (double)(float)s[2 * sc + 1];
        atoms[11 + 2 * sc] = n;
        atoms[12 + 2 * sc] = d;
        vnum += n;
        vden += d;
    }
    atoms[19] = (vden == 0.0) ? 1.0 : vnum / vden;

    // adm_tools.c:76 + adm.c:229-268
    double num = 0, den = 0;
    for (int sc = 0; sc < 4; ++sc) {
        const int *wn = e->win[sc];
        const float area = powf((wn[3] - wn[1]) * (wn[2] - wn[0]) / 32.0f, 1.0f / 3.0f);
        float num_s = 0.0f, den_s = 0.0f;
        for (int b = 0; b < 3; ++b) num_s += powf((float)s[8 + 6 * sc + b], 1.0f / 3.0f) + area;
        for (int b = 0; b < 3; ++b) den_s += powf((float)s[8 + 6 * sc + 3 + b], 1.0f / 3.0f) + area;
        num += num_s;
        den += den_s;
        atoms[2 + 2 * sc] = num_s;
        atoms[3 + 2 * sc] = den_s;
    }
    const double limit = 1e-10 * (e->w * e->h) / (1920.0 * 1080.0);
    atoms[0] = num < limit ? 0 : num;
    atoms[1] = den < limit ? 0 : den;

    // motion.c:59: float sum / (w*h) in float
    atoms[10] = has_motion ? (double)(float)((float)s[32] / (e->w * e->h)) : 0.0;
}

// psnr.c:51-55, ssim_tools.c:236-241, ms_ssim.c:268-312 from the reduced sums of one frame
void finish_metrics(const vmaf_hip_engine *e, const double *s, double *m)
{
    for (int k = 0; k < VMAF_HIP_METRICS; ++k) m[k] = 0.0;
    if (e->m_ansnr) {                                                   // ansnr.c:92-98
        const float sig = (float)s[58], noise = (float)s[59];
        m[3] = noise == 0 ? e->psnr_max : 10.0 * std::log10(sig / noise);
        const double eps = 1e-10;
        m[4] = std::min(10 * std::log10(e->psnr_peak * e->psnr_peak * e->w * e->h / std::max((double)noise, eps)), e->psnr_max);
    }
    if (e->m_psnr) {
        double noise = s[33];
        noise /= (e->w * e->h);
        const double eps = 1e-10;
        m[0] = std::min(10 * std::log10(e->psnr_peak * e->psnr_peak / std::max(noise, eps)), e->psnr_max);
    }
    if (e->m_ssim) m[1] = (double)(float)(s[34] / (double)((e->ssim_w - 10) * (e->ssim_h - 10)));
    if (e->m_ms) {
        static const float alphas[5] = { 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.1333f };   // ms_ssim.c:62-64
        static const float betas[5] = { 0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f };
        static const float gammas[5] = { 0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f };
        double msssim = 1.0;
        for (int i = 0; i < 5; ++i) {
            const double n = (double)((e->ms_w[i] - 10) * (e->ms_h[i] - 10));
            const float l = (float)(s[38 + 4 * i + 1] / n), c = (float)(s[38 + 4 * i + 2] / n), st = (float)(s[38 + 4 * i + 3] / n);
            // C's pow(): the float arguments widen to double (std::pow(float, float) would be powf, 4e-8 off in the score)
            msssim *= std::pow((double)l, (double)alphas[i]) * std::pow((double)c, (double)betas[i]) * std::pow((double)st, (double)gammas[i]);
        }
        m[2] = msssim;
    }
}

void geometry_init(Geometry *g, int width, int height, int max_batch)
{
    g->w = width; g->h = height; g->max_batch = max_batch;
    const size_t B = (size_t)max_batch;
    g->vw[0] = width; g->vh[0] = height;
    for (int s = 1; s < 4; ++s) { g->vw[s] = g->vw[s - 1] / 2; g->vh[s] = g->vh[s - 1] / 2; }      // vif.c:172-173
    g->aw[0] = width; g->ah[0] = height;
    for (int s = 1; s < 5; ++s) { g->aw[s] = (g->aw[s - 1] + 1) / 2; g->ah[s] = (g->ah[s - 1] + 1) / 2; }  // adm.c:135-136
    for (int s = 0; s < 4; ++s) {
        g->rfac[2 * s] = adm_rfactor(s, 1);
        g->rfac[2 * s + 1] = adm_rfactor(s, 2);
        const int bw = g->aw[s + 1], bh = g->ah[s + 1];
        const int left = (int)(bw * 0.1 - 0.5), top = (int)(bh * 0.1 - 0.5);                  // adm_tools.c:54-57
        g->win[s][0] = left; g->win[s][1] = top; g->win[s][2] = bw - left; g->win[s][3] = bh - top;
    }
    size_t off = 0;
    for (int s = 0; s < 4; ++s) { g->tiles_vif[s] = vk_vif_tiles(g->vw[s], g->vh[s]); g->part_off_vif[s] = off; off += B * g->tiles_vif[s] * 2; }
    for (int s = 0; s < 4; ++s) { g->tiles_adm[s] = vk_adm_tiles(g->aw[s], g->ah[s]); g->part_off_adm[s] = off; off += B * g->tiles_adm[s] * 6; }
    g->tiles_mot = vk_motion_tiles(width, height); g->part_off_mot = off; off += B * g->tiles_mot;
    g->part_off_vifx = off; off += B * 8;
    g->part_total = off;
}

int alloc_planes(float **p, size_t floats) { HIP_TRY(hipMalloc((void **)p, floats * sizeof(float))); return 0; }

// carve n elements, 256-byte aligned, from a buffer; only sizes the layout when base is null
template <typename T> T *carve(unsigned char *base, size_t *off, size_t n)
{
    const size_t at = (*off + 255) & ~(size_t)255;
    *off = at + n * sizeof(T);
    return base ? reinterpret_cast<T *>(base + at) : nullptr;
}

// Where a chain reads its current scale from: the caller's picture at scale 0 (fp32 planes, or raw 8/16-bit luma that the
// kernels widen in their loads), the fp32 pyramid buffers afterwards.  Strides in samples.
struct PlaneCursor { const void *ref, *dis; int sample; size_t frame_stride; int row_stride; };

// One VIF scale: statistic + (sc < 3) the next scale's planes, both from one read of the current planes (vif.c:157-269)
int launch_vif_scale(const Geometry *g, const FeatureBuffers *b, int order, int sc, PlaneCursor *c, int nframes, hipStream_t s)
{
    float *nr = sc < 3 ? b->vif_ref[sc + 1] : nullptr, *nd = sc < 3 ? b->vif_dis[sc + 1] : nullptr;
    const size_t nfs = sc < 3 ? (size_t)g->vw[sc + 1] * g->vh[sc + 1] : 0;
    const int nrs = sc < 3 ? g->vw[sc + 1] : 0;
    HIP_TRY(vk_launch_vif_stat(sc, order, c->ref, c->dis, c->sample, c->frame_stride, c->row_stride, g->vw[sc], g->vh[sc], nframes,
                               b->partials + g->part_off_vif[sc], nr, nd, nfs, nrs, b->vif_nd, s));
    if (b->vif_nd)                                                 // vif.c:265-266 in vif_sum_s's order; the planes are reused by the next scale
        HIP_TRY(vk_launch_vif_sums(b->vif_nd, g->vw[sc], g->vh[sc], nframes, b->vif_rowsum, b->partials + g->part_off_vifx, 8, sc, s));
    *c = { nr, nd, 4, nfs, nrs };
    return 0;
}

// One ADM scale (adm.c:132-247); its LL bands are the next scale's input
int launch_adm_scale(const Geometry *g, const FeatureBuffers *b, int order, int sc, PlaneCursor *c, int nframes, hipStream_t s)
{
    const int bw = g->aw[sc + 1], bh = g->ah[sc + 1];
    const size_t lfs = (size_t)bw * bh;
    float *lr = sc < 3 ? b->adm_ref[sc + 1] : nullptr, *ld = sc < 3 ? b->adm_dis[sc + 1] : nullptr;
    HIP_TRY(vk_launch_adm_scale(sc, order, c->ref, c->dis, c->sample, c->frame_stride, c->row_stride, g->aw[sc], g->ah[sc], lr, ld, lfs, bw,
                                g->rfac[2 * sc], g->rfac[2 * sc + 1], g->win[sc][0], g->win[sc][1], g->win[sc][2], g->win[sc][3], nframes,
                                b->partials + g->part_off_adm[sc], s));
    *c = { lr, ld, 4, lfs, bw };
    return 0;
}

// Sections [0..32] of the reduction table: VIF num/den x 4 scales, ADM 6 x 4 scales, motion.  Returns the next index.
int fill_reduce_table(const Geometry *g, vk_reduce_table *tab, bool vif_sums_exact)
{
    int q = 0;
    for (int sc = 0; sc < 4; ++sc)
        for (int c = 0; c < 2; ++c) {
            if (vif_sums_exact) tab->d[q++] = { g->part_off_vifx, 1, 8, 2 * sc + c };   // one value per frame: vif_total_kernel's fp32 total
            else tab->d[q++] = { g->part_off_vif[sc], g->tiles_vif[sc], 2, c };
        }
    for (int sc = 0; sc < 4; ++sc)
        for (int c = 0; c < 6; ++c) tab->d[q++] = { g->part_off_adm[sc], g->tiles_adm[sc], 6, c };
    tab->d[q++] = { g->part_off_mot, 1, g->tiles_mot, 0 };    // one value per frame: motion_total_kernel's fp32 total
    return q;
}

int ensure_stage(vmaf_hip_engine *e, int slot);

// in_ref / in_dis: device planes of `sample` bytes per sample (4 = fp32 as a read_frame callback delivers them, 1 / 2 = raw
// luma as the YUV file holds it).  The feature kernels read them as they are; only the optional metrics, whose kernels
// take fp32 planes, get a widened copy of raw input.
int enqueue_batch(vmaf_hip_engine *e, const void *in_ref, const void *in_dis, int sample, size_t row_stride_bytes,
                  size_t frame_stride_bytes, int nframes, int continue_sequence)
{
    if (row_stride_bytes % sample || frame_stride_bytes % sample) return fail("strides must be multiples of the sample size (%d bytes)", sample);
    if (row_stride_bytes < (size_t)e->w * sample) return fail("row stride %zu smaller than a row", row_stride_bytes);
    if (nframes > 1 && frame_stride_bytes < row_stride_bytes * e->h) return fail("frame stride smaller than a frame");
    if (frame_stride_bytes == 0) frame_stride_bytes = row_stride_bytes * e->h;       // one frame: any stride will do
    if (row_stride_bytes * e->h > 0xFFFFFFFFu) return fail("one plane must stay below 4 GiB");
    const int in_rs = (int)(row_stride_bytes / sample);
    const size_t in_fs = frame_stride_bytes / sample;
    const bool has_carry = continue_sequence && e->carry_valid;
    if (continue_sequence && !e->carry_valid) return fail("continue_sequence set but the engine holds no previous frame");
    const bool any_metric = e->m_psnr || e->m_ssim || e->m_ms || e->m_ansnr;

    const int slot = (e->pend_head + e->pend_count) % kDepth;
    e->partials = e->partials_set[slot];
    e->sums_dev = e->sums_set[slot];
    if (e->timed) HIP_TRY(hipEventRecord(e->ev_t0[slot], e->s_main));
    // fp32 view of the input for the metric kernels (dense planes); the features never use it
    const float *d_ref = static_cast<const float *>(in_ref), *d_dis = static_cast<const float *>(in_dis);
    int rs = in_rs;
    size_t fs = in_fs;
    if (any_metric && sample != 4) {
        if (ensure_stage(e, slot)) return 1;
        HIP_TRY(vk_launch_widen(in_ref, sample, row_stride_bytes, frame_stride_bytes, e->stage_ref[slot], e->w, e->h, nframes, e->s_main));
        HIP_TRY(vk_launch_widen(in_dis, sample, row_stride_bytes, frame_stride_bytes, e->stage_dis[slot], e->w, e->h, nframes, e->s_main));
        d_ref = e->stage_ref[slot]; d_dis = e->stage_dis[slot];
        rs = e->w; fs = (size_t)e->w * e->h;
    }
    HIP_TRY(hipEventRecord(e->ev_in, e->s_main));             // inputs (staging copies) are ready at this point
    // When the ADM chain and the motion launch start.  In the reference's orders the scale-0 VIF launch saturates the
    // VALUs on its own, so they start after it and share the GPU with VIF scales 1-3, which are too small to fill it
    // (+1.4 % throughput, and the dominant kernel runs undisturbed).  In the fused order that launch is 35 % shorter and
    // leaves issue slots: there they start at once (+2 %).  Measured with bench.py, both ways, both orders.
    const bool adm_early = e->order == 2;
    if (adm_early) {
        HIP_TRY(hipStreamWaitEvent(e->s_adm, e->ev_in, 0));
        HIP_TRY(hipStreamWaitEvent(e->s_mot, e->ev_in, 0));
    }

    // ---- optional metrics on their own stream (combo.c:168-230: before the -128 offset, i.e. on the planes as given)
    const bool psnr_fused = e->m_psnr && e->m_ms;
    if (any_metric) {
        hipStream_t q = e->s_iqa;
        HIP_TRY(hipStreamWaitEvent(q, e->ev_in, 0));
        // with MS-SSIM on, its full-resolution launch forms the PSNR noise sums from the samples it stages (one read less)
        if (e->m_psnr && !psnr_fused) HIP_TRY(vk_launch_psnr(d_ref, d_dis, fs, rs, e->w, e->h, nframes, e->partials + e->part_off_psnr, q));
        if (e->m_ansnr) HIP_TRY(vk_launch_ansnr(d_ref, d_dis, fs, rs, e->w, e->h, nframes, e->partials + e->part_off_ansnr, q));
        if (e->m_ssim) {
            const float *sr = d_ref, *sd = d_dis;
            if (e->ssim_f > 1) {
                HIP_TRY(vk_launch_box_decimate_pair(d_ref, d_dis, fs, rs, e->w, e->h, e->ssim_f, e->ssim_ref, e->ssim_dis, e->ssim_w, e->ssim_h, nframes, q));
                sr = e->ssim_ref; sd = e->ssim_dis;
            } else if (rs != e->w || fs != (size_t)e->w * e->h) {
                return fail("ssim at decimation factor 1 needs dense input planes");
            }
            HIP_TRY(vk_launch_ssim_stat(sr, sd, e->ssim_w, e->ssim_h, nframes, 0, e->m_exact, e->partials + e->part_off_ssim, nullptr, q));
        }
        if (e->m_ms) {
            if (rs != e->w || fs != (size_t)e->w * e->h) return fail("ms_ssim needs dense input planes");
            const float *cr0 = d_ref, *cd0 = d_dis;
            for (int i = 0; i < 5; ++i) {
                HIP_TRY(vk_launch_ssim_stat(cr0, cd0, e->ms_w[i], e->ms_h[i], nframes, i < 4 ? 1 : 0, e->m_exact, e->partials + e->part_off_ms[i],   // ms_ssim.c:62-64: alpha = 0 below the last level
                                            i == 0 && psnr_fused ? e->partials + e->part_off_psnr : nullptr, q));
                if (i < 4) {
                    HIP_TRY(vk_launch_lpf_decimate(cr0, cd0, (size_t)e->ms_w[i] * e->ms_h[i], e->ms_w[i], e->ms_w[i], e->ms_h[i],
                                                   e->ms_ref[i + 1], e->ms_dis[i + 1], e->ms_w[i + 1], e->ms_h[i + 1], nframes, e->m_exact, q));
                    cr0 = e->ms_ref[i + 1]; cd0 = e->ms_dis[i + 1];
                }
            }
        }
        HIP_TRY(hipEventRecord(e->ev_iqa, q));
    }

    // ---- VIF chain (vif.c:130-275) on s_main
    PlaneCursor vc = { in_ref, in_dis, sample, in_fs, in_rs };
    for (int sc = 0; sc < 4; ++sc) {
        if (e->timed && sc == 0) HIP_TRY(hipEventRecord(e->ev_k0[slot], e->s_main));
        if (launch_vif_scale(e, e, e->order, sc, &vc, nframes, e->s_main)) return 1;
        if (e->timed && sc == 0) HIP_TRY(hipEventRecord(e->ev_k1[slot], e->s_main));
        if (sc == 0 && !adm_early) {
            HIP_TRY(hipEventRecord(e->ev_in, e->s_main));
            HIP_TRY(hipStreamWaitEvent(e->s_adm, e->ev_in, 0));
            HIP_TRY(hipStreamWaitEvent(e->s_mot, e->ev_in, 0));
        }
    }

    // ---- ADM chain (adm.c:125-254) on s_adm
    PlaneCursor ac = { in_ref, in_dis, sample, in_fs, in_rs };
    for (int sc = 0; sc < 4; ++sc)
        if (launch_adm_scale(e, e, e->order, sc, &ac, nframes, e->s_adm)) return 1;
    HIP_TRY(hipEventRecord(e->ev_adm, e->s_adm));

    // ---- motion (combo.c:309-326) on s_mot
    HIP_TRY(vk_launch_motion(e->order, in_ref, sample, frame_stride_bytes, (int)row_stride_bytes, e->w, e->h, nframes, has_carry ? 1 : 0, e->carry[e->carry_cur], e->carry[1 - e->carry_cur],
                             e->partials + e->part_off_mot, e->s_mot));
    HIP_TRY(hipEventRecord(e->ev_mot, e->s_mot));
    e->carry_cur = 1 - e->carry_cur;
    e->carry_valid = true;

    // ---- join, reduce, copy back.  Small batches: on s_red, so that s_main goes straight on to the next batch's scale-0 VIF
    // launch while the short ADM tail of this batch (three launches that cannot fill the GPU) is still running: at 30 frames
    // of 1080p per batch 29.8k -> 31.2k frames/s, within 3 % of the rate of large batches.  Large batches amortise that tail
    // anyway (+1 % at 120 frames) and keep the join on s_main, where the dominant kernel runs undisturbed by the
    // previous batch (its launch stretches from 2.2 to 2.7 ms otherwise).
    const bool tail_overlap = (size_t)nframes * e->w * e->h < (size_t)64 * 1920 * 1080;
    hipStream_t s_join = tail_overlap ? e->s_red : e->s_main;
    if (s_join != e->s_main) {
        HIP_TRY(hipEventRecord(e->ev_vif, e->s_main));
        HIP_TRY(hipStreamWaitEvent(s_join, e->ev_vif, 0));
    }
    HIP_TRY(hipStreamWaitEvent(s_join, e->ev_adm, 0));
    HIP_TRY(hipStreamWaitEvent(s_join, e->ev_mot, 0));
    if (any_metric) HIP_TRY(hipStreamWaitEvent(s_join, e->ev_iqa, 0));
    vk_reduce_table tab;
    int q = fill_reduce_table(e, &tab, e->vif_nd != nullptr);
    tab.d[q++] = { e->part_off_psnr, e->m_psnr ? (psnr_fused ? e->tiles_ms[0] : e->tiles_psnr) : 0, 1, 0 };
    for (int c = 0; c < 4; ++c) tab.d[q++] = { e->part_off_ssim, e->m_ssim ? e->tiles_ssim : 0, 4, c };
    for (int i = 0; i < 5; ++i)
        for (int c = 0; c < 4; ++c) tab.d[q++] = { e->part_off_ms[i], e->m_ms ? e->tiles_ms[i] : 0, 4, c };
    for (int c = 0; c < 2; ++c) tab.d[q++] = { e->part_off_ansnr, e->m_ansnr ? e->tiles_ansnr : 0, 2, c };
    HIP_TRY(vk_launch_reduce(e->partials, &tab, nframes, e->sums_dev, s_join));
    HIP_TRY(hipMemcpyAsync(e->sums_host[slot], e->sums_dev, sizeof(double) * VK_NSUMS * nframes, hipMemcpyDeviceToHost, s_join));
    if (e->timed) HIP_TRY(hipEventRecord(e->ev_t1[slot], s_join));
    HIP_TRY(hipEventRecord(e->ev_done[slot], s_join));
    return 0;
}

int check_call(vmaf_hip_engine *e, const void *a, const void *b, int nframes, const double *out)
{
    if (!e) return fail("null engine");
    if (!a || !b || !out) return fail("null pointer argument");
    if (nframes <= 0 || nframes > e->max_batch) return fail("nframes %d outside 1..%d", nframes, e->max_batch);
    if (e->pend_count == kDepth) return fail("%d async extracts are already in flight: synchronise first", kDepth);
    HIP_TRY(hipSetDevice(e->device));
    return 0;
}

void push_pending(vmaf_hip_engine *e, int nframes, int continue_sequence, double *atoms_out)
{
    vmaf_hip_engine::Pending &p = e->pend[(e->pend_head + e->pend_count) % kDepth];
    p.frames = nframes;
    p.first_has_motion = continue_sequence != 0;
    p.out = atoms_out;
    ++e->pend_count;
}

// Wait for the oldest call in flight and finish its frames on the host.
int finish_oldest(vmaf_hip_engine *e)
{
    const int slot = e->pend_head;
    const vmaf_hip_engine::Pending &p = e->pend[slot];
    HIP_TRY(hipEventSynchronize(e->ev_done[slot]));
    const double *sums = e->sums_host[slot];
    for (int f = 0; f < p.frames; ++f)
        finish_atoms(e, sums + (size_t)f * VK_NSUMS, f > 0 || p.first_has_motion, p.out + (size_t)f * VMAF_HIP_ATOMS);
    if (e->m_psnr || e->m_ssim || e->m_ms || e->m_ansnr) {
        e->metrics.resize((size_t)p.frames * VMAF_HIP_METRICS);
        for (int f = 0; f < p.frames; ++f)
            finish_metrics(e, sums + (size_t)f * VK_NSUMS, e->metrics.data() + (size_t)f * VMAF_HIP_METRICS);
    }
    e->last_slot = slot;
    e->pend_head = (e->pend_head + 1) % kDepth;
    --e->pend_count;
    return 0;
}

}  // namespace

// ---- the entry point that allocates nothing (include/vmaf_hip.h: vmaf_hip_extract_device_noalloc) ----
// Its device memory beyond the caller's inputs - the VIF pyramid, the ADM LL bands, the blurred plane the motion kernel
// leaves behind, every kernel's per-workgroup partial sums and the reduced sums - lives in this file-scope array.
// In host code the NAME `scratch` is only the host-side stand-in of the symbol, never a device address: each GPU holds its
// own copy at its own address, looked up once per device with hipGetSymbolAddress and kept in g_noalloc_ptr[device].
static __device__ __align__(256) unsigned char scratch[VMAF_HIP_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_noalloc_mu;
unsigned char *g_noalloc_ptr[kMaxDevices] = {};
// one call at a time per GPU: the calls of a GPU share its scratch array, and two that overlapped would silently mix their
// pyramids and partial sums.  A second call while one is running is refused with a message (no allocation, no waiting)
std::atomic_flag g_noalloc_busy[kMaxDevices];
struct NoallocBusy {
    std::atomic_flag *f = nullptr;
    bool take(int device) { if (g_noalloc_busy[device].test_and_set(std::memory_order_acquire)) return false; f = &g_noalloc_busy[device]; return true; }
    ~NoallocBusy() { if (f) f->clear(std::memory_order_release); }
};

// device address of this GPU's copy of `scratch` (device must be current)
int noalloc_scratch(int device, unsigned char **out)
{
    std::lock_guard<std::mutex> lock(g_noalloc_mu);
    if (!g_noalloc_ptr[device]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail("hipGetSymbolAddress returned a null address for the static scratch area");
        g_noalloc_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_noalloc_ptr[device];
    return 0;
}

struct NoallocLayout { FeatureBuffers b; float *carry = nullptr; size_t bytes = 0; };

NoallocLayout noalloc_layout(const Geometry &g, int nframes, unsigned char *base)
{
    NoallocLayout L;
    size_t off = 0;
    const size_t B = (size_t)nframes;
    for (int s = 1; s < 4; ++s) {
        L.b.vif_ref[s] = carve<float>(base, &off, B * g.vw[s] * g.vh[s]);
        L.b.vif_dis[s] = carve<float>(base, &off, B * g.vw[s] * g.vh[s]);
        L.b.adm_ref[s] = carve<float>(base, &off, B * g.aw[s] * g.ah[s]);
        L.b.adm_dis[s] = carve<float>(base, &off, B * g.aw[s] * g.ah[s]);
    }
    L.carry = carve<float>(base, &off, (size_t)g.w * g.h);
    L.b.partials = carve<double>(base, &off, g.part_total);
    L.b.sums_dev = carve<double>(base, &off, B * VK_NSUMS);
    L.bytes = off;
    return L;
}
}  // namespace

extern "C" {

const char *vmaf_hip_last_error(void) { return g_err; }

int vmaf_hip_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vmaf_hip_engine_create(vmaf_hip_engine **out, int device, int width, int height, int max_batch)
{
    if (!out) return fail("null out pointer");
    *out = nullptr;
    // below 17 the mirrored taps of the coarsest ADM/VIF scales leave the plane (undefined in the reference too)
    if (width < 32 || height < 32) return fail("frame %dx%d too small (minimum 32x32)", width, height);
    if (width > 16384 || height > 16384) return fail("frame %dx%d too large", width, height);
    if (max_batch < 1 || max_batch > 4096) return fail("max_batch %d outside 1..4096", max_batch);
    int ndev = 0;
    trace("create: start");
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail("device %d outside 0..%d", device, ndev - 1);
    trace("create: device count known (runtime initialised)");
    HIP_TRY(hipSetDevice(device));
    trace("create: device set");
    // the kernels' code object goes to the device on a helper thread while the streams are created here (otherwise the
    // first launch loads it: ~10 ms in front of the first batch of a short run); joined on every way out
    struct Preload {
        std::thread t;
        explicit Preload(int dev) : t([dev] { if (hipSetDevice(dev) == hipSuccess) (void)vk_preload(); trace("create: code object loaded"); }) {}
        ~Preload() { t.join(); }
    } preload(device);

    vmaf_hip_engine *e = new (std::nothrow) vmaf_hip_engine();
    if (!e) return fail("out of host memory");
    e->device = device;
    const size_t B = (size_t)max_batch;

    geometry_init(e, width, height, max_batch);

    int rc = 0;
    auto cleanup = [&]() { vmaf_hip_engine_destroy(e); return 1; };
    // VMAF_HIP_SERIAL=1 puts the three chains on one stream (clean per-kernel timings when profiling)
    const char *ser = getenv("VMAF_HIP_SERIAL");
    e->serial = ser && *ser && *ser != '0';
    if (hipStreamCreateWithFlags(&e->s_main, hipStreamNonBlocking) != hipSuccess) { fail("hipStreamCreate failed"); return cleanup(); }
    trace("create: first stream");
    if (e->serial) { e->s_adm = e->s_main; e->s_mot = e->s_main; e->s_copy = e->s_main; e->s_red = e->s_main; }
    else if (hipStreamCreateWithFlags(&e->s_red, hipStreamNonBlocking) != hipSuccess ||
             hipStreamCreateWithFlags(&e->s_adm, hipStreamNonBlocking) != hipSuccess ||
             hipStreamCreateWithFlags(&e->s_mot, hipStreamNonBlocking) != hipSuccess ||
             hipStreamCreateWithFlags(&e->s_copy, hipStreamNonBlocking) != hipSuccess) { fail("hipStreamCreate failed"); return cleanup(); }
    trace("create: other streams");
    hipEvent_t *evs[] = { &e->ev_in, &e->ev_adm, &e->ev_mot, &e->ev_vif, &e->ev_copy[0], &e->ev_copy[1] };
    for (auto ev : evs)
        if (hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) { fail("hipEventCreate failed"); return cleanup(); }
    for (int i = 0; i < kDepth; ++i) {
        hipEvent_t *tevs[] = { &e->ev_t0[i], &e->ev_t1[i], &e->ev_k0[i], &e->ev_k1[i], &e->ev_done[i] };
        for (auto ev : tevs)
            if (hipEventCreate(ev) != hipSuccess) { fail("hipEventCreate failed"); return cleanup(); }
    }

    trace("create: streams and events");
    // one device allocation for everything whose size is fixed at creation (an allocation, and above all a free, costs a
    // device synchronisation: 21 separate ones made create + destroy ~8 ms of a short run)
    for (int pass = 0; pass < 2 && !rc; ++pass) {
        size_t off = 0;
        for (int s = 1; s < 4; ++s) {
            e->vif_ref[s] = carve<float>(e->arena, &off, B * e->vw[s] * e->vh[s]);
            e->vif_dis[s] = carve<float>(e->arena, &off, B * e->vw[s] * e->vh[s]);
            e->adm_ref[s] = carve<float>(e->arena, &off, B * e->aw[s] * e->ah[s]);
            e->adm_dis[s] = carve<float>(e->arena, &off, B * e->aw[s] * e->ah[s]);
        }
        for (int i = 0; i < 2; ++i) e->carry[i] = carve<float>(e->arena, &off, (size_t)width * height);
        for (int i = 0; i < kDepth; ++i) e->sums_set[i] = carve<double>(e->arena, &off, B * VK_NSUMS);
        if (pass == 0 && hipMalloc((void **)&e->arena, off) != hipSuccess) rc = fail("hipMalloc(engine buffers, %zu bytes) failed", off);
    }
    e->sums_dev = e->sums_set[0];
    const size_t off = e->part_total;
    for (int i = 0; i < kDepth && !rc; ++i)
        if (hipMalloc((void **)&e->partials_set[i], off * sizeof(double)) != hipSuccess) rc = fail("hipMalloc(partials) failed");
    e->partials = e->partials_set[0];
    if (!rc && hipHostMalloc((void **)&e->sums_host[0], kDepth * B * VK_NSUMS * sizeof(double), hipHostMallocDefault) != hipSuccess) rc = fail("hipHostMalloc failed");
    for (int i = 1; i < kDepth && !rc; ++i) e->sums_host[i] = e->sums_host[0] + (size_t)i * B * VK_NSUMS;
    if (rc) return cleanup();
    trace("create: device and host buffers");
    *out = e;
    return 0;
}

void vmaf_hip_engine_destroy(vmaf_hip_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(e->arena);
    for (int i = 0; i < kDepth; ++i) {
        (void)hipFree(e->stage_ref[i]); (void)hipFree(e->stage_dis[i]);
        (void)hipFree(e->raw_ref[i]); (void)hipFree(e->raw_dis[i]);
    }
    (void)hipFree(e->seed_buf);
    (void)hipFree(e->vif_nd); (void)hipFree(e->vif_rowsum);
    (void)hipFree(e->ssim_ref); (void)hipFree(e->ssim_dis);
    for (int i = 1; i < 5; ++i) { (void)hipFree(e->ms_ref[i]); (void)hipFree(e->ms_dis[i]); }
    if (e->ev_iqa) (void)hipEventDestroy(e->ev_iqa);
    if (e->s_iqa && e->s_iqa != e->s_main) (void)hipStreamDestroy(e->s_iqa);
    for (int i = 0; i < kDepth; ++i) (void)hipFree(e->partials_set[i]);
    hipEvent_t evs[] = { e->ev_in, e->ev_adm, e->ev_mot, e->ev_vif, e->ev_copy[0], e->ev_copy[1] };
    for (auto ev : evs) if (ev) (void)hipEventDestroy(ev);
    for (int i = 0; i < kDepth; ++i) {
        if (i == 0 && e->sums_host[0]) (void)hipHostFree(e->sums_host[0]);
        hipEvent_t tevs[] = { e->ev_t0[i], e->ev_t1[i], e->ev_k0[i], e->ev_k1[i], e->ev_done[i] };
        for (auto ev : tevs) if (ev) (void)hipEventDestroy(ev);
    }
    if (!e->serial) {
        if (e->s_adm) (void)hipStreamDestroy(e->s_adm);
        if (e->s_mot) (void)hipStreamDestroy(e->s_mot);
        if (e->s_copy) (void)hipStreamDestroy(e->s_copy);
        if (e->s_red) (void)hipStreamDestroy(e->s_red);
    }
    if (e->s_main) (void)hipStreamDestroy(e->s_main);
    delete e;
}

int vmaf_hip_engine_set_conv_order(vmaf_hip_engine *e, int order)
{
    if (!e) return fail("null engine");
    if (order < 0 || order > 2) return fail("conv order %d is none of 0 (scalar), 1 (avx), 2 (fused multiply-add)", order);
    e->order = order;
    return 0;
}

static int enable_metrics_impl(vmaf_hip_engine *e, int do_psnr, int do_ssim, int do_ms_ssim, int do_ansnr, int ten_bit);

int vmaf_hip_engine_enable_metrics(vmaf_hip_engine *e, int do_psnr, int do_ssim, int do_ms_ssim, int ten_bit)
{
    return enable_metrics_impl(e, do_psnr, do_ssim, do_ms_ssim, e ? e->m_ansnr : 0, ten_bit);
}

int vmaf_hip_engine_enable_ansnr(vmaf_hip_engine *e, int ten_bit)
{
    return enable_metrics_impl(e, e ? e->m_psnr : 0, e ? e->m_ssim : 0, e ? e->m_ms : 0, 1, ten_bit);
}

static int enable_metrics_impl(vmaf_hip_engine *e, int do_psnr, int do_ssim, int do_ms_ssim, int do_ansnr, int ten_bit)
{
    if (!e) return fail("null engine");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    HIP_TRY(hipSetDevice(e->device));
    const size_t B = (size_t)e->max_batch;
    if (do_ssim) {
        // ssim.c:86: scale = max(1, round(min(w, h) / 256)); iqa/decimate.c:40-41: sw = w / f + (w & 1)
        const float a = (float)std::min(e->w, e->h) / 256.0f;
        const int r = (a - (int)a >= 0.5) ? (int)a + (a > 0.0f ? 1 : -1) : (int)a;
        e->ssim_f = std::max(1, r);
        e->ssim_w = e->ssim_f > 1 ? e->w / e->ssim_f + (e->w & 1) : e->w;
        e->ssim_h = e->ssim_f > 1 ? e->h / e->ssim_f + (e->h & 1) : e->h;
        if (e->ssim_w < 11 || e->ssim_h < 11) return fail("frame too small for the 11x11 SSIM window after decimation by %d", e->ssim_f);
    }
    if (do_ms_ssim) {
        int cw = e->w, ch = e->h;
        for (int i = 0; i < 5; ++i) {                                  // ms_ssim.c:170-181
            if (cw < 11 || ch < 11) return fail("ms_ssim: scale below 1x1 (frame %dx%d too small for 5 scales)", e->w, e->h);
            cw /= 2; ch /= 2;
        }
        e->ms_w[0] = e->w; e->ms_h[0] = e->h;
        for (int i = 1; i < 5; ++i) { e->ms_w[i] = e->ms_w[i - 1] / 2 + (e->ms_w[i - 1] & 1); e->ms_h[i] = e->ms_h[i - 1] / 2 + (e->ms_h[i - 1] & 1); }
    }
    if (!e->s_iqa) {
        if (e->serial) e->s_iqa = e->s_main;
        else HIP_TRY(hipStreamCreateWithFlags(&e->s_iqa, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&e->ev_iqa, hipEventDisableTiming));
    }
    if (do_ssim && e->ssim_f > 1 && !e->ssim_ref) {
        if (alloc_planes(&e->ssim_ref, B * e->ssim_w * e->ssim_h) || alloc_planes(&e->ssim_dis, B * e->ssim_w * e->ssim_h)) return 1;
    }
    if (do_ms_ssim && !e->ms_ref[1]) {
        for (int i = 1; i < 5; ++i)
            if (alloc_planes(&e->ms_ref[i], B * e->ms_w[i] * e->ms_h[i]) || alloc_planes(&e->ms_dis[i], B * e->ms_w[i] * e->ms_h[i])) return 1;
    }
    // the partial-sum buffer grows by the metric sections
    size_t off = e->part_total;
    if (!e->tiles_psnr) {                                              // room for either producer of the PSNR partial sums
        e->tiles_psnr = vk_psnr_tiles(e->h); e->part_off_psnr = off; off += B * std::max(e->tiles_psnr, vk_ssim_tiles(e->w, e->h));
    }
    if (do_ansnr && !e->tiles_ansnr) { e->tiles_ansnr = vk_ansnr_tiles(e->w, e->h); e->part_off_ansnr = off; off += B * e->tiles_ansnr * 2; }
    if (do_ssim && !e->tiles_ssim) { e->tiles_ssim = vk_ssim_tiles(e->ssim_w, e->ssim_h); e->part_off_ssim = off; off += B * e->tiles_ssim * 4; }
    if (do_ms_ssim && !e->tiles_ms[0])
        for (int i = 0; i < 5; ++i) { e->tiles_ms[i] = vk_ssim_tiles(e->ms_w[i], e->ms_h[i]); e->part_off_ms[i] = off; off += B * e->tiles_ms[i] * 4; }
    if (off != e->part_total) {
        HIP_TRY(hipStreamSynchronize(e->s_main));
        HIP_TRY(hipStreamSynchronize(e->s_red));
        for (int i = 0; i < kDepth; ++i) {
            (void)hipFree(e->partials_set[i]);
            e->partials_set[i] = nullptr;
            HIP_TRY(hipMalloc((void **)&e->partials_set[i], off * sizeof(double)));
        }
        e->partials = e->partials_set[0];
        e->part_total = off;
    }
    e->m_psnr = do_psnr != 0; e->m_ssim = do_ssim != 0; e->m_ms = do_ms_ssim != 0; e->m_ansnr = do_ansnr != 0;
    e->psnr_peak = ten_bit ? 255.75 : 255.0;                           // combo.c:171-186
    e->psnr_max = ten_bit ? 72.0 : 60.0;
    return 0;
}

int vmaf_hip_engine_set_metrics_exact(vmaf_hip_engine *e, int exact)
{
    if (!e) return fail("null engine");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    e->m_exact = exact ? 1 : 0;
    return 0;
}

int vmaf_hip_engine_set_sums_exact(vmaf_hip_engine *e, int exact)
{
    if (!e) return fail("null engine");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    HIP_TRY(hipSetDevice(e->device));
    if (exact && !e->vif_nd) {
        const size_t B = (size_t)e->max_batch;
        if (hipMalloc((void **)&e->vif_nd, B * 2 * e->w * e->h * sizeof(float)) != hipSuccess) { e->vif_nd = nullptr; return fail("hipMalloc(num / den planes) failed"); }
        if (hipMalloc((void **)&e->vif_rowsum, B * 2 * e->h * sizeof(float)) != hipSuccess) {
            (void)hipFree(e->vif_nd); e->vif_nd = nullptr; e->vif_rowsum = nullptr;
            return fail("hipMalloc(row sums) failed");
        }
    } else if (!exact && e->vif_nd) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(e->vif_nd); (void)hipFree(e->vif_rowsum);
        e->vif_nd = nullptr; e->vif_rowsum = nullptr;
    }
    return 0;
}

int vmaf_hip_engine_get_metrics(vmaf_hip_engine *e, double *out, int nframes)
{
    if (!e || !out) return fail("null argument");
    if (!(e->m_psnr || e->m_ssim || e->m_ms || e->m_ansnr)) return fail("no metric enabled on this engine");
    if (nframes < 0 || (size_t)nframes * VMAF_HIP_METRICS > e->metrics.size()) return fail("the last call scored %zu frames, %d asked for", e->metrics.size() / VMAF_HIP_METRICS, nframes);
    std::copy(e->metrics.begin(), e->metrics.begin() + (size_t)nframes * VMAF_HIP_METRICS, out);
    return 0;
}

int vmaf_hip_engine_sync(vmaf_hip_engine *e)
{
    if (!e) return fail("null engine");
    HIP_TRY(hipSetDevice(e->device));
    while (e->pend_count)
        if (finish_oldest(e)) return 1;
    HIP_TRY(hipStreamSynchronize(e->s_main));
    HIP_TRY(hipStreamSynchronize(e->s_red));
    return 0;
}

int vmaf_hip_engine_debug_read_plane(vmaf_hip_engine *e, int kind, int level, int frame, float *out, size_t out_floats,
                                     int *width, int *height)
{
    if (!e || !out) return fail("null argument");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    if (kind < 0 || kind > 3 || level < 1 || level > 3) return fail("kind %d / level %d outside 0..3 / 1..3", kind, level);
    if (frame < 0 || frame >= e->max_batch) return fail("frame %d outside the batch", frame);
    const bool vif = kind < 2;
    const int w = vif ? e->vw[level] : e->aw[level], h = vif ? e->vh[level] : e->ah[level];
    const float *src = (kind == 0 ? e->vif_ref : kind == 1 ? e->vif_dis : kind == 2 ? e->adm_ref : e->adm_dis)[level];
    if ((size_t)w * h > out_floats) return fail("plane %dx%d does not fit %zu floats", w, h, out_floats);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy(out, src + (size_t)frame * w * h, sizeof(float) * w * h, hipMemcpyDeviceToHost));
    if (width) *width = w;
    if (height) *height = h;
    return 0;
}

int vmaf_hip_engine_debug_read_sums(vmaf_hip_engine *e, int frame, double *out)
{
    if (!e || !out) return fail("null argument");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    if (frame < 0 || frame >= e->max_batch) return fail("frame %d outside the batch", frame);
    std::copy(e->sums_host[e->last_slot] + (size_t)frame * VK_NSUMS, e->sums_host[e->last_slot] + (size_t)frame * VK_NSUMS + VMAF_HIP_DEBUG_SUMS, out);
    return 0;
}

int vmaf_hip_engine_sync_oldest(vmaf_hip_engine *e)
{
    if (!e) return fail("null engine");
    if (!e->pend_count) return fail("no async extract in flight");
    HIP_TRY(hipSetDevice(e->device));
    return finish_oldest(e);
}

int vmaf_hip_extract_device_async(vmaf_hip_engine *e, const float *d_ref, const float *d_dis, size_t row_stride_bytes,
                                  size_t frame_stride_bytes, int nframes, int continue_sequence, double *atoms_out)
{
    if (check_call(e, d_ref, d_dis, nframes, atoms_out)) return 1;
    if (enqueue_batch(e, d_ref, d_dis, 4, row_stride_bytes, frame_stride_bytes, nframes, continue_sequence)) return 1;
    push_pending(e, nframes, continue_sequence, atoms_out);
    return 0;
}

int vmaf_hip_extract_device(vmaf_hip_engine *e, const float *d_ref, const float *d_dis, size_t row_stride_bytes,
                            size_t frame_stride_bytes, int nframes, int continue_sequence, double *atoms_out)
{
    if (vmaf_hip_extract_device_async(e, d_ref, d_dis, row_stride_bytes, frame_stride_bytes, nframes, continue_sequence, atoms_out))
        return 1;
    return vmaf_hip_engine_sync(e);
}

// The slot the next asynchronous call takes.  Its staging buffers are free: the call that used them last (two calls ago)
// has been synchronised by the caller, or check_call would have refused this one.
static int next_slot(const vmaf_hip_engine *e) { return (e->pend_head + e->pend_count) % kDepth; }

// Called after the H2D copies of a host-pointer call were put on s_copy: the kernels on s_main wait for them, and only
// for them - the copies of the next call overlap these kernels.
static int copies_done(vmaf_hip_engine *e, int slot)
{
    if (e->s_copy == e->s_main) return 0;
    HIP_TRY(hipEventRecord(e->ev_copy[slot], e->s_copy));
    HIP_TRY(hipStreamWaitEvent(e->s_main, e->ev_copy[slot], 0));
    return 0;
}

int vmaf_hip_extract_host_async(vmaf_hip_engine *e, const float *h_ref, const float *h_dis, size_t row_stride_bytes,
                          size_t frame_stride_bytes, int nframes, int continue_sequence, double *atoms_out)
{
    if (check_call(e, h_ref, h_dis, nframes, atoms_out)) return 1;
    if (row_stride_bytes < (size_t)e->w * 4) return fail("row stride %zu smaller than a row", row_stride_bytes);
    const int slot = next_slot(e);
    if (ensure_stage(e, slot)) return 1;
    // stage densely (row stride = w floats) so the kernels see one geometry
    const size_t dense_row = (size_t)e->w * 4, dense_frame = dense_row * e->h;
    if (row_stride_bytes == dense_row && (nframes == 1 || frame_stride_bytes == dense_frame)) {
        HIP_TRY(hipMemcpyAsync(e->stage_ref[slot], h_ref, dense_frame * nframes, hipMemcpyHostToDevice, e->s_copy));
        HIP_TRY(hipMemcpyAsync(e->stage_dis[slot], h_dis, dense_frame * nframes, hipMemcpyHostToDevice, e->s_copy));
    } else {
        for (int f = 0; f < nframes; ++f) {
            const char *sr = (const char *)h_ref + (size_t)f * frame_stride_bytes, *sd = (const char *)h_dis + (size_t)f * frame_stride_bytes;
            HIP_TRY(hipMemcpy2DAsync((char *)e->stage_ref[slot] + f * dense_frame, dense_row, sr, row_stride_bytes, dense_row, e->h,
                                     hipMemcpyHostToDevice, e->s_copy));
            HIP_TRY(hipMemcpy2DAsync((char *)e->stage_dis[slot] + f * dense_frame, dense_row, sd, row_stride_bytes, dense_row, e->h,
                                     hipMemcpyHostToDevice, e->s_copy));
        }
    }
    if (copies_done(e, slot)) return 1;
    if (enqueue_batch(e, e->stage_ref[slot], e->stage_dis[slot], 4, dense_row, dense_frame, nframes, continue_sequence)) return 1;
    push_pending(e, nframes, continue_sequence, atoms_out);
    return 0;
}

int vmaf_hip_extract_host(vmaf_hip_engine *e, const float *h_ref, const float *h_dis, size_t row_stride_bytes,
                          size_t frame_stride_bytes, int nframes, int continue_sequence, double *atoms_out)
{
    if (vmaf_hip_extract_host_async(e, h_ref, h_dis, row_stride_bytes, frame_stride_bytes, nframes, continue_sequence, atoms_out))
        return 1;
    return vmaf_hip_engine_sync(e);
}

// ---- raw 8/16-bit luma in, widened on the device (file_io.c:86-172) ----

static int check_raw(vmaf_hip_engine *e, int bytes_per_sample, size_t row_stride_bytes, size_t frame_stride_bytes, int nframes)
{
    if (bytes_per_sample != 1 && bytes_per_sample != 2) return fail("bytes_per_sample %d is neither 1 nor 2", bytes_per_sample);
    if (row_stride_bytes < (size_t)e->w * bytes_per_sample) return fail("row stride %zu smaller than a row", row_stride_bytes);
    if (row_stride_bytes % bytes_per_sample || frame_stride_bytes % bytes_per_sample) return fail("strides must be multiples of the sample size");
    if (nframes > 1 && frame_stride_bytes < row_stride_bytes * e->h) return fail("frame stride smaller than a frame");
    return 0;
}

// fp32 staging set of a pending slot, allocated on first use: the fp32 host path copies into it, and raw input is widened
// into it when the optional metrics (whose kernels take fp32 planes) are on
namespace {
int ensure_stage(vmaf_hip_engine *e, int slot)
{
    if (e->stage_ref[slot]) return 0;
    const size_t floats = (size_t)e->max_batch * e->w * e->h;
    return alloc_planes(&e->stage_ref[slot], floats) || alloc_planes(&e->stage_dis[slot], floats);
}
}  // namespace

int vmaf_hip_extract_device_raw_async(vmaf_hip_engine *e, const void *d_ref, const void *d_dis, int bytes_per_sample,
                                      size_t row_stride_bytes, size_t frame_stride_bytes, int nframes, int continue_sequence,
                                      double *atoms_out)
{
    if (check_call(e, d_ref, d_dis, nframes, atoms_out)) return 1;
    if (check_raw(e, bytes_per_sample, row_stride_bytes, frame_stride_bytes, nframes)) return 1;
    if (enqueue_batch(e, d_ref, d_dis, bytes_per_sample, row_stride_bytes, frame_stride_bytes, nframes, continue_sequence)) return 1;
    push_pending(e, nframes, continue_sequence, atoms_out);
    return 0;
}

int vmaf_hip_extract_host_raw_async(vmaf_hip_engine *e, const void *h_ref, const void *h_dis, int bytes_per_sample,
                                    size_t row_stride_bytes, size_t frame_stride_bytes, int nframes, int continue_sequence,
                                    double *atoms_out)
{
    if (check_call(e, h_ref, h_dis, nframes, atoms_out)) return 1;
    if (check_raw(e, bytes_per_sample, row_stride_bytes, frame_stride_bytes, nframes)) return 1;
    const size_t dense_row = (size_t)e->w * bytes_per_sample, dense_frame = dense_row * e->h;
    const int slot = next_slot(e);
    if (!e->raw_ref[slot]) {
        const size_t bytes = (size_t)e->max_batch * e->w * e->h * 2;
        if (hipMalloc((void **)&e->raw_ref[slot], bytes) != hipSuccess || hipMalloc((void **)&e->raw_dis[slot], bytes) != hipSuccess)
            return fail("hipMalloc(raw luma staging, 2 x %zu bytes) failed", bytes);
    }
    if (row_stride_bytes == dense_row && (nframes == 1 || frame_stride_bytes == dense_frame)) {
        HIP_TRY(hipMemcpyAsync(e->raw_ref[slot], h_ref, dense_frame * nframes, hipMemcpyHostToDevice, e->s_copy));
        HIP_TRY(hipMemcpyAsync(e->raw_dis[slot], h_dis, dense_frame * nframes, hipMemcpyHostToDevice, e->s_copy));
    } else {
        for (int f = 0; f < nframes; ++f) {
            const char *sr = (const char *)h_ref + (size_t)f * frame_stride_bytes, *sd = (const char *)h_dis + (size_t)f * frame_stride_bytes;
            HIP_TRY(hipMemcpy2DAsync(e->raw_ref[slot] + f * dense_frame, dense_row, sr, row_stride_bytes, dense_row, e->h, hipMemcpyHostToDevice, e->s_copy));
            HIP_TRY(hipMemcpy2DAsync(e->raw_dis[slot] + f * dense_frame, dense_row, sd, row_stride_bytes, dense_row, e->h, hipMemcpyHostToDevice, e->s_copy));
        }
    }
    if (copies_done(e, slot)) return 1;
    if (enqueue_batch(e, e->raw_ref[slot], e->raw_dis[slot], bytes_per_sample, dense_row, dense_frame, nframes, continue_sequence)) return 1;
    push_pending(e, nframes, continue_sequence, atoms_out);
    return 0;
}

int vmaf_hip_extract_host_raw(vmaf_hip_engine *e, const void *h_ref, const void *h_dis, int bytes_per_sample, size_t row_stride_bytes,
                              size_t frame_stride_bytes, int nframes, int continue_sequence, double *atoms_out)
{
    if (vmaf_hip_extract_host_raw_async(e, h_ref, h_dis, bytes_per_sample, row_stride_bytes, frame_stride_bytes, nframes,
                                        continue_sequence, atoms_out))
        return 1;
    return vmaf_hip_engine_sync(e);
}

int vmaf_hip_extract_device_raw(vmaf_hip_engine *e, const void *d_ref, const void *d_dis, int bytes_per_sample, size_t row_stride_bytes,
                                size_t frame_stride_bytes, int nframes, int continue_sequence, double *atoms_out)
{
    if (vmaf_hip_extract_device_raw_async(e, d_ref, d_dis, bytes_per_sample, row_stride_bytes, frame_stride_bytes, nframes,
                                          continue_sequence, atoms_out))
        return 1;
    return vmaf_hip_engine_sync(e);
}

// One-frame motion launch without predecessor: only its carry_out side effect is used (the blurred plane the next
// extract call with continue_sequence = 1 compares its first frame with).
static int motion_seed(vmaf_hip_engine *e, const void *d_ref, int sample, size_t row_stride_bytes)
{
    HIP_TRY(vk_launch_motion(e->order, d_ref, sample, 0, (int)row_stride_bytes, e->w, e->h, 1, 0, e->carry[e->carry_cur],
                             e->carry[1 - e->carry_cur], e->partials + e->part_off_mot, e->s_main));
    e->carry_cur = 1 - e->carry_cur;
    e->carry_valid = true;
    HIP_TRY(hipStreamSynchronize(e->s_main));
    return 0;
}

// a host frame lands in a one-plane device buffer (allocated on first use) and is blurred from there
static int motion_seed_from_host(vmaf_hip_engine *e, const void *h_ref, int sample, size_t row_stride_bytes)
{
    HIP_TRY(hipSetDevice(e->device));
    if (!e->seed_buf && hipMalloc(&e->seed_buf, (size_t)e->w * e->h * 4) != hipSuccess) return fail("hipMalloc(motion seed plane) failed");
    const size_t dense_row = (size_t)e->w * sample;
    HIP_TRY(hipMemcpy2DAsync(e->seed_buf, dense_row, h_ref, row_stride_bytes, dense_row, e->h, hipMemcpyHostToDevice, e->s_main));
    return motion_seed(e, e->seed_buf, sample, dense_row);
}

int vmaf_hip_motion_seed_host_raw(vmaf_hip_engine *e, const void *h_ref, int bytes_per_sample, size_t row_stride_bytes)
{
    if (!e || !h_ref) return fail("null argument");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    if (check_raw(e, bytes_per_sample, row_stride_bytes, 0, 1)) return 1;
    return motion_seed_from_host(e, h_ref, bytes_per_sample, row_stride_bytes);
}

void *vmaf_hip_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { fail("hipHostMalloc(%zu) failed", bytes); return nullptr; }
    return p;
}

void vmaf_hip_host_free(void *p) { if (p) (void)hipHostFree(p); }

int vmaf_hip_motion_seed_device(vmaf_hip_engine *e, const float *d_ref, size_t row_stride_bytes)
{
    if (!e || !d_ref) return fail("null argument");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    if (row_stride_bytes % 4 || row_stride_bytes < (size_t)e->w * 4) return fail("bad row stride %zu", row_stride_bytes);
    HIP_TRY(hipSetDevice(e->device));
    return motion_seed(e, d_ref, 4, row_stride_bytes);
}

int vmaf_hip_motion_seed_device_raw(vmaf_hip_engine *e, const void *d_ref, int bytes_per_sample, size_t row_stride_bytes)
{
    if (!e || !d_ref) return fail("null argument");
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    if (check_raw(e, bytes_per_sample, row_stride_bytes, 0, 1)) return 1;
    HIP_TRY(hipSetDevice(e->device));
    return motion_seed(e, d_ref, bytes_per_sample, row_stride_bytes);
}

int vmaf_hip_motion_seed_host(vmaf_hip_engine *e, const float *h_ref, size_t row_stride_bytes)
{
    if (!e || !h_ref) return fail("null argument");
    if (row_stride_bytes < (size_t)e->w * 4) return fail("bad row stride %zu", row_stride_bytes);
    if (e->pend_count) return fail("a previous async extract has not been synchronised");
    return motion_seed_from_host(e, h_ref, 4, row_stride_bytes);
}

// Device side of the SVR, kept per GPU between calls: the model (support vectors and coefficients, ~12 KB for vmaf_v0.6.1)
// is uploaded when it differs from the one already there, the feature / score buffers only grow.  A caller that fuses
// batch after batch (bench.py, a serving loop) pays two small copies and one launch per call instead of four hipMalloc /
// hipFree pairs.  One mutex per GPU: calls on one GPU are serialised, calls on different GPUs are independent.
namespace {
struct SvmDevice {
    std::mutex mu;
    std::vector<double> sv, coef;                                  // host copy of what d_sv / d_coef hold
    double *d_sv = nullptr, *d_coef = nullptr, *d_x = nullptr, *d_out = nullptr;
    size_t cap_n = 0, cap_dim = 0;                                 // d_x holds cap_n * cap_dim doubles, d_out cap_n
};
SvmDevice g_svm[kMaxDevices];
}  // namespace

int vmaf_hip_svm_predict(int device, const double *sv, const double *coef, int n_sv, int dim, double gamma, double rho,
                         const double *x, int n, double *out)
{
    if (!sv || !coef || !x || !out) return fail("null pointer argument");
    if (n_sv <= 0 || dim <= 0 || n < 0) return fail("bad sizes n_sv=%d dim=%d n=%d", n_sv, dim, n);
    if (n == 0) return 0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev || device >= kMaxDevices) return fail("device %d outside 0..%d", device, std::min(ndev, kMaxDevices) - 1);
    HIP_TRY(hipSetDevice(device));
    SvmDevice &c = g_svm[device];
    std::lock_guard<std::mutex> lock(c.mu);
    const size_t nsv = (size_t)n_sv * dim;
    if (c.sv.size() != nsv || c.coef.size() != (size_t)n_sv || !std::equal(sv, sv + nsv, c.sv.begin()) || !std::equal(coef, coef + n_sv, c.coef.begin())) {
        (void)hipFree(c.d_sv); (void)hipFree(c.d_coef);
        c.d_sv = c.d_coef = nullptr;
        c.sv.clear(); c.coef.clear();
        HIP_TRY(hipMalloc((void **)&c.d_sv, sizeof(double) * nsv));
        HIP_TRY(hipMalloc((void **)&c.d_coef, sizeof(double) * n_sv));
        HIP_TRY(hipMemcpy(c.d_sv, sv, sizeof(double) * nsv, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c.d_coef, coef, sizeof(double) * n_sv, hipMemcpyHostToDevice));
        c.sv.assign(sv, sv + nsv);
        c.coef.assign(coef, coef + n_sv);
    }
    if ((size_t)n * dim > c.cap_n * c.cap_dim || (size_t)n > c.cap_n) {
        (void)hipFree(c.d_x); (void)hipFree(c.d_out);
        c.d_x = c.d_out = nullptr;
        c.cap_n = c.cap_dim = 0;
        const size_t cap = std::max<size_t>((size_t)n, 1024);
        HIP_TRY(hipMalloc((void **)&c.d_x, sizeof(double) * cap * dim));
        HIP_TRY(hipMalloc((void **)&c.d_out, sizeof(double) * cap));
        c.cap_n = cap; c.cap_dim = (size_t)dim;
    }
    HIP_TRY(hipMemcpy(c.d_x, x, sizeof(double) * (size_t)n * dim, hipMemcpyHostToDevice));
    HIP_TRY(vk_launch_svm(c.d_sv, c.d_coef, n_sv, dim, gamma, rho, c.d_x, n, c.d_out, nullptr));
    HIP_TRY(hipMemcpy(out, c.d_out, sizeof(double) * n, hipMemcpyDeviceToHost));
    return 0;
}

// ---- the entry point that allocates nothing (scratch area and layout: above, before the extern "C" block) ----
size_t vmaf_hip_noalloc_scratch_needed(int width, int height, int nframes)
{
    if (width < 32 || height < 32 || width > 16384 || height > 16384 || nframes < 1 || nframes > 4096) return 0;
    Geometry g;
    geometry_init(&g, width, height, nframes);
    return noalloc_layout(g, nframes, nullptr).bytes;
}

int vmaf_hip_extract_device_noalloc(int device, void *stream, const float *d_ref, const float *d_dis, int width, int height,
                                    size_t row_stride_bytes, size_t frame_stride_bytes, int nframes, int conv_order, double *atoms_out)
{
    if (!d_ref || !d_dis || !atoms_out) return fail("null pointer argument");
    if (width < 32 || height < 32) return fail("frame %dx%d too small (minimum 32x32)", width, height);
    if (width > 16384 || height > 16384) return fail("frame %dx%d too large", width, height);
    if (nframes < 1 || nframes > 4096) return fail("nframes %d outside 1..4096", nframes);
    if (conv_order < 0 || conv_order > 2) return fail("conv order %d is none of 0 (scalar), 1 (avx), 2 (fused multiply-add)", conv_order);
    if (row_stride_bytes % 4 || frame_stride_bytes % 4) return fail("strides must be multiples of 4 bytes");
    if (row_stride_bytes < (size_t)width * 4) return fail("row stride %zu smaller than a row", row_stride_bytes);
    if (nframes > 1 && frame_stride_bytes < row_stride_bytes * height) return fail("frame stride smaller than a frame");
    if (device < 0 || device >= kMaxDevices) return fail("device %d outside 0..%d", device, kMaxDevices - 1);
    Geometry g;
    geometry_init(&g, width, height, nframes);
    const size_t need = noalloc_layout(g, nframes, nullptr).bytes;
    if (need > (size_t)VMAF_HIP_NOALLOC_SCRATCH_BYTES)
        return fail("%d frames of %dx%d need %zu bytes of scratch, the static area holds %zu: pass fewer frames per call or use an engine",
                    nframes, width, height, need, (size_t)VMAF_HIP_NOALLOC_SCRATCH_BYTES);
    NoallocBusy busy;
    if (!busy.take(device))
        return fail("vmaf_hip_extract_device_noalloc: another call is using GPU %d's static scratch area (calls on one GPU must not overlap)", device);
    HIP_TRY(hipSetDevice(device));
    unsigned char *base = nullptr;
    if (noalloc_scratch(device, &base)) return 1;
    const NoallocLayout L = noalloc_layout(g, nframes, base);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int rs = (int)(row_stride_bytes / 4);
    const size_t fs = frame_stride_bytes / 4;

    // the engine's launches, one after the other on the caller's stream.  Nothing in the scratch area has to start at
    // zero: every workgroup stores its partial sums, reduce_kernel stores every sum, and the pyramid planes are written
    // before they are read - the previous call's contents are simply overwritten.
    PlaneCursor vc = { d_ref, d_dis, 4, fs, rs };
    for (int sc = 0; sc < 4; ++sc)
        if (launch_vif_scale(&g, &L.b, conv_order, sc, &vc, nframes, s)) return 1;
    PlaneCursor ac = { d_ref, d_dis, 4, fs, rs };
    for (int sc = 0; sc < 4; ++sc)
        if (launch_adm_scale(&g, &L.b, conv_order, sc, &ac, nframes, s)) return 1;
    HIP_TRY(vk_launch_motion(conv_order, d_ref, 4, fs * 4, rs * 4, width, height, nframes, 0, L.carry, L.carry, L.b.partials + g.part_off_mot, s));
    vk_reduce_table tab;
    int q = fill_reduce_table(&g, &tab, false);
    for (; q < VK_NSUMS; ++q) tab.d[q] = { 0, 0, 1, 0 };          // optional metrics: not computed here
    HIP_TRY(vk_launch_reduce(L.b.partials, &tab, nframes, L.b.sums_dev, s));

    // sums -> atoms through a small stack buffer (synchronous copies: atoms_out is complete on return)
    constexpr int kChunk = 16;
    double sums[kChunk * VK_NSUMS];
    for (int f0 = 0; f0 < nframes; f0 += kChunk) {
        const int n = std::min(kChunk, nframes - f0);
        HIP_TRY(hipMemcpyAsync(sums, L.b.sums_dev + (size_t)f0 * VK_NSUMS, sizeof(double) * VK_NSUMS * n, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int f = 0; f < n; ++f)
            finish_atoms(&g, sums + (size_t)f * VK_NSUMS, f0 + f > 0, atoms_out + (size_t)(f0 + f) * VMAF_HIP_ATOMS);
    }
    return 0;
}

double vmaf_hip_last_kernel_ms(vmaf_hip_engine *e, int which)
{
    if (!e) return -1.0;
    if (!e->timed) { e->timed = true; return -1.0; }          // first call arms the events for the next batch
    float ms = -1.0f;
    hipError_t err = which == 1 ? hipEventElapsedTime(&ms, e->ev_k0[e->last_slot], e->ev_k1[e->last_slot]) : hipEventElapsedTime(&ms, e->ev_t0[e->last_slot], e->ev_t1[e->last_slot]);
    return err == hipSuccess ? (double)ms : -1.0;
}

}  // extern "C"
