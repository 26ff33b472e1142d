/*
 * iqa_oracle.c - CPU restatement of the reference's optional per-frame metrics PSNR, SSIM and MS-SSIM.
 *
 * TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the
 * checker; never linked into the product (vmaf_amd/).  Pinned against the reference's own sources compiled into
 * oracle/_ref/libvmaf_ref_feature.so (tests/test_oracle_vs_ref.py) and against tests/golden/exec.json.
 *
 * Follows, in the reference (christosbampis/vmaf):
 *   feature/src/psnr.c:37-61                    compute_psnr
 *   feature/src/ssim.c:52-182                   compute_ssim (scale factor, box low-pass decimation, Gaussian window)
 *   feature/src/ms_ssim.c:118-326               compute_ms_ssim (5 scales, 9/7 low-pass decimation by 2, Wang exponents)
 *   feature/src/iqa/ssim_tools.c:92-243         _iqa_ssim, default-argument path (l, c, s means)
 *   feature/src/iqa/convolve.c:85-150           _iqa_convolve, IQA_CONVOLVE_1D path ("valid" separable, double sums)
 *   feature/src/iqa/convolve.c:219-268          _iqa_filter_pixel (2-D kernel, KBND_SYMMETRIC at the edges)
 *   feature/src/iqa/decimate.c:37-59            _iqa_decimate (output size w/f + (w&1))
 *   feature/src/iqa/math_utils.c:36-50          _round, _max, _min
 * Arithmetic types are kept expression by expression (float products, double running sums, float stores).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "vmaf_oracle.h"

#define GAUSSIAN_LEN 11
static const float g_gauss[GAUSSIAN_LEN] = {                       /* ssim_tools.h: g_gaussian_window_h == _v */
    0.001028f, 0.007599f, 0.036001f, 0.109361f, 0.213006f, 0.266012f, 0.213006f, 0.109361f, 0.036001f, 0.007599f, 0.001028f
};
#define LPF_LEN 9
static const float g_lpf[LPF_LEN][LPF_LEN] = {                     /* ms_ssim.c:41-51, the 2-D table _iqa_decimate uses */
   { 0.000714f,-0.000450f,-0.002090f, 0.007132f, 0.016114f, 0.007132f,-0.002090f,-0.000450f, 0.000714f},
   {-0.000450f, 0.000283f, 0.001316f,-0.004490f,-0.010146f,-0.004490f, 0.001316f, 0.000283f,-0.000450f},
   {-0.002090f, 0.001316f, 0.006115f,-0.020867f,-0.047149f,-0.020867f, 0.006115f, 0.001316f,-0.002090f},
   { 0.007132f,-0.004490f,-0.020867f, 0.071207f, 0.160885f, 0.071207f,-0.020867f,-0.004490f, 0.007132f},
   { 0.016114f,-0.010146f,-0.047149f, 0.160885f, 0.363505f, 0.160885f,-0.047149f,-0.010146f, 0.016114f},
   { 0.007132f,-0.004490f,-0.020867f, 0.071207f, 0.160885f, 0.071207f,-0.020867f,-0.004490f, 0.007132f},
   {-0.002090f, 0.001316f, 0.006115f,-0.020867f,-0.047149f,-0.020867f, 0.006115f, 0.001316f,-0.002090f},
   {-0.000450f, 0.000283f, 0.001316f,-0.004490f,-0.010146f,-0.004490f, 0.001316f, 0.000283f,-0.000450f},
   { 0.000714f,-0.000450f,-0.002090f, 0.007132f, 0.016114f, 0.007132f,-0.002090f,-0.000450f, 0.000714f},
};
static const float g_alphas[5] = { 0.0000f, 0.0000f, 0.0000f, 0.0000f, 0.1333f };   /* ms_ssim.c:62-64 */
static const float g_betas[5]  = { 0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f };
static const float g_gammas[5] = { 0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f };

const float *oracle_ssim_window(void) { return g_gauss; }
const float *oracle_ms_ssim_lpf(void) { return &g_lpf[0][0]; }

/* psnr.c:37-61; peak 255 / cap 60 for 8-bit formats, 255.75 / 72 for the 10-bit ones (combo.c:171-186) */
double oracle_psnr(const float *ref, const float *dis, int w, int h, int stride, double peak, double psnr_max)
{
    double noise = 0;
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            const float diff = ref[i * stride + j] - dis[i * stride + j];
            noise += diff * diff;                                  /* float product, double sum */
        }
    noise /= (w * h);
    const double eps = 1e-10;
    const double v = 10 * log10(peak * peak / (noise > eps ? noise : eps));
    return v < psnr_max ? v : psnr_max;
}

/* convolve.c:35-42 */
static float sym_pixel(const float *img, int w, int h, int x, int y)
{
    if (x < 0) x = -1 - x; else if (x >= w) x = (w - (x - w)) - 1;
    if (y < 0) y = -1 - y; else if (y >= h) y = (h - (y - h)) - 1;
    return img[y * w + x];
}

/* convolve.c:219-268 with a kw x kh 2-D kernel, scale factor kscale */
static float filter_pixel(const float *img, int w, int h, int x, int y, const float *k, int kw, int kh, float kscale)
{
    const int uc = kw / 2, vc = kh / 2, kw_even = (kw & 1) ? 0 : 1, kh_even = (kh & 1) ? 0 : 1;
    const int edge = x < uc || y < vc || x >= w - uc || y >= h - vc;
    double sum = 0.0;
    int ko = 0;
    for (int v = -vc; v <= vc - kh_even; ++v)
        for (int u = -uc; u <= uc - kw_even; ++u, ++ko)
            sum += (edge ? sym_pixel(img, w, h, x + u, y + v) : img[(y + v) * w + x + u]) * k[ko];
    return (float)(sum * kscale);
}

/* decimate.c:37-59 into a fresh buffer */
static float *decimate(const float *img, int w, int h, int factor, const float *k, int kw, int kh, int *rw, int *rh)
{
    const int sw = w / factor + (w & 1), sh = h / factor + (h & 1);
    float *dst = (float *)malloc(sizeof(float) * sw * sh);
    for (int y = 0; y < sh; ++y)
        for (int x = 0; x < sw; ++x) dst[y * sw + x] = filter_pixel(img, w, h, x * factor, y * factor, k, kw, kh, 1.0f);
    *rw = sw; *rh = sh;
    return dst;
}

/* convolve.c:85-150, 11-tap window both ways, result (w-10) x (h-10), dense */
static void convolve_valid(const float *img, int w, int h, float *dst)
{
    const int uc = GAUSSIAN_LEN / 2, dst_w = w - GAUSSIAN_LEN + 1, dst_h = h - GAUSSIAN_LEN + 1;
    float *cache = (float *)calloc((size_t)w * h, sizeof(float));
    for (int y = -uc; y < dst_h + uc; ++y)
        for (int x = 0; x < dst_w; ++x) {
            double sum = 0.0;
            const int off = (y + uc) * w + x + uc;
            for (int u = -uc, ko = 0; u <= uc; ++u, ++ko) sum += img[off + u] * g_gauss[ko];
            cache[off] = (float)(sum * 1.0f);
        }
    for (int x = 0; x < dst_w; ++x)
        for (int y = 0; y < dst_h; ++y) {
            double sum = 0.0;
            const int off = (y + uc) * w + x + uc;
            for (int v = -uc, ko = 0; v <= uc; ++v, ++ko) sum += cache[off + v * w] * g_gauss[ko];
            dst[y * dst_w + x] = (float)(sum * 1.0f);
        }
    free(cache);
}

/* ssim_tools.c:92-243, args == NULL.  out = { ssim, l_mean, c_mean, s_mean } as floats widened to double */
static void iqa_ssim_default(const float *ref, const float *cmp, int w, int h, double out[4])
{
    const float K1 = 0.01f, K2 = 0.03f;
    const int L = 255;
    const float C1 = (K1 * L) * (K1 * L), C2 = (K2 * L) * (K2 * L);
    const size_t n = (size_t)w * h;
    float *ref_mu = malloc(n * 4), *cmp_mu = malloc(n * 4), *rss = malloc(n * 4), *css = malloc(n * 4), *sb = malloc(n * 4);
    float *t = malloc(n * 4);
    convolve_valid(ref, w, h, ref_mu);
    convolve_valid(cmp, w, h, cmp_mu);
    for (size_t i = 0; i < n; ++i) t[i] = ref[i] * ref[i];
    convolve_valid(t, w, h, rss);
    for (size_t i = 0; i < n; ++i) t[i] = cmp[i] * cmp[i];
    convolve_valid(t, w, h, css);
    for (size_t i = 0; i < n; ++i) t[i] = ref[i] * cmp[i];
    convolve_valid(t, w, h, sb);
    free(t);
    w -= GAUSSIAN_LEN - 1; h -= GAUSSIAN_LEN - 1;
    double ssim_sum = 0.0, l_sum = 0.0, c_sum = 0.0, s_sum = 0.0;
    for (int i = 0; i < w * h; ++i) {
        rss[i] -= ref_mu[i] * ref_mu[i];
        css[i] -= cmp_mu[i] * cmp_mu[i];
        rss[i] = (0.0 > rss[i]) ? 0.0 : rss[i];                   /* MAX(0.0, x), stored as float */
        css[i] = (0.0 > css[i]) ? 0.0 : css[i];
        sb[i] -= ref_mu[i] * cmp_mu[i];
    }
    for (int i = 0; i < w * h; ++i) {
        const double srsc = sqrt(rss[i] * css[i]);                 /* float product, double sqrt */
        const double l = (2.0 * ref_mu[i] * cmp_mu[i] + C1) / (ref_mu[i] * ref_mu[i] + cmp_mu[i] * cmp_mu[i] + C1);
        const double c = (2.0 * srsc + C2) / (rss[i] + css[i] + C2);
        const double s = (sb[i] + C2 / 2.0) / (srsc + C2 / 2.0);
        ssim_sum += l * c * s;
        l_sum += l; c_sum += c; s_sum += s;
    }
    out[0] = (float)(ssim_sum / (double)(w * h));
    out[1] = (float)(l_sum / (double)(w * h));
    out[2] = (float)(c_sum / (double)(w * h));
    out[3] = (float)(s_sum / (double)(w * h));
    free(ref_mu); free(cmp_mu); free(rss); free(css); free(sb);
}

static int iqa_round(float a)                                      /* math_utils.c:36-40 */
{
    const int sign_a = a > 0.0f ? 1 : -1;
    return a - (int)a >= 0.5 ? (int)a + sign_a : (int)a;
}

int oracle_ssim_scale(int w, int h)                                /* ssim.c:86 */
{
    const int m = w < h ? w : h, r = iqa_round((float)m / 256.0f);
    return r > 1 ? r : 1;
}

/* ssim.c:52-182.  out = { ssim, l, c, s }.  Returns non-zero when the decimated image is smaller than the window. */
int oracle_ssim(const float *ref, const float *cmp, int w, int h, int stride, double out[4])
{
    float *rf = malloc(sizeof(float) * w * h), *cf = malloc(sizeof(float) * w * h);
    for (int y = 0; y < h; ++y) {
        memcpy(rf + (size_t)y * w, ref + (size_t)y * stride, sizeof(float) * w);
        memcpy(cf + (size_t)y * w, cmp + (size_t)y * stride, sizeof(float) * w);
    }
    const int scale = oracle_ssim_scale(w, h);
    if (scale > 1) {
        float *box = malloc(sizeof(float) * scale * scale);
        for (int i = 0; i < scale * scale; ++i) box[i] = 1.0f / (scale * scale);
        int sw, sh;
        float *r2 = decimate(rf, w, h, scale, box, scale, scale, &sw, &sh);
        float *c2 = decimate(cf, w, h, scale, box, scale, scale, &sw, &sh);
        free(rf); free(cf); free(box);
        rf = r2; cf = c2; w = sw; h = sh;
    }
    int rc = 0;
    if (w < GAUSSIAN_LEN || h < GAUSSIAN_LEN) rc = 1;
    else iqa_ssim_default(rf, cf, w, h, out);
    free(rf); free(cf);
    return rc;
}

/* ms_ssim.c:118-326 (wang = 1, gauss = 1).  out_lcs: 15 doubles l[5], c[5], s[5].  Returns 1 on "scale below 1x1". */
int oracle_ms_ssim(const float *ref, const float *cmp, int w, int h, int stride, double *score, double *out_lcs)
{
    int cw = w, ch = h;
    for (int idx = 0; idx < 5; ++idx) {
        if (cw < GAUSSIAN_LEN || ch < GAUSSIAN_LEN) return 1;
        cw /= 2; ch /= 2;
    }
    float *ri[5], *ci[5];
    ri[0] = malloc(sizeof(float) * w * h); ci[0] = malloc(sizeof(float) * w * h);
    for (int y = 0; y < h; ++y) {
        memcpy(ri[0] + (size_t)y * w, ref + (size_t)y * stride, sizeof(float) * w);
        memcpy(ci[0] + (size_t)y * w, cmp + (size_t)y * stride, sizeof(float) * w);
    }
    cw = w; ch = h;
    int ws[5], hs[5];
    ws[0] = w; hs[0] = h;
    for (int idx = 1; idx < 5; ++idx) {
        int nw, nh;
        ri[idx] = decimate(ri[idx - 1], cw, ch, 2, &g_lpf[0][0], LPF_LEN, LPF_LEN, &nw, &nh);
        ci[idx] = decimate(ci[idx - 1], cw, ch, 2, &g_lpf[0][0], LPF_LEN, LPF_LEN, &nw, &nh);
        cw = nw; ch = nh; ws[idx] = nw; hs[idx] = nh;
    }
    double msssim = 1.0;
    for (int idx = 0; idx < 5; ++idx) {
        double o[4];
        iqa_ssim_default(ri[idx], ci[idx], ws[idx], hs[idx], o);
        const float l = (float)o[1], c = (float)o[2], s = (float)o[3];
        msssim *= pow(l, g_alphas[idx]) * pow(c, g_betas[idx]) * pow(s, g_gammas[idx]);
        if (out_lcs) { out_lcs[idx] = l; out_lcs[5 + idx] = c; out_lcs[10 + idx] = s; }
    }
    for (int idx = 0; idx < 5; ++idx) { free(ri[idx]); free(ci[idx]); }
    *score = msssim;
    return 0;
}

/* ---------------------------------------------------------------- ANSNR (feature/src/ansnr.c:44-104, ansnr_tools.c)
 * 2-D filter path (ANSNR_OPT_FILTER_1D is not defined, ansnr_options.h): ref through the 3x3 binomial, dis through
 * the 5x5 kernel, mirrored borders (-i / 2n-i-1), float accumulation per kernel row then over rows
 * (ansnr_tools.c:159-199); sig = sum ref_filtr^2, noise = sum (ref_filtr - dis_filtd)^2 as float row sums
 * (ansnr_tools.c:60-88).  Every caller hands in the planes AFTER the -128 offset (ansnr.c:174-177, all.c:128-129). */
static const float g_ansnr_ref[9] = {
    1.0 / 16.0, 2.0 / 16.0, 1.0 / 16.0, 2.0 / 16.0, 4.0 / 16.0, 2.0 / 16.0, 1.0 / 16.0, 2.0 / 16.0, 1.0 / 16.0 };
static const float g_ansnr_dis[25] = {
    2.0 / 571.0,  7.0 / 571.0,  12.0 / 571.0,  7.0 / 571.0,  2.0 / 571.0,
    7.0 / 571.0,  31.0 / 571.0, 52.0 / 571.0,  31.0 / 571.0, 7.0 / 571.0,
    12.0 / 571.0, 52.0 / 571.0, 127.0 / 571.0, 52.0 / 571.0, 12.0 / 571.0,
    7.0 / 571.0,  31.0 / 571.0, 52.0 / 571.0,  31.0 / 571.0, 7.0 / 571.0,
    2.0 / 571.0,  7.0 / 571.0,  12.0 / 571.0,  7.0 / 571.0,  2.0 / 571.0 };

static void ansnr_filter2d(const float *f, int fw, const float *src, float *dst, int w, int h, int stride)
{
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            float accum = 0;
            for (int fi = 0; fi < fw; ++fi) {
                float inner = 0;
                for (int fj = 0; fj < fw; ++fj) {
                    int ii = i - fw / 2 + fi, jj = j - fw / 2 + fj;
                    if (ii < 0) ii = -ii; else if (ii >= h) ii = 2 * h - ii - 1;
                    if (jj < 0) jj = -jj; else if (jj >= w) jj = 2 * w - jj - 1;
                    inner += f[fi * fw + fj] * src[ii * stride + jj];
                }
                accum += inner;
            }
            dst[i * w + j] = accum;
        }
}

int oracle_ansnr(const float *ref_off, const float *dis_off, int w, int h, int stride, double peak, double psnr_max,
                 double *score, double *score_psnr)
{
    float *rf = malloc(sizeof(float) * w * h), *df = malloc(sizeof(float) * w * h);
    ansnr_filter2d(g_ansnr_ref, 3, ref_off, rf, w, h, stride);
    ansnr_filter2d(g_ansnr_dis, 5, dis_off, df, w, h, stride);
    float sig = 0, noise = 0;
    for (int i = 0; i < h; ++i) {
        float si = 0, ni = 0;
        for (int j = 0; j < w; ++j) {
            const float r = rf[i * w + j], d = df[i * w + j];
            si += r * r;
            ni += (r - d) * (r - d);
        }
        sig += si; noise += ni;
    }
    free(rf); free(df);
    *score = noise == 0 ? psnr_max : 10.0 * log10(sig / noise);
    const double eps = 1e-10;
    const double v = 10 * log10(peak * peak * w * h / (noise > eps ? noise : eps));
    *score_psnr = v < psnr_max ? v : psnr_max;
    return 0;
}
