/*
 * vmaf_oracle.c - CPU restatement of the reference's VMAF v0.6.1 per-frame feature path.
 *
 * TEST INFRASTRUCTURE ONLY (see vmaf_oracle.h).  Written from the reference's behaviour, each
 * function citing the file:line it follows.  Compile with -ffp-contract=off and without
 * fast-math: every multiply and add below is meant to round separately, as the reference's
 * x86-64 SSE/AVX build does (no FMA: wrapper/Makefile:47-55 passes only -mavx).
 *
 * Parity status: PINNED against oracle/_ref (tests/test_oracle_vs_ref.py), the committed
 * golden fixtures generated from it (tests/golden/), and the reference's own known-answer tests
 * for its flat and checkerboard clips (python/test/feature_test.py:566-719,
 * quality_runner_test.py:333-432; tests/test_reference_pinned.py, tests/test_oracle_golden.py).
 */
#include "vmaf_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(__x86_64__) || defined(__i386__)
#include <emmintrin.h>
#define ORACLE_HAVE_RCPSS 1
#endif

/* ------------------------------------------------------------------------------------------ */
/* tables                                                                                      */
/* ------------------------------------------------------------------------------------------ */

/* feature/src/vif_tools.c:82-87 (decimal literals, rounded to float by the compiler). */
static const float k_vif_filter[4][17] = {
    { 0.00745626912f, 0.0142655009f, 0.0250313189f, 0.0402820669f, 0.0594526194f, 0.0804751068f,
      0.0999041125f, 0.113746084f, 0.118773937f, 0.113746084f, 0.0999041125f, 0.0804751068f,
      0.0594526194f, 0.0402820669f, 0.0250313189f, 0.0142655009f, 0.00745626912f },
    { 0.0189780835f, 0.0558981746f, 0.120920904f, 0.192116052f, 0.224173605f, 0.192116052f,
      0.120920904f, 0.0558981746f, 0.0189780835f },
    { 0.054488685f, 0.244201347f, 0.402619958f, 0.244201347f, 0.054488685f },
    { 0.166378498f, 0.667243004f, 0.166378498f }
};
static const int k_vif_width[4] = { 17, 9, 5, 3 };               /* vif_tools.c:97 */

/* feature/src/motion_tools.h:24-29 */
static const float k_motion_filter[5] = { 0.054488685f, 0.244201342f, 0.402619947f, 0.244201342f, 0.054488685f };

const float *oracle_vif_filter(int scale, int *width) { if (width) *width = k_vif_width[scale]; return k_vif_filter[scale]; }
const float *oracle_motion_filter(void) { return k_motion_filter; }

/* Mirror rule used by every stencil on the path (vif_tools.c:301, convolution_internal.h:38-46,
 * adm_tools.c:241-248,321-324): below 0 reflect without repeating the edge, at/after n reflect
 * repeating it. */
static inline int mirror(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - i - 1 : i); }

/* ------------------------------------------------------------------------------------------ */
/* separable convolution                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* One output sample, taps accumulated strictly left to right starting from 0
 * (vif_tools.c:295-306 / convolution_internal.h:31-50). */
static inline float taps_sequential(const float *f, int fw, const float *p, int pstride, int center, int n)
{
    float acc = 0;
    int r = fw / 2;
    for (int k = 0; k < fw; ++k)
        acc += f[k] * p[(size_t)mirror(center - r + k, n) * pstride];
    return acc;
}

/* One output sample in the interior order of the reference's AVX kernels
 * (convolution_avx.c:225-270 for 9 taps per pass, :347-392, :408-437; generic 3-tap :116-205):
 * the filter is consumed in passes of up to 9 taps; inside a pass tap k goes to partial sum k%4,
 * the pass value is (s0+s2)+(s1+s3), and passes are added in order. No mirroring (interior only). */
static inline float taps_avx(const float *f, int fw, const float *p, int pstride, int center)
{
    int r = fw / 2;
    const float *q = p + (ptrdiff_t)(center - r) * pstride;
    float total = 0;
    for (int x = 0; x < fw; x += 9) {
        int n = fw - x < 9 ? fw - x : 9;
        float s[4] = { 0, 0, 0, 0 };
        for (int k = 0; k < n; ++k) {
            float prod = f[x + k] * q[(size_t)(x + k) * pstride];
            /* the specialised kernels assign the first product of each partial sum; the
             * generic one adds it to +0 - identical except for the sign of a zero result */
            s[k & 3] = (k < 4) ? prod : s[k & 3] + prod;
        }
        float a = s[0] + s[2];
        float b = s[1] + s[3];
        float pass = a + b;
        total = (x == 0) ? (0.0f + pass) : total + pass;
    }
    return total;
}

static inline int floor8(int n) { return n - n % 8; }
static inline int ceil8(int n) { return n % 8 ? n + (8 - n % 8) : n; }

void oracle_conv_sep(const float *filter, int fw, const float *src, float *dst,
                     int w, int h, int src_stride, int dst_stride, int order)
{
    int r = fw / 2;
    int tstride = ceil8(w);                                   /* convolution_avx.c:801 */
    float *tmp = (float *)malloc(sizeof(float) * (size_t)tstride * h);

    /* Region in which the AVX path uses its vector kernels (convolution_avx.c:800-836). */
    int wv = floor8(w);
    int jv_end = wv - ceil8(r + 1);

    /* vertical pass: src -> tmp */
    for (int i = 0; i < h; ++i) {
        int row_vec = (order == ORACLE_ORDER_AVX) && i >= r && i < h - r;
        for (int j = 0; j < w; ++j) {
            if (row_vec && j < wv)
                tmp[(size_t)i * tstride + j] = taps_avx(filter, fw, src + j, src_stride, i);
            else
                tmp[(size_t)i * tstride + j] = taps_sequential(filter, fw, src + j, src_stride, i, h);
        }
    }
    /* horizontal pass: tmp -> dst */
    for (int i = 0; i < h; ++i) {
        const float *row = tmp + (size_t)i * tstride;
        for (int j = 0; j < w; ++j) {
            int col_vec = (order == ORACLE_ORDER_AVX) && j >= r && j < r + jv_end;
            dst[(size_t)i * dst_stride + j] = col_vec ? taps_avx(filter, fw, row, 1, j)
                                                      : taps_sequential(filter, fw, row, 1, j, w);
        }
    }
    free(tmp);
}

/* ------------------------------------------------------------------------------------------ */
/* VIF                                                                                         */
/* ------------------------------------------------------------------------------------------ */

/* vif_tools.c:33-75: degree-8 polynomial in the mantissa; -inf at 0, NaN below. */
float oracle_log2f_approx(float x)
{
    static const float poly[9] = { -0.012671635276421f, 0.064841182402670f, -0.157048836463065f,
                                   0.257167726303123f, -0.353800560300520f, 0.480131410397451f,
                                   -0.721314327952201f, 1.442694803896991f, 0.0f };
    if (x == 0) return -INFINITY;
    if (x < 0) return NAN;
    uint32_t bits;
    memcpy(&bits, &x, 4);
    uint32_t mant_bits = (bits & 0x007FFFFFu) | 0x3F800000u;
    float mant;
    memcpy(&mant, &mant_bits, 4);
    float e = (float)((int32_t)((bits & 0x7F800000u) >> 23) - 127);
    float t = mant - 1.0f, v = 0;
    for (int i = 0; i < 9; ++i) v = v * t + poly[i];
    return e + v;
}

/* vif_tools.c:161-179: per-row fp32 sums, then fp32 sum of the rows. */
static float sum_rows_then_total(const float *x, int w, int h, int stride)
{
    float total = 0;
    for (int i = 0; i < h; ++i) {
        float row = 0;
        for (int j = 0; j < w; ++j) row += x[(size_t)i * stride + j];
        total += row;
    }
    return total;
}

static int vif_impl(const float *ref, const float *dis, int w, int h, int stride, int order,
                    double *score, double *score_num, double *score_den, double scores[8], double exact[8]);

int oracle_vif(const float *ref, const float *dis, int w, int h, int stride, int order,
               double *score, double *score_num, double *score_den, double scores[8])
{
    return vif_impl(ref, dis, w, h, stride, order, score, score_num, score_den, scores, NULL);
}

/* Test aid, no counterpart in the reference: the per-pixel num / den values of vif_statistic_s (bit-identical to the
 * reference's, see tests/test_oracle_vs_ref.py) summed in extended precision instead of vif_sum_s's fp32 row-then-total
 * order.  exact[2s], exact[2s+1] = sum of num, den of scale s, correct to ~1e-16 relative: what a GPU kernel that
 * accumulates the same per-pixel values in fp64 must reproduce to ~1e-12, i.e. a pin on EVERY pixel's value. */
int oracle_vif_exact_sums(const float *ref, const float *dis, int w, int h, int stride, int order, double exact[8])
{
    double s, n, d, sc[8];
    return vif_impl(ref, dis, w, h, stride, order, &s, &n, &d, sc, exact);
}

static int vif_impl(const float *ref, const float *dis, int w, int h, int stride, int order,
                    double *score, double *score_num, double *score_den, double scores[8], double exact[8])
{
    size_t n0 = (size_t)w * h;
    float *buf = (float *)malloc(sizeof(float) * n0 * 12);
    if (!buf) return 1;
    float *ref_s = buf, *dis_s = buf + n0;                 /* current-scale inputs for scale > 0 */
    float *mu1 = buf + 2 * n0, *mu2 = buf + 3 * n0;
    float *xx = buf + 4 * n0, *yy = buf + 5 * n0, *xy = buf + 6 * n0;
    float *xxf = buf + 7 * n0, *yyf = buf + 8 * n0, *xyf = buf + 9 * n0;
    float *num = buf + 10 * n0, *den = buf + 11 * n0;

    const float *cr = ref, *cd = dis;
    int cs = stride;

    for (int s = 0; s < 4; ++s) {
        const float *f = k_vif_filter[s];
        int fw = k_vif_width[s];

        if (s > 0) {
            /* vif.c:157-186: blur the previous scale with THIS scale's filter, keep even samples */
            oracle_conv_sep(f, fw, cr, mu1, w, h, cs, w, order);
            oracle_conv_sep(f, fw, cd, mu2, w, h, cs, w, order);
            int w2 = w / 2, h2 = h / 2;
            for (int i = 0; i < h2; ++i)
                for (int j = 0; j < w2; ++j) {      /* vif_tools.c:146-159 */
                    ref_s[(size_t)i * w2 + j] = mu1[(size_t)(2 * i) * w + 2 * j];
                    dis_s[(size_t)i * w2 + j] = mu2[(size_t)(2 * i) * w + 2 * j];
                }
            w = w2; h = h2; cr = ref_s; cd = dis_s; cs = w;
        }

        oracle_conv_sep(f, fw, cr, mu1, w, h, cs, w, order);        /* vif.c:189-190 */
        oracle_conv_sep(f, fw, cd, mu2, w, h, cs, w, order);
        for (int i = 0; i < h; ++i)                                  /* vif.c:197, vif_tools.c:181-212 */
            for (int j = 0; j < w; ++j) {
                float a = cr[(size_t)i * cs + j], b = cd[(size_t)i * cs + j];
                xx[(size_t)i * w + j] = a * a;
                yy[(size_t)i * w + j] = b * b;
                xy[(size_t)i * w + j] = a * b;
            }
        oracle_conv_sep(f, fw, xx, xxf, w, h, w, w, order);          /* vif.c:199-201 */
        oracle_conv_sep(f, fw, yy, yyf, w, h, w, w, order);
        oracle_conv_sep(f, fw, xy, xyf, w, h, w, w, order);

        /* vif_tools.c:214-269 (the mu products of vif.c:195 are formed here, same fp32 values) */
        const float sigma_nsq = 2;
        const float sigma_max_inv = (float)(4.0 / (255.0 * 255.0));
        for (size_t p = 0; p < (size_t)w * h; ++p) {
            float m1 = mu1[p], m2 = mu2[p];
            float m1sq = m1 * m1, m2sq = m2 * m2, m12 = m1 * m2;
            float s1 = xxf[p] - m1sq;
            float s2 = yyf[p] - m2sq;
            float s12 = xyf[p] - m12;
            float nv, dv;
            if (s1 < sigma_nsq) {
                float prod = s2 * sigma_max_inv;
                nv = (float)(1.0 - (double)prod);           /* double literal in the reference */
                dv = 1.0f;
            } else {
                float sv = (s2 + sigma_nsq) * s1;
                if (s12 < 0) nv = 0.0f;
                else {
                    float g = sv - s12 * s12;
                    nv = oracle_log2f_approx(sv / g);
                }
                dv = oracle_log2f_approx(1.0f + s1 / sigma_nsq);
            }
            num[p] = nv; den[p] = dv;
        }
        scores[2 * s] = sum_rows_then_total(num, w, h, w);            /* vif.c:265-269 */
        scores[2 * s + 1] = sum_rows_then_total(den, w, h, w);
        if (exact) {
            long double en = 0.0L, ed = 0.0L;
            for (size_t p = 0; p < (size_t)w * h; ++p) { en += num[p]; ed += den[p]; }
            exact[2 * s] = (double)en;
            exact[2 * s + 1] = (double)ed;
        }
    }

    double tn = 0, td = 0;                                            /* vif.c:277-291 */
    for (int s = 0; s < 4; ++s) { tn += scores[2 * s]; td += scores[2 * s + 1]; }
    if (score_num) *score_num = tn;
    if (score_den) *score_den = td;
    if (score) *score = (td == 0.0) ? 1.0 : tn / td;
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* ADM                                                                                         */
/* ------------------------------------------------------------------------------------------ */

static const float k_db2_lo[4] = { 0.482962913144690f, 0.836516303737469f, 0.224143868041857f, -0.129409522550921f };
static const float k_db2_hi[4] = { -0.129409522550921f, -0.224143868041857f, 0.836516303737469f, -0.482962913144690f };

/* adm_tools.h:69-115: Watson et al. DWT quantisation step for luma, 7/9 basis. theta 1 = h/v, 2 = d. */
static float dwt_quant_step_y(int lambda, int theta)
{
    static const float a = 0.495f, k = 0.466f, f0 = 0.401f;
    static const float g[4] = { 1.501f, 1.0f, 0.534f, 1.0f };
    static const float amp[6][4] = {
        { 0.62171f, 0.67234f, 0.72709f, 0.67234f }, { 0.34537f, 0.41317f, 0.49428f, 0.41317f },
        { 0.18004f, 0.22727f, 0.28688f, 0.22727f }, { 0.091401f, 0.11792f, 0.15214f, 0.11792f },
        { 0.045943f, 0.059758f, 0.077727f, 0.059758f }, { 0.023013f, 0.030018f, 0.039156f, 0.030018f } };
    /* the reference's adm_tools.h:19 spells pi with 30 digits; same double */
    float r = (float)(3.0f * 1080 * 3.14159265358979323846264338327 / 180.0);
    float t = (float)log10(pow(2.0, lambda + 1) * f0 * g[theta] / r);
    float q = (float)(2.0 * a * pow(10.0, k * t * t) / amp[lambda][theta]);
    return q;
}

void oracle_adm_csf_rfactors(float out[8])
{
    for (int s = 0; s < 4; ++s) {
        out[2 * s] = 1.0f / dwt_quant_step_y(s, 1);
        out[2 * s + 1] = 1.0f / dwt_quant_step_y(s, 2);
    }
}

typedef struct { float *a, *v, *h, *d; } bands_t;

/* adm_tools.c:293-393: one level of the decimating 2-D db2 DWT.  band naming per adm_tools.h:27-32. */
static void dwt2(const float *src, int w, int h, int stride, bands_t out, int ostride)
{
    int w2 = (w + 1) / 2, h2 = (h + 1) / 2;
    float *lo = (float *)malloc(sizeof(float) * w * 2), *hi = lo + w;
    for (int i = 0; i < h2; ++i) {
        for (int j = 0; j < w; ++j) {
            float al = 0, ah = 0;
            for (int t = 0; t < 4; ++t) {
                float px = src[(size_t)mirror(2 * i - 1 + t, h) * stride + j];
                al += k_db2_lo[t] * px;
                ah += k_db2_hi[t] * px;
            }
            lo[j] = al; hi[j] = ah;
        }
        for (int j = 0; j < w2; ++j) {
            float ll = 0, lh = 0, hl = 0, hh = 0;
            for (int t = 0; t < 4; ++t) {
                int jj = mirror(2 * j - 1 + t, w);
                ll += k_db2_lo[t] * lo[jj];
                lh += k_db2_hi[t] * lo[jj];
                hl += k_db2_lo[t] * hi[jj];
                hh += k_db2_hi[t] * hi[jj];
            }
            out.a[(size_t)i * ostride + j] = ll;   /* lo-V lo-H */
            out.v[(size_t)i * ostride + j] = lh;   /* lo-V hi-H */
            out.h[(size_t)i * ostride + j] = hl;   /* hi-V lo-H */
            out.d[(size_t)i * ostride + j] = hh;   /* hi-V hi-H */
        }
    }
    free(lo);
}

/* Stage-level hook for tests: the approximation band of one DWT level, i.e. the plane compute_adm hands to the next
 * scale (adm.c:241-247).  out: ((w+1)/2) x ((h+1)/2), dense. */
void oracle_adm_dwt2_ll(const float *src, int w, int h, int stride, float *out)
{
    const int w2 = (w + 1) / 2, h2 = (h + 1) / 2;
    float *scratch = (float *)malloc(sizeof(float) * 3 * w2 * h2);
    bands_t b = { out, scratch, scratch + w2 * h2, scratch + 2 * w2 * h2 };
    dwt2(src, w, h, stride, b, w2);
    free(scratch);
}

static inline float div_rcp(float n, float d, int exact_rcp)
{
#ifdef ORACLE_HAVE_RCPSS
    if (exact_rcp) {                                         /* adm_tools.c:34-40 */
        float xi = _mm_cvtss_f32(_mm_rcp_ss(_mm_load_ss(&d)));
        return n * (xi + xi * (1.0f - d * xi));
    }
#endif
    (void)exact_rcp;
    return n / d;
}

/* adm_tools.c:51-77 */
static float sum_cube_center(const float *x, int w, int h, int stride)
{
    const double border = 0.1;                               /* adm_options.h ADM_BORDER_FACTOR */
    int left = (int)(w * border - 0.5), top = (int)(h * border - 0.5);
    int right = w - left, bottom = h - top;
    float total = 0;
    for (int i = top; i < bottom; ++i) {
        float row = 0;
        for (int j = left; j < right; ++j) {
            float v = fabsf(x[(size_t)i * stride + j]);
            row += v * v * v;
        }
        total += row;
    }
    return powf(total, 1.0f / 3.0f) + powf((bottom - top) * (right - left) / 32.0f, 1.0f / 3.0f);
}

int oracle_adm(const float *ref, const float *dis, int w, int h, int stride, int exact_rcp,
               double *score, double *score_num, double *score_den, double scores[8])
{
    double limit = 1e-10 * (w * h) / (1920.0 * 1080.0);      /* adm.c:53-57 */
    int bw = (w + 1) / 2, bh = (h + 1) / 2;
    size_t nb = (size_t)bw * bh;
    float *buf = (float *)malloc(sizeof(float) * nb * 24);
    if (!buf) return 1;
    float *p = buf;
    bands_t rb = { p, p + nb, p + 2 * nb, p + 3 * nb }; p += 4 * nb;
    bands_t db = { p, p + nb, p + 2 * nb, p + 3 * nb }; p += 4 * nb;
    float *csf_o[3], *csf_r[3], *csf_a[3], *cm_r[3];
    for (int b = 0; b < 3; ++b) { csf_o[b] = p; p += nb; }
    for (int b = 0; b < 3; ++b) { csf_r[b] = p; p += nb; }
    for (int b = 0; b < 3; ++b) { csf_a[b] = p; p += nb; }
    for (int b = 0; b < 3; ++b) { cm_r[b] = p; p += nb; }
    float *ref_ll = p; p += nb;
    float *dis_ll = p; p += nb;
    float *mta = p; p += nb;

    float rf[8];
    oracle_adm_csf_rfactors(rf);
    const float cos1sq = (float)(cos(1.0 * 3.1415926535897932384626433832795028841971693993751 / 180.0) *
                                 cos(1.0 * 3.1415926535897932384626433832795028841971693993751 / 180.0));
    const float eps = 1e-30f;

    const float *cr = ref, *cd = dis;
    int cs = stride;
    double num = 0, den = 0;

    for (int s = 0; s < 4; ++s) {
        dwt2(cr, w, h, cs, rb, bw);
        dwt2(cd, w, h, cs, db, bw);
        w = (w + 1) / 2; h = (h + 1) / 2;

        const float *ob[3] = { rb.h, rb.v, rb.d }, *tb[3] = { db.h, db.v, db.d };
        float rfac[3] = { rf[2 * s], rf[2 * s], rf[2 * s + 1] };

        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                size_t q = (size_t)i * bw + j;
                float o[3] = { ob[0][q], ob[1][q], ob[2][q] }, t[3] = { tb[0][q], tb[1][q], tb[2][q] };
                float rest[3];
                for (int b = 0; b < 3; ++b) {                    /* adm_tools.c:110-120 */
                    float k = div_rcp(t[b], o[b] + eps, exact_rcp);
                    k = k < 0.0f ? 0.0f : (k > 1.0f ? 1.0f : k);
                    rest[b] = k * o[b];
                }
                float dp = o[0] * t[0] + o[1] * t[1];            /* adm_tools.c:143-147 */
                float om = o[0] * o[0] + o[1] * o[1];
                float tm = t[0] * t[0] + t[1] * t[1];
                int aligned = (dp >= 0.0f) && (dp * dp >= cos1sq * om * tm);
                for (int b = 0; b < 3; ++b) {
                    float r = aligned ? t[b] : rest[b];
                    float a = t[b] - r;
                    csf_o[b][q] = rfac[b] * o[b];                /* adm_tools.c:177-206 */
                    csf_r[b][q] = rfac[b] * r;
                    csf_a[b][q] = rfac[b] * a;
                }
            }

        /* adm_tools.c:208-259: per band 9 taps row-major, bands added in order h, v, d */
        for (int i = 0; i < h; ++i)
            for (int j = 0; j < w; ++j) {
                float thr = 0;
                for (int b = 0; b < 3; ++b) {
                    float acc = 0;
                    for (int di = 0; di < 3; ++di)
                        for (int dj = 0; dj < 3; ++dj) {
                            float c = (di == 1 && dj == 1) ? 1.0f / 15.0f : 1.0f / 30.0f;
                            int ii = mirror(i - 1 + di, h), jj = mirror(j - 1 + dj, w);
                            acc += c * fabsf(csf_a[b][(size_t)ii * bw + jj]);
                        }
                    thr += acc;
                }
                mta[(size_t)i * bw + j] = thr;
            }
        for (int i = 0; i < h; ++i)                              /* adm_tools.c:261-291 */
            for (int j = 0; j < w; ++j) {
                size_t q = (size_t)i * bw + j;
                for (int b = 0; b < 3; ++b) {
                    float x = fabsf(csf_r[b][q]) - mta[q];
                    cm_r[b][q] = x < 0.0f ? 0.0f : x;
                }
            }

        float num_s = 0, den_s = 0;                              /* adm.c:229-238 */
        for (int b = 0; b < 3; ++b) num_s += sum_cube_center(cm_r[b], w, h, bw);
        for (int b = 0; b < 3; ++b) den_s += sum_cube_center(csf_o[b], w, h, bw);
        num += num_s; den += den_s;
        scores[2 * s] = num_s; scores[2 * s + 1] = den_s;

        for (int i = 0; i < h; ++i) {                            /* adm.c:241-247 */
            memcpy(ref_ll + (size_t)i * bw, rb.a + (size_t)i * bw, sizeof(float) * w);
            memcpy(dis_ll + (size_t)i * bw, db.a + (size_t)i * bw, sizeof(float) * w);
        }
        cr = ref_ll; cd = dis_ll; cs = bw;
    }

    num = num < limit ? 0 : num;                                 /* adm.c:256-268 */
    den = den < limit ? 0 : den;
    if (score) *score = (den == 0.0) ? 1.0 : num / den;
    if (score_num) *score_num = num;
    if (score_den) *score_den = den;
    free(buf);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* motion + frame loop                                                                         */
/* ------------------------------------------------------------------------------------------ */

double oracle_sad(const float *a, const float *b, int w, int h, int a_stride, int b_stride)
{
    float total = 0;                                             /* motion.c:44-60 */
    for (int i = 0; i < h; ++i) {
        float row = 0;
        for (int j = 0; j < w; ++j)
            row += fabs(a[(size_t)i * a_stride + j] - b[(size_t)i * b_stride + j]);
        total += row;
    }
    return (double)(float)(total / (w * h));
}

int oracle_frame(const float *ref_raw, const float *dis_raw, int w, int h, int stride,
                 float *prev_blur, int first_frame, int order, double *atoms)
{
    size_t n = (size_t)w * h;
    float *buf = (float *)malloc(sizeof(float) * n * 3);
    if (!buf) return 1;
    float *r = buf, *d = buf + n, *blur = buf + 2 * n;
    for (int i = 0; i < h; ++i)                                  /* combo.c:236-237, file_io.h:26 */
        for (int j = 0; j < w; ++j) {
            r[(size_t)i * w + j] = ref_raw[(size_t)i * stride + j] + -128.0f;
            d[(size_t)i * w + j] = dis_raw[(size_t)i * stride + j] + -128.0f;
        }
    double sc, sn, sd, scores[8];
    int ret = oracle_adm(r, d, w, h, w, 0, &sc, &sn, &sd, scores);        /* combo.c:240-269 */
    atoms[ATOM_ADM_NUM] = sn; atoms[ATOM_ADM_DEN] = sd;
    for (int k = 0; k < 8; ++k) atoms[ATOM_ADM_NUM_SCALE0 + k] = scores[k];

    oracle_conv_sep(k_motion_filter, 5, r, blur, w, h, w, w, order);      /* combo.c:309-326 */
    atoms[ATOM_MOTION] = first_frame ? 0.0 : oracle_sad(prev_blur, blur, w, h, w, w);
    memcpy(prev_blur, blur, sizeof(float) * n);

    ret |= oracle_vif(r, d, w, h, w, order, &sc, &sn, &sd, scores);       /* combo.c:336-364 */
    for (int k = 0; k < 8; ++k) atoms[ATOM_VIF_NUM_SCALE0 + k] = scores[k];
    atoms[ATOM_VIF] = sc;
    free(buf);
    return ret;
}
