/*
 * vmaf_oracle.h - CPU restatement of the reference's VMAF v0.6.1 feature path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker.  The product (libvmaf_amd.so) never does.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every function here bit-for-bit
 * (planes) / to 1e-12 relative (scalars) against the reference's own C sources compiled by
 * oracle/Makefile into oracle/_ref/ (scalar and AVX convolution paths), and against the
 * reference's pinned flat-image numbers (python/test/feature_test.py:702-719).
 *
 * All planes are fp32, row-major, stride given in PIXELS.  "off" planes are luma - 128
 * (reference: wrapper/src/combo.c:236-237).
 */
#ifndef VMAF_ORACLE_H_
#define VMAF_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Summation order of the separable convolutions (SURVEY.md 8a note N1). */
enum { ORACLE_ORDER_SCALAR = 0,   /* reference with disable_avx: taps accumulated left to right */
       ORACLE_ORDER_AVX    = 1 }; /* reference AVX path: 4 interleaved partial sums in the interior */

/* Atom row layout (20 doubles per frame), the insertion order of wrapper/src/combo.c:260-364. */
enum { ATOM_ADM_NUM = 0, ATOM_ADM_DEN = 1, ATOM_ADM_NUM_SCALE0 = 2, /* num/den pairs to index 9 */
       ATOM_MOTION = 10, ATOM_VIF_NUM_SCALE0 = 11,                  /* num/den pairs to index 18 */
       ATOM_VIF = 19, ATOM_COUNT = 20 };

/* feature/src/vif_tools.c:271-331 (scalar) / feature/src/common/convolution_avx.c:787-837 (AVX order). */
void oracle_conv_sep(const float *filter, int fwidth, const float *src, float *dst,
                     int w, int h, int src_stride, int dst_stride, int order);

/* feature/src/vif.c:43-297.  scores[2s], scores[2s+1] = num, den of scale s.  Returns 0 on success. */
int oracle_vif(const float *ref_off, const float *dis_off, int w, int h, int stride,
               int order, double *score, double *score_num, double *score_den, double scores[8]);

/* feature/src/adm.c:51-275 with border_factor 0.1.  exact_rcp: 0 = plain division (portable),
 * 1 = the reference's x86 rcpss + one Newton step (adm_tools.c:34-38; x86 only). */
int oracle_adm(const float *ref_off, const float *dis_off, int w, int h, int stride, int exact_rcp,
               double *score, double *score_num, double *score_den, double scores[8]);

/* feature/src/motion.c:44-60: mean absolute difference of two (blurred) planes. */
double oracle_sad(const float *a, const float *b, int w, int h, int a_stride, int b_stride);

/* One iteration of the per-frame loop, wrapper/src/combo.c:147-371, for un-offset luma planes.
 * prev_blur (w*h floats, stride w) is read when !first_frame and always overwritten with this
 * frame's blurred ref.  atoms receives ATOM_COUNT doubles. */
int oracle_frame(const float *ref_raw, const float *dis_raw, int w, int h, int stride,
                 float *prev_blur, int first_frame, int order, double *atoms);

/* Test aid: the same per-pixel num / den values summed in extended precision (exact[2s], exact[2s+1]). */
int oracle_vif_exact_sums(const float *ref_off, const float *dis_off, int w, int h, int stride, int order, double exact[8]);

/* The filter tables (vif_tools.c:82-97, motion_tools.h:24-29) for tests that want them. */
const float *oracle_vif_filter(int scale, int *width);
const float *oracle_motion_filter(void);

/* 8 values 1/Q: (scale0 hv, scale0 d, scale1 hv, ...) - adm_tools.c:190-192, adm_tools.h:105-115. */
void oracle_adm_csf_rfactors(float out[8]);

/* vif_tools.c:46-75 */
float oracle_log2f_approx(float x);

/* stage-level hook: LL band of one db2 DWT level (adm_tools.c:293-393), ((w+1)/2) x ((h+1)/2) dense */
void oracle_adm_dwt2_ll(const float *src, int w, int h, int stride, float *out);

/* ---- optional per-frame metrics (oracle/iqa_oracle.c): psnr.c, ssim.c, ms_ssim.c, iqa/ ---- */
double oracle_psnr(const float *ref, const float *dis, int w, int h, int stride, double peak, double psnr_max);
int oracle_ssim_scale(int w, int h);
int oracle_ssim(const float *ref, const float *cmp, int w, int h, int stride, double out[4]);           /* ssim, l, c, s */
int oracle_ms_ssim(const float *ref, const float *cmp, int w, int h, int stride, double *score, double *out_lcs /* 15 or NULL */);
int oracle_ansnr(const float *ref_off, const float *dis_off, int w, int h, int stride, double peak, double psnr_max,
                 double *score, double *score_psnr);
const float *oracle_ssim_window(void);      /* 11 taps */
const float *oracle_ms_ssim_lpf(void);      /* 9 x 9 */

#ifdef __cplusplus
}
#endif
#endif
