// vmaf_kernels.h - internal launch interface between the engine (vmaf_engine.cpp) and the HIP kernels.
#ifndef VMAF_KERNELS_H_
#define VMAF_KERNELS_H_

#include <hip/hip_runtime_api.h>
#include <stddef.h>

// Per-frame raw sums produced on the device, finished into the 20 atoms on the host:
//   [0..7]   VIF  num,den of scale 0..3
//   [8..31]  ADM  scale s: num_h,num_v,num_d,den_h,den_v,den_d  (cube sums over the centre window)
//   [32]     motion: sum |blur_prev - blur_cur|
// optional metrics (zero sections when not enabled):
//   [33]     PSNR: sum (ref - dis)^2
//   [34..37] SSIM: sum l*c*s, sum l, sum c, sum s over the valid window of the decimated planes
//   [38..57] MS-SSIM scale 0..4: the same four sums each
//   [58,59]  ANSNR: sig = sum ref_filtr^2, noise = sum (ref_filtr - dis_filtd)^2
#define VK_NSUMS 60

typedef struct {
    size_t offset;   // first double of this section in the partial buffer
    int ntiles;      // blocks per frame in the producing kernel
    int k;           // doubles per block
    int comp;        // which of the k this sum takes
} vk_reduce_desc;

typedef struct { vk_reduce_desc d[VK_NSUMS]; } vk_reduce_table;

#ifdef __cplusplus
extern "C" {
#endif

int vk_vif_tiles(int w, int h);
int vk_adm_tiles(int w, int h);     // w,h = INPUT size of the scale
int vk_motion_tiles(int w, int h);
// loads the kernels' code object onto the current device (otherwise the first launch does, ~10 ms)
hipError_t vk_preload(void);

// order: 0 = the reference's scalar tap order (disable_avx), 1 = its AVX order (default on x86), 2 = fused multiply-adds.
// ref / dis at scale 0: the caller's picture, sample_bytes 4 = fp32 planes, 1 = 8-bit samples, 2 = little-endian 16-bit words
// (value / 4.0), widened in the kernels' loads (file_io.c:112-116,159-163); strides in samples.  Scales 1-3: fp32 only.
hipError_t vk_launch_vif_stat(int scale, int order, const void *ref, const void *dis, int sample_bytes, size_t frame_stride, int row_stride,
                              int w, int h, int nframes, double *partial, float *oref, float *odis, size_t out_frame_stride,
                              int out_row_stride, float *nd, hipStream_t s);
// nd (null = off): per-pixel num / den planes [frame][2][h][w] written by the launch above; vk_launch_vif_sums adds them in
// vif_sum_s's order (fp32 per row, then rows) -> out[frame * out_stride + 2 * scale + {0 num, 1 den}] as doubles.
// rowsum: nframes * 2 * h floats of scratch
hipError_t vk_launch_vif_sums(const float *nd, int w, int h, int nframes, float *rowsum, double *out, size_t out_stride, int scale, hipStream_t s);
hipError_t vk_launch_adm_scale(int scale, int order, const void *ref, const void *dis, int sample_bytes, size_t in_frame_stride, int in_row_stride,
                               int w, int h, float *ll_ref, float *ll_dis, size_t ll_frame_stride, int ll_row_stride,
                               float rfac_hv, float rfac_d, int left, int top, int right, int bottom, int nframes,
                               double *partial, hipStream_t s);
// ref: the caller's picture; sample_bytes 4 = fp32 planes, 1 = 8-bit samples, 2 = little-endian 16-bit words (value / 4.0),
// widened in the kernel's loads (file_io.c:112-116,159-163).  partial: vk_motion_tiles() doubles per frame, [0] = the total
hipError_t vk_launch_motion(int order, const void *ref, int sample_bytes, size_t frame_stride_bytes, int row_stride_bytes, int w, int h,
                            int nframes, int has_carry, const float *carry_in, float *carry_out, double *partial, hipStream_t s);
// raw luma (1 or 2 bytes per sample, 2 = little-endian words / 4.0) -> dense fp32 planes [nframes][h][w]
hipError_t vk_launch_widen(const void *src, int bytes_per_sample, size_t src_row_bytes, size_t src_frame_bytes, float *dst,
                           int w, int h, int nframes, hipStream_t s);
// optional metrics (psnr.c, ssim.c, ms_ssim.c, iqa/): all on un-offset planes
int vk_psnr_tiles(int h);
int vk_ssim_tiles(int w, int h);    // w, h = size of the planes _iqa_ssim sees
hipError_t vk_launch_psnr(const float *ref, const float *dis, size_t frame_stride, int row_stride, int w, int h, int nframes,
                          double *partial, hipStream_t s);
hipError_t vk_launch_box_decimate(const float *src, size_t src_frame_stride, int src_row_stride, int w, int h, int f, float *dst,
                                  int sw, int sh, int nframes, hipStream_t s);
hipError_t vk_launch_box_decimate_pair(const float *ref, const float *dis, size_t src_frame_stride, int src_row_stride, int w, int h, int f,
                                       float *oref, float *odis, int sw, int sh, int nframes, hipStream_t s);
// exact: the reference's own expression for every window sum (float product, widened, added) instead of v_fma_f64
hipError_t vk_launch_lpf_decimate(const float *ref, const float *dis, size_t src_frame_stride, int src_row_stride, int w, int h,
                                  float *oref, float *odis, int sw, int sh, int nframes, int exact, hipStream_t s);
int vk_ansnr_tiles(int w, int h);
hipError_t vk_launch_ansnr(const float *ref, const float *dis, size_t frame_stride, int row_stride, int w, int h, int nframes,
                           double *partial, hipStream_t s);
// cs_only: only the contrast and structure sums (MS-SSIM levels whose luminance exponent is 0)
hipError_t vk_launch_ssim_stat(const float *ref, const float *cmp, int w, int h, int nframes, int cs_only, int exact, double *partial,
                               double *psnr_partial, hipStream_t s);
hipError_t vk_launch_reduce(const double *partial, const vk_reduce_table *tab, int nframes, double *sums, hipStream_t s);
hipError_t vk_launch_svm(const double *sv, const double *coef, int n_sv, int dim, double gamma, double rho, const double *x,
                         int n, double *out, hipStream_t s);

#ifdef __cplusplus
}
#endif
#endif
