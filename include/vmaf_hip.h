/*
 * vmaf_hip.h - thin C-ABI over the hand-written gfx950 (MI355X) HIP kernels of the VMAF v0.6.1
 * feature-extraction + fusion hot path.  Plain pointers and sizes only; no C++/torch types.
 *
 * These entry points are what a host layer binds INSTEAD of the reference's per-frame C feature
 * functions; each cites the reference interface it replaces (paths relative to the reference repo).
 * The library-level drop-in, compute_vmaf(), is declared in include/libvmaf.h.
 *
 * Atom row layout (VMAF_HIP_ATOMS = 20 doubles per frame) is the insertion order of
 * wrapper/src/combo.c:260-364:
 *   [0] adm_num [1] adm_den [2..9] adm_{num,den}_scale0..3 [10] motion
 *   [11..18] vif_{num,den}_scale0..3 [19] vif
 *
 * Error convention: functions return 0 on success, non-zero on failure (the convention of the
 * reference's C feature functions, combo.c:240-243); vmaf_hip_last_error() gives the message.
 * Nothing here falls back to a CPU path: without a usable HIP device every call fails.
 *
 * Threading contract (the reference's compute_vmaf is single-threaded and not re-entrant: it writes the process-global
 * `cpu`, libvmaf.cpp:56-61, and libsvm's static line buffer, svm.cpp:330-331; this library has no such globals):
 *   - An engine is NOT internally locked: at most one thread may be inside a call on a given engine at a time (calls on
 *     one engine from several threads need the caller's own mutex).  The thread may change from call to call.
 *   - Different engines are independent - own streams, device buffers, pending slots - and may be created, used and
 *     destroyed concurrently from different threads, on the same or on different devices
 *     (tests/test_gpu_parity.py::test_engines_on_concurrent_host_threads).  vmaf_hip_engine_destroy waits for the whole
 *     device to go idle, so it may stall (never corrupt) other engines' calls on that device.
 *   - vmaf_hip_last_error() is per calling thread.  vmaf_hip_host_alloc/free and vmaf_hip_device_count keep no state;
 *     vmaf_hip_svm_predict keeps the model and its work buffers on the GPU between calls behind one lock per GPU: all
 *     three may be called from any thread at any time.
 *   - Every call makes the engine's device current on the calling thread (hipSetDevice) and leaves it so.
 *   - vmaf_hip_extract_device_noalloc uses one static scratch area per GPU: calls on the same GPU must not overlap - a
 *     call that arrives while another one is running on that GPU is refused with an error (see its comment); it is
 *     independent of engines.
 * Debug / profiling hooks live in include/vmaf_hip_debug.h.
 */
#ifndef VMAF_HIP_H_
#define VMAF_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only these entry points are exported */

#define VMAF_HIP_ATOMS 20

typedef struct vmaf_hip_engine vmaf_hip_engine;

/* Message of the last failure on the calling thread ("" if none). */
const char *vmaf_hip_last_error(void);

/* Number of HIP devices visible (0 when there is none / no driver). */
int vmaf_hip_device_count(void);

/*
 * One engine = one GPU, one frame geometry, one video stream's motion state.
 * Replaces the per-video buffers combo() allocates (wrapper/src/combo.c:100-144) and the per-frame
 * work areas of compute_vif (feature/src/vif.c:104) and compute_adm (feature/src/adm.c:101):
 * everything is allocated once here and stays resident in HBM.
 * max_batch = most frames a single extract call may carry (device scratch scales with it).
 */
int vmaf_hip_engine_create(vmaf_hip_engine **out, int device, int width, int height, int max_batch);
void vmaf_hip_engine_destroy(vmaf_hip_engine *e);

/*
 * Summation order of the separable convolutions (VIF filters, motion blur).  The reference has two and
 * picks one per call: its AVX kernels (default on x86) or the scalar loops (compute_vmaf's disable_avx,
 * wrapper/src/libvmaf.cpp:56-61; the process-global `cpu` read by feature/src/vif_tools.c:279 and
 * feature/src/common/convolution.c:82).  The two differ at 1e-7 relative per pixel, which VIF's per-pixel
 * branches can amplify to several 1e-3 VMAF on noisy content, so the engine reproduces both bit for bit
 * per pixel: order 1 = AVX order (default), 0 = scalar order.
 * Order 2 is this library's own opt-in fast mode and has no counterpart in the reference: the sequential order with
 * every multiply-add fused (one rounding per tap instead of two), and the same for ADM's db2 taps.  Not bit-identical
 * to either reference order; on the repository's fixtures features land within 1.2e-5 and VMAF within 6e-4 of the AVX
 * order (the reference's own two orders are up to 5e-5 / 3.7e-3 apart), for about a third more throughput.
 */
int vmaf_hip_engine_set_conv_order(vmaf_hip_engine *e, int order);

/*
 * Plane sums of VIF in the reference's own order (feature/src/vif_tools.c:161-179, vif_sum_s: one fp32 accumulator per row,
 * columns left to right, then the row sums top to bottom in fp32).  By default the VIF kernels add their per-pixel num /
 * den values - bit-identical to the reference's - in fp64 and reduce deterministically: the raw atoms vif_num_scaleN /
 * vif_den_scaleN agree with the reference to ~1e-6 relative (they only ever feed ratios).  With exact = 1 the kernels also
 * leave num and den as planes (the reference's own intermediate, vif.c:258-262; 8 bytes per pixel of the current scale in
 * device memory, allocated here) and a second pass adds them as vif_sum_s does: the eight per-scale atoms and `vif` are then
 * bit-identical to the reference's in the selected convolution order, for about a quarter more time per frame.  compute_vmaf
 * sets it with disable_avx (the "most literal" mode).  Not while calls are in flight.
 */
int vmaf_hip_engine_set_sums_exact(vmaf_hip_engine *e, int exact);

/*
 * Extract the 20 atoms of `nframes` CONSECUTIVE frames.  Replaces, per frame, the body of the
 * reference loop wrapper/src/combo.c:236-364:
 *   offset_image(-128) x2          (feature/src/common/file_io.c:177-198)
 *   compute_adm(...)               (feature/src/adm.c:51, prototype combo.c:41)
 *   convolution_f32_c(FILTER_5..)  (feature/src/common/convolution.c:78) + compute_motion (motion.c:65)
 *   compute_vif(...)               (feature/src/vif.c:43, prototype combo.c:45)
 * Inputs are UN-OFFSET fp32 luma planes exactly as a read_frame callback delivers them
 * (8-bit units, libvmaf.h read_frame contract): frame f starts at base + f*frame_stride_bytes,
 * rows are row_stride_bytes apart (multiple of 4).
 * continue_sequence = 0: frame 0 of this call starts a video (motion = 0, combo.c:312-315);
 *                   = 1: frame 0 follows the last frame of the previous call on this engine (or the
 *                        frame given to vmaf_hip_motion_seed_*), whose blurred ref is still in HBM.
 * atoms_out: HOST pointer, nframes * VMAF_HIP_ATOMS doubles.
 * _device: ref/dis are device pointers on the engine's GPU.  _host: host pointers (staged by the engine).
 */
int vmaf_hip_extract_device(vmaf_hip_engine *e, const float *d_ref, const float *d_dis,
                            size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                            int continue_sequence, double *atoms_out);
int vmaf_hip_extract_host(vmaf_hip_engine *e, const float *h_ref, const float *h_dis,
                          size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                          int continue_sequence, double *atoms_out);

/*
 * Asynchronous form of vmaf_hip_extract_device for benchmarking and pipelining: enqueue only;
 * atoms land in atoms_out (host, ideally pinned) once vmaf_hip_engine_sync() returns.
 */
int vmaf_hip_extract_device_async(vmaf_hip_engine *e, const float *d_ref, const float *d_dis,
                                  size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                                  int continue_sequence, double *atoms_out);
int vmaf_hip_extract_host_async(vmaf_hip_engine *e, const float *h_ref, const float *h_dis,
                                size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                                int continue_sequence, double *atoms_out);
int vmaf_hip_engine_sync(vmaf_hip_engine *e);
/*
 * Up to TWO asynchronous extract calls may be in flight on one engine, so that the host can enqueue batch k+1 while
 * batch k is still running and the GPU never idles between batches.  vmaf_hip_engine_sync() finishes all of them in
 * order; vmaf_hip_engine_sync_oldest() waits for the oldest only (its atoms_out is complete on return).  A third
 * async call, a motion seed or vmaf_hip_engine_enable_metrics with calls in flight is an error.
 */
int vmaf_hip_engine_sync_oldest(vmaf_hip_engine *e);

/*
 * Raw-luma forms of the four extract calls: the planes hold the samples as the YUV file does -
 * bytes_per_sample 1 = 8-bit, 2 = 16-bit little-endian words (the *10le formats) - and the GPU does the
 * conversion the reference does on the host while reading: (float)px, or (float)px / 4.0 for words, no masking
 * to 10 bits (feature/src/common/file_io.c:86-172, read_image_b2s / read_image_w2s called from
 * feature/src/common/frame.c:44-78).  The conversion happens in the loads of the kernels that read the picture (VIF
 * scale 0, ADM scale 0, motion): the samples never exist as fp32 planes in device memory, a quarter (half) of the fp32
 * bytes cross PCIe and come out of HBM.  (Only the optional metrics, whose kernels take fp32 planes, get a widened copy.)
 * Strides in bytes, multiples of bytes_per_sample.  Results are bit-identical to the fp32 calls on the converted planes.
 */
int vmaf_hip_extract_host_raw(vmaf_hip_engine *e, const void *h_ref, const void *h_dis, int bytes_per_sample,
                              size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                              int continue_sequence, double *atoms_out);
int vmaf_hip_extract_device_raw(vmaf_hip_engine *e, const void *d_ref, const void *d_dis, int bytes_per_sample,
                                size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                                int continue_sequence, double *atoms_out);
int vmaf_hip_extract_host_raw_async(vmaf_hip_engine *e, const void *h_ref, const void *h_dis, int bytes_per_sample,
                                    size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                                    int continue_sequence, double *atoms_out);
int vmaf_hip_extract_device_raw_async(vmaf_hip_engine *e, const void *d_ref, const void *d_dis, int bytes_per_sample,
                                      size_t row_stride_bytes, size_t frame_stride_bytes, int nframes,
                                      int continue_sequence, double *atoms_out);

/*
 * Page-locked host memory for read_frame buffers, so that the H2D copies of
 * vmaf_hip_extract_host_async really are asynchronous.  Replaces aligned_malloc for the frame
 * buffers of combo() (wrapper/src/combo.c:116-144, feature/src/common/alloc.h); 4096-byte aligned.
 */
void *vmaf_hip_host_alloc(size_t bytes);
void vmaf_hip_host_free(void *p);

/*
 * Motion halo for frame-range sharding: blur one ref frame (the frame just before this shard's
 * first frame) into the engine's "previous blurred frame" slot, so that the next extract call with
 * continue_sequence = 1 gets an exact motion value for its first frame.  Replaces the carried
 * prev_blur_buf state of combo.c:309-326.
 */
int vmaf_hip_motion_seed_device(vmaf_hip_engine *e, const float *d_ref, size_t row_stride_bytes);
int vmaf_hip_motion_seed_host(vmaf_hip_engine *e, const float *h_ref, size_t row_stride_bytes);
int vmaf_hip_motion_seed_host_raw(vmaf_hip_engine *e, const void *h_ref, int bytes_per_sample, size_t row_stride_bytes);
int vmaf_hip_motion_seed_device_raw(vmaf_hip_engine *e, const void *d_ref, int bytes_per_sample, size_t row_stride_bytes);

/*
 * Optional per-frame metrics of the reference's frame loop (wrapper/src/combo.c:168-230): PSNR
 * (feature/src/psnr.c:37-61), SSIM (feature/src/ssim.c:52-182) and MS-SSIM (feature/src/ms_ssim.c:118-326), computed
 * on the planes as given (before the -128 offset), on their own stream next to the VMAF features.  ten_bit selects
 * the PSNR peak and cap (255.75 / 72 instead of 255 / 60, combo.c:171-186).  Call before the first extract, or between
 * synchronised calls.  SSIM at decimation factor 1 (min(w,h) < 384) and MS-SSIM need dense input planes.
 * vmaf_hip_engine_get_metrics copies the metrics of the frames of the extract call that was synchronised last (by
 * vmaf_hip_engine_sync or vmaf_hip_engine_sync_oldest; another call may still be in flight):
 * out[nframes][VMAF_HIP_METRICS] = psnr, ssim, ms_ssim, ansnr, anpsnr (0 where not enabled).
 * vmaf_hip_engine_enable_ansnr adds ANSNR / ANPSNR (feature/src/ansnr.c:44-104, the atoms `ansnr` and `anpsnr` of
 * feature/src/all.c:148-176), computed like every caller of compute_ansnr does on the planes after the -128 offset.
 *
 * Arithmetic of SSIM / MS-SSIM: by default every window sum takes its products through v_fma_f64 (the product of two
 * floats is exact in double, so a term enters the double running sum without the reference's intermediate rounding of
 * img * k to float, iqa/convolve.c:129,144,262) and the per-pixel sqrt and divisions are hardware estimates + Newton steps:
 * same tap order, same accumulator and result types, but not the reference's expression - scores within 1.3e-7 of it.
 * vmaf_hip_engine_set_metrics_exact(e, 1) selects the reference's own expressions instead (float product, widened, added;
 * IEEE double sqrt and division): per-pixel l, c, s bit-identical to iqa/ssim_tools.c:193-201, the scores equal to the
 * reference's up to the order of the final double sum (~1e-15), for about half the metric throughput.  compute_vmaf sets
 * it together with the scalar convolution order when disable_avx is given: that flag is the "most literal" mode.
 */
#define VMAF_HIP_METRICS 5
int vmaf_hip_engine_enable_metrics(vmaf_hip_engine *e, int do_psnr, int do_ssim, int do_ms_ssim, int ten_bit);
int vmaf_hip_engine_set_metrics_exact(vmaf_hip_engine *e, int exact);
int vmaf_hip_engine_enable_ansnr(vmaf_hip_engine *e, int ten_bit);
int vmaf_hip_engine_get_metrics(vmaf_hip_engine *e, double *out, int nframes);

/*
 * The entry point that allocates nothing.  No counterpart in the reference: it exists for callers that cannot call
 * hipMalloc - code under HIP graph capture, tight serving loops, harnesses that forbid allocation - and so cannot
 * create an engine.  Same result as vmaf_hip_engine_create + vmaf_hip_extract_device(continue_sequence = 0) +
 * vmaf_hip_engine_destroy on the same frames, bit for bit (same kernels, same reduction): the 20 atoms of `nframes`
 * consecutive frames starting a video (motion of frame 0 is 0).  d_ref / d_dis: device pointers on `device`, un-offset
 * fp32 luma as for vmaf_hip_extract_device; conv_order as vmaf_hip_engine_set_conv_order; stream: a hipStream_t on that
 * device (NULL = the default stream) that all launches go to; atoms_out: host, complete on return.
 * Everything the kernels need beyond the inputs (pyramid planes, LL bands, partial and reduced sums) lives in ONE static
 * device array of VMAF_HIP_NOALLOC_SCRATCH_BYTES per GPU, held from library load to process end.  The call makes no
 * hipMalloc / hipFree / hipHostMalloc and creates no stream or event.  A request that does not fit is refused with a
 * message (vmaf_hip_noalloc_scratch_needed tells in advance; 0 = invalid arguments): 32 MiB hold e.g. two 1920x1080
 * frames, 6 of 1280x720 or 33 of 576x324 per call; one 3840x2160 frame (74 MB) does not fit.  All calls on one GPU share that GPU's array: calls that
 * overlap in time on the same GPU (other threads, other streams) would race, so a call that finds another one running on
 * its GPU fails with a message instead (a per-GPU flag, held from the first launch to the final synchronisation); callers
 * serialise.  The array costs its 32 MiB of device memory on every GPU the process touches, whether this entry point is
 * ever called or not.
 */
#define VMAF_HIP_NOALLOC_SCRATCH_BYTES (32u << 20)
size_t vmaf_hip_noalloc_scratch_needed(int width, int height, int nframes);
int vmaf_hip_extract_device_noalloc(int device, void *stream, const float *d_ref, const float *d_dis, int width, int height,
                                    size_t row_stride_bytes, size_t frame_stride_bytes, int nframes, int conv_order,
                                    double *atoms_out);

/*
 * Batched nu-SVR RBF prediction.  Replaces the per-frame svm_predict() call site
 * wrapper/src/vmaf.cpp:468 (wrapper/src/svm.cpp:305 -> :229 -> k_function RBF :179-219):
 *   out[f] = sum_i coef[i] * exp(-gamma * |x[f] - sv[i]|^2) - rho        (all fp64)
 * sv is dense [n_sv][dim] (entries omitted from the sparse libsvm text are 0), x is [n][dim].
 * All pointers are HOST pointers.
 */
int vmaf_hip_svm_predict(int device, const double *sv, const double *coef, int n_sv, int dim,
                         double gamma, double rho, const double *x, int n, double *out);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* VMAF_HIP_H_ */
