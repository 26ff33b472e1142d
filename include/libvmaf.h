/*
 * libvmaf.h - the library-level drop-in boundary of the MI355X VMAF engine.
 *
 * compute_vmaf() has the signature, argument meaning, stdout lines and log files of the
 * reference's wrapper/src/libvmaf.h:26 (implementation wrapper/src/libvmaf.cpp:28-67 ->
 * wrapper/src/vmaf.cpp:631-753): a program linked against the reference's libvmaf.a
 * (wrapper/src/main.cpp:109, FFmpeg's libvmaf filter) links against libvmaf_amd.so unchanged.
 *
 *   fmt            yuv420p yuv422p yuv444p yuv420p10le yuv422p10le yuv444p10le (informational here,
 *                  as in the reference where it only selects the PSNR peak, combo.c:171-186)
 *   read_frame     called on the caller's thread, once per frame, in order; writes UN-OFFSET luma as
 *                  fp32 in 8-bit units into ref_data / main_data (rows stride_byte apart,
 *                  stride_byte = 4*width rounded up to 32); temp_data is 2 planes of scratch.
 *                  Returns 0 = frame delivered, 2 = end of stream, 1 = failure (combo.c:149-157)
 *   model_path     <name>.pkl (protocol-0 pickle); the libsvm model is read from <model_path>.model
 *   log_path       NULL = no log; log_fmt "json" = JSON, anything else = XML (vmaf.cpp:675-750)
 *   disable_avx    0: sum the filter taps in the order of the reference's AVX kernels (its default on x86);
 *                  non-zero: in the order of its scalar loops (libvmaf.cpp:56-61) - both reproduced bit for bit
 *   enable_transform | phone_model   apply the model's score_transform (libvmaf.cpp:37-39)
 *   do_psnr / do_ssim / do_ms_ssim   add the per-frame metrics psnr / ssim / ms_ssim to the logs (combo.c:168-230,
 *                  vmaf.cpp:595-597), computed on the GPU next to the VMAF features (vmaf_hip_engine_enable_metrics)
 *   pool_method    NULL | "mean" | "min" | "harmonic_mean"
 * Returns the pooled VMAF score.  Failures are reported as in the reference, by a C++ exception
 * derived from std::exception (wrapper/src/vmaf.h:144-151) that wrapper/src/main.cpp:219-230 catches.
 *
 * Frames are gathered into batches, shipped to the GPU(s) and scored by the kernels behind
 * include/vmaf_hip.h.  Environment knobs: VMAF_HIP_DEVICES="0,1,.." (default "0"), VMAF_HIP_BATCH=N.
 */
#ifndef LIBVMAF_H_
#define LIBVMAF_H_

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only these entry points are exported */

double compute_vmaf(char* fmt, int width, int height, int (*read_frame)(float *ref_data, float *main_data, float *temp_data, int stride_byte, void *user_data), void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method);

/*
 * Same call for callers that cannot catch C++ exceptions (C programs, cgo/ctypes/JNI bindings).
 * No counterpart in the reference (its compute_vmaf lets exceptions escape the extern "C" frame,
 * libvmaf.cpp:63).  Returns 0 and stores the pooled score in *score, or returns non-zero and copies
 * the message into errbuf (when errbuf != NULL).
 */
int compute_vmaf_noexcept(double *score, char* fmt, int width, int height, int (*read_frame)(float *ref_data, float *main_data, float *temp_data, int stride_byte, void *user_data), void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method, char *errbuf, int errbuf_len);

/*
 * compute_vmaf with raw-luma hand-over.  No counterpart in the reference, whose file-backed callback
 * converts every sample to fp32 on one host thread (feature/src/common/frame.c:26-129 -> file_io.c:86-172) before
 * the library sees it.  Here the callback copies the luma plane of ref and dis AS THE FILE HOLDS IT - 8-bit samples,
 * or 16-bit little-endian words for the *10le formats (decided from fmt) - into the library's page-locked buffers
 * (dense rows, stride_byte = width * bytes per sample) and skips the chroma itself; the GPU widens to fp32
 * (/ 4.0 for words, no masking) and applies the -128 offset.  Same return values as read_frame (0 / 2 / 1), same
 * arguments, outputs, logs and exceptions as compute_vmaf; scores are bit-identical to the fp32 path.
 */
double compute_vmaf_raw(char* fmt, int width, int height, int (*read_frame_raw)(void *ref_luma, void *main_luma, int stride_byte, void *user_data), void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method);
int compute_vmaf_raw_noexcept(double *score, char* fmt, int width, int height, int (*read_frame_raw)(void *ref_luma, void *main_luma, int stride_byte, void *user_data), void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method, char *errbuf, int errbuf_len);

/*
 * compute_vmaf_raw for sources that can SEEK (regular files): frame-range sharding of one video over several GPUs
 * (no counterpart in the reference, whose only parallelism is one process per ref/dis pair,
 * python/src/vmaf/tools/misc.py:246-276).  The caller states the number of frames and hands over a positional callback:
 * read_frame_raw_at(ref_luma, main_luma, stride_byte, frame_index, user_data) fills the two raw luma planes of frame
 * `frame_index` (0-based; layout as for compute_vmaf_raw) and returns 0, or non-zero on failure.  It is called from
 * SEVERAL THREADS AT ONCE - one per device listed in VMAF_HIP_DEVICES ("0,1,..,7"; default "0"; an id may repeat) - each
 * walking its own contiguous range [g*F/G, (g+1)*F/G) in order, after one extra call for frame start-1 (the motion halo:
 * the shard blurs that ref frame itself).  Shards exchange nothing; motion2 (vmaf.cpp:375-379) and the fusion run once
 * after all of them.  Scores, logs and stdout are identical, digit for digit, to compute_vmaf_raw on the same frames.
 */
double compute_vmaf_raw_seekable(char* fmt, int width, int height, long nframes, int (*read_frame_raw_at)(void *ref_luma, void *main_luma, int stride_byte, long frame_index, void *user_data), void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method);
int compute_vmaf_raw_seekable_noexcept(double *score, char* fmt, int width, int height, long nframes, int (*read_frame_raw_at)(void *ref_luma, void *main_luma, int stride_byte, long frame_index, void *user_data), void *user_data, char *model_path, char *log_path, char *log_fmt, int disable_clip, int disable_avx, int enable_transform, int phone_model, int do_psnr, int do_ssim, int do_ms_ssim, char *pool_method, char *errbuf, int errbuf_len);

/*
 * The feature names of a model in model order (model_dict['feature_names'], wrapper/src/vmaf.cpp:79-81), e.g.
 * "VMAF_feature_adm2_score": *n_features = their number D; names_out (may be NULL) receives them separated by '\n',
 * NUL-terminated, when names_len is large enough (otherwise the call fails and *n_features still holds D).
 * Returns 0 on success.
 */
int vmaf_amd_model_feature_names(const char *model_path, int *n_features, char *names_out, int names_len,
                                 char *errbuf, int errbuf_len);

/*
 * Fusion half only (wrapper/src/vmaf.cpp:351-559): atoms [nframes][20] (layout in vmaf_hip.h) ->
 * per-frame VMAF scores and the derived feature vectors named by the model, in model order
 * (features_out: [nframes][features_capacity], may be NULL; the first D columns of a row are written, D =
 * vmaf_amd_model_feature_names' n_features; features_capacity < D is an error, nothing is written).
 * The SVR runs on `device`.  Returns 0 on success.
 * _v2 is the form with the capacity argument.  The symbol without the suffix keeps the prototype it was first shipped
 * with (no capacity: features_out rows are exactly D wide, the caller sizes them from vmaf_amd_model_feature_names), so
 * that a program built against that prototype still gets the arguments it passes; new code uses _v2.
 */
int vmaf_amd_predict_from_atoms_v2(const char *model_path, const double *atoms, int nframes, int disable_clip,
                                   int enable_transform, int device, double *vmaf_out, double *features_out,
                                   int features_capacity, char *errbuf, int errbuf_len);
int vmaf_amd_predict_from_atoms(const char *model_path, const double *atoms, int nframes, int disable_clip,
                                int enable_transform, int device, double *vmaf_out, double *features_out,
                                char *errbuf, int errbuf_len);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif

#endif /* LIBVMAF_H_ */
