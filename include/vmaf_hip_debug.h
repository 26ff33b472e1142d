/*
 * vmaf_hip_debug.h - test and profiling hooks of libvmaf_amd.so.  Not part of the product interface
 * (include/vmaf_hip.h, include/libvmaf.h): used by tests/ and bench.py only, may change without notice.
 */
#ifndef VMAF_HIP_DEBUG_H_
#define VMAF_HIP_DEBUG_H_

#include "vmaf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

/*
 * Stage-level test hook, the counterpart of the reference's *_OPT_DEBUG_DUMP paths (feature/src/vif.c:227-263,
 * adm.c:147-228): copies one intermediate plane of the LAST synchronised extract call to the host.
 * kind 0 / 1 = the ref / dis plane compute_vif works on at scale `level` (vif.c:157-186: blurred + decimated),
 * kind 2 / 3 = the ref / dis LL band compute_adm hands to scale `level` (adm.c:241-247); level 1..3; frame = index in
 * the batch.  The planes of the other stages never exist in memory here (they live in LDS / registers).
 */
int vmaf_hip_engine_debug_read_plane(vmaf_hip_engine *e, int kind, int level, int frame, float *out, size_t out_floats,
                                     int *width, int *height);

/*
 * The raw fp64 device sums of one frame of the LAST synchronised extract call, before the host rounds them into the
 * atoms the way the reference does ((float) casts, powf, ...): out[0..7] VIF num, den of scale 0..3, out[8..31] ADM cube
 * sums (scale s: num h,v,d, den h,v,d), out[32] motion SAD.  The kernels accumulate their per-pixel fp32 values in fp64,
 * so these pin every pixel's value against an oracle that sums the same values exactly (oracle_vif_exact_sums).
 */
#define VMAF_HIP_DEBUG_SUMS 33
int vmaf_hip_engine_debug_read_sums(vmaf_hip_engine *e, int frame, double *out);

/*
 * Kernel-level timing hook for bench.py: HIP-event time in milliseconds of the LAST extract call's
 * kernels, measured on the engine's own stream.  which: 0 = whole batch, 1 = VIF scale-0 statistic
 * kernel (the dominant kernel).  Returns < 0 when nothing has run.
 */
double vmaf_hip_last_kernel_ms(vmaf_hip_engine *e, int which);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* VMAF_HIP_DEBUG_H_ */
