/*
 * Glue for building the REFERENCE's feature sources into oracle/_ref/libvmaf_ref_feature.so.
 * TEST INFRASTRUCTURE ONLY.  Contains no reference code: it only provides the process-global
 * `cpu` that the reference's libvmaf.cpp:26 defines (consumers: vif_tools.c:29, convolution.c:24)
 * and a setter so a test can choose the AVX or the scalar convolution path, exactly like
 * compute_vmaf()'s disable_avx flag does (libvmaf.cpp:56-61).
 */
#include "common/cpu.h"

enum vmaf_cpu cpu = VMAF_CPU_NONE;

/* mode 0 = scalar (disable_avx), 1 = cpu_autodetect() (AVX when the host has it). Returns the value set. */
int ref_set_cpu(int autodetect)
{
    cpu = autodetect ? cpu_autodetect() : VMAF_CPU_NONE;
    return (int)cpu;
}
