// Micro-benchmark: issue rate of the fp64-path instructions the optional-metric kernels use (gfx950).
// hipcc --offload-arch=gfx950 -O3 tools/fp64_ubench.hip -o tools/fp64_ubench && ./tools/fp64_ubench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, double a, float b, int iters)
{
    double x[8];
    float f[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001 + i; f[i] = threadIdx.x * 0.5f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(a));
                else if (MODE == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
                else if (MODE == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x[i]) : "v"(f[i]));
                else if (MODE == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(x[i]));
                else if (MODE == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(b));
                else if (MODE == 5) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "s"(a), "v"(a));
                else asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += x[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int blocks, int iters, double *d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(d, 1.0001, 1.0001f, iters);
    (void)hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(d, 1.0001, 1.0001f, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double winst = (double)blocks * 4 * iters * 64;
    // cycles per wave instruction and SIMD at 2.4 GHz with `blocks`/256 workgroups of 4 waves per CU
    const double per_simd = winst / 1024.0;
    printf("%-28s blocks=%5d  %.3f ms  %.3f T wave-inst/s  ~%.2f cycles/inst/SIMD @2.4GHz\n", name, blocks, ms, winst / ms / 1e9, ms * 1e-3 * 2.4e9 / per_simd);
}

int main()
{
    double *d; (void)hipMalloc(&d, 256 * 8192 * sizeof(double));
    for (int blocks : {256 * 4, 256 * 2}) {
        run<4>("v_mul_f32", blocks, 1000, d);
        run<0>("v_fma_f64", blocks, 1000, d);
        run<5>("v_fma_f64 (sgpr operand)", blocks, 1000, d);
        run<1>("v_add_f64", blocks, 1000, d);
        run<6>("v_mul_f64", blocks, 1000, d);
        run<2>("v_cvt_f64_f32", blocks, 1000, d);
        run<3>("v_cvt_f32_f64", blocks, 1000, d);
    }
    return 0;
}
