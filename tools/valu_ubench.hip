// Micro-benchmark: VALU issue rate on gfx950 for the instruction kinds the stencil kernels use.
// hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o tools/valu_ubench && ./tools/valu_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, float a, float b, int iters)
{
    extern __shared__ float occ_limiter[];      // dynamic LDS only limits the resident workgroups per CU
    if (iters < 0) occ_limiter[threadIdx.x] = a;
    float x[8];
    float2_ p[4];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int i = 0; i < 4; ++i) p[i] = float2_{x[2 * i], x[2 * i + 1]};
    float2_ pa = {a, a}, pb = {b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {          // v_fma_f32 x8
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
            } else if (MODE == 1) {   // v_mul_f32 + v_add_f32 (4 each)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[2 * i]) : "v"(a));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2 * i + 1]) : "v"(b));
                }
            } else if (MODE == 2) {   // v_pk_fma_f32 x4 (8 lanes-flops pairs)
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pa), "v"(pb));
            } else if (MODE == 4) {   // v_mul_f32 with a 32-bit literal (8-byte encoding) + v_add_f32, what the compiler emits for constexpr taps
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mul_f32 %0, 0x3de8f3b5, %0" : "+v"(x[2 * i]));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2 * i + 1]) : "v"(b));
                }
            } else if (MODE == 5) {   // v_mul_f32 with an SGPR operand + v_add_f32
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("v_mul_f32 %0, %1, %0" : "+v"(x[2 * i]) : "s"(a));
                    asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[2 * i + 1]) : "v"(b));
                }
            } else {                  // v_pk_mul_f32 + v_pk_add_f32 (2 each)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[2 * i]) : "v"(pa));
                    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[2 * i + 1]) : "v"(pb));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static int g_lds = 0;
template <int MODE> void run(const char *name, int blocks, int iters, float *d, int elems_per_iter)
{
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, g_lds>>>(d, 1.0001f, 0.0001f, iters);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, g_lds>>>(d, 1.0001f, 0.0001f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lane_ops = (double)blocks * 256 * iters * 8 * elems_per_iter;   // float results produced
    printf("%-28s lds=%5d blocks=%5d  %.3f ms  %.2f T float-results/s\n", name, g_lds, blocks, ms, lane_ops / ms / 1e9);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8192 * sizeof(float));
    // dynamic LDS per workgroup -> resident workgroups per CU (4 waves each): 0 -> 8, 40 KB -> 4, 52 KB -> 3, 64 KB -> 2
    for (int lds : {0, 40 * 1024, 52 * 1024, 64 * 1024}) {
        g_lds = lds;
        const int blocks = 8192;
        run<0>("v_fma_f32", blocks, 2000, d, 8);
        run<1>("v_mul_f32+v_add_f32", blocks, 2000, d, 8);
        run<4>("v_mul_f32(literal)+v_add_f32", blocks, 2000, d, 8);
        run<5>("v_mul_f32(sgpr)+v_add_f32", blocks, 2000, d, 8);
        run<3>("v_pk_mul_f32+v_pk_add_f32", blocks, 2000, d, 8);
    }
    return 0;
}
