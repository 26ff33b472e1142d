// Micro-benchmark: can the matrix core serve as a second fp32 MULTIPLIER port for the VALU-bound VIF stencil?
//   v_mfma_f32_4x4x1_16b_f32 D, A, B, 0: 16 blocks of 4x4 outer products; lane l = 4g + r gets
//   D[i] = A[4g + i] * B[l], i = 0..3 - four products tap_i * x per lane and instruction, if A holds four filter taps
//   in every group of four lanes and B the lane's own sample.
// Two questions, answered on the GPU:
//   1. are those products bit-identical to v_mul_f32 (one rounding, RNE, no flush) for the values the path produces?
//   2. do MFMA instructions overlap with v_add_f32 of the same / other waves, i.e. is the multiply then free?
// hipcc --offload-arch=gfx950 -O3 tools/mfma_mul_ubench.hip -o tools/mfma_mul_ubench && ./tools/mfma_mul_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include <random>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void exact_kernel(const float *taps /*4 per group*/, const float *x, float *out_mfma, float *out_mul, int n)
{
    // one wave per 64 samples
    const int lane = threadIdx.x & 63, base = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    if (base + lane >= n) return;
    const float a = taps[lane & 3];
    const float b = x[base + lane];
    const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    const f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, z, 0, 0, 0);
    for (int i = 0; i < 4; ++i) {
        out_mfma[(size_t)(base + lane) * 4 + i] = d[i];
        float p;
        const float t = taps[i];
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(t), "v"(b));
        out_mul[(size_t)(base + lane) * 4 + i] = p;
    }
}

// MODE 0: 4 v_mul + 4 v_add per step   MODE 1: 1 mfma(4 products) + 4 v_add   MODE 2: 4 v_add only   MODE 3: mfma only
// MODE 4: 2 v_mul + half an mfma (1 mfma per two steps) + 4 v_add  (a balanced split)
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float *out, float a, float b, int iters)
{
    float acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 0.001f + i;
    float x = threadIdx.x * 0.5f;
    const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    f4 d0 = z, d1 = z;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {
                float p[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(a), "v"(acc[4 + i]));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(p[i]));
            } else if (MODE == 1) {
                f4 d;
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(x));
                // consume the PREVIOUS result: software pipelining hides the MFMA latency
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(d0[i]));
                d0 = d;
            } else if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(b));
            } else if (MODE == 3) {
                f4 d;
                asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(x));
                d1 = d0; d0 = d;
            } else {
                float p[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[i]) : "v"(a), "v"(acc[4 + i]));
                if (r & 1) {
                    f4 d;
                    asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(x));
                    d1 = d0; d0 = d;
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[i]) : "v"(p[i]));
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[2 + i]) : "v"(d1[(r & 1) * 2 + i]));
            }
        }
    }
    float s = d0[0] + d0[1] + d0[2] + d0[3] + d1[0] + d1[3];
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE> void run(const char *name, int blocks, int iters, float *d)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<blocks, 256>>>(d, 1.0001f, 0.0001f, iters);
    hipEventRecord(e0);
    rate_kernel<MODE><<<blocks, 256>>>(d, 1.0001f, 0.0001f, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double steps = (double)blocks * 4 * iters * 8;          // wave-steps (4 products + 4 adds each)
    printf("%-44s blocks=%5d  %.3f ms  %.3f T wave-steps/s\n", name, blocks, ms, steps / ms / 1e9);
}

int main()
{
    // ---- 1. exactness
    std::vector<float> taps = {0.00745626912f, 0.0142655009f, 0.0250313189f, 0.0402820669f};
    std::vector<float> taps2 = {0.0594526194f, 0.0804751068f, 0.0999041125f, 0.113746084f};
    std::vector<float> taps3 = {0.118773937f, 0.244201347f, 0.402619958f, 0.166378498f};
    std::vector<float> x;
    for (int v = 0; v < 1024; ++v) x.push_back(v / 4.0f - 128.0f);                       // every 10-bit sample after the offset
    for (int a = 0; a < 1024; a += 3) for (int b = 0; b < 1024; b += 7) x.push_back((a / 4.0f - 128.0f) * (b / 4.0f - 128.0f));   // products r*d
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> u(-300.0f, 300.0f);
    for (int i = 0; i < 200000; ++i) x.push_back(u(rng));                                 // filtered values (pyramid levels)
    for (int i = 0; i < 200000; ++i) { float v = u(rng); x.push_back(v * v); }
    for (int i = 0; i < 100000; ++i) { uint32_t bits = rng(); float v; memcpy(&v, &bits, 4); if (v == v) x.push_back(v); }   // any bit pattern (incl. denormals, inf)
    x.push_back(0.0f); x.push_back(-0.0f); x.push_back(1e-38f); x.push_back(1e-45f); x.push_back(3e38f);
    while (x.size() % 64) x.push_back(1.0f);
    const int n = (int)x.size();
    float *d_t, *d_x, *d_a, *d_b;
    hipMalloc(&d_t, 16); hipMalloc(&d_x, n * 4); hipMalloc(&d_a, (size_t)n * 16); hipMalloc(&d_b, (size_t)n * 16);
    hipMemcpy(d_x, x.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> ra((size_t)n * 4), rb((size_t)n * 4);
    long bad = 0, bad_finite_normal = 0;
    for (auto *tp : {&taps, &taps2, &taps3}) {
        hipMemcpy(d_t, tp->data(), 16, hipMemcpyHostToDevice);
        exact_kernel<<<(n / 64 + 3) / 4, 256>>>(d_t, d_x, d_a, d_b, n);
        hipMemcpy(ra.data(), d_a, (size_t)n * 16, hipMemcpyDeviceToHost);
        hipMemcpy(rb.data(), d_b, (size_t)n * 16, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < ra.size(); ++i) {
            uint32_t p, q; memcpy(&p, &ra[i], 4); memcpy(&q, &rb[i], 4);
            if (p != q && !(ra[i] != ra[i] && rb[i] != rb[i])) {
                ++bad;
                const float xv = x[i / 4], av = fabsf(rb[i]);
                const bool normal = fabsf(xv) >= 1.2e-38f && av >= 1.2e-38f && av < 3e38f;
                if (normal) ++bad_finite_normal;
                if (bad <= 10) printf("  mismatch: tap %g * x %g (%a): mfma %a  v_mul %a\n", (*tp)[i % 4], xv, xv, ra[i], rb[i]);
            }
        }
    }
    printf("exactness: %d samples x 12 taps, %ld mismatches (%ld with normal input and result)\n", n, bad, bad_finite_normal);

    // ---- 2. rates at the occupancy of the VIF kernel (4 workgroups of 4 waves per CU = 4 waves per SIMD) and at 2 per SIMD
    float *d; hipMalloc(&d, 256 * 8192 * sizeof(float));
    for (int blocks : {256 * 4, 256 * 2}) {
        run<0>("4 v_mul + 4 v_add", blocks, 2000, d);
        run<1>("1 mfma_4x4x1 (4 products) + 4 v_add", blocks, 2000, d);
        run<2>("4 v_add only", blocks, 2000, d);
        run<3>("1 mfma_4x4x1 only", blocks, 2000, d);
        run<4>("2 v_mul + 1/2 mfma + 4 v_add", blocks, 2000, d);
    }
    return 0;
}
