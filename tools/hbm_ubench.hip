// Micro-benchmark: what HBM delivers on this GPU for the access patterns of the HBM-bound kernels (ADM scale 0, motion):
// read-only streams (dword and dwordx4 per lane), read + 1/4 write, copy.  hipcc --offload-arch=gfx950 -O3 tools/hbm_ubench.hip -o tools/hbm_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T>
__global__ __launch_bounds__(256) void read_kernel(const T *__restrict__ src, size_t n, float *__restrict__ out, int per_thread)
{
    // a workgroup reads per_thread consecutive 256-element rows (like a tile's rows), all loads in flight
    const size_t base = (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
    float acc = 0.0f;
    for (int i = 0; i < per_thread; i += 8) {
        T v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = base + (size_t)(i + k) * 256 < n ? src[base + (size_t)(i + k) * 256] : T{};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += *reinterpret_cast<float *>(&v[k]);
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(256) void read_write_kernel(const float4 *__restrict__ src, size_t n, float4 *__restrict__ dst, int per_thread)
{
    // reads 4 float4, writes 1 (the ratio of ADM scale 0: two planes in, two quarter-size LL bands out)
    const size_t base = (size_t)blockIdx.x * 256 * per_thread + threadIdx.x;
    for (int i = 0; i < per_thread; i += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = base + (size_t)(i + k) * 256 < n ? src[base + (size_t)(i + k) * 256] : float4{};
        float4 a = v[0], b = v[4];
#pragma unroll
        for (int k = 1; k < 4; ++k) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; b.x += v[4 + k].x; b.y += v[4 + k].y; b.z += v[4 + k].z; b.w += v[4 + k].w; }
        const size_t o = ((size_t)blockIdx.x * 256 * per_thread + (size_t)i * 256) / 4 + threadIdx.x;
        dst[o] = a; dst[o + 256] = b;
    }
}

template <typename F> static double time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main()
{
    const size_t bytes = (size_t)2 << 30;                      // 2 GiB: the two planes of 120 1080p frames
    void *src, *dst; float *out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc((void **)&out, 1 << 24));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
    for (int pt : { 8, 16, 32, 64 }) {
        size_t n4 = bytes / 4, n16 = bytes / 16;
        double t = time_ms([&] { read_kernel<float><<<dim3((unsigned)(n4 / (256 * pt))), 256>>>((const float *)src, n4, out, pt); }, 20);
        printf("read dword    x%-2d per thread: %7.1f us  %6.2f TB/s\n", pt, t * 1e3, bytes / t / 1e9);
        t = time_ms([&] { read_kernel<float4><<<dim3((unsigned)(n16 / (256 * pt))), 256>>>((const float4 *)src, n16, out, pt); }, 20);
        printf("read dwordx4  x%-2d per thread: %7.1f us  %6.2f TB/s\n", pt, t * 1e3, bytes / t / 1e9);
        t = time_ms([&] { read_write_kernel<<<dim3((unsigned)(n16 / (256 * pt))), 256>>>((const float4 *)src, n16, (float4 *)dst, pt); }, 20);
        printf("read + 1/4 wr x%-2d per thread: %7.1f us  %6.2f TB/s (of %.2f GB moved)\n", pt, t * 1e3, bytes * 1.25 / t / 1e9, bytes * 1.25 / 1e9);
    }
    double t = time_ms([&] { CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, 0)); }, 20);
    printf("hipMemcpy D2D: %7.1f us  %6.2f TB/s (read + write)\n", t * 1e3, 2.0 * bytes / t / 1e9);
    return 0;
}
