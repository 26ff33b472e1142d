// Micro-benchmark: the load pattern of the VIF scale-0 vertical pass (thread = one of 256 columns, 24 rows x 2 planes in flight,
// 240-column tiles with an 8-column halo either side) for 8-bit, 16-bit and fp32 samples, with nothing else in the kernel:
// does the sample width change what the load path delivers?  1920x1080, 120 frames per launch.
//   hipcc --offload-arch=gfx950 -O3 tools/subdword_ubench.hip -o tools/subdword_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename T> __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, int x, int roff);
template <> __device__ __forceinline__ float ld<float>(__amdgpu_buffer_rsrc_t r, int x, int roff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, x * 4, roff, 0)); }
template <> __device__ __forceinline__ float ld<unsigned char>(__amdgpu_buffer_rsrc_t r, int x, int roff) { return (float)__builtin_amdgcn_raw_buffer_load_b8(r, x, roff, 0); }
template <> __device__ __forceinline__ float ld<unsigned short>(__amdgpu_buffer_rsrc_t r, int x, int roff) { return (float)__builtin_amdgcn_raw_buffer_load_b16(r, x * 2, roff, 0); }

// FMT: typed buffer loads - the descriptor names the sample format (8 or 16 bit, USCALED) and the load path converts to float
__device__ float llvm_amdgcn_raw_ptr_buffer_load_format_f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.ptr.buffer.load.format.f32");
template <typename T>
__global__ __launch_bounds__(256, 4) void tile_read_fmt(const T *__restrict__ ref, const T *__restrict__ dis, int w, int h, float *__restrict__ out)
{
    constexpr int ROWS = 24;
    const int ntx = w / 240;
    const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
    const int x0 = tx * 240 - 8, y0 = ty * 8 - 8;
    int xc = x0 + (int)threadIdx.x;
    xc = xc < 0 ? -xc : xc; xc = xc >= w ? 2 * w - xc - 1 : xc;
    const size_t fo = (size_t)blockIdx.z * w * h;
    const int word3 = 4 | (2 << 12) | ((sizeof(T) == 1 ? 1 : 2) << 15);        // dst_sel_x = X, num_format USCALED, data_format 8 / 16
    const auto br = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(ref + fo), 0, (unsigned)(w * h * sizeof(T)), word3);
    const auto bd = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(dis + fo), 0, (unsigned)(w * h * sizeof(T)), word3);
    float a[ROWS], b[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        int y = y0 + i; y = y < 0 ? -y : y; y = y >= h ? 2 * h - y - 1 : y;
        const int roff = y * w * (int)sizeof(T);
        a[i] = llvm_amdgcn_raw_ptr_buffer_load_format_f32(br, xc * (int)sizeof(T), roff, 0);
        b[i] = llvm_amdgcn_raw_ptr_buffer_load_format_f32(bd, xc * (int)sizeof(T), roff, 0);
    }
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc += a[i] * b[i];
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}

// PAIR: 16-bit samples fetched as the aligned dword that holds the lane's sample (two lanes share one), half picked by shift
template <typename T, int ROWS, bool PAIR>
__global__ __launch_bounds__(256, 4) void tile_read(const T *__restrict__ ref, const T *__restrict__ dis, int w, int h, float *__restrict__ out)
{
    const int ntx = w / 240;
    const int tx = blockIdx.x % ntx, ty = blockIdx.x / ntx;
    const int x0 = tx * 240 - 8, y0 = ty * 8 - 8;
    int xc = x0 + (int)threadIdx.x;
    xc = xc < 0 ? -xc : xc; xc = xc >= w ? 2 * w - xc - 1 : xc;
    const size_t fo = (size_t)blockIdx.z * w * h;
    const auto br = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(ref + fo), 0, (unsigned)(w * h * sizeof(T)), 0x00020000);
    const auto bd = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(dis + fo), 0, (unsigned)(w * h * sizeof(T)), 0x00020000);
    float a[ROWS], b[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        int y = y0 + i; y = y < 0 ? -y : y; y = y >= h ? 2 * h - y - 1 : y;
        const int roff = y * w * (int)sizeof(T);
        if (PAIR) {
            const unsigned ua = __builtin_amdgcn_raw_buffer_load_b32(br, (xc * 2) & ~3, roff, 0);
            const unsigned ub = __builtin_amdgcn_raw_buffer_load_b32(bd, (xc * 2) & ~3, roff, 0);
            const int sh = (xc & 1) * 16;
            a[i] = (float)((ua >> sh) & 0xffffu);
            b[i] = (float)((ub >> sh) & 0xffffu);
        } else {
            a[i] = ld<T>(br, xc, roff);
            b[i] = ld<T>(bd, xc, roff);
        }
    }
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) acc += a[i] * b[i];
    if (acc == 12345.678f) out[blockIdx.x] = acc;
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
    const int w = 1920, h = 1080, frames = 120;
    const size_t n = (size_t)w * h * frames;
    void *ref, *dis; float *out;
    CK(hipMalloc(&ref, n * 4)); CK(hipMalloc(&dis, n * 4)); CK(hipMalloc((void **)&out, 1 << 24));
    CK(hipMemset(ref, 1, n * 4)); CK(hipMemset(dis, 2, n * 4));
    const dim3 g((w / 240) * (h / 8), 1, frames);
    for (int rep = 0; rep < 2; ++rep) {
        double t8 = time_ms([&] { tile_read<unsigned char, 24, false><<<g, 256>>>((const unsigned char *)ref, (const unsigned char *)dis, w, h, out); }, 20);
        double t16 = time_ms([&] { tile_read<unsigned short, 24, false><<<g, 256>>>((const unsigned short *)ref, (const unsigned short *)dis, w, h, out); }, 20);
        double t16p = time_ms([&] { tile_read<unsigned short, 24, true><<<g, 256>>>((const unsigned short *)ref, (const unsigned short *)dis, w, h, out); }, 20);
        double t8f = time_ms([&] { tile_read_fmt<unsigned char><<<g, 256>>>((const unsigned char *)ref, (const unsigned char *)dis, w, h, out); }, 20);
        double t16f = time_ms([&] { tile_read_fmt<unsigned short><<<g, 256>>>((const unsigned short *)ref, (const unsigned short *)dis, w, h, out); }, 20);
        printf("typed (format) loads, conversion in the load path: u8 %.1f us, u16 %.1f us\n", t8f * 1e3, t16f * 1e3);
        double t32 = time_ms([&] { tile_read<float, 24, false><<<g, 256>>>((const float *)ref, (const float *)dis, w, h, out); }, 20);
        printf("24 rows x 2 planes per thread, 1920x1080 x %d frames: u8 %.1f us, u16 %.1f us, u16 as dword pairs %.1f us, f32 %.1f us\n", frames, t8 * 1e3, t16 * 1e3, t16p * 1e3, t32 * 1e3);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
